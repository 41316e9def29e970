from sm_hpss_mtl_amd.lib.cython_impl.tools import *  # noqa: F401,F403
