from sm_hpss_mtl_amd.lib.preprocessing import *  # noqa: F401,F403
