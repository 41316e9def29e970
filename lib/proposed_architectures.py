from sm_hpss_mtl_amd.lib.proposed_architectures import *  # noqa: F401,F403
