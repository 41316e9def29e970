import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.synth import synth_clips
from oracle import frontend as ofe
fe = Frontend(FrontendConfig())
for secs in (3, 10, 60):
    y = synth_clips(1, seed=secs, n_samples=16000*secs)
    try:
        res = fe.run(torch.from_numpy(y).cuda(), W=68, shift=34)
        torch.cuda.synchronize()
        fv = res["fv"].cpu().numpy()[0]
        msg = "fv %s patches %s" % (fv.shape, tuple(res["patches"].shape))
        if secs <= 10:
            ref = ofe.featuregram(y[0], "LogMelHarmPercSpec")
            msg += " max|dfv| %.2e" % np.max(np.abs(fv - ref))
        print(secs, "s:", msg, flush=True)
    except Exception as e:
        print(secs, "s: FAILED", type(e).__name__, str(e)[:200], flush=True)
