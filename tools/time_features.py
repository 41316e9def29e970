"""GPU timing of the single feature kernel (features_clip_kernel) by phases: SMH_FEAT_STOP = 1 (walk), 2 (+ dB / clip / write),
3 (+ statistics), 0 (whole kernel, + layer 0).  Inputs are the bench's (B = 1024, 17 x 17)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import _lib
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.pipeline import HotPath
from sm_hpss_mtl_amd.synth import synth_clips

B = 1024
fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
hp = HotPath(fe, model, B, 16000)
audio = torch.from_numpy(np.tile(synth_clips(64, seed=1000), (B // 64, 1))).cuda()
hp.step(audio); torch.cuda.synchronize()
lib, h = fe.lib, fe._h
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
w0 = C.c_void_p(lib.smh_model_w0_ptr(model._h))
def feat():
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.smh_features_l0_f32(h, p(hp.S), p(hp.harm), p(hp.perc), hp.layout, B, hp.T, 68, 68, p(hp.fv), None, w0, p(hp.x0p), p(hp.maxkeys), st))
def t(reps=20):
    for _ in range(3): feat()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); feat(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for rnd in range(2):
  for label, env in (("half-clip WGs", {}), ("one WG per clip", {"SMH_FEAT_NOSPLIT": "1"})):
    for k in ("SMH_FEAT_NOSPLIT", "SMH_FEAT_NOPAIR"): os.environ.pop(k, None)
    os.environ.update(env)
    prev = 0.0
    for stop, name in ((1, "walk"), (2, "dB+clip+write"), (3, "statistics"), (0, "layer 0")):
        os.environ["SMH_FEAT_STOP"] = str(stop)
        v = t()
        print("%-16s through %-14s %.4f ms   (+%.4f)" % (label, name, v, v - prev), flush=True)
        prev = v
os.environ.pop("SMH_FEAT_STOP", None)
