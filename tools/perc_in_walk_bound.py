"""VERDICT r2 item 2, measured: what would "the percussive median inside the feature walk" cost and save?

Today (B = 1024, 17 x 17):  median kernel (harm + perc) -> features_half_kernel (reads S, harm, perc).
Proposal:                   median kernel (harm only)   -> feature walk that carries the l_perc sorted window itself and
                            never reads (nor does anyone write) perc: 81 MB of writes + 81 MB of reads per step gone.
This script times both sides of the proposal on the bench batch:
  * the harm-only block-split median in the blocked layout (smh_median_time_ex_f32: a real entry point, valid output);
  * the feature kernel with the proposal's instruction stream in place (timing probe SMH_FEAT_PROBE_PERC = n, outputs invalid):
    perc not read, 2 n `v_med3_f32` per bin step (n per frame of a lane's frame pair; the block-split scheme needs 17.6
    selection instructions per output at window 17 -- measured), 16 more bins of S per segment for the window's warm-up.
    The probe keeps 2 registers for its window where the real thing needs 80 per frame: it is the OPTIMISTIC side of the estimate.
Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import _lib  # noqa: E402
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig  # noqa: E402
from sm_hpss_mtl_amd.model import B3MTL  # noqa: E402
from sm_hpss_mtl_amd.pipeline import HotPath  # noqa: E402
from sm_hpss_mtl_amd.synth import bench_clips  # noqa: E402

B, REPS = 1024, 200
fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
hp = HotPath(fe, model, B, 16000)
audio = torch.from_numpy(bench_clips(B, 0)).cuda()
lib, h = fe.lib, fe._h
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for _ in range(60):
    hp.step(audio)


def timed(fn):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / REPS * 1e3, 1)  # us


def median_both():
    _lib.check(lib.smh_hpss_median_ex_f32(h, p(hp.S), B, fe.K, hp.T, 17, 17, p(hp.harm), p(hp.perc), 2, st))


def median_harm_only():
    assert _lib.check(lib.smh_median_time_ex_f32(h, p(hp.S), B, fe.K, hp.T, 17, p(hp.harm), 2, st)) == 2


def features():
    _lib.check(lib.smh_features_l0_f32(h, p(hp.S), p(hp.harm), p(hp.perc), 2, B, hp.T, 68, 68, p(hp.fv), None,
                                       C.c_void_p(lib.smh_model_w0_ptr(model._h)), p(hp.x0p), p(hp.maxkeys), st))


res = {"batch": B, "windows": "17x17", "unit": "us per launch"}
res["median_harm_and_perc"] = timed(median_both)
res["median_harm_only"] = timed(median_harm_only)
median_both()
res["features_today"] = timed(features)
os.environ["SMH_ENABLE_PROBES"] = "1"
for n in (0, 6, 12, 18):
    if n:
        os.environ["SMH_FEAT_PROBE_PERC"] = str(n)
    else:
        os.environ["SMH_FEAT_PROBE_PERC"] = "0"
    stderr = os.dup(2)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)  # the probe announces itself on every launch
    try:
        res["features_probe_n%d" % n] = timed(features)
    finally:
        os.dup2(stderr, 2)
        os.close(devnull)
os.environ.pop("SMH_FEAT_PROBE_PERC")
res["today_median_plus_features"] = round(res["median_harm_and_perc"] + res["features_today"], 1)
res["proposal_median_plus_features_n18"] = round(res["median_harm_only"] + res["features_probe_n18"], 1)
print(json.dumps(res))
