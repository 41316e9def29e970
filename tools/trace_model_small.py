"""Where the time of the B3_MTL forward goes at a given batch (default: BASELINE config 3's 256 patches from PATCHES): kernel
entry -> block loop, the 24-block loop, loop -> exit, from the TRACE instantiation's s_memtime / s_memrealtime stamps."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
from_x0 = len(sys.argv) > 2 and sys.argv[2] == "x0"
m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
x = torch.randn((N, 2, 68, 32), device="cuda") if from_x0 else torch.randn((N, 68, 240), device="cuda")
out = torch.empty((N, m.out_dim), device="cuda")
run = (lambda: m.forward_from_x0(x, out=out)) if from_x0 else (lambda: m.forward_device(x, out=out))
f = m.lib.smh_internal_tcn_trace
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
t_end = time.time() + 1.0
while time.time() < t_end:
    for _ in range(200): run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print("N = %d (%s): %.1f us per launch, back to back, plain build" % (N, "layer-0 partials" if from_x0 else "patches", e0.elapsed_time(e1) / 200 * 1e3))
assert f(1, None, 0) == 0
for _ in range(50): run()
buf2 = np.zeros(2 * 4 * 4096, dtype=np.uint64)
assert f(0, buf2.ctypes.data, buf2.size) == 0
nwg = min(256, (N + 3) // 4 if N > 256 * 1 else N)  # workgroups (G = ceil(N / 256) patches each)
G = max(1, (N + 255) // 256); nwg = min(256, (N + G - 1) // G)
for h in (buf2[:4 * 4096], buf2[4 * 4096:]):
    ee = h[4 * 3000:4 * 3000 + 2 * nwg].astype(np.int64).reshape(nwg, 2)
    lp = h[4 * 2000:4 * (2000 + nwg)].astype(np.int64).reshape(nwg, 4)
    if ee[:, 0].min() == 0: continue
    us = (lp[:, 3] - lp[:, 1]) / 100.0
    print("  %d workgroups: entry -> loop %.1f us, loop %.1f us (%.0f k cycles, %.2f GHz), loop -> exit %.1f us, first entry -> last exit %.1f us" % (
        nwg, np.median(lp[:, 1] - ee[:, 0]) / 100.0, np.median(us), np.median(lp[:, 2] - lp[:, 0]) / 1e3,
        np.median((lp[:, 2] - lp[:, 0]) / (us * 1000.0)), np.median(ee[:, 1] - lp[:, 3]) / 100.0, (ee[:, 1].max() - ee[:, 0].min()) / 100.0))
