"""Timeline of features_half_kernel: s_memrealtime stamps per wave at the phase boundaries (smh_internal_feat_trace, tools only).
Prints how long the phases last per wave and workgroup, how many workgroups share a CU over time, and the whole-kernel span."""
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
for _ in range(20): hp.step(audio)
torch.cuda.synchronize()
lib = fe.lib
f = lib.smh_internal_feat_trace
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
assert f(1, None, 0) == 0
for _ in range(5): hp.step(audio)
nwg = 16 * ((B + 7) // 8)
buf = np.zeros(nwg * 8 * 8, dtype=np.uint64)
assert f(0, buf.ctypes.data, buf.size) == 0
r = buf.reshape(nwg, 8, 8).astype(np.int64)
t = r[:, :, :6] / 100.0  # us
hw = r[:, 0, 7]
t0 = t[:, :, 0].min()
names = ["walk (own segment)", "wait + dB/clip/write", "statistics", "barrier", "layer 0 (incl. idle waves)"]
print("kernel: first wave start -> last wave end %.1f us; %d workgroups" % (t[:, :, 5].max() - t0, nwg))
for i, nm in enumerate(names):
    d = t[:, :, i + 1] - t[:, :, i]
    print("  %-28s per wave: median %6.2f  mean %6.2f  p90 %6.2f us" % (nm, np.median(d), d.mean(), np.percentile(d, 90)))
wg = t[:, :, 5].max(axis=1) - t[:, :, 0].min(axis=1)
print("  workgroup lifetime: median %.2f mean %.2f p90 %.2f us; sum of lifetimes / (256 CUs x span) = %.2f resident per CU" % (
    np.median(wg), wg.mean(), np.percentile(wg, 90), wg.sum() / (256 * (t[:, :, 5].max() - t0))))
l0 = t[:, :, 5] - t[:, :, 4]
busy = l0[:, :5]
print("  layer 0, waves that hold a tile task (0..4): median %.2f mean %.2f p90 %.2f us; idle waves: median %.2f" % (
    np.median(busy), busy.mean(), np.percentile(busy, 90), np.median(l0[:, 5:])))
start = t[:, :, 0].min(axis=1) - t0
order = np.argsort(start)
print("  workgroup starts: %d within the first 2 us, then at (us): %s ..." % ((start < 2).sum(), np.round(np.sort(start)[(start < 2).sum():(start < 2).sum() + 12], 1)))
# phases of all workgroups on a time grid: how many are in which phase
grid = np.arange(0, t[:, :, 5].max() - t0, 2.0)
wt = t.mean(axis=1) - t0  # per-workgroup mean stamps
occ = np.zeros((len(grid), 5))
for i in range(5):
    for g, x in enumerate(grid):
        occ[g, i] = ((wt[:, i] <= x) & (wt[:, i + 1] > x)).sum()
print("  time(us): workgroups in [walk, write, statistics, barrier, layer 0]")
for g in range(0, len(grid), 4):
    print("   %5.0f: %s" % (grid[g], occ[g].astype(int)))
np.save(os.environ.get("TRACE_OUT", "/tmp/feat_trace.npy"), r)
