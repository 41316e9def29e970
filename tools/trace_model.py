"""Timeline of the B3_MTL forward's skewed block schedule: s_memtime stamps of every (block, tile) task of workgroup 0
(smh_internal_tcn_trace, tools only).  Prints per-wave busy fractions, the mean wait / compute time per task and the block rate."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd import _lib
N = 1024
m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
x0p = torch.randn((N, 2, 68, 32), device="cuda")
out = torch.empty((N, m.out_dim), device="cuda")
lib = _lib.load() if hasattr(_lib, "load") else m.lib
f = lib.smh_internal_tcn_trace
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
import time
t_end = time.time() + 2.0
while time.time() < t_end:
    for _ in range(200): m.forward_from_x0(x0p, out=out)
    torch.cuda.synchronize()
assert f(1, None, 0) == 0
for _ in range(50): m.forward_from_x0(x0p, out=out)
units, nb = 17, 24
buf2 = np.zeros(2 * 4 * 4096, dtype=np.uint64)
assert f(0, buf2.ctypes.data, buf2.size) == 0
h0, h1 = buf2[:4 * 4096], buf2[4 * 4096:]
e0 = h0[4 * 3000:4 * 3000 + 512].astype(np.int64).reshape(256, 2); e1 = h1[4 * 3000:4 * 3000 + 512].astype(np.int64).reshape(256, 2)
if e0[:, 0].min() > e1[:, 0].min(): e0, e1, h0, h1 = e1, e0, h1, h0
print('two consecutive launches: last exit of one -> first entry of the next %.1f us; first entry -> first entry %.1f us' % ((e1[:, 0].min() - e0[:, 1].max()) / 100.0, (e1[:, 0].min() - e0[:, 0].min()) / 100.0))
buf = h1
allc = buf[4 * 2000:4 * (2000 + 256)].astype(np.int64).reshape(256, 4)
cyc = allc[:, 2] - allc[:, 0]; us = (allc[:, 3] - allc[:, 1]) / 100.0
ghz = cyc / (us * 1000.0)
print('all 256 workgroups: loop cycles min %d median %d max %d; loop us min %.1f median %.1f max %.1f; clock GHz min %.3f median %.3f max %.3f' % (cyc.min(), np.median(cyc), cyc.max(), us.min(), np.median(us), us.max(), ghz.min(), np.median(ghz), ghz.max()))
print('loop start spread (100 MHz ticks): %d, end spread %d, first start -> last end %.1f us' % (allc[:, 1].max() - allc[:, 1].min(), allc[:, 3].max() - allc[:, 3].min(), (allc[:, 3].max() - allc[:, 1].min()) / 100.0))
for x in range(8):
    k = np.arange(256) % 8 == x
    print('  workgroups %d mod 8: median cycles %d, median us %.1f, clock %.3f' % (x, np.median(cyc[k]), np.median(us[k]), np.median(ghz[k])))
ee = buf[4 * 3000:4 * 3000 + 512].astype(np.int64).reshape(256, 2)
t_first = ee[:, 0].min()
print('kernel entry spread %.1f us; entry -> loop start median %.1f us; loop median %.1f us; loop end -> exit median %.1f us; first entry -> last exit %.1f us' % (
    (ee[:, 0].max() - t_first) / 100.0, np.median(allc[:, 1] - ee[:, 0]) / 100.0, np.median(us), np.median(ee[:, 1] - allc[:, 3]) / 100.0, (ee[:, 1].max() - t_first) / 100.0))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    ev0.record(); m.forward_from_x0(x0p, out=out); ev1.record(); torch.cuda.synchronize(); ts.append(ev0.elapsed_time(ev1) * 1000)
print('event-timed launch (stamps on): median %.1f us' % np.median(ts))
print('tasks that found a dependency unfinished (all workgroups): %d of %d' % (int(buf[4 * 3900]), 256 * 17 * 24))
clk = allc[0]
print('block loop of wave 0: %d s_memtime ticks, %d s_memrealtime ticks (100 MHz) -> %.1f us, s_memtime at %.3f GHz' % (clk[2] - clk[0], clk[3] - clk[1], (clk[3] - clk[1]) / 100.0, (clk[2] - clk[0]) / ((clk[3] - clk[1]) * 10.0)))
if not (int(os.environ.get('SMH_TCN_TUNE', '0')) & 128): sys.exit(0)
r = buf[:8 * units * nb].reshape(-1, 8).astype(np.int64)
np.save(os.environ.get("TRACE_OUT", "/tmp/tcn_trace.npy"), r)
st = r[:, :6]; wv = r[:, 6]; fe = r[:, 7]
names = ["top -> first 8 products + publish", "-> rest of the dilated conv issued", "-> task taken, flags sampled, epilogue", "-> next operands issued", "-> 1x1 conv issued, rows stored"]
d = np.diff(st, axis=1)
print("per-task segment cycles (median / mean / p90), %d tasks, fetched ahead %.1f%%" % (len(r), 100.0 * fe.mean()))
for i, nm in enumerate(names):
    print("  %-45s %6.0f %6.0f %6.0f" % (nm, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90)))
print("  task total                                    %6.0f %6.0f" % (np.median(st[:, 5] - st[:, 0]), (st[:, 5] - st[:, 0]).mean()))
nd = 6  # dilations per stack
for dcl in range(nd):
    k = ((np.arange(len(r)) // units) % nd) == dcl
    print("  blocks of dilation %2d: fetched ahead %5.1f %%, task total median %5.0f, next-operands segment median %5.0f" % (
        1 << dcl, 100.0 * fe[k].mean(), np.median((st[:, 5] - st[:, 0])[k]), np.median(d[k, 3])))
for w in range(8):
    k = np.where(wv == w)[0]
    gaps = st[k[1:], 0] - st[k[:-1], 5]
    print("  wave %d: %d tasks, between tasks median %.0f mean %.0f" % (w, len(k), np.median(gaps), gaps.mean()))
