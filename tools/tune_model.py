"""GPU timing of the B3_MTL forward kernel by parts (SMH_TCN_BLOCKS override), one process."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.model import B3MTL
N = 1024
m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
x = torch.randn((N, 68, 240), device="cuda")
out = torch.empty((N, m.out_dim), device="cuda")
X0 = os.environ.get("TUNE_X0") == "1"   # the bench path: start from the layer-0 partials (N, 2, 68, 32)
x0p = torch.randn((N, 2, 68, 32), device="cuda")
trunk = torch.empty((N, 68, 32), device="cuda")
def fwd():
    if X0: m.forward_from_x0(x0p, out=out, trunk=trunk)
    else: m.forward_device(x, out=out)
def t(nb, rounds=12):
    if nb is None: os.environ.pop("SMH_TCN_BLOCKS", None)
    else: os.environ["SMH_TCN_BLOCKS"] = str(nb)
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fwd(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[3:]))
os.environ['SMH_TCN_NOHEADS']='1'
print('no heads, blocks=0: %.4f ms' % t(0)); print('no heads, blocks=24: %.4f ms' % t(24))
os.environ.pop('SMH_TCN_NOHEADS')
if os.environ.get('TUNE_SHORT'):
    print('blocks=0 %.4f  blocks=24 %.4f' % (t(0), t(24))); sys.exit(0)
for nb in (None, 0, 2, 8, 16, 24):
    print("blocks=%s  %.4f ms" % (nb, t(nb)), flush=True)
for n in (256, 512, 2048):
    x2 = torch.randn((n, 68, 240), device="cuda"); o2 = torch.empty((n, m.out_dim), device="cuda")
    os.environ.pop("SMH_TCN_BLOCKS", None)
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m.forward_device(x2, out=o2); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("N=%d  %.4f ms" % (n, float(np.median(ts[3:]))), flush=True)
