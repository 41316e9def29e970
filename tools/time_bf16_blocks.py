import os, sys, torch, numpy as np, time
sys.path.insert(0, os.getcwd())
from sm_hpss_mtl_amd.model import B3MTL
import ctypes as C
m = B3MTL(n_feat=240, patch_size=68, n_classes=5, seed=0)
x0p = torch.randn(1024, 2, 68, 32, device="cuda")
out = torch.empty(1024, m.out_dim, device="cuda")
from sm_hpss_mtl_amd import _lib
st = _lib.current_stream
def run():
    _lib.check(m.lib.smh_model_forward_x0_bf16(m._h, C.c_void_p(x0p.data_ptr()), 1024, C.c_void_p(out.data_ptr()), 1, st()))
m._sync_weights()
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print("blocks=%s  %.1f us" % (os.environ.get("SMH_TCN_BLOCKS", "24"), e0.elapsed_time(e1) / 200 * 1e3))
