"""The B3_MTL forward alone (N patches from the layer-0 partials, default 1024), a few launches: a target for rocprofv3 --pmc passes
(tools/gpu/r3_valu_split.sh).  With the timing probe SMH_TCN_BLOCKS=k (SMH_ENABLE_PROBES=1) only k residual blocks run.
Prints the average launch time over the last REPS launches (HIP events on the launch stream)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.model import B3MTL
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
x = torch.randn((N, 2, 68, 32), device="cuda")
out = torch.empty((N, m.out_dim), device="cuda")
for _ in range(5):
    m.forward_from_x0(x, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    m.forward_from_x0(x, out=out)
e1.record()
torch.cuda.synchronize()
m.check_status()
print("N=%d tune=%s: %.2f us per launch" % (N, os.environ.get("SMH_TCN_TUNE", "-"), 1e3 * e0.elapsed_time(e1) / REPS))
