"""GPU timing of the B3_MTL inference kernel over batch sizes, skew schedule (SMH_TCN_SKEW=2: whenever it can run) against barrier schedule (=0)
and the default choice (the variable is read at every launch).  Steady state: a pre-roll, then 60 back-to-back launches between two events per point."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.model import B3MTL

W = int(os.environ.get("TIME_W", "68"))
m = B3MTL(n_feat=240, patch_size=W, n_classes=3, seed=0)
sizes = [int(v) for v in os.environ.get("TIME_N", "1,16,48,128,256,384,510,768,1024,2048,4096").split(",")]
big = torch.randn((1024, W, 240), device="cuda")
for _ in range(300):  # clocks up
    m.forward_device(big)
torch.cuda.synchronize()
for n in sizes:
    x = torch.randn((n, W, 240), device="cuda")
    out = torch.empty((n, m.out_dim), device="cuda")
    res = {}
    for sk in ("2", "0", "1"):  # forced skew / barrier / the rule of launch_forward
        os.environ["SMH_TCN_SKEW"] = sk
        for _ in range(20):
            m.forward_device(x, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(60):
            m.forward_device(x, out=out)
        b.record()
        torch.cuda.synchronize()
        res[sk] = a.elapsed_time(b) / 60 * 1e3
    # the barrier schedule without the two-wave split of a lone last-round tile (SMH_TCN_SPLIT=0)
    os.environ["SMH_TCN_SKEW"], os.environ["SMH_TCN_SPLIT"] = "0", "0"
    for _ in range(20):
        m.forward_device(x, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(60):
        m.forward_device(x, out=out)
    b.record()
    torch.cuda.synchronize()
    del os.environ["SMH_TCN_SPLIT"]
    print("W=%d N=%5d  skew %8.1f us   barrier %8.1f us (unsplit %8.1f)   default %8.1f us" % (W, n, res["2"], res["0"], a.elapsed_time(b) / 60 * 1e3, res["1"]), flush=True)
