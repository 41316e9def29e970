"""diagnostic: 16-wave skew schedule vs the barrier schedule, by number of residual blocks (timing probe SMH_TCN_BLOCKS)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import b3_mtl
from sm_hpss_mtl_amd.model import B3MTL
W, N = int(sys.argv[1]), int(sys.argv[2])
w = b3_mtl.init_weights(seed=5, n_feat=240, patch_size=W, n_classes=3, randomize_bn=True)
m = B3MTL(n_feat=240, patch_size=W, n_classes=3); m.set_weights_dict(w)
x = torch.from_numpy(np.random.default_rng(3).standard_normal((N, W, 240)).astype(np.float32)).cuda()
os.environ["SMH_ENABLE_PROBES"] = "1"
for nb in (0, 1, 2, 3, 4, 5, 8, 24):
    os.environ["SMH_TCN_BLOCKS"] = str(nb)
    res = {}
    for sk in ("0", "2"):
        os.environ["SMH_TCN_SKEW"] = sk
        tr = torch.empty((N, W, 32), device="cuda")
        m.forward_device(x, trunk=tr); torch.cuda.synchronize()
        try:
            m.check_status()
        except RuntimeError as e:
            print("  status:", sk, str(e)[:80])
        res[sk] = tr.cpu().numpy()
    d = np.abs(res["0"] - res["2"])
    bad = np.argwhere(d > 0)
    print("blocks", nb, "max diff", d.max(), "n bad", len(bad), "bad patches", sorted(set(bad[:, 0].tolist()))[:8], "bad rows", sorted(set(bad[:, 1].tolist()))[:20])
