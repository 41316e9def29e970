"""GPU timing of one B3_MTL training step (forward-train + heads + backward + optimiser) at patch width W and N patches:
python tools/time_train_w.py W N [K].  Target of `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.batching import make_labels_3class
W, N = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = B3MTL(n_feat=240, patch_size=W, n_classes=3, TR_STEPS=100, seed=0)
x = torch.randn((N, W, 240), device="cuda")
lab = make_labels_3class(N // 3, np.zeros(N // 3))
y = m.pack_targets({k: v[:N] if len(v) >= N else np.resize(v, (N,) + v.shape[1:]) for k, v in lab.items()})
if y.shape[0] < N:
    y = torch.cat([y, y[: N - y.shape[0]]])
for _ in range(3):
    m.train_on_batch(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    m.train_on_batch(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("W=%d N=%d: %.3f ms per training step -> %.0f patches/s" % (W, N, dt * 1e3, N / dt), flush=True)
