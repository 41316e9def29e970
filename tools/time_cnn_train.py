"""Time one training step (forward + backward + Adam) of the Doukhan MTL baseline at the reference's batch (3 x 16 = 48)
and at 192 patches; prints patches/s and the fraction of the f32 MFMA peak (3x the forward flops)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sm_hpss_mtl_amd.cnn_models import CnnMTL  # noqa: E402

# forward GFLOP per patch (2 x MACs of every Conv2D / Dense at the reference's input size)
SHAPES = {"Doukhan": ((240, 68), 1.906), "Papakostas": ((402, 68), 0.755), "Jang": ((514, 68), 0.479)}
PEAK = 157.3e12
kinds = sys.argv[1:] or ["Doukhan"]

for kind, N in [(k, n) for k in kinds for n in (48, 192)]:
    (H, W), FWD_GFLOP = SHAPES[kind]
    m = CnnMTL(kind, (H, W, 1), seed=0)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(size=(N, H, W)).astype(np.float32)).cuda()
    c = np.arange(N) % 3
    y = m.pack_targets([(c == 1).astype(np.float32)[:, None], (c == 0).astype(np.float32)[:, None],
                        rng.uniform(size=(N, 2)).astype(np.float32), np.eye(3, dtype=np.float32)[c]])
    for _ in range(3):
        m.train_on_batch(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        m.train_on_batch(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%s N=%d: %.2f ms/step = %.0f patches/s, %.1f%% of the f32 MFMA peak (3 x forward flops)"
          % (kind, N, dt * 1e3, N / dt, 100 * 3 * FWD_GFLOP * 1e9 * N / dt / PEAK), flush=True)
