"""GPU timing of the signal-conditioning stage (normalise -> rms -> removeSilence -> normalise) on the bench batch.
Prints one JSON line: ms per pass, clips/s, algorithmic GB/s (read x + write out = 8 bytes per sample)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import silence as sil

B, N = int(os.environ.get("B", 1024)), int(os.environ.get("N", 16000))
rng = np.random.default_rng(0)
x = rng.standard_normal((B, N)).astype(np.float32)
x[0::2, N // 8: N // 8 + 3000] *= 1e-4
x[0::2, N // 2: N // 2 + 3000] *= 1e-4
d = torch.from_numpy(x).cuda()
res = {}
for name, fn in (("preprocess_signal", lambda: sil.preprocess_signal(d, 16000, 25, 10)),
                 ("normalize", lambda: sil.normalize(d)),
                 ("rms", lambda: sil.rms(d, 400, 160))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    res[name] = {"ms": round(ms, 4), "clips_per_s": round(B / ms * 1e3), "algo_GBps": round(8.0 * B * N / ms / 1e6, 1)}
print(json.dumps({"B": B, "N": N, **res}), flush=True)
