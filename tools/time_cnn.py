"""GPU timing of the Conv2D MTL baselines' forward pass at the reference shapes.  One JSON line per model."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.cnn_models import CnnMTL

FLOPS = {"Doukhan": 1.897e9, "Papakostas": 0.755e9, "Jang": None}
N = int(os.environ.get("N", 256))
for kind, shape in (("Doukhan", (240, 68, 1)), ("Jang", (514, 68, 1)), ("Papakostas", (402, 68, 1))):
    if os.environ.get("ONLY") and os.environ["ONLY"] != kind:
        continue
    m = CnnMTL(kind, shape, seed=0)
    x = torch.randn((N,) + shape[:2], device="cuda")
    for dtype in ("f32", "bf16"):
        for _ in range(2):
            m.forward_device(x, dtype=dtype)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 5
        e0.record()
        for _ in range(K):
            m.forward_device(x, dtype=dtype)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        r = {"model": kind, "dtype": dtype, "N": N, "ms": round(ms, 3), "patches_per_s": round(N / ms * 1e3)}
        if FLOPS[kind]:
            r["TFLOPs"] = round(FLOPS[kind] * N / ms / 1e9, 2)
            if dtype == "f32":
                r["frac_f32_mfma_peak"] = round(FLOPS[kind] * N / ms / 1e9 / 157.3, 3)
        print(json.dumps(r), flush=True)
