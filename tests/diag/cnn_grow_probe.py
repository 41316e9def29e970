"""diagnostic: the 'growing trainer' flow of tests/test_cnn_train_gpu.py, several times: which weights differ bit-wise between
a trainer grown 48 -> 60 and one pre-sized to 60, after each step?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tests.test_cnn_train_gpu import _batch, _model
x, y = _batch(60, 30, 68, 5)
for rep in range(3):
    res = []
    for presize in (False, True):
        m, w = _model(30, 68, seed=13)
        if presize:
            m._get_trainer(60)
        snaps = []
        for _ in range(2):
            m.train_on_batch(x[:24], {k: v[:24] for k, v in y.items()}, drop=None, drop_heads=None)
            snaps.append({k: v.copy() for k, v in m.get_weights_dict().items()})
        m.train_on_batch(x, y, drop=None, drop_heads=None)
        snaps.append({k: v.copy() for k, v in m.get_weights_dict().items()})
        res.append(snaps)
    for step in range(3):
        bad = [(k, float(np.abs(res[0][step][k] - res[1][step][k]).max())) for k in res[0][step] if not np.array_equal(res[0][step][k], res[1][step][k])]
        print("rep", rep, "after step", step + 1, "differing:", bad[:6], "..." if len(bad) > 6 else "", len(bad))
