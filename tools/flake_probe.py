import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_training_gpu as T
from sm_hpss_mtl_amd.model import B3MTL
w, x, y, _, _ = T._problem(3, 96, seed=13)
def run(presize):
    m = B3MTL(n_feat=240, patch_size=68, n_classes=3, TR_STEPS=10)
    m.set_weights_dict(w)
    if presize: m._get_trainer(96)
    small = {k: v[:48] for k, v in y.items()}
    for _ in range(2): m.train_on_batch(x[:48], small, drop_tcn=None, drop_heads=None)
    m.train_on_batch(x, y, drop_tcn=None, drop_heads=None)
    return m.get_weights_dict()
for rep in range(6):
    a, b = run(rep % 2 == 0), run(True)
    worst = max(((np.abs(a[k] - b[k]).max() / (1e-3 * np.abs(b[k] - w[k]).max() + 1e-7)), k) for k in a)
    print("rep %d presize=%s: worst ratio to the test's tolerance %.2f at %s" % (rep, rep % 2 == 0, worst[0], worst[1]))
