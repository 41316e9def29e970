"""diagnostic: which gradient tensors differ bit-wise between two identical training steps (run to run, and between trainer
capacities), for the Conv2D baseline trainer (Doukhan) and for B3_MTL."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tests.test_cnn_train_gpu import _batch, _model


def grads_cnn(cap, N, reps=3):
    x, y = _batch(60, 30, 68, 5)
    m, w = _model(30, 68, seed=13)
    m._get_trainer(cap)
    out = []
    for _ in range(reps):
        m.train_on_batch(x[:N], {k: v[:N] for k, v in y.items()}, drop=None, drop_heads=None, apply=False)
        torch.cuda.synchronize()
        out.append(m._grad_tensor().clone())
    return m, out


def report(name, a, b, spec):
    o = 0
    bad = []
    for n, shape in spec:
        k = int(np.prod(shape))
        if not torch.equal(a[o:o + k], b[o:o + k]):
            d = (a[o:o + k] - b[o:o + k]).abs().max().item()
            bad.append((n, d, a[o:o + k].abs().max().item()))
        o += k
    print(name, "differing tensors:", bad if bad else "none")


m, g48 = grads_cnn(48, 24)
spec = [(t[0], t[1]) for t in m.tensor_specs()] if hasattr(m, "tensor_specs") else None
if spec is None:
    spec = [(k, v.shape) for k, v in m.get_weights_dict().items()]
report("cnn cap48 run0 vs run1", g48[0], g48[1], spec)
report("cnn cap48 run0 vs run2", g48[0], g48[2], spec)
_, g60 = grads_cnn(60, 24)
report("cnn cap48 vs cap60     ", g48[0], g60[0], spec)
_, g60b = grads_cnn(60, 60)
_, g60c = grads_cnn(60, 60)
report("cnn N=60 two trainers  ", g60b[0], g60c[0], spec)

# B3_MTL
from tests.test_training_gpu import _problem
from sm_hpss_mtl_amd.model import B3MTL
w, x, y, _, _ = _problem(3, 48, seed=9)
mb = B3MTL(n_feat=240, patch_size=68, n_classes=3, TR_STEPS=10)
mb.set_weights_dict(w)
gs = []
for _ in range(3):
    mb.train_on_batch(x, y, drop_tcn=None, drop_heads=None, apply=False)
    torch.cuda.synchronize()
    gs.append(mb._grad_tensor().clone())
specb = [(n, s) for n, s, _, _ in mb._spec]
report("b3mtl run0 vs run1     ", gs[0], gs[1], specb)
report("b3mtl run0 vs run2     ", gs[0], gs[2], specb)
