"""Debug probe: gradient error of the Doukhan training step per tensor for a few input sizes."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_cnn_train_gpu import _batch, _model  # noqa: E402
from oracle import cnn_mtl_train  # noqa: E402

for H, W, N in [(240, 68, 2), (240, 68, 12), (120, 68, 4), (160, 68, 4), (200, 68, 4), (128, 68, 3), (136, 68, 3)]:
    m, w = _model(H, W)
    x, y = _batch(N, H, W, 1)
    ref = cnn_mtl_train.forward_backward(x, y, w)
    m.train_on_batch(x, y, drop=None, drop_heads=None, apply=False)
    g = m.gradients()
    out = []
    for name in ("conv4/kernel", "bn4/gamma", "conv3/kernel", "bn3/gamma", "conv2/kernel", "conv1/kernel", "fc1/kernel"):
        r = ref["grads"][name]
        out.append("%s %.1e" % (name.split("/")[0], np.abs(g[name] - r).max() / np.abs(r).max()))
    print(H, W, N, " ".join(out), flush=True)
