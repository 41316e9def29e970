"""The training oracle of the Doukhan MTL baseline (oracle/cnn_mtl_train.py) against the inference oracle and
finite differences (CPU)."""
import numpy as np
import pytest

from oracle import cnn_mtl, cnn_mtl_train


def _batch(N, H, W, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(N, H, W)).astype(np.float32)
    c = rng.integers(0, 3, size=N)
    y = {"S": (c == 1).astype(np.float32)[:, None], "M": (c == 0).astype(np.float32)[:, None],
         "R": rng.uniform(0, 1, size=(N, 2)).astype(np.float32), "3C": np.eye(3, dtype=np.float32)[c]}
    return x, y


def test_training_forward_matches_inference_oracle_on_batch_statistics():
    N, H, W = 5, 30, 68
    w = cnn_mtl.init_doukhan(seed=3, H=H, W=W)
    x, y = _batch(N, H, W, 1)
    out = cnn_mtl_train.forward_backward(x, y, w)
    # feed the batch statistics back as moving statistics: the inference graph must reproduce the training features
    w2 = dict(w)
    sizes = {"bn1": N * 27 * 64, "bn2": N * 11 * 30, "bn3": N * 9 * 28, "bn4": N * 3 * 12}
    for p, (mean, var) in out["bn_batch"].items():
        if p in sizes:  # fused layers record the unbiased variance
            var = var * (sizes[p] - 1.0) / sizes[p]
        w2[p + "/moving_mean"], w2[p + "/moving_variance"] = mean.astype(np.float32), var.astype(np.float32)
    _, feat = cnn_mtl.forward_doukhan(x[..., None], w2, return_features=True)
    np.testing.assert_allclose(feat, out["features"], atol=2e-4)
    assert set(out["losses"]) == {"S", "M", "R", "3C"} and 0.0 <= out["acc"] <= 1.0


def test_gradients_against_finite_differences():
    N, H, W = 4, 24, 68
    w = {k: v.astype(np.float64) for k, v in cnn_mtl.init_doukhan(seed=5, H=H, W=W).items()}
    x, y = _batch(N, H, W, 2)
    rng = np.random.default_rng(0)
    drop = [(rng.uniform(size=(N, 512)) < 1 - r) / (1 - r) for r in cnn_mtl_train.FC_DROP]
    dh = {n: (rng.uniform(size=(N, 16)) < 0.6) / 0.6 for n in ("S", "M", "R")}
    base = cnn_mtl_train.forward_backward(x, y, w, drop=drop, drop_heads=dh)
    for name in ("conv1/kernel", "conv3/kernel", "bn2/gamma", "fc1/kernel", "fc3_bn/beta", "S/dense/kernel", "R/out/kernel", "3C/bias"):
        g = base["grads"][name]
        idx = np.unravel_index(int(np.argmax(np.abs(g))), g.shape)
        eps = 1e-7
        vals = []
        for sgn in (1, -1):
            w2 = dict(w)
            a = w[name].copy()
            a[idx] += sgn * eps
            w2[name] = a
            vals.append(cnn_mtl_train.forward_backward(x, y, w2, drop=drop, drop_heads=dh)["loss"])
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert fd == pytest.approx(g[idx], rel=1e-3, abs=1e-7), name


def test_adam_first_step_is_sign_descent():
    w = {"a/kernel": np.array([1.0, -2.0, 3.0]), "a/moving_mean": np.array([0.5])}
    g = {"a/kernel": np.array([0.3, -0.1, 1e-3])}
    nw, m, v = cnn_mtl_train.adam_step(w, g, {}, {}, {"a": (np.array([1.5]), np.array([2.0]))}, step=1, lr=1e-2)
    np.testing.assert_allclose(nw["a/kernel"], w["a/kernel"] - 1e-2 * np.sign(g["a/kernel"]), atol=1e-4)
    np.testing.assert_allclose(nw["a/moving_mean"], [0.99 * 0.5 + 0.01 * 1.5])


def test_papakostas_training_forward_matches_inference_oracle():
    N, H, W = 5, 61, 68
    w = cnn_mtl.init_papakostas(seed=4, H=H, W=W, fc=64)
    x, y = _batch(N, H, W, 3)
    out = cnn_mtl_train.forward_backward(x, y, w, kind="Papakostas")
    w2 = dict(w)
    for p, (mean, var) in out["bn_batch"].items():  # Dense layers only: population variance
        w2[p + "/moving_mean"], w2[p + "/moving_variance"] = mean.astype(np.float32), var.astype(np.float32)
    _, feat = cnn_mtl.forward_papakostas(x[..., None], w2, return_features=True)
    np.testing.assert_allclose(feat, out["features"], atol=2e-4)
    g = out["grads"]
    assert np.abs(g["conv1/bias"]).max() > 0 and np.abs(g["conv2/kernel"]).max() > 0  # no BatchNorm: real bias gradients
    nw = cnn_mtl_train.sgd_step(w, g, out["bn_batch"], 1e-3)
    assert np.allclose(nw["conv3/bias"], w["conv3/bias"] - 1e-3 * g["conv3/bias"])


def test_jang_training_forward_matches_inference_oracle():
    N, W = 3, 12
    w = cnn_mtl.init_jang(seed=2, W=W, mel_init=False)
    x, y = _batch(N, 514, W, 4)
    out = cnn_mtl_train.forward_backward(x, y, w, kind="Jang")
    w2 = dict(w)
    sizes = {"bn1": N * 240 * 12, "bn2": N * 120 * 6, "bn3": N * 60 * 3}
    for p, (mean, var) in out["bn_batch"].items():
        if p in sizes:
            var = var * (sizes[p] - 1.0) / sizes[p]
        w2[p + "/moving_mean"], w2[p + "/moving_variance"] = mean.astype(np.float32), var.astype(np.float32)
    _, (_, feat) = cnn_mtl.forward_jang(x[..., None], w2, return_features=True)
    np.testing.assert_allclose(feat, out["features"], atol=2e-4)
    names = cnn_mtl_train.l2_names("Jang", w)
    assert len(names) == 240 + 3 + 2 + 1 + 3 and out["l2"] == pytest.approx(0.01 * sum(float((w[k].astype(np.float64) ** 2).sum()) for k in names))
    assert np.abs(out["grads"]["harm_melCl7/kernel"]).max() > 0 and np.abs(out["grads"]["perc_melCl119/kernel"]).max() > 0
