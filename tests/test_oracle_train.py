"""CPU: pin the training-step oracle (oracle/b3_mtl_train.py) against torch autograd on the same formulas."""
import numpy as np
import pytest

from oracle import b3_mtl, b3_mtl_train as tr

torch = pytest.importorskip("torch")


def _torch_loss(x, y, w, ncls, drop_tcn, drop_heads, lw, nb_stacks, n_dil):
    F = torch.nn.functional
    tw = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(tr.TRAINABLE_SKIP)) for k, v in w.items()}
    h = torch.tensor(x, dtype=torch.float64)
    N, T, _ = h.shape

    def conv(inp, k, b, d):
        return F.conv1d(inp.permute(0, 2, 1), k.permute(2, 1, 0), b, padding=d * (k.shape[0] // 2), dilation=d).permute(0, 2, 1)

    h = conv(h, tw["tcn/initial_conv/kernel"], tw["tcn/initial_conv/bias"], 1)
    for bi, (p, d) in enumerate(tr.block_names(nb_stacks, n_dil)):
        u = conv(h, tw[p + "/conv/kernel"], tw[p + "/conv/bias"], d)
        r = torch.relu(u)
        yn = r / (r.abs().amax(dim=2, keepdim=True) + b3_mtl.NORM_EPS)
        yn = yn * torch.tensor(drop_tcn[:, bi][:, None, :])
        h = h + conv(yn, tw[p + "/conv1x1/kernel"], tw[p + "/conv1x1/bias"], 1)
    flat = torch.relu(h).reshape(N, -1)
    total = 0.0
    for name, odim, act in b3_mtl.head_spec(ncls):
        hd = flat @ tw[name + "/dense/kernel"] + tw[name + "/dense/bias"]
        xhat = (hd - hd.mean(0)) / torch.sqrt(hd.var(0, unbiased=False) + b3_mtl.BN_EPS)
        a = torch.relu(xhat * tw[name + "/bn/gamma"] + tw[name + "/bn/beta"]) * torch.tensor(drop_heads[name])
        zo = a @ tw[name + "/out/kernel"] + tw[name + "/out/bias"]
        t = torch.tensor(np.asarray(y[name], np.float64).reshape(N, odim))
        if act == "sigmoid":
            oc = torch.clamp(torch.sigmoid(zo), tr.KERAS_EPS, 1 - tr.KERAS_EPS)
            l = -(t * torch.log(oc + tr.KERAS_EPS) + (1 - t) * torch.log(1 - oc + tr.KERAS_EPS)).mean()
        else:
            l = ((zo - t) ** 2).mean()
        total = total + lw.get(name, 1.0) * l + tr.L2 * (tw[name + "/dense/kernel"] ** 2).sum()
    p = torch.softmax(flat @ tw["3C/kernel"] + tw["3C/bias"], dim=1)
    t3 = torch.tensor(np.asarray(y["3C"], np.float64))
    total = total + lw.get("3C", 1.0) * (-(t3 * torch.log(p)).sum(1)).mean()
    total.backward()
    return float(total), {k: v.grad.numpy() for k, v in tw.items() if v.requires_grad}


@pytest.mark.parametrize("ncls", [3, 5])
def test_training_gradients_match_torch_autograd(ncls):
    rng = np.random.default_rng(ncls)
    N, T, Fd, C, nb_stacks, n_dil = 6, 11, 7, 4, 2, 3
    w = b3_mtl.init_weights(seed=1, n_feat=Fd, patch_size=T, n_classes=ncls, nb_filters=C, nb_stacks=nb_stacks,
                            n_dil=n_dil, randomize_bn=True)
    x = rng.standard_normal((N, T, Fd))
    heads = b3_mtl.head_spec(ncls)
    y = {n: (rng.random((N, od)) > 0.5).astype(float) if act == "sigmoid" else rng.random((N, od)) for n, od, act in heads}
    y["3C"] = np.eye(ncls)[rng.integers(0, ncls, N)]
    nblk = nb_stacks * n_dil
    drop_tcn = (rng.random((N, nblk, C)) > 0.3) / 0.7
    drop_heads = {n: (rng.random((N, 16)) > 0.4) / 0.6 for n, _, _ in heads}
    lw = {"S": 0.7, "R": 1.3}
    res = tr.forward_backward(x, y, w, ncls, drop_tcn, drop_heads, lw, nb_stacks, n_dil)
    ref_loss, ref_g = _torch_loss(x, y, w, ncls, drop_tcn, drop_heads, lw, nb_stacks, n_dil)
    assert abs(res["loss"] - ref_loss) < 1e-9 * max(1, abs(ref_loss))
    for k, gv in ref_g.items():
        np.testing.assert_allclose(res["grads"][k], gv, rtol=1e-7, atol=1e-10, err_msg=k)


def test_sgd_clipnorm_momentum_and_decay():
    w = {"a/kernel": np.array([3.0, 4.0]), "S/bn/moving_mean": np.array([1.0]), "S/bn/moving_variance": np.array([2.0])}
    g = {"a/kernel": np.array([30.0, 40.0])}  # norm 50 -> clipped to 1
    nw, nv = tr.sgd_step(w, g, {}, {"S": (np.array([0.0]), np.array([1.0]))}, lr=0.1)
    np.testing.assert_allclose(nv["a/kernel"], [-0.06, -0.08])
    np.testing.assert_allclose(nw["a/kernel"], [2.94, 3.92])
    np.testing.assert_allclose(nw["S/bn/moving_mean"], [0.99])
    np.testing.assert_allclose(nw["S/bn/moving_variance"], [1.99])
    nw2, nv2 = tr.sgd_step(nw, {"a/kernel": np.array([0.3, 0.4])}, nv, {"S": (np.array([0.0]), np.array([1.0]))}, lr=0.1)
    np.testing.assert_allclose(nv2["a/kernel"], 0.9 * np.array([-0.06, -0.08]) - 0.1 * np.array([0.3, 0.4]))
    assert abs(tr.exponential_decay(30, 0.002, 30, 0.1) - 0.0002) < 1e-15


def test_adam_and_nadam_steps_vs_torch_optim():
    """oracle adam_step / nadam_step (Keras 2.x formulas) against torch.optim on the CPU.  torch.optim.NAdam implements
    the same Dozat momentum schedule (momentum_decay = Keras' schedule_decay = 0.004): agreement to 1e-9.  torch's Adam
    puts epsilon outside the bias correction (sqrt(v)/sqrt(1-b2^t) + eps, Keras: sqrt(v) + eps inside the folded step
    size), an O(eps) difference: 1e-6."""
    import torch
    from oracle import b3_mtl_train as tr
    rng = np.random.default_rng(0)
    w = {"a": rng.standard_normal(5), "b": rng.standard_normal((3, 2))}

    def run(step, opt_cls, **kw):
        w0 = {k: v.copy() for k, v in w.items()}
        tw = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w0.items()}
        opt = opt_cls(list(tw.values()), lr=0.002, eps=1e-7, **kw)
        st = {}
        for _ in range(6):
            g = {k: rng.standard_normal(v.shape) for k, v in w0.items()}
            for k in tw:
                tw[k].grad = torch.tensor(g[k])
            opt.step()
            w0, st = step(w0, g, st, 0.002)
        return max(np.abs(w0[k] - tw[k].detach().numpy()).max() for k in w0)

    assert run(tr.nadam_step, torch.optim.NAdam, momentum_decay=0.004) < 1e-9
    assert run(tr.adam_step, torch.optim.Adam) < 1e-6
