"""numpy (float64) restatement of ONE TRAINING STEP of B3_MTL (TEST INFRASTRUCTURE, see oracle/__init__).

What the reference runs inside `model.fit` (Proposed_Work_Results.py:298-307) for the model compiled at
lib/proposed_architectures.py:156-165: losses S,M(,N): binary_crossentropy, R: mean_squared_error,
3C: categorical_crossentropy, optional loss_weights, l2(0.01) on the Dense(16) kernels
(MTL_modifications, :46,60,73), SGD(momentum 0.9, clipnorm=1, lr = 0.002 * 0.1**(step/(3*TR_STEPS))).
Training-mode layers restated from Keras 2.x / keras-tcn 2.3 ("parity unpinned", like the forward):
  SpatialDropout1D(rate) after the channel normalisation of every TCN block (drops whole channels per
  sample: mask (N, 1, C), kept values scaled by 1/(1-rate));  BatchNormalization with batch statistics
  (population variance), momentum 0.99 moving averages, eps 1e-3;  Dropout(0.4) in every head.
Random masks are INPUTS here so that the HIP path can be compared on identical masks.

The gradient arithmetic itself is pinned in tests/test_oracle_train.py against torch autograd (CPU).
"""
from __future__ import annotations

import numpy as np

from .b3_mtl import BN_EPS, NORM_EPS, head_spec

KERAS_EPS = 1e-7
BN_MOMENTUM = 0.99
L2 = 0.01


def _conv_same(x, kernel, bias, d):
    """x (N,T,Cin) f64, kernel (k,Cin,Cout): y[t] = sum_j x[t+(j-k//2)d] @ W[j] + b."""
    N, T, _ = x.shape
    k = kernel.shape[0]
    y = np.zeros((N, T, kernel.shape[2]))
    for j in range(k):
        off = (j - k // 2) * d
        lo, hi = max(0, -off), min(T, T - off)
        if lo < hi:
            y[:, lo:hi] += x[:, lo + off:hi + off] @ kernel[j]
    return y + bias


def _conv_same_backward(x, kernel, d, dy):
    """returns dx, dkernel, dbias for y = conv_same(x)."""
    N, T, _ = x.shape
    k = kernel.shape[0]
    dx = np.zeros_like(x)
    dk = np.zeros_like(kernel)
    for j in range(k):
        off = (j - k // 2) * d
        lo, hi = max(0, -off), min(T, T - off)
        if lo < hi:
            xs = x[:, lo + off:hi + off]
            g = dy[:, lo:hi]
            dk[j] = np.einsum("ntc,nto->co", xs, g)
            dx[:, lo + off:hi + off] += g @ kernel[j].T
    return dx, dk, dy.sum(axis=(0, 1))


def block_names(nb_stacks=3, n_dil=8):
    return [("tcn/s%d_d%d" % (s, 2 ** i), 2 ** i) for s in range(nb_stacks) for i in range(n_dil)]


def forward_backward(x, y, w, n_classes=3, drop_tcn=None, drop_heads=None, loss_weights=None, nb_stacks=3, n_dil=8):
    """One training forward + backward.

    x (N,T,F); y: dict name -> targets ('S','M',['N'],'R','3C' one-hot); w: weights dict (oracle.b3_mtl order);
    drop_tcn: (N, n_blocks, C) multiplicative masks (0 or 1/(1-rate)) or None; drop_heads: dict head -> (N,16).
    Returns dict(loss, losses{name}, acc, grads{name}, bn_batch{head: (mean, var)}).
    """
    w = {k: np.asarray(v, np.float64) for k, v in w.items()}
    x = np.asarray(x, np.float64)
    N, T, _ = x.shape
    heads = head_spec(n_classes)
    lw = {n: 1.0 for n, _, _ in heads}
    lw["3C"] = 1.0
    if loss_weights:
        lw.update(loss_weights)
    blocks = block_names(nb_stacks, n_dil)
    # ---------------- forward ----------------
    saved = []
    h = _conv_same(x, w["tcn/initial_conv/kernel"], w["tcn/initial_conv/bias"], 1)
    for bi, (p, d) in enumerate(blocks):
        u = _conv_same(h, w[p + "/conv/kernel"], w[p + "/conv/bias"], d)
        r = np.maximum(u, 0.0)
        mx = r.max(axis=2, keepdims=True)
        m = mx + NORM_EPS
        yn = r / m
        mask = np.ones((N, 1, r.shape[2])) if drop_tcn is None else np.asarray(drop_tcn, np.float64)[:, bi][:, None, :]
        ynd = yn * mask
        z = _conv_same(ynd, w[p + "/conv1x1/kernel"], w[p + "/conv1x1/bias"], 1)
        saved.append((h, u, r, mx, m, yn, mask, ynd))
        h = h + z
    trunk_pre = h
    trunk = np.maximum(h, 0.0)
    flat = trunk.reshape(N, -1)
    out, cache = {}, {}
    losses = {}
    for name, odim, act in heads:
        hd = flat @ w[name + "/dense/kernel"] + w[name + "/dense/bias"]
        mean = hd.mean(axis=0)
        var = hd.var(axis=0)  # population variance
        xhat = (hd - mean) / np.sqrt(var + BN_EPS)
        bn = xhat * w[name + "/bn/gamma"] + w[name + "/bn/beta"]
        a = np.maximum(bn, 0.0)
        dm = np.ones_like(a) if not drop_heads or name not in drop_heads else np.asarray(drop_heads[name], np.float64)
        ad = a * dm
        zo = ad @ w[name + "/out/kernel"] + w[name + "/out/bias"]
        if act == "sigmoid":
            o = 1.0 / (1.0 + np.exp(-zo))
            oc = np.clip(o, KERAS_EPS, 1 - KERAS_EPS)
            t = np.asarray(y[name], np.float64).reshape(N, odim)
            losses[name] = float(np.mean(-(t * np.log(oc + KERAS_EPS) + (1 - t) * np.log(1 - oc + KERAS_EPS))))
        else:
            o = zo
            t = np.asarray(y[name], np.float64).reshape(N, odim)
            losses[name] = float(np.mean((o - t) ** 2))
        out[name] = o
        cache[name] = (hd, mean, var, xhat, bn, a, dm, ad, zo, t)
    logits = flat @ w["3C/kernel"] + w["3C/bias"]
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    p = e / e.sum(axis=1, keepdims=True)
    t3 = np.asarray(y["3C"], np.float64).reshape(N, n_classes)
    pc = np.clip(p / p.sum(axis=1, keepdims=True), KERAS_EPS, 1 - KERAS_EPS)
    losses["3C"] = float(np.mean(-np.sum(t3 * np.log(pc), axis=1)))
    out["3C"] = p
    reg = sum(L2 * float(np.sum(w[n + "/dense/kernel"] ** 2)) for n, _, _ in heads)
    total = sum(lw[k] * v for k, v in losses.items()) + reg
    acc = float(np.mean(p.argmax(1) == t3.argmax(1)))

    # ---------------- backward ----------------
    g = {k: np.zeros_like(v) for k, v in w.items()}
    dflat = np.zeros_like(flat)
    for name, odim, act in heads:
        hd, mean, var, xhat, bn, a, dm, ad, zo, t = cache[name]
        if act == "sigmoid":
            o = out[name]
            oc = np.clip(o, KERAS_EPS, 1 - KERAS_EPS)
            inside = (o > KERAS_EPS) & (o < 1 - KERAS_EPS)
            doc = -(t / (oc + KERAS_EPS) - (1 - t) / (1 - oc + KERAS_EPS)) / (N * odim)
            dzo = doc * inside * o * (1 - o)
        else:
            dzo = 2.0 * (zo - t) / (N * odim)
        dzo = dzo * lw[name]
        g[name + "/out/kernel"] = ad.T @ dzo
        g[name + "/out/bias"] = dzo.sum(axis=0)
        da = (dzo @ w[name + "/out/kernel"].T) * dm
        dbn = da * (bn > 0)
        g[name + "/bn/gamma"] = (dbn * xhat).sum(axis=0)
        g[name + "/bn/beta"] = dbn.sum(axis=0)
        dxhat = dbn * w[name + "/bn/gamma"]
        inv = 1.0 / np.sqrt(var + BN_EPS)
        dhd = inv / N * (N * dxhat - dxhat.sum(axis=0) - xhat * (dxhat * xhat).sum(axis=0))
        g[name + "/dense/kernel"] = flat.T @ dhd + 2 * L2 * w[name + "/dense/kernel"]
        g[name + "/dense/bias"] = dhd.sum(axis=0)
        dflat += dhd @ w[name + "/dense/kernel"].T
    dlog = (p - t3) / N * lw["3C"]
    g["3C/kernel"] = flat.T @ dlog
    g["3C/bias"] = dlog.sum(axis=0)
    dflat += dlog @ w["3C/kernel"].T
    dh = dflat.reshape(trunk.shape) * (trunk_pre > 0)
    for bi in range(len(blocks) - 1, -1, -1):
        p_, d = blocks[bi]
        h_in, u, r, mx, m, yn, mask, ynd = saved[bi]
        dz = dh
        dynd, dk2, db2 = _conv_same_backward(ynd, w[p_ + "/conv1x1/kernel"], 1, dz)
        g[p_ + "/conv1x1/kernel"], g[p_ + "/conv1x1/bias"] = dk2, db2
        dyn = dynd * mask
        # yn = r / m, m = max_c r + eps ; the max gradient is shared equally among tied maxima (tf.reduce_max)
        s1 = (dyn * r).sum(axis=2, keepdims=True)
        is_max = (r == mx)
        share = is_max / is_max.sum(axis=2, keepdims=True)
        dr = dyn / m - share * s1 / (m * m) * (r > 0)  # d|r|/dr = sign(r) (0 at r = 0)
        du = dr * (u > 0)
        dxc, dk1, db1 = _conv_same_backward(h_in, w[p_ + "/conv/kernel"], d, du)
        g[p_ + "/conv/kernel"], g[p_ + "/conv/bias"] = dk1, db1
        dh = dh + dxc
    _, dk0, db0 = _conv_same_backward(x, w["tcn/initial_conv/kernel"], 1, dh)
    g["tcn/initial_conv/kernel"], g["tcn/initial_conv/bias"] = dk0, db0
    bn_batch = {name: (cache[name][1], cache[name][2]) for name, _, _ in heads}
    return dict(loss=float(total), losses=losses, acc=acc, grads=g, bn_batch=bn_batch, outputs=out)


TRAINABLE_SKIP = ("moving_mean", "moving_variance")


def sgd_step(w, grads, velocity, bn_batch, lr, momentum=0.9, clipnorm=1.0):
    """Keras SGD(momentum, clipnorm): every gradient tensor is clipped to norm <= clipnorm on its own;
    v = momentum*v - lr*g ; w += v.  BN moving statistics follow the momentum-0.99 rule."""
    new_w, new_v = {}, {}
    for k, val in w.items():
        val = np.asarray(val, np.float64)
        if k.endswith(TRAINABLE_SKIP):
            head = k.split("/")[0]
            mean, var = bn_batch[head]
            tgt = mean if k.endswith("moving_mean") else var
            new_w[k] = BN_MOMENTUM * val + (1 - BN_MOMENTUM) * tgt
            new_v[k] = np.zeros_like(val)
            continue
        gk = grads[k]
        nrm = np.sqrt(np.sum(gk * gk))
        if clipnorm is not None and nrm > clipnorm:
            gk = gk * (clipnorm / nrm)
        v = momentum * np.asarray(velocity.get(k, 0.0), np.float64) - lr * gk
        new_v[k] = v
        new_w[k] = val + v
    return new_w, new_v


def exponential_decay(step, initial=0.002, decay_steps=1, rate=0.1):
    """tf.keras ExponentialDecay (staircase False): initial * rate ** (step / decay_steps)."""
    return initial * rate ** (step / float(decay_steps))


def adam_step(w, grads, state, lr, names=None, beta_1=0.9, beta_2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam (2.x): m, v moments; w -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps).
    state: dict(t=int, m={}, v={}); names: tensors to update (None = all trainable)."""
    t = state.get("t", 0) + 1
    m, v = dict(state.get("m", {})), dict(state.get("v", {}))
    alpha = lr * np.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
    new_w = {k: np.asarray(val, np.float64) for k, val in w.items()}
    for k in (names if names is not None else [k for k in w if not k.endswith(TRAINABLE_SKIP)]):
        g = np.asarray(grads[k], np.float64)
        m[k] = beta_1 * np.asarray(m.get(k, 0.0)) + (1 - beta_1) * g
        v[k] = beta_2 * np.asarray(v.get(k, 0.0)) + (1 - beta_2) * g * g
        new_w[k] = new_w[k] - alpha * m[k] / (np.sqrt(v[k]) + eps)
    return new_w, dict(t=t, m=m, v=v)


def nadam_step(w, grads, state, lr, names=None, beta_1=0.9, beta_2=0.999, eps=1e-7, schedule_decay=0.004):
    """tf.keras.optimizers.Nadam (2.x) -- the optimiser of the sub-model fine-tuning in
    DAFx12_Speech_Music_Detection_B3_MTL_v2.py:524-526 ("parity unpinned": restated from the published Keras source):
      u_t = b1 (1 - 0.5 * 0.96^(decay t)),  u_{t+1} likewise;  M_t = prod_{i<=t} u_i (a running product kept by the optimiser)
      g' = g / (1 - M_t);  m_t = b1 m + (1-b1) g;  m' = m_t / (1 - M_t u_{t+1});  v_t = b2 v + (1-b2) g^2;  v' = v_t / (1 - b2^t)
      w -= lr * ((1 - u_t) g' + u_{t+1} m') / (sqrt(v') + eps)."""
    t = state.get("t", 0) + 1
    m, v = dict(state.get("m", {})), dict(state.get("v", {}))
    u_t = beta_1 * (1.0 - 0.5 * 0.96 ** (schedule_decay * t))
    u_t1 = beta_1 * (1.0 - 0.5 * 0.96 ** (schedule_decay * (t + 1)))
    ms_new = state.get("m_schedule", 1.0) * u_t
    ms_next = ms_new * u_t1
    new_w = {k: np.asarray(val, np.float64) for k, val in w.items()}
    for k in (names if names is not None else [k for k in w if not k.endswith(TRAINABLE_SKIP)]):
        g = np.asarray(grads[k], np.float64)
        gp = g / (1.0 - ms_new)
        m[k] = beta_1 * np.asarray(m.get(k, 0.0)) + (1 - beta_1) * g
        mp = m[k] / (1.0 - ms_next)
        v[k] = beta_2 * np.asarray(v.get(k, 0.0)) + (1 - beta_2) * g * g
        vp = v[k] / (1.0 - beta_2 ** t)
        new_w[k] = new_w[k] - lr * ((1.0 - u_t) * gp + u_t1 * mp) / (np.sqrt(vp) + eps)
    return new_w, dict(t=t, m=m, v=v, m_schedule=ms_new)
