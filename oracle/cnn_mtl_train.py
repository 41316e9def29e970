"""torch (float64, CPU, autograd) restatement of ONE TRAINING STEP of the Doukhan MTL baseline -- TEST INFRASTRUCTURE.

What `model.fit` runs per batch for the model compiled at lib/proposed_architectures.py:448-506:
  graph :453-492 in training mode -- BatchNormalization on batch statistics (eps 1e-3; population variance in the
  normalisation), Dropout 0.2/0.3/0.4/0.5 behind the four Dense blocks, MTL heads of :25-80 (Dropout 0.4, l2(0.01)
  on the Dense(16) kernels), losses S,M: binary_crossentropy, R: mean_squared_error, 3C: categorical_crossentropy,
  optimizer Adam(lr=1e-4) (:499-500; Keras defaults beta_1 0.9, beta_2 0.999, epsilon 1e-7).
Keras / TensorFlow are absent here: layer semantics are restated from their published definitions ("parity
unpinned", like the inference graphs of oracle/cnn_mtl.py); the gradients are torch autograd's.
  * moving statistics: momentum 0.99; Conv2D feature maps (4-D) go through the fused BatchNorm, whose moving variance
    takes the UNBIASED batch variance (M/(M-1)); Dense outputs (2-D) use the population variance.
  * max-pooling gradient goes to the first maximum of each window (torch's rule = TensorFlow's).
  * Keras losses clip probabilities to [1e-7, 1 - 1e-7] (BCE adds another 1e-7 inside the logs).
Random masks are INPUTS (0 or 1/(1-rate)) so that the HIP path can be compared on identical masks.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as Fnn

from .b3_mtl import head_spec
from .cnn_mtl import same_pads

BN_EPS = 1e-3
KERAS_EPS = 1e-7
BN_MOMENTUM = 0.99
L2 = 0.01
FC_DROP = (0.2, 0.3, 0.4, 0.5)


def _bn_train(x, w, p, stats, fused):
    """x (..., C): normalise over every axis but the last with batch statistics; record (mean, var for the moving update)."""
    C = x.shape[-1]
    flat = x.reshape(-1, C)
    M = flat.shape[0]
    mean = flat.mean(dim=0)
    var = ((flat - mean) ** 2).mean(dim=0)
    stats[p] = (mean.detach().numpy().copy(), (var * (M / (M - 1.0) if fused else 1.0)).detach().numpy().copy())
    return (x - mean) / torch.sqrt(var + BN_EPS) * w[p + "/gamma"] + w[p + "/beta"]


def _conv(x, w, p):
    """x NHWC, 'valid', stride 1."""
    k = w[p + "/kernel"].permute(3, 2, 0, 1)  # HWIO -> OIHW
    y = Fnn.conv2d(x.permute(0, 3, 1, 2), k, w[p + "/bias"])
    return y.permute(0, 2, 3, 1)


def _pool(x, pool, same):
    xn = x.permute(0, 3, 1, 2)
    if same:
        (t, b), (l, r) = same_pads(xn.shape[2], pool[0], pool[0]), same_pads(xn.shape[3], pool[1], pool[1])
        xn = Fnn.pad(xn, (l, r, t, b), value=float("-inf"))
    return Fnn.max_pool2d(xn, pool, pool).permute(0, 2, 3, 1)


def forward_backward(x, y, w, n_classes=3, drop=None, drop_heads=None, loss_weights=None, dtype=np.float64):
    """x (N, H, W) images; y: dict name -> targets; w: weights dict (oracle.cnn_mtl.init_doukhan names);
    drop: list of four (N, 512) masks or None; drop_heads: dict head -> (N, 16) or None.
    Returns dict(loss, losses, acc, l2, grads (without the l2 term), bn_batch{name: (mean, var_for_moving)}).
    dtype=np.float32 runs the same graph in single precision: the distance between the two runs is the noise floor
    (ReLU gates and pooling arg-maxima that flip under rounding) a float32 implementation is entitled to."""
    np64 = dtype
    W = {k: torch.tensor(np.asarray(v, np64), requires_grad=not k.endswith(("moving_mean", "moving_variance")))
         for k, v in w.items()}
    xt = torch.tensor(np.asarray(x, np64))[..., None]
    N = xt.shape[0]
    stats = {}
    h = torch.relu(_bn_train(_conv(xt, W, "conv1"), W, "bn1", stats, True))
    h = _pool(h, (2, 2), False)
    h = torch.relu(_bn_train(_conv(h, W, "conv2"), W, "bn2", stats, True))
    h = torch.relu(_bn_train(_conv(h, W, "conv3"), W, "bn3", stats, True))
    h = _pool(h, (2, 2), True)
    h = torch.relu(_bn_train(_conv(h, W, "conv4"), W, "bn4", stats, True))
    h = _pool(h, (1, 12), False)
    h = h.reshape(N, -1)
    for i in range(4):
        p = "fc%d" % (i + 1)
        h = torch.relu(_bn_train(h @ W[p + "/kernel"] + W[p + "/bias"], W, p + "_bn", stats, False))
        if drop is not None:
            h = h * torch.tensor(np.asarray(drop[i], np64))
    feat = h
    heads = head_spec(n_classes)
    lw = {n: 1.0 for n, _, _ in heads}
    lw["3C"] = 1.0
    if loss_weights:
        lw.update(loss_weights)
    losses, total = {}, 0.0
    for name, odim, act in heads:
        hd = _bn_train(feat @ W[name + "/dense/kernel"] + W[name + "/dense/bias"], W, name + "/bn", stats, False)
        a = torch.relu(hd)
        if drop_heads is not None and name in drop_heads:
            a = a * torch.tensor(np.asarray(drop_heads[name], np64))
        zo = a @ W[name + "/out/kernel"] + W[name + "/out/bias"]
        t = torch.tensor(np.asarray(y[name], np64)).reshape(N, odim)
        if act == "sigmoid":
            oc = torch.clamp(torch.sigmoid(zo), KERAS_EPS, 1 - KERAS_EPS)
            l = torch.mean(-(t * torch.log(oc + KERAS_EPS) + (1 - t) * torch.log(1 - oc + KERAS_EPS)))
        else:
            l = torch.mean((zo - t) ** 2)
        losses[name] = l
        total = total + lw[name] * l
    logits = feat @ W["3C/kernel"] + W["3C/bias"]
    p3 = torch.softmax(logits, dim=1)
    t3 = torch.tensor(np.asarray(y["3C"], np64)).reshape(N, n_classes)
    pc = torch.clamp(p3 / p3.sum(dim=1, keepdim=True), KERAS_EPS, 1 - KERAS_EPS)
    losses["3C"] = torch.mean(-torch.sum(t3 * torch.log(pc), dim=1))
    total = total + lw["3C"] * losses["3C"]
    total.backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in W.items()
             if v.requires_grad}
    l2 = float(sum(L2 * float((W[n + "/dense/kernel"].detach() ** 2).sum()) for n, _, _ in heads))
    acc = float((p3.argmax(1) == t3.argmax(1)).double().mean())
    return dict(loss=float(total.detach()), losses={k: float(v.detach()) for k, v in losses.items()}, acc=acc, l2=l2, grads=grads,
                bn_batch=stats, features=feat.detach().numpy())


def adam_step(w, grads, m, v, bn_batch, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7, n_classes=3):
    """Keras Adam on every trainable tensor (l2 term of the head Dense(16) kernels added to the gradient), BN moving
    statistics by the momentum-0.99 rule.  `step` counts from 1.  Returns (new_w, new_m, new_v)."""
    alpha = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    l2_names = {n + "/dense/kernel" for n, _, _ in head_spec(n_classes)}
    nw, nm, nv = {}, {}, {}
    for k, val in w.items():
        val = np.asarray(val, np.float64)
        if k.endswith(("moving_mean", "moving_variance")):
            mean, var = bn_batch[k.rsplit("/", 1)[0]]
            nw[k] = BN_MOMENTUM * val + (1 - BN_MOMENTUM) * (mean if k.endswith("moving_mean") else var)
            continue
        g = grads[k] + (2 * L2 * val if k in l2_names else 0.0)
        nm[k] = beta1 * np.asarray(m.get(k, 0.0)) + (1 - beta1) * g
        nv[k] = beta2 * np.asarray(v.get(k, 0.0)) + (1 - beta2) * g * g
        nw[k] = val - alpha * nm[k] / (np.sqrt(nv[k]) + eps)
    return nw, nm, nv
