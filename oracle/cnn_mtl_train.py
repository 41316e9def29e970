"""torch (float64, CPU, autograd) restatement of ONE TRAINING STEP of the Doukhan and Papakostas MTL baselines -- TEST
INFRASTRUCTURE.

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
  * Papakostas (:539-580): Conv2D stride 2 / 'same', tf.nn.local_response_normalization(depth_radius 5, alpha 1e-4,
    beta 0.75, bias 1) + ReLU, 3x3 stride-2 'same' max-pooling (overlapping windows), Dropout 0.5 behind the two Dense
    blocks, optimizer SGD(ExponentialDecay(1e-3, 700, 0.1)) without momentum.
  * Jang (:695-757): 240 trainable mel-scale kernels (tanh), 'same' 3x3 convolutions, Dropout 0.4 on the feature maps
    BEFORE each 2x2 'same' pooling, l2(0.01) on every kernel incl. '3C', optimizer Adam(1e-3).
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


def _conv_s(x, w, p, stride, same):
    k = w[p + "/kernel"].permute(3, 2, 0, 1)
    xn = x.permute(0, 3, 1, 2)
    if same:
        (t, b), (l, r) = same_pads(xn.shape[2], k.shape[2], stride), same_pads(xn.shape[3], k.shape[3], stride)
        xn = Fnn.pad(xn, (l, r, t, b))
    return Fnn.conv2d(xn, k, w[p + "/bias"], stride=stride).permute(0, 2, 3, 1)


def _pool_s(x, k, s, same):
    xn = x.permute(0, 3, 1, 2)
    if same:
        (t, b), (l, r) = same_pads(xn.shape[2], k, s), same_pads(xn.shape[3], k, s)
        xn = Fnn.pad(xn, (l, r, t, b), value=float("-inf"))
    return Fnn.max_pool2d(xn, k, s).permute(0, 2, 3, 1)


def _lrn(x, radius=5, alpha=1e-4, beta=0.75):
    """tf.nn.local_response_normalization over the last axis, bias 1."""
    C = x.shape[-1]
    sq = Fnn.pad(x * x, (radius, radius))
    ssum = sum(sq[..., k:k + C] for k in range(2 * radius + 1))
    return x / (1.0 + alpha * ssum) ** beta


def _trunk_doukhan(xt, W, stats, drop, np64):
    N = xt.shape[0]
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
    return h


def _trunk_papakostas(xt, W, stats, drop, np64):
    """lib/proposed_architectures.py:539-571 in training mode (Dropout 0.5 behind both Dense blocks)."""
    N = xt.shape[0]
    h = torch.relu(_lrn(_conv_s(xt, W, "conv1", 2, False)))
    h = _pool_s(h, 3, 2, True)
    h = torch.relu(_lrn(_conv_s(h, W, "conv2", 2, False)))
    h = _pool_s(h, 3, 2, True)
    h = torch.relu(_conv_s(h, W, "conv3", 1, True))
    h = _pool_s(h, 3, 2, True)
    h = h.reshape(N, -1)
    for i in range(2):
        p = "fc%d" % (i + 1)
        h = torch.relu(_bn_train(h @ W[p + "/kernel"] + W[p + "/bias"], W, p + "_bn", stats, False))
        if drop is not None:
            h = h * torch.tensor(np.asarray(drop[i], np64))
    return h


def _trunk_jang(xt, W, stats, drop, np64, n_mels=120, n_fft=512, fs=16000):
    """lib/proposed_architectures.py:695-745 in training mode: mel-scale layers (:622-646, tanh), three
    Conv-BN-ReLU-Dropout(0.4)-MaxPool blocks, two Dense-BN-ReLU-Dropout(0.4) blocks.  drop: five masks."""
    from .cnn_mtl import mel_filter_bins
    N = xt.shape[0]
    K = n_fft // 2 + 1
    _, bins = mel_filter_bins(fs, n_fft, n_mels)
    xn = xt.permute(0, 3, 1, 2)  # (N, 1, 2K, W)
    rows = []
    for half, top in (("harm", 0), ("perc", K)):
        for i in range(n_mels):
            k = W["%s_melCl%d/kernel" % (half, i)]  # (width, t_dim, 1, 3)
            band = xn[:, :, top + int(bins[i, 0]):top + int(bins[i, 1]) + 1]
            rows.append(Fnn.conv2d(band, k.permute(3, 2, 0, 1), padding=(0, k.shape[1] // 2)))  # (N, 3, 1, W)
    h = torch.tanh(torch.cat(rows, dim=2)).permute(0, 2, 3, 1)  # (N, 2*n_mels, W, 3)
    for i in range(3):
        h = torch.relu(_bn_train(_conv_s(h, W, "conv%d" % (i + 1), 1, True), W, "bn%d" % (i + 1), stats, True))
        if drop is not None:
            h = h * torch.tensor(np.asarray(drop[i], np64)).reshape(h.shape)
        h = _pool_s(h, 2, 2, True)
    h = h.reshape(N, -1)
    for i in range(2):
        p = "fc%d" % (i + 1)
        h = torch.relu(_bn_train(h @ W[p + "/kernel"] + W[p + "/bias"], W, p + "_bn", stats, False))
        if drop is not None:
            h = h * torch.tensor(np.asarray(drop[3 + i], np64))
    return h


def l2_names(kind, w, n_classes=3):
    """Tensors carrying kernel_regularizer=l2(): the heads' Dense(16) kernels; for Jang also every mel-scale, Conv2D
    and Dense kernel and the '3C' kernel (:630, :639, :713-747)."""
    names = {n + "/dense/kernel" for n, _, _ in head_spec(n_classes)}
    if kind == "Jang":
        names |= {k for k in w if k.endswith("/kernel") and (k.startswith(("conv", "fc", "3C")) or "_melCl" in k)}
    return names


def forward_backward(x, y, w, n_classes=3, drop=None, drop_heads=None, loss_weights=None, dtype=np.float64, kind="Doukhan"):
    """x (N, H, W) images; y: dict name -> targets; w: weights dict (oracle.cnn_mtl.init_doukhan / init_papakostas
    names); drop: list of (N, width) masks of the Dense blocks or None; drop_heads: dict head -> (N, 16) or None.
    Returns dict(loss, losses, acc, l2, grads (without the l2 term), bn_batch{name: (mean, var_for_moving)}).
    dtype=np.float32 runs the same graph in single precision: the distance between the two runs is the noise floor
    (ReLU gates and pooling arg-maxima that flip under rounding) a float32 implementation is entitled to."""
    np64 = dtype
    W = {k: torch.tensor(np.asarray(v, np64), requires_grad=not k.endswith(("moving_mean", "moving_variance")))
         for k, v in w.items()}
    xt = torch.tensor(np.asarray(x, np64))[..., None]
    N = xt.shape[0]
    stats = {}
    feat = {"Doukhan": _trunk_doukhan, "Papakostas": _trunk_papakostas, "Jang": _trunk_jang}[kind](xt, W, stats, drop, np64)
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
    l2 = float(sum(L2 * float((W[k].detach() ** 2).sum()) for k in l2_names(kind, w, n_classes)))
    acc = float((p3.argmax(1) == t3.argmax(1)).double().mean())
    return dict(loss=float(total.detach()), losses={k: float(v.detach()) for k, v in losses.items()}, acc=acc, l2=l2, grads=grads,
                bn_batch=stats, features=feat.detach().numpy())


def sgd_step(w, grads, bn_batch, lr, n_classes=3):
    """Keras SGD without momentum (Papakostas, :572-574) + the l2 term of the head kernels + BN moving statistics."""
    l2n = l2_names("Papakostas", w, n_classes)
    nw = {}
    for k, val in w.items():
        val = np.asarray(val, np.float64)
        if k.endswith(("moving_mean", "moving_variance")):
            mean, var = bn_batch[k.rsplit("/", 1)[0]]
            nw[k] = BN_MOMENTUM * val + (1 - BN_MOMENTUM) * (mean if k.endswith("moving_mean") else var)
        else:
            nw[k] = val - lr * (grads[k] + (2 * L2 * val if k in l2n else 0.0))
    return nw


def adam_step(w, grads, m, v, bn_batch, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7, n_classes=3, kind="Doukhan"):
    """Keras Adam on every trainable tensor (l2 term of the head Dense(16) kernels added to the gradient), BN moving
    statistics by the momentum-0.99 rule.  `step` counts from 1.  Returns (new_w, new_m, new_v)."""
    alpha = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    l2n = l2_names(kind, w, n_classes)
    nw, nm, nv = {}, {}, {}
    for k, val in w.items():
        val = np.asarray(val, np.float64)
        if k.endswith(("moving_mean", "moving_variance")):
            mean, var = bn_batch[k.rsplit("/", 1)[0]]
            nw[k] = BN_MOMENTUM * val + (1 - BN_MOMENTUM) * (mean if k.endswith("moving_mean") else var)
            continue
        g = grads[k] + (2 * L2 * val if k in l2n else 0.0)
        nm[k] = beta1 * np.asarray(m.get(k, 0.0)) + (1 - beta1) * g
        nv[k] = beta2 * np.asarray(v.get(k, 0.0)) + (1 - beta2) * g * g
        nw[k] = val - alpha * nm[k] / (np.sqrt(nv[k]) + eps)
    return nw, nm, nv
