"""numpy restatement of B3_MTL = `get_Lemaire_MTL_model` (TEST INFRASTRUCTURE, see oracle/__init__).

Follows lib/proposed_architectures.py:85-170 (graph), :25-80 (MTL heads; the first x_mu / x_smr
blocks at :55-58 / :68-71 are dead -- overwritten at :60 / :73 -- so they are not in the graph) and,
for the 5-class variant, 5_class_classification.py:150-215 (extra head N, R -> Dense(3)) and :220-308.

`tcn.TCN` is third-party (keras-tcn, not vendored, version unpinned).  The positional signature used
at proposed_architectures.py:144 -- (nb_filters, kernel_size, nb_stacks, dilations, activation,
padding, use_skip_connections, dropout_rate, return_sequences) with activation='norm_relu' -- is the
keras-tcn 2.3.x API, whose published algorithm is restated here ("parity unpinned"):

    x = Conv1D(nb_filters, 1, padding)(inputs)
    for s in stacks: for d in dilations:
        y = Conv1D(nb_filters, k, dilation_rate=d, padding)(x)
        y = relu(y);  y = y / (max_over_channels(|y|) + 1e-5)          # 'norm_relu'
        y = SpatialDropout1D(rate)(y)                                  # identity at inference
        y = Conv1D(nb_filters, 1, 'same')(y)
        x = x + y
    x = relu(x)                                                        # use_skip_connections=False

Keras defaults restated: glorot_uniform kernels / zero biases; BatchNormalization eps=1e-3,
momentum=0.99, gamma=1, beta=0, moving_mean=0, moving_var=1; l2() = 0.01.

Weight layout (Keras conventions): Conv1D kernel (k, C_in, C_out); Dense kernel (in, out); the
Flatten of (T, 32) is row-major (t major, channel minor).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-3
NORM_EPS = 1e-5


def head_spec(n_classes: int):
    """[(name, out_dim, activation)] in Keras output order, softmax head last
    (proposed_architectures.py:154 ; 5_class_classification.py:286)."""
    if n_classes == 5:
        return [("S", 1, "sigmoid"), ("M", 1, "sigmoid"), ("N", 1, "sigmoid"), ("R", 3, "linear")]
    return [("S", 1, "sigmoid"), ("M", 1, "sigmoid"), ("R", 2, "linear")]


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_weights(seed: int = 0, n_feat: int = 240, patch_size: int = 68, n_classes: int = 3,
                 nb_filters: int = 32, kernel_size: int = 3, nb_stacks: int = 3, n_dil: int = 8,
                 randomize_bn: bool = False):
    """Ordered dict name -> float32 array.  `randomize_bn` perturbs BN statistics/affine and biases so
    that parity tests exercise every term (a freshly initialised Keras model has trivial BN/bias)."""
    rng = np.random.default_rng(seed)
    C = nb_filters
    w = {}
    w["tcn/initial_conv/kernel"] = _glorot(rng, (1, n_feat, C), n_feat, C)
    w["tcn/initial_conv/bias"] = np.zeros(C, np.float32)
    for s in range(nb_stacks):
        for i in range(n_dil):
            d = 2 ** i
            p = f"tcn/s{s}_d{d}"
            w[p + "/conv/kernel"] = _glorot(rng, (kernel_size, C, C), kernel_size * C, kernel_size * C)
            w[p + "/conv/bias"] = np.zeros(C, np.float32)
            w[p + "/conv1x1/kernel"] = _glorot(rng, (1, C, C), C, C)
            w[p + "/conv1x1/bias"] = np.zeros(C, np.float32)
    D = patch_size * C
    w["3C/kernel"] = _glorot(rng, (D, n_classes), D, n_classes)
    w["3C/bias"] = np.zeros(n_classes, np.float32)
    for name, odim, _ in head_spec(n_classes):
        w[f"{name}/dense/kernel"] = _glorot(rng, (D, 16), D, 16)
        w[f"{name}/dense/bias"] = np.zeros(16, np.float32)
        w[f"{name}/bn/gamma"] = np.ones(16, np.float32)
        w[f"{name}/bn/beta"] = np.zeros(16, np.float32)
        w[f"{name}/bn/moving_mean"] = np.zeros(16, np.float32)
        w[f"{name}/bn/moving_variance"] = np.ones(16, np.float32)
        w[f"{name}/out/kernel"] = _glorot(rng, (16, odim), 16, odim)
        w[f"{name}/out/bias"] = np.zeros(odim, np.float32)
    if randomize_bn:
        for k in w:
            if k.endswith("/bias") or k.endswith("/beta") or k.endswith("moving_mean"):
                w[k] = rng.normal(0, 0.1, size=w[k].shape).astype(np.float32)
            elif k.endswith("/gamma") or k.endswith("moving_variance"):
                w[k] = rng.uniform(0.5, 1.5, size=w[k].shape).astype(np.float32)
    return w


def conv1d_same(x, kernel, bias, dilation=1):
    """Keras Conv1D(padding='same', dilation_rate=d), odd k: y[t] = sum_j x[t+(j-k//2)d] @ W[j] + b.
    x: (N, T, Cin) -> (N, T, Cout).  float32 storage, float64 accumulation."""
    N, T, _ = x.shape
    k = kernel.shape[0]
    y = np.zeros((N, T, kernel.shape[2]), np.float64)
    x64 = x.astype(np.float64)
    for j in range(k):
        off = (j - k // 2) * dilation
        lo, hi = max(0, -off), min(T, T - off)
        if lo < hi:
            y[:, lo:hi] += x64[:, lo + off : hi + off] @ kernel[j].astype(np.float64)
    return (y + bias.astype(np.float64)).astype(np.float32)


def tcn_forward(x, w, nb_stacks=3, n_dil=8, return_blocks=False):
    blocks = []
    x = conv1d_same(x, w["tcn/initial_conv/kernel"], w["tcn/initial_conv/bias"])
    if return_blocks:
        blocks.append(x)
    for s in range(nb_stacks):
        for i in range(n_dil):
            d = 2 ** i
            p = f"tcn/s{s}_d{d}"
            y = conv1d_same(x, w[p + "/conv/kernel"], w[p + "/conv/bias"], d)
            y = np.maximum(y, np.float32(0))
            y = y / (np.max(np.abs(y), axis=2, keepdims=True) + np.float32(NORM_EPS))
            y = conv1d_same(y.astype(np.float32), w[p + "/conv1x1/kernel"], w[p + "/conv1x1/bias"])
            x = (x + y).astype(np.float32)
            if return_blocks:
                blocks.append(x)
    x = np.maximum(x, np.float32(0))
    return (x, blocks) if return_blocks else x


def _dense(x, k, b):
    return (x.astype(np.float64) @ k.astype(np.float64) + b.astype(np.float64)).astype(np.float32)


def init_head_weights(w, rng, D: int, n_classes: int = 3):
    """Append the '3C' classifier and the MTL heads of `MTL_modifications` on a D-wide feature vector."""
    w["3C/kernel"] = _glorot(rng, (D, n_classes), D, n_classes)
    w["3C/bias"] = np.zeros(n_classes, np.float32)
    for name, odim, _ in head_spec(n_classes):
        w[f"{name}/dense/kernel"] = _glorot(rng, (D, 16), D, 16)
        w[f"{name}/dense/bias"] = np.zeros(16, np.float32)
        w[f"{name}/bn/gamma"] = np.ones(16, np.float32)
        w[f"{name}/bn/beta"] = np.zeros(16, np.float32)
        w[f"{name}/bn/moving_mean"] = np.zeros(16, np.float32)
        w[f"{name}/bn/moving_variance"] = np.ones(16, np.float32)
        w[f"{name}/out/kernel"] = _glorot(rng, (16, odim), 16, odim)
        w[f"{name}/out/bias"] = np.zeros(odim, np.float32)


def mtl_heads(flat, w, n_classes: int = 3):
    """lib/proposed_architectures.py:25-80 + the '3C' softmax on the same features: list in Keras output order
    [S, M, (N,) R, 3C]."""
    outs = []
    for name, _, act in head_spec(n_classes):
        h = _dense(flat, w[f"{name}/dense/kernel"], w[f"{name}/dense/bias"])
        h = (h - w[f"{name}/bn/moving_mean"]) / np.sqrt(w[f"{name}/bn/moving_variance"] + np.float32(BN_EPS))
        h = h * w[f"{name}/bn/gamma"] + w[f"{name}/bn/beta"]
        h = np.maximum(h, np.float32(0)).astype(np.float32)
        o = _dense(h, w[f"{name}/out/kernel"], w[f"{name}/out/bias"])
        if act == "sigmoid":
            o = (1.0 / (1.0 + np.exp(-o.astype(np.float64)))).astype(np.float32)
        outs.append(o)
    logits = _dense(flat, w["3C/kernel"], w["3C/bias"]).astype(np.float64)
    logits -= logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    outs.append((e / e.sum(axis=1, keepdims=True)).astype(np.float32))
    return outs


# ---------------------------------------------------------------------------------------------------
# The later keras-tcn residual block (2.8 / 3.x), kept selectable because the reference does not pin keras-tcn:
#   ResidualBlock(dilation d):  y = x
#       for k in (0, 1):  y = Conv1D(nb_filters, kernel_size, dilation_rate=d, padding)(y);  y = relu(y);  SpatialDropout1D
#       shortcut = x if channels match else Conv1D(nb_filters, 1, 'same')(x)          ('matching_conv1D', first block only)
#       out = relu(shortcut + y)
#   TCN: no initial 1x1 convolution, no final activation of its own; with use_skip_connections=False and
#   return_sequences=True the output is the last block's `out`.   [recollection of the published keras-tcn source; "parity
#   unpinned".  NB that release line dropped the 'norm_relu' activation and moved `activation` behind `padding` in the
#   signature, so the reference's positional call (proposed_architectures.py:144) only binds under the 2.3.x API restated
#   above -- block_variant 1 exists for a maintainer who ports that one call, not because the reference can run on it.]
# ---------------------------------------------------------------------------------------------------
def init_weights_v2(seed: int = 0, n_feat: int = 240, patch_size: int = 68, n_classes: int = 3, nb_filters: int = 32,
                    kernel_size: int = 3, nb_stacks: int = 3, n_dil: int = 8, randomize_bn: bool = False):
    rng = np.random.default_rng(seed)
    C = nb_filters
    w = {}
    cin = n_feat
    for s in range(nb_stacks):
        for i in range(n_dil):
            p = f"tcn/s{s}_d{2 ** i}"
            w[p + "/conv0/kernel"] = _glorot(rng, (kernel_size, cin, C), kernel_size * cin, kernel_size * C)
            w[p + "/conv0/bias"] = np.zeros(C, np.float32)
            w[p + "/conv1/kernel"] = _glorot(rng, (kernel_size, C, C), kernel_size * C, kernel_size * C)
            w[p + "/conv1/bias"] = np.zeros(C, np.float32)
            if cin != C:
                w[p + "/matching/kernel"] = _glorot(rng, (1, cin, C), cin, C)
                w[p + "/matching/bias"] = np.zeros(C, np.float32)
            cin = C
    init_head_weights(w, rng, patch_size * C, n_classes)
    if randomize_bn:
        for k in w:
            if k.endswith("/bias") or k.endswith("/beta") or k.endswith("moving_mean"):
                w[k] = rng.normal(0, 0.1, size=w[k].shape).astype(np.float32)
            elif k.endswith("/gamma") or k.endswith("moving_variance"):
                w[k] = rng.uniform(0.5, 1.5, size=w[k].shape).astype(np.float32)
    return w


def tcn_forward_v2(x, w, nb_stacks=3, n_dil=8):
    for s in range(nb_stacks):
        for i in range(n_dil):
            d, p = 2 ** i, f"tcn/s{s}_d{2 ** i}"
            y = np.maximum(conv1d_same(x, w[p + "/conv0/kernel"], w[p + "/conv0/bias"], d), np.float32(0))
            y = np.maximum(conv1d_same(y, w[p + "/conv1/kernel"], w[p + "/conv1/bias"], d), np.float32(0))
            sc = conv1d_same(x, w[p + "/matching/kernel"], w[p + "/matching/bias"]) if (p + "/matching/kernel") in w else x
            x = np.maximum((sc + y).astype(np.float32), np.float32(0))
    return x


def forward(x, w, n_classes: int = 3, return_trunk: bool = False):
    """Inference forward.  x: (N, T, F) float32.  Returns list in Keras output order
    [S, M, (N,) R, 3C] (proposed_architectures.py:154).  The residual-block variant follows the weights: dicts from
    `init_weights_v2` (tensors '.../conv0/kernel') take the two-convolution block."""
    x = np.asarray(x, dtype=np.float32)
    trunk = tcn_forward_v2(x, w) if "tcn/s0_d1/conv0/kernel" in w else tcn_forward(x, w)
    flat = trunk.reshape(trunk.shape[0], -1)  # Flatten: (T, C) row-major
    outs = mtl_heads(flat, w, n_classes)
    if return_trunk:
        return outs, trunk
    return outs


def flops_per_patch(T=68, F=240, C=32, k=3, blocks=24, n_out=51) -> float:
    """2*[T*F*C + blocks*T*(k*C*C + C*C) + T*C*n_out] = 14.64 MFLOP at the defaults."""
    return 2.0 * (T * F * C + blocks * T * (k * C * C + C * C) + T * C * n_out)
