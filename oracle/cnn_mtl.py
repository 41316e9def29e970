"""numpy restatement of the Conv2D MTL baselines -- TEST INFRASTRUCTURE.

  get_Doukhan_MTL_model      lib/proposed_architectures.py:425-511
  get_Papakostas_MTL_model   lib/proposed_architectures.py:516-588
  get_Jang_MTL_model         lib/proposed_architectures.py:650-764  (mel_scale_layer :622-646,
                                                                     get_kernel_initializer :595-618)
  MTL_modifications          lib/proposed_architectures.py:25-80    (oracle/b3_mtl.mtl_heads)

Inference arithmetic only (Dropout = identity, BatchNormalization with moving statistics).  TensorFlow/Keras
are absent here, so the layer semantics are restated from their published definitions ("parity unpinned" for
the graphs); the primitives -- TF 'SAME'/'VALID' geometry, max-pool padding, local response normalisation --
are pinned against torch.nn.functional on the CPU in tests/test_oracle_cnn.py.

Conventions (Keras): images NHWC, Conv2D kernel (kh, kw, Cin, Cout), Dense kernel (in, out); Flatten of
(H, W, C) is row-major; 'same' padding puts the odd extra row/column at the bottom/right
(pad_before = total // 2); MaxPooling2D 'same' pads with -inf; BatchNormalization eps = 1e-3;
tf.nn.local_response_normalization(x, depth_radius=5, alpha=1e-4, beta=0.75) has bias = 1:
    y = x / (1 + alpha * sum_{|c'-c| <= 5} x_{c'}^2) ** beta.
float32 storage, float64 accumulation then one rounding (BLAS-order independent).
"""
from __future__ import annotations

import numpy as np

from . import b3_mtl
from . import frontend as ofe

BN_EPS = 1e-3


# ---- primitives --------------------------------------------------------------------------------------------------
def out_size(n, k, s, padding):
    return -(-n // s) if padding == "same" else (n - k) // s + 1


def same_pads(n, k, s):
    total = max((out_size(n, k, s, "same") - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _pad_hw(x, kh, kw, sh, sw, padding, value=0.0):
    if padding == "valid":
        return x
    (t, b), (l, r) = same_pads(x.shape[1], kh, sh), same_pads(x.shape[2], kw, sw)
    return np.pad(x, ((0, 0), (t, b), (l, r), (0, 0)), constant_values=value)


def conv2d(x, kernel, bias=None, strides=(1, 1), padding="valid"):
    """x (N, H, W, Cin) float32, kernel (kh, kw, Cin, Cout) -> (N, OH, OW, Cout) float32."""
    kh, kw, cin, cout = kernel.shape
    sh, sw = strides
    xp = _pad_hw(np.asarray(x, np.float32), kh, kw, sh, sw, padding)
    N, H, W, _ = xp.shape
    oh, ow = (H - kh) // sh + 1, (W - kw) // sw + 1
    y = np.zeros((N, oh, ow, cout), np.float64)
    k64 = kernel.astype(np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (oh - 1) * sh + 1:sh, j:j + (ow - 1) * sw + 1:sw, :].astype(np.float64)
            y += patch @ k64[i, j]
    if bias is not None:
        y += bias.astype(np.float64)
    return y.astype(np.float32)


def maxpool2d(x, pool, strides, padding="valid"):
    ph, pw = pool
    sh, sw = strides
    xp = _pad_hw(np.asarray(x, np.float32), ph, pw, sh, sw, padding, value=-np.inf)
    N, H, W, C = xp.shape
    oh, ow = (H - ph) // sh + 1, (W - pw) // sw + 1
    y = np.full((N, oh, ow, C), -np.inf, np.float32)
    for i in range(ph):
        for j in range(pw):
            y = np.maximum(y, xp[:, i:i + (oh - 1) * sh + 1:sh, j:j + (ow - 1) * sw + 1:sw, :])
    return y


def lrn(x, depth_radius=5, alpha=1e-4, beta=0.75, bias=1.0):
    x = np.asarray(x, np.float32)
    sq = x.astype(np.float64) ** 2
    C = x.shape[-1]
    cs = np.concatenate([np.zeros(x.shape[:-1] + (1,)), np.cumsum(sq, axis=-1)], axis=-1)
    lo = np.maximum(np.arange(C) - depth_radius, 0)
    hi = np.minimum(np.arange(C) + depth_radius + 1, C)
    ssum = cs[..., hi] - cs[..., lo]
    return (x / (bias + alpha * ssum) ** beta).astype(np.float32)


def batchnorm(x, w, p):
    scale = w[p + "/gamma"] / np.sqrt(w[p + "/moving_variance"] + np.float32(BN_EPS))
    return ((x - w[p + "/moving_mean"]) * scale + w[p + "/beta"]).astype(np.float32)


def relu(x):
    return np.maximum(x, np.float32(0))


def dense(x, w, p):
    return b3_mtl._dense(x, w[p + "/kernel"], w[p + "/bias"])


# ---- weight initialisation (seeded stand-ins for the Keras initialisers; values only need to be generic) --------
def _bn(w, p, C):
    w[p + "/gamma"] = np.ones(C, np.float32)
    w[p + "/beta"] = np.zeros(C, np.float32)
    w[p + "/moving_mean"] = np.zeros(C, np.float32)
    w[p + "/moving_variance"] = np.ones(C, np.float32)


def _conv(w, rng, p, kh, kw, cin, cout, bias=True):
    fan_in, fan_out = kh * kw * cin, kh * kw * cout
    w[p + "/kernel"] = b3_mtl._glorot(rng, (kh, kw, cin, cout), fan_in, fan_out)
    if bias:
        w[p + "/bias"] = np.zeros(cout, np.float32)


def _dense_w(w, rng, p, cin, cout):
    w[p + "/kernel"] = b3_mtl._glorot(rng, (cin, cout), cin, cout)
    w[p + "/bias"] = np.zeros(cout, np.float32)


def _randomize(w, rng):
    for k in w:
        if k.endswith("/bias") or k.endswith("/beta") or k.endswith("moving_mean"):
            w[k] = rng.normal(0, 0.1, size=w[k].shape).astype(np.float32)
        elif k.endswith("/gamma") or k.endswith("moving_variance"):
            w[k] = rng.uniform(0.5, 1.5, size=w[k].shape).astype(np.float32)


# ---- Doukhan et al. (MIREX 2018) MTL --------------------------------------------------------------------------
def doukhan_shapes(H, W):
    """Feature-map sizes after every spatial layer; the last entry is the Flatten width."""
    h, w_ = H - 3, W - 4                                   # conv 4x5 valid
    h, w_ = (h - 2) // 2 + 1, (w_ - 2) // 2 + 1            # pool 2x2 valid
    h, w_ = h - 2, w_ - 2                                  # conv 3x3 valid
    h, w_ = h - 2, w_ - 2                                  # conv 3x3 valid
    h, w_ = -(-h // 2), -(-w_ // 2)                        # pool 2x2 same
    h, w_ = h - 2, w_ - 2                                  # conv 3x3 valid
    w_ = (w_ - 12) // 12 + 1                               # pool 1x12 valid
    return h, w_, h * w_ * 256


def init_doukhan(seed=0, H=240, W=68, n_classes=3, randomize=True):
    rng = np.random.default_rng(seed)
    w = {}
    for i, (kh, kw, cin, cout) in enumerate([(4, 5, 1, 64), (3, 3, 64, 128), (3, 3, 128, 128), (3, 3, 128, 256)]):
        _conv(w, rng, f"conv{i + 1}", kh, kw, cin, cout)
        _bn(w, f"bn{i + 1}", cout)
    D = doukhan_shapes(H, W)[2]
    for i in range(4):
        _dense_w(w, rng, f"fc{i + 1}", D if i == 0 else 512, 512)
        _bn(w, f"fc{i + 1}_bn", 512)
    b3_mtl.init_head_weights(w, rng, 512, n_classes)
    if randomize:
        _randomize(w, rng)
    return w


def forward_doukhan(x, w, n_classes=3, return_features=False):
    """x (N, 2F, W, 1) -> [S, M, R, 3C]   (proposed_architectures.py:448-492)."""
    x = np.asarray(x, np.float32)
    x = relu(batchnorm(conv2d(x, w["conv1/kernel"], w["conv1/bias"]), w, "bn1"))
    x = maxpool2d(x, (2, 2), (2, 2), "valid")
    x = relu(batchnorm(conv2d(x, w["conv2/kernel"], w["conv2/bias"]), w, "bn2"))
    x = relu(batchnorm(conv2d(x, w["conv3/kernel"], w["conv3/bias"]), w, "bn3"))
    x = maxpool2d(x, (2, 2), (2, 2), "same")
    x = relu(batchnorm(conv2d(x, w["conv4/kernel"], w["conv4/bias"]), w, "bn4"))
    x = maxpool2d(x, (1, 12), (1, 12), "valid")
    x = x.reshape(x.shape[0], -1)
    for i in range(4):
        x = relu(batchnorm(dense(x, w, f"fc{i + 1}"), w, f"fc{i + 1}_bn"))
    outs = b3_mtl.mtl_heads(x, w, n_classes)
    return (outs, x) if return_features else outs


# ---- Papakostas & Giannakopoulos (2018) MTL --------------------------------------------------------------------
def papakostas_shapes(H, W):
    h, w_ = (H - 5) // 2 + 1, (W - 5) // 2 + 1             # conv 5x5 s2 valid
    h, w_ = -(-h // 2), -(-w_ // 2)                        # pool 3x3 s2 same
    h, w_ = (h - 3) // 2 + 1, (w_ - 3) // 2 + 1            # conv 3x3 s2 valid
    h, w_ = -(-h // 2), -(-w_ // 2)                        # pool
    h, w_ = -(-h // 2), -(-w_ // 2)                        # conv same, pool
    return h, w_, h * w_ * 512


def init_papakostas(seed=0, H=402, W=68, n_classes=3, randomize=True, fc=4096):
    rng = np.random.default_rng(seed)
    w = {}
    _conv(w, rng, "conv1", 5, 5, 1, 96)
    _conv(w, rng, "conv2", 3, 3, 96, 384)
    _conv(w, rng, "conv3", 3, 3, 384, 512)
    D = papakostas_shapes(H, W)[2]
    _dense_w(w, rng, "fc1", D, fc)
    _bn(w, "fc1_bn", fc)
    _dense_w(w, rng, "fc2", fc, fc)
    _bn(w, "fc2_bn", fc)
    b3_mtl.init_head_weights(w, rng, fc, n_classes)
    if randomize:
        _randomize(w, rng)
    return w


def forward_papakostas(x, w, n_classes=3, return_features=False):
    """x (N, 2K, W, 1) -> [S, M, R, 3C]   (proposed_architectures.py:539-571)."""
    x = np.asarray(x, np.float32)
    x = relu(lrn(conv2d(x, w["conv1/kernel"], w["conv1/bias"], (2, 2), "valid")))
    x = maxpool2d(x, (3, 3), (2, 2), "same")
    x = relu(lrn(conv2d(x, w["conv2/kernel"], w["conv2/bias"], (2, 2), "valid")))
    x = maxpool2d(x, (3, 3), (2, 2), "same")
    x = relu(conv2d(x, w["conv3/kernel"], w["conv3/bias"], (1, 1), "same"))
    x = maxpool2d(x, (3, 3), (2, 2), "same")
    x = x.reshape(x.shape[0], -1)
    x = relu(batchnorm(dense(x, w, "fc1"), w, "fc1_bn"))
    x = relu(batchnorm(dense(x, w, "fc2"), w, "fc2_bn"))
    outs = b3_mtl.mtl_heads(x, w, n_classes)
    return (outs, x) if return_features else outs


# ---- Jang et al. (2019) mel-scale CNN, MTL --------------------------------------------------------------------
def mel_filter_bins(fs=16000, n_fft=512, n_mels=120):
    """proposed_architectures.py:681-691: first/last bin with a positive weight of every Slaney mel filter."""
    M = ofe.mel_basis(fs, n_fft, n_mels)
    bins = np.zeros((n_mels, 2), np.int64)
    for i in range(n_mels):
        nz = np.where(M[i] > 0)[0]
        bins[i] = nz[0], nz[-1]
    return M, bins


def jang_shapes(W, n_mels=120):
    h, w_ = 2 * n_mels, W
    for _ in range(3):
        h, w_ = -(-h // 2), -(-w_ // 2)
    return h, w_, h * w_ * 128


def init_jang(seed=0, W=68, n_classes=3, n_mels=120, t_dim=5, n_fft=512, fs=16000, randomize=True, mel_init=True):
    """Mel-scale kernels start from the mel weights repeated over time and over the 3 output channels
    (get_kernel_initializer, :613-616); they are trainable, so `mel_init=False` draws generic values instead."""
    rng = np.random.default_rng(seed)
    M, bins = mel_filter_bins(fs, n_fft, n_mels)
    w = {}
    for half in ("harm", "perc"):
        for i in range(n_mels):
            kwid = int(bins[i, 1] - bins[i, 0] + 1)
            if mel_init:
                k = np.repeat(M[i, bins[i, 0]:bins[i, 1] + 1][:, None], t_dim, axis=1)[:, :, None, None]
                k = np.repeat(k, 3, axis=3).astype(np.float32)
            else:
                k = rng.normal(0, 0.3, size=(kwid, t_dim, 1, 3)).astype(np.float32)
            w[f"{half}_melCl{i}/kernel"] = k
    for i, (cin, cout) in enumerate([(3, 32), (32, 64), (64, 128)]):
        _conv(w, rng, f"conv{i + 1}", 3, 3, cin, cout)
        _bn(w, f"bn{i + 1}", cout)
    D = jang_shapes(W, n_mels)[2]
    _dense_w(w, rng, "fc1", D, 2048)
    _bn(w, "fc1_bn", 2048)
    _dense_w(w, rng, "fc2", 2048, 1024)
    _bn(w, "fc2_bn", 1024)
    b3_mtl.init_head_weights(w, rng, 1024, n_classes)
    if randomize:
        _randomize(w, rng)
    return w


def mel_scale_layer(x_half, w, name, bins, t_dim=5):
    """proposed_architectures.py:622-646: per mel filter, Cropping2D to its bins, Conv2D(3, (width, t_dim),
    strides=(width, 1), 'same', no bias) -> one output row; rows concatenated; tanh.
    x_half (N, K, W, 1) -> (N, n_mels, W, 3)."""
    rows = []
    for i in range(len(bins)):
        band = x_half[:, bins[i, 0]:bins[i, 1] + 1]
        kwid = int(bins[i, 1] - bins[i, 0] + 1)
        y = conv2d(band, w[f"{name}_melCl{i}/kernel"], None, (kwid, 1), "same")
        assert y.shape[1] == 1
        rows.append(y)
    return np.tanh(np.concatenate(rows, axis=1).astype(np.float64)).astype(np.float32)


def forward_jang(x, w, n_classes=3, n_mels=120, n_fft=512, fs=16000, return_features=False):
    """x (N, 2K, W, 1) with K = n_fft/2 + 1 -> [S, M, R, 3C]   (proposed_architectures.py:695-747)."""
    x = np.asarray(x, np.float32)
    K = n_fft // 2 + 1
    _, bins = mel_filter_bins(fs, n_fft, n_mels)
    mel = np.concatenate([mel_scale_layer(x[:, :K], w, "harm", bins), mel_scale_layer(x[:, K:], w, "perc", bins)],
                         axis=1)
    x = mel
    for i in range(3):
        x = conv2d(x, w[f"conv{i + 1}/kernel"], w[f"conv{i + 1}/bias"], (1, 1), "same")
        x = relu(batchnorm(x, w, f"bn{i + 1}"))
        x = maxpool2d(x, (2, 2), (2, 2), "same")
    x = x.reshape(x.shape[0], -1)
    x = relu(batchnorm(dense(x, w, "fc1"), w, "fc1_bn"))
    x = relu(batchnorm(dense(x, w, "fc2"), w, "fc2_bn"))
    outs = b3_mtl.mtl_heads(x, w, n_classes)
    return (outs, (mel, x)) if return_features else outs


def flops_per_patch(kind, H, W):
    """2 x multiply-adds of the conv + dense layers (the MFMA work) of one patch."""
    f = 0
    if kind == "doukhan":
        h, w_ = H - 3, W - 4
        f += h * w_ * 64 * 20
        h, w_ = (h - 2) // 2 + 1, (w_ - 2) // 2 + 1
        h, w_ = h - 2, w_ - 2
        f += h * w_ * 128 * 9 * 64
        h, w_ = h - 2, w_ - 2
        f += h * w_ * 128 * 9 * 128
        h, w_ = -(-h // 2), -(-w_ // 2)
        h, w_ = h - 2, w_ - 2
        f += h * w_ * 256 * 9 * 128
        f += doukhan_shapes(H, W)[2] * 512 + 3 * 512 * 512 + 512 * 51
    return 2 * f
