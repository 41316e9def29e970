"""CPU pins of the Conv2D-baseline oracle (oracle/cnn_mtl.py): the primitives against torch.nn.functional (a plain
fp32/fp64 reference of the same ops), the graphs against their published shapes / parameter counts."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cnn_mtl as oc


def _t(x):  # NHWC numpy -> NCHW torch float64
    return torch.from_numpy(np.ascontiguousarray(x)).double().permute(0, 3, 1, 2)


def _n(y):  # NCHW torch -> NHWC numpy
    return y.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("H,W,kh,kw,sh,sw,padding", [
    (12, 11, 4, 5, 1, 1, "valid"), (13, 10, 3, 3, 1, 1, "same"), (14, 9, 5, 5, 2, 2, "valid"),
    (10, 12, 3, 3, 2, 2, "same"), (7, 20, 7, 5, 7, 1, "same"), (9, 9, 2, 2, 2, 2, "same")])
def test_conv2d_tf_geometry_vs_torch(H, W, kh, kw, sh, sw, padding):
    rng = np.random.default_rng(H * W)
    x = rng.standard_normal((2, H, W, 3)).astype(np.float32)
    k = rng.standard_normal((kh, kw, 3, 5)).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    y = oc.conv2d(x, k, b, (sh, sw), padding)
    xt = _t(x)
    if padding == "same":  # TF: the odd extra row / column goes to the bottom / right
        (t, bt), (l, r) = oc.same_pads(H, kh, sh), oc.same_pads(W, kw, sw)
        xt = F.pad(xt, (l, r, t, bt))
    ref = F.conv2d(xt, torch.from_numpy(k).double().permute(3, 2, 0, 1), torch.from_numpy(b).double(), stride=(sh, sw))
    assert y.shape == tuple(_n(ref).shape) and y.shape[1] == oc.out_size(H, kh, sh, padding)
    assert np.max(np.abs(y - _n(ref))) <= 1e-5


@pytest.mark.parametrize("H,W,ph,pw,sh,sw,padding", [(11, 9, 2, 2, 2, 2, "valid"), (11, 9, 2, 2, 2, 2, "same"),
                                                      (10, 13, 3, 3, 2, 2, "same"), (5, 28, 1, 12, 1, 12, "valid")])
def test_maxpool_vs_torch(H, W, ph, pw, sh, sw, padding):
    x = np.random.default_rng(1).standard_normal((2, H, W, 4)).astype(np.float32)
    y = oc.maxpool2d(x, (ph, pw), (sh, sw), padding)
    xt = _t(x)
    if padding == "same":
        (t, b), (l, r) = oc.same_pads(H, ph, sh), oc.same_pads(W, pw, sw)
        xt = F.pad(xt, (l, r, t, b), value=float("-inf"))
    ref = F.max_pool2d(xt, (ph, pw), (sh, sw))
    assert np.array_equal(y, _n(ref).astype(np.float32))


def test_lrn_vs_torch():
    x = np.random.default_rng(2).standard_normal((2, 5, 6, 40)).astype(np.float32) * 3
    y = oc.lrn(x)
    # torch divides alpha by the window size n = 2*radius + 1 and uses k for TF's bias
    ref = F.local_response_norm(_t(x), size=11, alpha=1e-4 * 11, beta=0.75, k=1.0)
    assert np.max(np.abs(y - _n(ref))) <= 1e-6


def test_published_shapes_and_parameter_counts():
    assert oc.doukhan_shapes(240, 68) == (55, 1, 14080)          # flatten 14 080
    w = oc.init_doukhan(H=240, W=68)
    n = sum(v.size for v in w.values())
    conv = 4 * 5 * 64 + 64 + 3 * 3 * 64 * 128 + 128 + 3 * 3 * 128 * 128 + 128 + 3 * 3 * 128 * 256 + 256
    bn = 4 * (64 + 128 + 128 + 256) + 4 * 4 * 512
    fc = 14080 * 512 + 512 + 3 * (512 * 512 + 512)
    heads = 512 * 3 + 3 + 3 * (512 * 16 + 16 + 64) + (16 + 1) * 2 + 16 * 2 + 2
    assert n == conv + bn + fc + heads
    assert abs(oc.flops_per_patch("doukhan", 240, 68) - 1.9e9) < 0.1e9  # SURVEY: ~1.9 GFLOP / patch
    assert oc.papakostas_shapes(402, 68) == (13, 2, 13312)
    assert oc.jang_shapes(68) == (30, 9, 34560)
    M, bins = oc.mel_filter_bins(16000, 512, 120)
    assert M.shape == (120, 257) and np.all(bins[:, 1] >= bins[:, 0]) and (bins[:, 1] - bins[:, 0] + 1).max() == 13


def test_graphs_run_and_outputs_are_well_formed():
    rng = np.random.default_rng(3)
    for kind, H, W in (("doukhan", 40, 68), ("papakostas", 66, 40), ("jang", 514, 12)):
        x = rng.standard_normal((2, H, W, 1)).astype(np.float32)
        if kind == "doukhan":
            outs = oc.forward_doukhan(x, oc.init_doukhan(1, H, W))
        elif kind == "papakostas":
            outs = oc.forward_papakostas(x, oc.init_papakostas(1, H, W, fc=64))
        else:
            outs = oc.forward_jang(x, oc.init_jang(1, W, mel_init=False))
        assert [o.shape for o in outs] == [(2, 1), (2, 1), (2, 2), (2, 3)]
        assert np.allclose(outs[3].sum(axis=1), 1, atol=1e-6) and np.all((outs[0] > 0) & (outs[0] < 1))


def test_jang_mel_layer_with_mel_initialiser_is_a_smoothed_mel_projection():
    """With the Constant(mel weights) initialiser the layer computes tanh(sum over 5 frames of the mel band energy)."""
    rng = np.random.default_rng(4)
    x = np.abs(rng.standard_normal((1, 257, 9, 1))).astype(np.float32) * 0.01
    w = oc.init_jang(0, 9, mel_init=True, randomize=False)
    M, bins = oc.mel_filter_bins()
    y = oc.mel_scale_layer(x, w, "harm", bins)
    mel = M.astype(np.float64) @ x[0, :, :, 0].astype(np.float64)           # (120, 9)
    pad = np.pad(mel, ((0, 0), (2, 2)))
    smooth = sum(pad[:, d:d + 9] for d in range(5))
    assert np.max(np.abs(y[0, :, :, 0] - np.tanh(smooth))) <= 1e-6 and np.array_equal(y[..., 0], y[..., 2])
