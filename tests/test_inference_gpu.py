"""GPU parity of the dense file-level inference: the 501-wide zero-padded median (bit-exact) and
the batched hop-1 patch -> B3_MTL head track against the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import b3_mtl
from oracle import frontend as ofe
from oracle import inference as oinf

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k", [(3000, 501), (300, 501), (1000, 5), (257, 1), (100000, 501), (50, 99)])
def test_medfilt_bit_exact(n, k):
    from sm_hpss_mtl_amd import inference as inf
    rng = np.random.default_rng(n)
    x = rng.random(n).astype(np.float32)
    x[rng.integers(0, n, n // 5)] = x[0]          # ties
    x[rng.integers(0, n, n // 10)] *= -1.0        # generic medfilt: negative values order correctly too
    assert np.array_equal(inf.medfilt(x, k), oinf.medfilt(x, k))
    xb = np.stack([x, x[::-1].copy()])
    yb = inf.medfilt(torch.from_numpy(xb).cuda(), k)
    assert isinstance(yb, torch.Tensor) and np.array_equal(yb.cpu().numpy()[1], oinf.medfilt(xb[1], k))
    with pytest.raises(ValueError):
        inf.medfilt(x, 500)


def test_smooth_labels_signature():
    from sm_hpss_mtl_amd import inference as inf
    p = np.random.default_rng(0).random(2000).astype(np.float32)
    sm, lab = inf.smooth_labels(p, None, 501, smooth_type="prediction")
    ref_sm, ref_lab = oinf.smooth_labels(p, 501)
    assert np.array_equal(sm, ref_sm) and np.array_equal(lab, ref_lab)


@pytest.mark.parametrize("head", ["M", "S"])
def test_patch_probabilities_vs_oracle(head):
    """A 3-second 'file' walked in batches of 120 frames (two full batches, a short one that is tiled)."""
    from sm_hpss_mtl_amd import inference as inf
    from sm_hpss_mtl_amd.model import B3MTL
    from sm_hpss_mtl_amd.synth import synth_clips
    y = synth_clips(1, seed=3, n_samples=48000)[0]
    fv = ofe.featuregram(y, "LogMelHarmPercSpec")
    assert fv.shape == (240, 298)
    w = b3_mtl.init_weights(seed=2, n_feat=240, patch_size=68, n_classes=3, randomize_bn=True)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
    m.set_weights_dict(w)
    got = inf.patch_probabilities(fv, m, 68, 1, output=head, batch_frames=120)
    ref = oinf.patch_probabilities(fv, w, 68, 1, output=head, batch_frames=120)
    assert got.shape == ref.shape and got.shape[0] == 2 * (120 - 68) + len(ofe.patch_starts(116, 68, 1))
    assert np.max(np.abs(got - ref)) <= 1e-4


@pytest.mark.parametrize("W,shift,Tc", [(68, 1, 1000), (68, 3, 517), (99, 1, 400), (99, 7, 99 + 70), (68, 1, 69), (68, 5, 68)])
def test_forward_dense_equals_forward_on_built_patches(W, shift, Tc):
    """smh_model_forward_dense_f32 (layer 0 once per frame, every patch a window of it) against the same patches built by
    extract_patches and run through smh_model_forward_f32: same count (tools.extract_patches' grid, incl. 0 patches at Tc = W for an
    even W), outputs within 2e-5."""
    import torch
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.model import B3MTL
    rng = np.random.default_rng(W + shift + Tc)
    fv = torch.from_numpy(rng.standard_normal((240, Tc)).astype(np.float32)).cuda()
    m = B3MTL(n_feat=240, patch_size=W, n_classes=3, seed=1)
    w = b3_mtl.init_weights(seed=5, n_feat=240, patch_size=W, n_classes=3, randomize_bn=True)
    m.set_weights_dict(w)
    fe = Frontend(FrontendConfig())
    got = m.forward_dense(fv, shift)
    x = fe.extract_patches(fv[None], W, shift, time_major=True)
    assert got.shape == (x.shape[0], m.out_dim) and x.shape[0] == len(ofe.patch_starts(Tc, W, shift))
    if x.shape[0]:
        ref = m.forward_device(x)
        torch.cuda.synchronize()
        assert float((got - ref).abs().max()) <= 2e-5
    with pytest.raises(ValueError):
        m.forward_dense(fv[:, :W - 1].contiguous(), shift)


def test_patch_probabilities_dense_and_patch_paths_agree(monkeypatch):
    from sm_hpss_mtl_amd import inference as inf
    from sm_hpss_mtl_amd.model import B3MTL
    rng = np.random.default_rng(11)
    fv = (rng.standard_normal((240, 2500)) * 10 - 40).astype(np.float32)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=3)
    dense = inf.patch_probabilities(fv, m, 68, 1, output="M", batch_frames=1000)
    monkeypatch.setenv("SMH_DENSE_PATCHES", "1")
    built = inf.patch_probabilities(fv, m, 68, 1, output="M", batch_frames=1000)
    assert dense.shape == built.shape == (2 * (1000 - 68) + (500 - 68),) and np.max(np.abs(dense - built)) <= 2e-5


def test_head_sub_model_as_the_dafx_driver_builds_it():
    """DAFx12...:518-523: Model(trained_model.input, trained_model.get_layer('M').output).predict(x)."""
    from sm_hpss_mtl_amd.lib.proposed_architectures import Model, get_Lemaire_MTL_model
    trained_model, _ = get_Lemaire_MTL_model(100, 240, 3, 99, seed=4)
    x = np.random.default_rng(1).standard_normal((5, 99, 240)).astype(np.float32)
    full = trained_model.predict(x)
    for name in ("M", "S"):
        sub = Model(trained_model.input, trained_model.get_layer(name).output)
        got = sub.predict(x=x)
        assert got.shape == (5, 1) and np.array_equal(got, full[trained_model.output_names.index(name)])
    with pytest.raises(ValueError):
        trained_model.get_layer("dense_7")
    with pytest.raises(TypeError):
        Model(None, trained_model.get_layer("M").output)
