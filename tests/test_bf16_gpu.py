"""Mixed-precision B3_MTL forward (bf16 matrix-core operands, f32 accumulation / residual stream / normalisation):
BASELINE config 5.  It is NOT the parity path; this file states how far it is from the f32 path and the oracle.
  dtype="bf16"        hi + lo split operands, three bf16 products per f32 product: inside SURVEY 8(d')'s bf16 tolerance
                      (abs 2e-2, argmax agreement >= 99.5 %); measured ~1e-3.
  dtype="bf16_plain"  one bf16 per operand: 3.5-5e-2 through the 24 normalised blocks -- outside; kept for measurement."""
import numpy as np
import pytest
import torch

from oracle import b3_mtl

pytestmark = pytest.mark.gpu

# measured on seeded random weights with perturbed BatchNorm statistics (the hardest case for rounding: 24 blocks of
# 'divide by the channel maximum'): max |output difference| 2-4e-2, argmax of the 3C head identical on > 99.5 %
TOL_ABS = 6e-2          # bf16_plain
TOL_SPLIT = 2e-2        # bf16 (split operands): the SURVEY 8(d') tolerance
MIN_AGREE = 0.995


def _model(ncls, W, seed):
    from sm_hpss_mtl_amd.model import B3MTL
    w = b3_mtl.init_weights(seed=seed, n_feat=240, patch_size=W, n_classes=ncls, randomize_bn=True)
    m = B3MTL(n_feat=240, patch_size=W, n_classes=ncls, seed=0)
    m.set_weights_dict(w)
    return m, w


@pytest.mark.parametrize("ncls,W,N,seed", [(3, 68, 1024, 7), (5, 68, 301, 1), (3, 99, 37, 2)])
def test_bf16_forward_close_to_f32(ncls, W, N, seed):
    m, w = _model(ncls, W, seed)
    x = torch.randn((N, W, 240), device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))
    ref = m.forward_device(x)                      # f32 path (itself within 1e-4 of the oracle)
    small = np.concatenate(b3_mtl.forward(x[:8].cpu().numpy(), w, ncls), axis=1)   # the numpy oracle
    for dtype, tol in (("bf16", TOL_SPLIT), ("bf16_plain", TOL_ABS)):
        got = m.forward_device(x, dtype=dtype)
        torch.cuda.synchronize()
        assert got.shape == ref.shape and torch.isfinite(got).all()
        err = float((got - ref).abs().max())
        agree = float((got[:, -ncls:].argmax(1) == ref[:, -ncls:].argmax(1)).float().mean())
        print("%s vs f32: max abs %.3e, argmax agreement %.4f" % (dtype, err, agree))
        assert err <= tol and agree >= MIN_AGREE, (dtype, err, agree)
        assert np.max(np.abs(got[:8].cpu().numpy() - small)) <= tol


def test_bf16_operands_follow_weight_updates():
    m, w = _model(3, 68, 3)
    x = torch.randn((16, 68, 240), device="cuda")
    a = m.forward_device(x, dtype="bf16").clone()
    w2 = {k: (v * np.float32(0.5) if k.endswith("3C/kernel") else v) for k, v in w.items()}
    m.set_weights_dict(w2)
    b = m.forward_device(x, dtype="bf16")
    ref = m.forward_device(x)
    torch.cuda.synchronize()
    assert not torch.equal(a, b) and float((b - ref).abs().max()) <= TOL_SPLIT
    with pytest.raises(ValueError):
        m.forward_device(x, dtype="fp8")


@pytest.mark.parametrize("ncls,seed", [(3, 5), (5, 6)])
def test_bf16_from_layer0_partials_end_to_end(ncls, seed):
    """BASELINE config 5's shape on the bench fast path: f32 front end -> layer-0 partials (exact f32, feature kernel) ->
    the bf16-operand network (smh_model_forward_x0_bf16), against the all-f32 path of the same clips and the oracle."""
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.pipeline import HotPath
    from sm_hpss_mtl_amd.synth import synth_clips
    m, w = _model(ncls, 68, seed)
    fe = Frontend(FrontendConfig(l_harm=21, l_perc=11))
    B = 64
    audio = torch.from_numpy(synth_clips(B, seed=40 + seed)).cuda()
    ref = HotPath(fe, m, B, 16000, model_dtype="f32", keep_patches=True)
    out32 = ref.step(audio).clone()
    for dtype, tol in (("bf16", TOL_SPLIT),):
        hp = HotPath(fe, m, B, 16000, model_dtype=dtype)
        assert hp.fuse_l0 and hp.patches is None   # starts from the partials: no patch tensor on this path
        got = hp.step(audio)
        torch.cuda.synchronize()
        err = float((got - out32).abs().max())
        agree = float((got[:, -ncls:].argmax(1) == out32[:, -ncls:].argmax(1)).float().mean())
        print("%s from x0 vs f32: max abs %.3e, argmax agreement %.4f" % (dtype, err, agree))
        assert err <= tol and agree >= MIN_AGREE
    small = np.concatenate(b3_mtl.forward(ref.patches[:8].cpu().numpy(), w, ncls), axis=1)
    assert np.max(np.abs(got[:8].cpu().numpy() - small)) <= TOL_SPLIT
