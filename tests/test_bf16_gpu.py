"""B3_MTL forward on the bf16 matrix pipe (BASELINE config 5 reads "mixed bf16 CNN + fp32 HPSS").  Stated plainly:
  dtype="bf16" is SPLIT-operand arithmetic -- every f32 operand travels as hi + lo (two bf16 values, 16 mantissa bits), every
  product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with f32 accumulation, residual stream and normalisation.  That is
  f32-grade arithmetic carried by bf16 instructions (three of them per f32 product), NOT an 8-bit-mantissa network: it is held
  here to 2e-4 of the f32 kernel (measured 7-8e-5), two orders inside SURVEY 8(d')'s 2e-2.
  A network with ONE bf16 per operand (C ABI smh_model_forward_bf16_ex(split = 0); not offered by the Python surface) lands
  3.5-5e-2 from f32 after 24 blocks of 'divide by the channel maximum' -- outside the 2e-2 tolerance.  The last test states that
  distance so the claim stays measured; it is the reason config 5 is served by the split form."""
import numpy as np
import pytest
import torch

from oracle import b3_mtl

pytestmark = pytest.mark.gpu

# measured on seeded random weights with perturbed BatchNorm statistics (the hardest case for rounding: 24 blocks of
# 'divide by the channel maximum'): max |output difference| 2-4e-2, argmax of the 3C head identical on > 99.5 %
TOL_SPLIT = 1e-4        # "bf16" = split operands (measured 4.9-9.0e-5 in round 4; SURVEY 8(d') would allow 2e-2)
SINGLE_BF16_BAND = (2e-2, 8e-2)   # one bf16 per operand: measured 3.5-5e-2, i.e. OUTSIDE SURVEY 8(d')'s 2e-2
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
    for dtype, tol in (("bf16", TOL_SPLIT),):
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
    for bad in ("fp8", "bf16_plain"):   # the single-bf16 network is not part of the Python surface
        with pytest.raises(ValueError):
            m.forward_device(x, dtype=bad)


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


def test_single_bf16_operands_miss_the_tolerance():
    """Why config 5 is not served by a plain bf16 network: operands rounded to one bf16 (the C ABI's split = 0, reachable for
    measurement only) end 3.5-5e-2 from the f32 kernel on the hardest seeded weights -- outside SURVEY 8(d')'s 2e-2."""
    import ctypes as C
    from sm_hpss_mtl_amd import _lib
    m, _ = _model(3, 68, 7)
    x = torch.randn((1024, 68, 240), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    ref = m.forward_device(x)
    got = torch.empty_like(ref)
    _lib.check(m.lib.smh_model_forward_bf16_ex(m._h, C.c_void_p(x.data_ptr()), 1024, C.c_void_p(got.data_ptr()), 0,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "smh_model_forward_bf16_ex")
    torch.cuda.synchronize()
    err = float((got - ref).abs().max())
    print("single bf16 operands vs f32: max abs %.3e" % err)
    assert SINGLE_BF16_BAND[0] < err < SINGLE_BF16_BAND[1]
