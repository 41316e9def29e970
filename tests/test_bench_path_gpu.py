"""The TIMED configuration under the oracle: exactly `HotPath.step` (what bench.py times) at bench.py's size.

bench.py runs  smh_stft_mag_f32 -> smh_hpss_median_ex_f32(harm_layout = 2: 16-frame blocks) -> smh_features_l0_f32
(features_clip_kernel with the layer-0 partials) -> smh_model_forward_x0_f32  on 1024 clips with 17 x 17 medians.
Every stage of that very sequence is compared with the CPU oracle here, at B = 1024, for the bench's (17, 17) and the
reference's (21, 11) windows, through the persistent and the non-persistent median kernel:
  * medians: bit-exact against the oracle's selection (= scipy.ndimage.median_filter, pinned on CPU) on the device's own S;
  * featuregram: abs 1e-3 dB on EVERY bin against `featuregram_from_S` fed the device's own S (SURVEY 8d');
  * logits: abs 1e-4 + identical argmax against the numpy B3_MTL fed the device's own standardised patches, and the
    full chain from audio against the committed golden (tests/golden/bench_golden.npz) within bench.py's tolerance.
"""
import os

import numpy as np
import pytest
import torch

from oracle import b3_mtl, frontend as ofe

pytestmark = pytest.mark.gpu


def _skip_if_forced(*names):
    """Tests that assert WHICH implementation the bench path took are skipped, not failed, when an environment switch forces another
    one (tools/gpu/r*_variants.sh)."""
    import os
    forced = [n for n in names if os.environ.get(n)]
    if forced:
        pytest.skip("implementation forced by " + ", ".join(forced))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 1024
SPREAD = (0, 1, 63, 64, 511, 700, 1022, 1023)  # clips spread over the batch (and over the workgroups of every kernel)


def _hot_path(lh, lp, keep_patches=False, seed=0, **kw):
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.model import B3MTL
    from sm_hpss_mtl_amd.pipeline import HotPath
    fe = Frontend(FrontendConfig(l_harm=lh, l_perc=lp))
    model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=seed)
    return HotPath(fe, model, B, 16000, patch=68, keep_patches=keep_patches, **kw), model


def _bench_audio(rank=0):
    from sm_hpss_mtl_amd.synth import bench_clips
    base = bench_clips(B, rank)  # bench.py's batch of rank `rank`: 1024 DISTINCT clips
    return base, torch.from_numpy(base).cuda()


@pytest.mark.parametrize("lh,lp,persist", [(17, 17, None), (17, 17, "1"), (17, 17, "0"), (21, 11, None), (21, 11, "0"),
                                            (21, 11, "1")])
def test_timed_median_variant_is_bit_exact_at_bench_size(lh, lp, persist, monkeypatch):
    """`hpss_median_split_kernel<lh, lp>` with the 16-frame-blocked harmonic output (harm_layout = 2), the instantiation
    the headline number is measured on: every one of the 1024 clips against the oracle, bit for bit."""
    if persist is not None:
        monkeypatch.setenv("SMH_MEDIAN_PERSIST", persist)
    _skip_if_forced("SMH_FEAT_TWO_KERNELS", "SMH_FEAT_TAPS", "SMH_MEDIAN_NOSPLIT")
    hp, _ = _hot_path(lh, lp)
    base, audio = _bench_audio()
    hp.step(audio)
    torch.cuda.synchronize()
    assert hp.want_layout == 2 and hp.layout == 2, "bench configuration must take the blocked harmonic layout"
    S = hp.S.cpu().numpy()
    ref_h = torch.from_numpy(np.stack([ofe.median_time(S[i], lh) for i in range(B)])).cuda()
    ref_p = torch.from_numpy(np.stack([ofe.median_freq(S[i], lp) for i in range(B)])).cuda()
    assert torch.equal(hp.harm_bkt(), ref_h), "harmonic medians (blocked layout) differ from the oracle"
    assert torch.equal(hp.perc, ref_p), "percussive medians differ from the oracle"


@pytest.mark.parametrize("lh,lp", [(17, 17), (21, 11)])
def test_timed_feature_kernel_vs_oracle_from_device_S(lh, lp):
    """features_clip_kernel as bench.py launches it (blocked harm, layer-0 partials, no patches): the featuregram it
    writes against the oracle started from the device's OWN S -- abs 1e-3 dB on 100 % of the bins."""
    _skip_if_forced("SMH_FEAT_TWO_KERNELS", "SMH_FEAT_TAPS", "SMH_MEDIAN_NOSPLIT")
    hp, _ = _hot_path(lh, lp)
    base, audio = _bench_audio()
    hp.step(audio)
    torch.cuda.synchronize()
    assert hp.layout == 2 and hp.patches is None
    S, fv = hp.S.cpu().numpy(), hp.fv.cpu().numpy()
    for i in SPREAD:
        ref = ofe.featuregram_from_S(S[i], "LogMelHarmPercSpec", l_harm=lh, l_perc=lp)
        assert fv[i].shape == ref.shape == (240, 98)
        assert np.max(np.abs(fv[i] - ref)) <= 1e-3, (i, float(np.max(np.abs(fv[i] - ref))))
        for half in (slice(0, 120), slice(120, 240)):  # per-array top-dB floor (lib/preprocessing.py:420,422)
            assert abs(fv[i][half].max() - ref[half].max()) <= 1e-3
            assert fv[i][half].min() >= fv[i][half].max() - 80.0 - 1e-3


@pytest.mark.parametrize("lh,lp", [(17, 17), (21, 11)])
def test_timed_sequence_logits_vs_oracle(lh, lp):
    """Logits of the exact bench sequence.  (a) bit-identical to the same sequence with the patch tap on; (b) abs 1e-4 +
    same argmax against the numpy B3_MTL fed those patches; (c) standardised patches abs 1e-4 against the oracle's
    StandardScaler + extract_patches fed the device's own featuregram; (d) the full chain from audio against the
    committed golden within bench.py's tolerance."""
    import bench
    hp, model = _hot_path(lh, lp)
    hpt, model_t = _hot_path(lh, lp, keep_patches=True)
    base, audio = _bench_audio()
    got = hp.step(audio).clone()
    got_t = hpt.step(audio)
    torch.cuda.synchronize()
    assert hp.patches is None and hpt.patches is not None
    assert torch.equal(got, got_t) and torch.equal(hp.fv, hpt.fv)
    w = model.get_weights_dict()
    idx = list(SPREAD)
    patches = hpt.patches.cpu().numpy()[idx]
    fv = hp.fv.cpu().numpy()
    for j, i in enumerate(idx):
        refp = ofe.tcn_input(ofe.feature_patches(fv[i], 68, 68))
        assert np.max(np.abs(patches[j] - refp[0])) <= 1e-4
    ref = np.concatenate(b3_mtl.forward(patches, w), axis=1)
    out = got.cpu().numpy()
    assert np.max(np.abs(out[idx] - ref)) <= 1e-4
    assert np.array_equal(out[idx][:, -3:].argmax(1), ref[:, -3:].argmax(1))
    g = np.load(os.path.join(ROOT, "tests", "golden", "bench_golden.npz"))
    gold = g["logits_%dx%d" % (lh, lp)][0]
    n = gold.shape[0]
    assert np.max(np.abs(out[:n] - gold)) <= bench.GOLDEN_LOGIT_TOL
    assert np.array_equal(out[:n, -3:].argmax(1), gold[:, -3:].argmax(1))
    tail, r0 = g["logits_tail_%dx%d" % (lh, lp)][0], int(g["tail_row"])   # rows 64, 65: the second part of the distinct batch
    assert np.max(np.abs(out[r0:r0 + tail.shape[0]] - tail)) <= bench.GOLDEN_LOGIT_TOL
    # a clip gives the same bits wherever it sits in the batch (moved to the other end, among different neighbours)
    rolled = hp.step(torch.roll(audio, 5, dims=0))
    assert torch.equal(torch.roll(got, 5, dims=0), rolled)


def test_other_ranks_clips_match_their_golden():
    """bench.py --gpus N gives rank r the clips of seed 1000 + r: rank 3's first clips against the golden."""
    import bench
    hp, _ = _hot_path(17, 17)
    _, audio = _bench_audio(rank=3)
    out = hp.step(audio).cpu().numpy()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "bench_golden.npz"))["logits_17x17"][3]
    assert np.max(np.abs(out[:gold.shape[0]] - gold)) <= bench.GOLDEN_LOGIT_TOL


def test_feature_kernel_residency():
    """The feature kernel of the bench path keeps THREE workgroups per CU (78 VGPRs, 49 KB of LDS: DESIGN 4.3); one more
    register class or an LDS copy of the layer-0 weights and it drops to two, 113-118 us -> ~130 us, without any test noticing."""
    import ctypes as C
    from sm_hpss_mtl_amd import _lib
    lib = _lib.require_gpu()
    f = lib.smh_internal_feat_residency
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int]
    assert f(120, 98) == 3

