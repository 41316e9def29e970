"""Ragged batches and the device generators on the GPU.

* `Frontend.run_ragged` (`smh_frontend_ragged_f32`): clips of different lengths in one call; every clip must get bit for bit
  what `Frontend.run` gives it alone and -- for runs of equal lengths -- inside an equal-length batch; against the oracle within
  the stated tolerances.
* `generators.generator` / `test_file_wise_generator` on synthetic .npy "files": the device path (files of a batch decided from
  their lengths, one ragged pass) against the sequential reference loop (`oracle.generators`) driven by the per-file functions.
"""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import frontend as ofe, generators as ogen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    return Frontend(FrontendConfig())


def test_ragged_equals_per_clip_and_equal_length_batches_bit_for_bit(fe):
    from sm_hpss_mtl_amd.synth import synth_clips
    # 1 s clips (the LDS-image kernels), an odd number of samples, below one patch width with even and odd T (tile-if-short), files
    # beyond the LDS image with even and odd T (tiled medians, streaming feature kernels), a run of three equal clips in the middle
    lens = [16000, 16000, 12345, 30000, 16000, 16000, 16000, 5001, 8000, 160000, 47998, 48160, 26000]
    clips = [synth_clips(1, seed=40 + i, n_samples=n)[0] for i, n in enumerate(lens)]
    res = fe.run_ragged(clips, W=68, shift=34)
    torch.cuda.synchronize()
    assert res["T"] == [ofe.num_frames(n, 400, 160) for n in lens]
    assert {t & 1 for t in res["T"] if t > 161} == {0, 1} and {t & 1 for t in res["T"] if t < 68} == {0, 1}
    for i, c in enumerate(clips):
        one = fe.run(torch.from_numpy(c).cuda()[None], W=68, shift=34)
        assert res["n_patches"][i] == one["n_patches"] == len(ofe.patch_starts(ofe.tile_if_short(np.zeros((1, res["T"][i])), 68).shape[1], 68, 34))
        assert torch.equal(res["fv"][i], one["fv"][0]), ("fv", i, lens[i])
        assert torch.equal(res["patches"][i], one["patches"]), ("patches", i, lens[i])
    batch = fe.run(torch.from_numpy(np.stack(clips[4:7])).cuda(), W=68, shift=34)
    nP = batch["n_patches"]
    for k, i in enumerate((4, 5, 6)):
        assert torch.equal(res["fv"][i], batch["fv"][k]) and torch.equal(res["patches"][i], batch["patches"][k * nP:(k + 1) * nP])
    for i in (2, 7, 8, 3, 10, 11):  # against the oracle: short, tile-if-short and streamed clips
        ref = ofe.featuregram(clips[i], "LogMelHarmPercSpec")
        got = res["fv"][i].cpu().numpy()
        assert np.mean(np.abs(got - ref) <= 1e-3) >= 0.98 and np.max(np.abs(got - ref)) <= 2e-2, (i, np.max(np.abs(got - ref)))
        # patches from the DEVICE's featuregram: what is under test here is the scaler and the patch grid
        pref = ofe.tcn_input(ofe.feature_patches(got.astype(np.float32), 68, 34))
        assert pref.shape == tuple(res["patches"][i].shape), (i, pref.shape)
        assert np.max(np.abs(res["patches"][i].cpu().numpy() - pref)) <= 1e-4, i
    assert fe.run_ragged([], W=68, shift=34)["fv"] == []
    with pytest.raises(ValueError):
        fe.run_ragged([clips[0], clips[0][:300]], W=68, shift=34)  # shorter than n_fft


def _ragged_raw(fe, clips, W, shift, work_bytes=None, patches=True):
    """smh_frontend_ragged_f32 through ctypes with a workspace of the caller's choosing."""
    import ctypes as C
    from sm_hpss_mtl_amd import _lib
    from sm_hpss_mtl_amd.frontend import _ptr, _stream
    B = len(clips)
    lens = [len(c) for c in clips]
    offs, o = [], 0
    for n in lens:
        offs.append(o)
        o += (n + 3) // 4 * 4
    host = np.zeros(o, np.float32)
    for c, n, of in zip(clips, lens, offs):
        host[of:of + n] = c
    audio = torch.from_numpy(host).cuda()
    h_off, h_len = (C.c_longlong * B)(*offs), (C.c_int * B)(*lens)
    fv_off, p_off = (C.c_longlong * (B + 1))(), (C.c_longlong * (B + 1))()
    hT, hnP = (C.c_int * B)(), (C.c_int * B)()
    work = C.c_size_t()
    _lib.check(fe.lib.smh_frontend_ragged_sizes(fe._h, h_off, h_len, B, W if patches else 0, shift if patches else 0, fv_off, p_off, hT, hnP, C.byref(work)))
    fv = torch.full((int(fv_off[B]),), float("nan"), device="cuda")
    pt = torch.full((max(int(p_off[B]), 1), W, 2 * fe.rows), float("nan"), device="cuda") if patches else None
    nbytes = work.value if work_bytes is None else work_bytes
    wk = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    rc = fe.lib.smh_frontend_ragged_f32(fe._h, _ptr(audio), h_off, h_len, B, W if patches else 0, shift if patches else 0, _ptr(fv),
                                        _ptr(pt) if patches else None, _ptr(wk), wk.numel(), _stream())
    return rc, fv, pt, work.value, [int(x) for x in fv_off], [int(x) for x in p_off]


def test_ragged_sub_batches_and_stream_order(fe):
    """A workspace smaller than the batch needs makes the call run in sub-batches (same bits); results consumed on the caller's
    stream right behind the call are complete; a workspace below one clip's need is refused; featuregrams without patches."""
    from sm_hpss_mtl_amd import _lib
    from sm_hpss_mtl_amd.synth import synth_clips
    forced = [n for n in ("SMH_FEAT_TWO_KERNELS", "SMH_MEDIAN_NOSPLIT", "SMH_RAGGED_PERFILE", "SMH_FEAT_TAPS") if os.environ.get(n)]
    if forced:  # these switches send every clip through smh_frontend_f32 alone: no tables, no sub-batches to test
        pytest.skip("ragged kernels switched off by " + ", ".join(forced))
    rng = np.random.default_rng(2)
    lens = [int(rng.integers(6000, 70000)) // 2 * 2 for _ in range(23)] + [16000, 16000]
    clips = [synth_clips(1, seed=300 + i, n_samples=n)[0] for i, n in enumerate(lens)]
    rc, fv, pt, need, fv_off, p_off = _ragged_raw(fe, clips, 68, 34)
    torch.cuda.synchronize()
    assert rc == 0 and not torch.isnan(fv).any() and not torch.isnan(pt[:p_off[-1]]).any()  # every element was written
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # not the default stream, a third of the workspace: sub-batches on the CALLER's stream
        rc2, fv2, pt2, _, _, _ = _ragged_raw(fe, clips, 68, 34, work_bytes=need // 3 // 256 * 256)
        s2 = float(fv2.sum())
    torch.cuda.synchronize()
    assert rc2 == 0 and torch.equal(fv, fv2) and torch.equal(pt, pt2) and s2 == float(fv.sum())
    rc3, *_ = _ragged_raw(fe, clips, 68, 34, work_bytes=4096)
    assert rc3 == _lib.SMH_E_WORKSPACE
    torch.cuda.synchronize()
    rc4, fv4, _, _, _, _ = _ragged_raw(fe, clips, 68, 34, patches=False)
    torch.cuda.synchronize()
    assert rc4 == 0 and torch.equal(fv4, fv)


def test_ragged_other_geometries_and_configurations():
    """Patch geometries of the reference's drivers (W 68 / 99 / 249, incl. clips shorter than a patch: tile-if-short inside the streaming
    kernels) and the other window pair / feature names, each clip against the same clip alone; patches against the oracle's scaler and
    patch grid on the device's featuregram."""
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.synth import synth_clips
    lens = [16000, 30000, 33000, 39998, 60000, 9000, 52345]
    clips = [synth_clips(1, seed=700 + i, n_samples=n)[0] for i, n in enumerate(lens)]
    for cfg, geoms in ((FrontendConfig(), ((99, 34), (249, 24))), (FrontendConfig(l_harm=17, l_perc=17), ((68, 68),)),
                       (FrontendConfig(n_mels=0, log_db=False), ((68, 34),)), (FrontendConfig(log_db=False), ((68, 68),)),
                       (FrontendConfig(n_fft=512, n_mels=0), ((68, 68),))):
        fe = Frontend(cfg)
        for W, shift in geoms:
            res = fe.run_ragged(clips, W=W, shift=shift)
            torch.cuda.synchronize()
            for i, c in enumerate(clips):
                one = fe.run(torch.from_numpy(c).cuda()[None], W=W, shift=shift)
                assert torch.equal(res["fv"][i], one["fv"][0]), (cfg, W, i)
                assert res["n_patches"][i] == one["n_patches"] and torch.equal(res["patches"][i], one["patches"]), (cfg, W, i)
                got = res["fv"][i].cpu().numpy()
                pref = ofe.tcn_input(ofe.feature_patches(got.astype(np.float32), W, shift))
                assert pref.shape == tuple(res["patches"][i].shape), (cfg, W, i, pref.shape)
                if pref.size:
                    assert np.max(np.abs(res["patches"][i].cpu().numpy() - pref)) <= 1e-4, (cfg, W, i)


def _params(tmp, sub):
    m = "Lemaire_et_al_MTL"
    return {"Model": m, "classes": {0: "music", 1: "speech", 2: "speech_music"}, "feature_opDir": str(tmp / sub), "W": 68, "W_shift": 24,
            "n_fft": {m: 400}, "n_mels": {m: 120}, "featName": {m: "LogMelHarmPercSpec"}, "frame_level_scaling": False,
            "skewness_vector": None, "data_augmentation_with_noise": False, "Tw": 25, "Ts": 10,
            "l_harm": {m: 21}, "l_perc": {m: 11}}


def _dataset(tmp):
    from sm_hpss_mtl_amd.synth import synth_clips
    folder = tmp / "data"
    rng = np.random.default_rng(3)
    names = {"speech": [], "music": []}
    for cls, n_files in (("speech", 7), ("music", 6)):
        os.makedirs(folder / cls, exist_ok=True)
        for i in range(n_files):
            n = int(rng.integers(9000, 52000))
            x = synth_clips(1, seed=900 + 31 * i + (0 if cls == "speech" else 500), n_samples=n)[0]
            np.save(folder / cls / ("%s%02d.npy" % (cls[:2], i)), x)
            names[cls].append("%s%02d.npy" % (cls[:2], i))
    mix = [{"speech": names["speech"][i % 7], "music": names["music"][(2 * i) % 6], "SMR": [-5, 0, 5, 10, 15, 20][i % 6]} for i in range(8)]
    return str(folder), {"speech": names["speech"], "music": names["music"], "speech+music": mix}


def test_device_generator_yields_the_batches_of_the_sequential_loop(tmp_path):
    from sm_hpss_mtl_amd import generators as gen
    from sm_hpss_mtl_amd.lib import preprocessing as pp
    folder, files = _dataset(tmp_path)
    Pd, Pr = _params(tmp_path, "feat_dev"), _params(tmp_path, "feat_ref")
    np.random.seed(7)
    dev = gen.generator(Pd, folder, copy.deepcopy(files), 16)
    got = [next(dev) for _ in range(4)]
    np.random.seed(7)
    ref = ogen.reference_generator(Pr, folder, copy.deepcopy(files), 16, pp.get_featuregram, pp.get_feature_patches)
    want = [next(ref) for _ in range(4)]
    for (xb, yb), (xr, yr) in zip(got, want):
        assert isinstance(xb, torch.Tensor) and xb.is_cuda and xb.dtype == torch.float32 and tuple(xb.shape) == (48, 68, 240) == xr.shape
        # the fused ragged pass and the per-file wrapper functions standardise with the same arithmetic in different kernels
        assert np.max(np.abs(xb.cpu().numpy() - xr)) <= 2e-4
        for k in ("R", "S", "M", "3C"):
            np.testing.assert_array_equal(np.asarray(yb[k], np.float64), np.asarray(yr[k], np.float64))
    # both wrote the reference's .npy feature cache; the cached featuregrams agree to the last bit (same kernels)
    for cls in ("speech", "music", "speech_music"):
        a, b = sorted(os.listdir(tmp_path / "feat_dev" / cls)), sorted(os.listdir(tmp_path / "feat_ref" / cls))
        assert a == b and len(a) > 0
        for f in a:
            assert np.array_equal(np.load(tmp_path / "feat_dev" / cls / f), np.load(tmp_path / "feat_ref" / cls / f)), (cls, f)
    # a second generator over the now cached features yields the same first batch (cache hits take the patch-only path)
    np.random.seed(7)
    again = next(gen.generator(Pd, folder, copy.deepcopy(files), 16))
    assert np.max(np.abs(again[0].cpu().numpy() - got[0][0].cpu().numpy())) <= 2e-4
    # ... and feeds model.fit directly
    from sm_hpss_mtl_amd.lib.proposed_architectures import get_Lemaire_MTL_model
    model, _ = get_Lemaire_MTL_model(TR_STEPS=2, N_MELS=240, n_classes=3, patch_size=68, seed=0)
    h = model.fit(gen.generator(Pd, folder, copy.deepcopy(files), 16), steps_per_epoch=2, epochs=1, verbose=0)
    assert np.isfinite(h.history["loss"]).all()


def test_fit_draws_the_next_batch_on_a_side_stream_with_identical_results(tmp_path, monkeypatch):
    """fit(generator) builds batch s + 1 on a side stream while step s trains (training.py: fetch / adopt).  Same generator
    sequence, same seeds, bit-reproducible gradients: the weights after two epochs of three steps equal the one-stream run's."""
    from sm_hpss_mtl_amd import generators as gen
    from sm_hpss_mtl_amd.lib.proposed_architectures import get_Lemaire_MTL_model
    folder, files = _dataset(tmp_path)

    def run(tag):
        P = _params(tmp_path, tag)  # a feature cache of its own: both runs compute every featuregram from the audio
        np.random.seed(11)
        torch.manual_seed(11)
        model, _ = get_Lemaire_MTL_model(TR_STEPS=3, N_MELS=240, n_classes=3, patch_size=68, seed=0)
        model.deterministic_gradients = True
        h = model.fit(gen.generator(P, folder, copy.deepcopy(files), 16), steps_per_epoch=3, epochs=2, verbose=0)
        torch.cuda.synchronize()
        return h.history["loss"], model.get_weights_dict()

    monkeypatch.setenv("SMH_FIT_PREFETCH", "0")
    loss0, w0 = run("feat_pf0")
    monkeypatch.setenv("SMH_FIT_PREFETCH", "1")
    loss1, w1 = run("feat_pf1")
    assert loss0 == loss1
    for k in w0:
        assert np.array_equal(w0[k], w1[k]), k


def test_device_resident_featuregram_cache(tmp_path, monkeypatch):
    """generators._cached_featuregram: a cached .npy featuregram is uploaded once and served from the device afterwards -- same
    batches as with the cache off (SMH_FV_CACHE_GB=0), a rewritten file is read again, the byte budget evicts the oldest entry."""
    from sm_hpss_mtl_amd import generators as gen
    folder, files = _dataset(tmp_path)
    P = _params(tmp_path, "feat_dc")

    def batches(n):
        np.random.seed(5)
        torch.manual_seed(5)
        g = gen.generator(P, folder, copy.deepcopy(files), 16)
        return [next(g) for _ in range(n)]

    monkeypatch.setenv("SMH_FV_CACHE_GB", "0")
    batches(3)  # computes and writes the .npy cache of every file the three batches touch: from here on all runs take the cached path
    gen.fv_cache_clear()
    off = batches(3)
    assert len(gen._FV_CACHE) == 0
    monkeypatch.setenv("SMH_FV_CACHE_GB", "1")
    on1 = batches(3)   # fills the device cache
    n_entries = len(gen._FV_CACHE)
    assert n_entries > 0
    on2 = batches(3)   # served from it
    assert len(gen._FV_CACHE) == n_entries
    for a, b, c in zip(off, on1, on2):
        assert torch.equal(a[0], b[0]) and torch.equal(a[0], c[0])
        assert all(np.array_equal(a[1][k], c[1][k]) for k in a[1])
    # a rewritten file is a new entry (size / mtime are part of the key)
    path = next(iter(gen._FV_CACHE))[0]
    fv = np.load(path)
    np.save(path, (fv * 0.5).astype(np.float32))
    os.utime(path, ns=(os.stat(path).st_atime_ns, os.stat(path).st_mtime_ns + 10_000_000))
    t = gen._cached_featuregram(path)
    assert torch.equal(t.cpu(), torch.from_numpy(fv * 0.5)) and len(gen._FV_CACHE) == n_entries + 1
    # a byte budget of the largest of three featuregrams: the newest file is always resident, the total never exceeds the budget,
    # and what does not fit beside it has been evicted oldest first
    gen.fv_cache_clear()
    paths = sorted(str(q) for q in (tmp_path / "feat_dc").rglob("*.npy"))[:3]
    sizes = [np.load(q).nbytes for q in paths]
    budget = max(sizes)
    monkeypatch.setenv("SMH_FV_CACHE_GB", repr(budget / 2 ** 30))
    for q in paths:
        gen._cached_featuregram(q)
        keys = [k[0] for k in gen._FV_CACHE]
        assert keys[-1] == os.path.abspath(q) and gen._FV_CACHE_BYTES[0] <= budget
        assert keys == [os.path.abspath(r) for r in paths[paths.index(q) + 1 - len(keys):paths.index(q) + 1]]  # a suffix of the files so far
    assert sum(sizes) > budget and len(gen._FV_CACHE) < 3
    gen.fv_cache_clear()


def test_file_wise_generator_device_path(tmp_path):
    from sm_hpss_mtl_amd import generators as gen
    from sm_hpss_mtl_amd.lib import preprocessing as pp
    folder, files = _dataset(tmp_path)
    P = _params(tmp_path, "feat")
    sp, mu = folder + "/speech/" + files["speech"][2], folder + "/music/" + files["music"][1]
    for args, lab in (((sp, "", None), 1), (("", mu, None), 0), ((sp, mu, 10), 2)):
        x, y = gen.test_file_wise_generator(P, *args)
        xr, yr = gen.test_file_wise_generator(P, *args, featuregram_fn=pp.get_featuregram, patches_fn=pp.get_feature_patches)
        assert tuple(x.shape) == xr.shape and x.shape[1:] == (68, 240) and np.array_equal(y, yr) and np.all(y[:, lab] == 1)
        assert np.max(np.abs(x.cpu().numpy() - xr)) <= 2e-4
    assert not os.path.exists(tmp_path / "feat" / "speech")  # save_feat=False (Proposed_Work_Results.py:465-469)
