"""Host logic of the generator counterparts (no GPU): `sm_hpss_mtl_amd.generators.generator` with the per-file callables
injected, batch by batch against `oracle.generators.reference_generator` -- the sequential restatement of
Proposed_Work_Results.py:49-270 -- under the same numpy random state: file order, refills of exhausted lists, the per-pop
reshuffle of the mixture list, carry-over of surplus patches, labels (the S = M = 0 rule of the mixture rows, the SMR ratio
pair), the (N, W, F) transpose and the noise draw.  `test_file_wise_generator`: label, shape and the hard-coded shift 68."""
import copy
import os

import numpy as np
import pytest

from oracle import generators as ogen
from sm_hpss_mtl_amd import generators as gen

F, W = 6, 5


def _params(tmp, noise=True):
    m = "Lemaire_et_al_MTL"
    return {"Model": m, "classes": {0: "music", 1: "speech", 2: "speech_music"}, "feature_opDir": str(tmp / "feat"), "W": W, "W_shift": 2,
            "n_fft": {m: 400}, "n_mels": {m: 120}, "featName": {m: "LogMelHarmPercSpec"}, "frame_level_scaling": False,
            "skewness_vector": None, "data_augmentation_with_noise": noise, "Tw": 25, "Ts": 10}


def _files(tmp):
    """12 speech / 9 music stand-in 'files' (they only have to exist); one listed file is missing on purpose."""
    folder = tmp / "data"
    names = {"speech": ["sp%02d.wav" % i for i in range(12)], "music": ["mu%02d.wav" % i for i in range(9)]}
    for cls, ns in names.items():
        os.makedirs(folder / cls, exist_ok=True)
        for n in ns:
            (folder / cls / n).write_bytes(b"x")
    os.remove(folder / "speech" / "sp03.wav")
    rng = np.random.default_rng(5)
    mix = [{"speech": names["speech"][int(rng.integers(12))], "music": names["music"][int(rng.integers(9))],
            "SMR": int(rng.choice([-5, 0, 5, 10, 15, 20]))} for _ in range(15)]
    return str(folder), {"speech": names["speech"], "music": names["music"], "speech+music": mix}


def _fv(PARAMS, classname, opdir, sp, mu, db, n_fft, n_mels, featName, save_feat=True):
    """Stand-in featuregram: a (F, T) ramp that encodes which file(s) it came from; T differs per file."""
    key = sum(map(ord, os.path.basename(sp) + os.path.basename(mu))) + (0 if db is None else 1000 + int(db))
    T = 5 + key % 9
    return (key + np.arange(F)[:, None] * 0.01 + np.arange(T)[None, :] * 1e-4).astype(np.float32)


def _patches(PARAMS, FV, patch_size, patch_shift, featName):
    starts = range(0, max(FV.shape[1] - patch_size + 1, 0), patch_shift)
    return np.stack([FV[:, s:s + patch_size] for s in starts]).astype(np.float64) if len(starts) else np.zeros((0, FV.shape[0], patch_size))


@pytest.mark.parametrize("noise", [False, True])
def test_generator_matches_the_sequential_reference_loop(tmp_path, noise):
    P = _params(tmp_path, noise)
    folder, files = _files(tmp_path)
    np.random.seed(123)
    ours = gen.generator(P, folder, copy.deepcopy(files), 7, featuregram_fn=_fv, patches_fn=_patches)
    got = [next(ours) for _ in range(9)]  # 9 batches of 7 per class: every list is exhausted and refilled several times
    np.random.seed(123)
    ref = ogen.reference_generator(P, folder, copy.deepcopy(files), 7, _fv, _patches)
    want = [next(ref) for _ in range(9)]
    for (xb, yb), (xr, yr) in zip(got, want):
        assert xb.shape == xr.shape == (21, W, F)
        np.testing.assert_array_equal(xb, xr)
        assert set(yb) == set(yr) == {"R", "S", "M", "3C"}
        for k in yr:
            np.testing.assert_array_equal(np.asarray(yb[k], dtype=np.float64), np.asarray(yr[k], dtype=np.float64))
    # mixture rows: S = M = 0 (Proposed_Work_Results.py:249-260), R is the (weaker, 1) pair
    y = got[0][1]
    assert y["S"][14:].sum() == 0 and y["M"][14:].sum() == 0 and y["S"][7:14].all() and y["M"][:7].all()
    assert np.all(np.max(y["R"][14:], axis=1) == 1.0) and np.all(y["R"][14:] > 0)


def test_generator_rejects_unbuilt_options(tmp_path):
    P = _params(tmp_path)
    folder, files = _files(tmp_path)
    P["frame_level_scaling"] = True
    with pytest.raises(ValueError):
        next(gen.generator(P, folder, files, 4, featuregram_fn=_fv, patches_fn=_patches))


def test_file_wise_generator_labels_and_hard_coded_shift(tmp_path):
    P = _params(tmp_path)
    seen = {}

    def patches(PARAMS, FV, patch_size, patch_shift, featName):
        seen["shift"] = patch_shift
        return _patches(PARAMS, FV, patch_size, 1, featName)

    x, y = gen.test_file_wise_generator(P, "a/sp1.wav", "", None, featuregram_fn=_fv, patches_fn=patches)
    assert seen["shift"] == 68 and P["W_shift"] == 2  # Proposed_Work_Results.py:474 ignores PARAMS['W_shift']
    assert x.shape[1:] == (W, F) and y.shape == (x.shape[0], 3) and np.all(y[:, 1] == 1)
    _, y = gen.test_file_wise_generator(P, "", "a/mu1.wav", None, featuregram_fn=_fv, patches_fn=patches)
    assert np.all(y[:, 0] == 1)
    _, y = gen.test_file_wise_generator(P, "a/sp1.wav", "a/mu1.wav", 10, featuregram_fn=_fv, patches_fn=patches)
    assert np.all(y[:, 2] == 1)
    assert gen.to_categorical([2, 0], 3).tolist() == [[0, 0, 1], [1, 0, 0]]
