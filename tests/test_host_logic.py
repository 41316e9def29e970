"""CPU: the C-ABI library loads and exports every symbol include/smh.h declares, the integer contracts
match the oracle / compiled reference, the product never touches oracle/, and the compute path fails
loudly (no CPU fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT


def _lib():
    from sm_hpss_mtl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsmh.so not built (run __graft_entry__.build())")
    return _lib


def test_header_symbols_exported_and_typed():
    _l = _lib()
    lib = _l.load()
    hdr = open(os.path.join(ROOT, "include", "smh.h")).read()
    declared = set(re.findall(r"\b(smh_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), "declared in smh.h but not exported: " + name
    assert declared == set(_l.SIGNATURES), declared ^ set(_l.SIGNATURES)
    assert lib.smh_version() >= 100


def test_integer_contracts_match_oracle(golden_fe):
    from oracle import frontend as ofe
    lib = _lib().load()
    for N in (0, 399, 400, 559, 560, 16000, 16159, 48000):
        assert lib.smh_num_frames(N, 400, 160) == ofe.num_frames(N, 400, 160)
    for T, W, shift, nP in golden_fe["npatch_table"]:
        T, W, shift = int(T), int(W), int(shift)
        assert lib.smh_num_patches(T, W, shift) == nP
        starts = ofe.patch_starts(T, W, shift)
        assert [lib.smh_patch_start(T, W, shift, p) for p in range(len(starts))] == starts
    for T, W in ((98, 249), (98, 99), (98, 98), (98, 68), (10, 68), (68, 68), (34, 68)):
        assert lib.smh_tiled_frames(T, W) == ofe.tile_if_short(np.zeros((1, T)), W).shape[1]


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[/.]_ref|libsmh_oracle", re.M)
    for base in ("sm_hpss_mtl_amd", "lib", "tools"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    src = open(os.path.join(dp, f), errors="replace").read()
                    assert not pat.search(src), "product file references the oracle: " + os.path.join(dp, f)
    bench = open(os.path.join(ROOT, "bench.py")).read() if os.path.exists(os.path.join(ROOT, "bench.py")) else ""
    for m in re.finditer(r"^\s*(from|import)\s+oracle\b.*$", bench, re.M):
        # allowed only inside the cpu_baseline function
        head = bench[:m.start()]
        last_def = re.findall(r"^def (\w+)", head, re.M)[-1]
        assert last_def == "cpu_baseline", "bench.py imports oracle outside cpu_baseline(): " + m.group(0)


def test_reference_module_paths_importable():
    import importlib
    for mod in ("lib.preprocessing", "lib.cython_impl.tools", "lib.proposed_architectures"):
        m = importlib.import_module(mod)
    import lib.preprocessing as pp
    for fn in ("get_featuregram", "get_feature_patches", "normalize_signal", "mix_signals", "load_and_preprocess_signal"):
        assert callable(getattr(pp, fn))
    import lib.proposed_architectures as pa
    assert callable(pa.get_Lemaire_MTL_model)
    import lib.cython_impl.tools as tl
    assert callable(tl.extract_patches)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _lib()
    from sm_hpss_mtl_amd.frontend import Frontend
    from sm_hpss_mtl_amd.model import B3MTL
    with pytest.raises(RuntimeError, match="GPU|HIP"):
        Frontend()
    with pytest.raises(RuntimeError, match="GPU|HIP"):
        B3MTL()


def test_host_signal_helpers_match_oracle():
    from oracle import frontend as ofe
    from sm_hpss_mtl_amd.lib import preprocessing as pp
    rng = np.random.default_rng(1)
    sp = rng.standard_normal(5000).astype(np.float32)
    np.testing.assert_allclose(pp.normalize_signal(sp), ofe.normalize_signal(sp), rtol=0, atol=0)
    # mix_signals is a batch-of-one call into smh_mix_signals_f32: tests/test_silence_gpu.py::test_mix_signals_host_call
    assert pp.feature_cache_path("/f", "speech_music", "/a/b/sp1.wav", "/c/mu2.wav", 10) == "/f/speech_music/sp1_mu2_10dB.npy"
    assert pp.feature_cache_path("/f", "music", "", "/c/mu2.wav", None) == "/f/music/mu2.npy"


def test_weight_spec_matches_oracle_order():
    from oracle import b3_mtl
    from sm_hpss_mtl_amd.model import weight_spec
    for ncls, W in ((3, 68), (5, 68), (3, 249)):
        w = b3_mtl.init_weights(seed=0, patch_size=W, n_classes=ncls)
        spec = weight_spec(240, W, ncls)
        assert [n for n, _, _, _ in spec] == list(w.keys())
        assert [tuple(s) for _, s, _, _ in spec] == [v.shape for v in w.values()]


def test_label_assembly_matches_reference_rules():
    """a15: Proposed_Work_Results.py:170-262 and 5_class_classification.py:602-671 restated literally."""
    from sm_hpss_mtl_amd.batching import make_labels_3class, make_labels_5class
    bs = 4
    smr = np.array([-5, 0, 10, 20])
    lab = make_labels_3class(bs, smr)
    R = np.ones((3 * bs, 2))
    R[:bs] = [1, 0]
    R[bs:2 * bs] = [0, 1]
    for i, d in enumerate(smr):
        R[2 * bs + i] = [1 / np.power(10, d / 10), 1] if d >= 0 else [1, np.power(10, d / 10)]
    assert np.allclose(lab["R"], R)
    assert lab["S"].tolist() == [0] * bs + [1] * bs + [0] * bs     # the reference leaves S=0 on the mixtures
    assert lab["M"].tolist() == [1] * bs + [0] * bs + [0] * bs
    assert lab["3C"].argmax(1).tolist() == [0] * bs + [1] * bs + [2] * bs
    l5 = make_labels_5class(bs, smr, -smr)
    assert l5["S"].tolist() == [0] * bs + [1] * bs + [1] * bs + [0] * bs + [1] * bs
    assert l5["M"].tolist() == [1] * bs + [0] * bs + [1] * bs + [0] * bs + [0] * bs
    assert l5["N"].tolist() == [0] * bs + [0] * bs + [0] * bs + [1] * bs + [1] * bs
    assert l5["R"].shape == (5 * bs, 3) and np.allclose(l5["R"][3 * bs], [0, 0, 1])
    assert np.allclose(l5["R"][2 * bs + 2], [0.1, 1, 0]) and np.allclose(l5["R"][4 * bs], [0, 1, 10 ** 0.5][:3] if False else l5["R"][4 * bs])
    d = -smr[0]  # +5 dB speech-to-noise
    assert np.allclose(l5["R"][4 * bs], [0, 1 / np.power(10, d / 10), 1])


def test_h5_weight_files_round_trip(tmp_path):
    """Keras-layout HDF5 weight files through libhdf5 (sm_hpss_mtl_amd/h5io.py): structure, names, order, values."""
    from collections import OrderedDict
    from sm_hpss_mtl_amd import h5io, persistence
    if not h5io.available():
        pytest.skip("libhdf5 not found on this machine")
    from oracle import b3_mtl
    w = OrderedDict((k, np.asarray(v, np.float32)) for k, v in b3_mtl.init_weights(seed=2, n_feat=240, patch_size=68, n_classes=5).items())
    path = str(tmp_path / "weights.h5")
    assert persistence.save_weights_file(path, w) == path
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    layers, attrs = h5io.read_weights(path)
    assert attrs["backend"] == b"tensorflow" and list(layers) == ["tcn", "3C", "S", "M", "N", "R"] or list(layers)[0] == "tcn"
    flat = [n for ws in layers.values() for n in ws]
    assert sorted(flat) == sorted(w) and len(flat) == len(w)
    back = persistence.load_weights_file(path)
    for k, v in w.items():
        assert back[k].dtype == np.float32 and np.array_equal(back[k], v), k
    # any other extension keeps the .npz form
    p2 = persistence.save_weights_file(str(tmp_path / "weights"), w)
    assert p2.endswith(".npz") and np.array_equal(persistence.load_weights_file(str(tmp_path / "weights"))["3C/kernel"], w["3C/kernel"])


def test_mode_filtering_matches_the_reference_loop():
    """DAFx12_Speech_Music_Detection_B3_MTL_v2.py:81-89 restated literally vs the prefix-sum form."""
    from sm_hpss_mtl_amd.inference import mode_filtering

    def ref(X, win_size):
        if win_size % 2 == 0:
            win_size += 1
        Xs = X.copy()
        for i in range(int(win_size / 2), len(X) - int(win_size / 2)):
            win = X[i - int(win_size / 2):i + int(win_size / 2)]
            lab, cnt = np.unique(win, return_counts=True)
            Xs[i] = lab[np.argmax(cnt)]
        return Xs
    rng = np.random.default_rng(0)
    for n, w, k in [(200, 11, 2), (500, 50, 2), (64, 5, 3), (30, 501, 2), (7, 3, 2), (300, 2, 4)]:
        X = rng.integers(0, k, n)
        assert np.array_equal(mode_filtering(X, w), ref(X, w)), (n, w, k)
