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


def _keras_style_artifacts(w, n_classes, T, F, shuffle_heads=True):
    """A weight file + architecture JSON laid out the way tf.keras 2.x writes them for the reference's B3_MTL graph
    (lib/proposed_architectures.py:85-170) [recollection of the format: root attribute layer_names listing EVERY layer,
    weight_names per layer, datasets at <layer>/<layer>/<weight>:0; Functional config with inbound_nodes].  Hidden layers carry
    Keras' auto-generated names (conv1d_N, dense_N, batch_normalization_N); the heads' hidden layers are deliberately NOT in
    S, M, R order in the file, so only the graph can tell which belongs to which output."""
    from collections import OrderedDict
    heads = [h for h in ("S", "M", "N", "R") if (h + "/dense/kernel") in w]
    layers, cfg = OrderedDict(), []

    def add(cls, name, inbound, weights=None, **conf):
        layers[name] = OrderedDict((name + "/" + k + ":0", v) for k, v in (weights or {}).items())
        cfg.append({"class_name": cls, "name": name, "config": dict(name=name, **conf), "inbound_nodes": [[[i, 0, 0, {}] for i in inbound]] if inbound else []})

    add("InputLayer", "input_1", [], batch_input_shape=[None, T, F])
    add("Conv1D", "tcn_initial_conv", ["input_1"], {"kernel": w["tcn/initial_conv/kernel"], "bias": w["tcn/initial_conv/bias"]}, filters=32)
    prev, n = "tcn_initial_conv", 0
    for s in range(3):
        for i in range(8):
            d, p = 2 ** i, "tcn/s%d_d%d" % (s, 2 ** i)
            dc = "tcn_dilated_conv_%d_tanh_s%d" % (d, s)
            add("Conv1D", dc, [prev], {"kernel": w[p + "/conv/kernel"], "bias": w[p + "/conv/bias"]}, filters=32, dilation_rate=[d])
            add("Activation", "activation_%d" % (2 * n + 1), [dc])
            add("Lambda", "lambda_%d" % (n + 1), ["activation_%d" % (2 * n + 1)])
            add("SpatialDropout1D", "tcn_spatial_dropout1d_%d_s%d_0.250000" % (d, s), ["lambda_%d" % (n + 1)], rate=0.25)
            c1 = "conv1d_%d" % (n + 1)
            add("Conv1D", c1, [cfg[-1]["name"]], {"kernel": w[p + "/conv1x1/kernel"], "bias": w[p + "/conv1x1/bias"]}, filters=32)
            add("Add", "add_%d" % (n + 1), [prev, c1])
            prev, n = "add_%d" % n if False else "add_%d" % (n + 1), n + 1
    add("Activation", "activation_49", [prev])
    add("Flatten", "flatten_1", ["activation_49"])
    # hidden layers of the heads: created S, (dead M block), M, (dead R block), R -> surviving auto-names skip numbers; listed
    # here in a scrambled order on purpose
    order = list(reversed(heads)) if shuffle_heads else heads
    num = {"S": 1, "M": 3, "N": 4, "R": 6}
    for h in order:
        add("Dense", "dense_%d" % num[h], ["flatten_1"], {"kernel": w[h + "/dense/kernel"], "bias": w[h + "/dense/bias"]}, units=16)
    for h in order:
        add("BatchNormalization", "batch_normalization_%d" % num[h], ["dense_%d" % num[h]],
            {"gamma": w[h + "/bn/gamma"], "beta": w[h + "/bn/beta"], "moving_mean": w[h + "/bn/moving_mean"], "moving_variance": w[h + "/bn/moving_variance"]})
        add("Activation", "activation_5%d" % num[h], ["batch_normalization_%d" % num[h]])
        add("Dropout", "dropout_%d" % num[h], ["activation_5%d" % num[h]], rate=0.4)
    for h in heads:
        add("Dense", h, ["dropout_%d" % num[h]], {"kernel": w[h + "/out/kernel"], "bias": w[h + "/out/bias"]}, units=int(w[h + "/out/kernel"].shape[1]))
    add("Dense", "3C", ["flatten_1"], {"kernel": w["3C/kernel"], "bias": w["3C/bias"]}, units=n_classes)
    arch = {"class_name": "Functional", "config": {"name": "model_1", "layers": cfg, "input_layers": [["input_1", 0, 0]],
                                                     "output_layers": [[h, 0, 0] for h in heads + ["3C"]]}}
    return layers, arch


@pytest.mark.parametrize("ncls", [3, 5])
def test_keras_written_weight_file_is_mapped_through_the_architecture_json(tmp_path, ncls):
    """`model_from_json(open(architechtureFile).read()); model.load_weights(weightFile)` on artifacts with
    Keras' own layer names (Proposed_Work_Results.py:370-384).  Host logic only: the mapping and the file format."""
    import json
    from collections import OrderedDict
    from sm_hpss_mtl_amd import h5io, persistence
    if not h5io.available():
        pytest.skip("libhdf5 not found on this machine")
    from oracle import b3_mtl
    w = OrderedDict((k, np.asarray(v, np.float32)) for k, v in b3_mtl.init_weights(seed=4, n_feat=240, patch_size=68, n_classes=ncls, randomize_bn=True).items())
    layers, arch = _keras_style_artifacts(w, ncls, 68, 240)
    wf, af = str(tmp_path / "fold0_model.h5"), str(tmp_path / "fold0_model.json")
    h5io.write_layers(wf, layers)
    json.dump(arch, open(af, "w"))
    read, _ = h5io.read_weights(wf)
    assert list(read) == list(layers) and read["activation_49"] == {} and "tcn_initial_conv/kernel" in read["tcn_initial_conv"]
    got = persistence.load_weights_file(wf)  # finds fold0_model.json next to the weights
    assert list(got) == list(w)              # canonical order
    for k in w:
        assert np.array_equal(got[k], w[k]), k
    # without the JSON the heads' hidden layers are taken in file order: this file scrambles them, so the result must differ
    os.remove(af)
    blind = persistence.load_weights_file(wf)
    assert not np.array_equal(blind["S/dense/kernel"], w["S/dense/kernel"]) and np.array_equal(blind["3C/kernel"], w["3C/kernel"])
    # a file in creation order maps without the JSON
    layers2, _ = _keras_style_artifacts(w, ncls, 68, 240, shuffle_heads=False)
    h5io.write_layers(str(tmp_path / "b.h5"), layers2)
    plain = persistence.load_weights_file(str(tmp_path / "b.h5"))
    assert all(np.array_equal(plain[k], w[k]) for k in w)
    # a foreign file is refused
    layers3 = OrderedDict(layers)
    layers3["conv2d_1"] = OrderedDict([("conv2d_1/kernel:0", np.zeros((3, 3, 1, 8), np.float32))])
    h5io.write_layers(str(tmp_path / "c.h5"), layers3)
    with pytest.raises(ValueError):
        persistence.load_weights_file(str(tmp_path / "c.h5"))
