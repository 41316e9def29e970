"""GPU parity of the Conv2D MTL baselines through the C ABI (smh_cnn_*) against the numpy
oracle on the same seeded weights and inputs.  Floating point on the f32 matrix cores: outputs within 1e-4 absolute
(SURVEY 8d'), feature vectors within 2e-4 of their largest entry, identical argmax of the 3C head."""
import numpy as np
import pytest
import torch

from oracle import cnn_mtl as oc

pytestmark = pytest.mark.gpu


def _model(kind, H, W, n_classes=3, **kw):
    from sm_hpss_mtl_amd.cnn_models import CnnMTL
    return CnnMTL(kind, (H, W, 1), n_classes=n_classes, seed=0, **kw)


def _oracle(kind, H, W, n_classes=3, fc=64, seed=5):
    if kind == "Doukhan":
        return oc.init_doukhan(seed, H, W, n_classes), oc.forward_doukhan
    if kind == "Papakostas":
        return oc.init_papakostas(seed, H, W, n_classes, fc=fc), oc.forward_papakostas
    return oc.init_jang(seed, W, n_classes, mel_init=False), oc.forward_jang


def _check(kind, H, W, N, n_classes=3, fc=64, feat_tol=2e-4):
    w, fwd = _oracle(kind, H, W, n_classes, fc)
    m = _model(kind, H, W, n_classes, fc_width=fc if kind == "Papakostas" else 0)
    assert m.weight_names() == list(w.keys())  # same tensors, same order, same Keras shapes
    assert [s for _, s, _ in m._spec] == [v.shape for v in w.values()]
    m.set_weights_dict(w)
    rng = np.random.default_rng(N + H)
    x = rng.standard_normal((N, H, W, 1)).astype(np.float32)
    ref, feat_ref = fwd(x, w, n_classes, return_features=True)
    if kind == "Jang":
        feat_ref = feat_ref[1]
    feats = torch.empty((N, m.feat_dim), device="cuda")
    out = m.forward_device(torch.from_numpy(x).cuda(), features=feats)
    torch.cuda.synchronize()
    got = [o.cpu().numpy() for o in m.split_outputs(out)]
    f = feats.cpu().numpy()
    assert f.shape == feat_ref.shape
    assert np.max(np.abs(f - feat_ref)) <= feat_tol * max(1.0, np.abs(feat_ref).max())
    for g, r in zip(got, ref):
        assert g.shape == r.shape and np.max(np.abs(g - r)) <= 1e-4
    assert np.array_equal(got[-1].argmax(1), ref[-1].argmax(1))
    return m


@pytest.mark.parametrize("H,W,N", [(40, 68, 3), (240, 68, 2), (64, 80, 5),
                                   (240, 249, 2)])  # the input of Proposed_Work_Results.py (W = 249): flatten 56 320
def test_doukhan_vs_oracle(H, W, N):
    _check("Doukhan", H, W, N)


@pytest.mark.parametrize("H,W,N,fc", [(66, 40, 3, 64), (402, 68, 2, 128), (402, 249, 1, 64)])  # the last: the driver's input
def test_papakostas_vs_oracle(H, W, N, fc):
    _check("Papakostas", H, W, N, fc=fc)


@pytest.mark.parametrize("W,N", [(12, 3), (68, 2), (249, 1)])  # 249: the driver's input (Proposed_Work_Results.py:795)
def test_jang_vs_oracle(W, N):
    _check("Jang", 514, W, N)


def test_five_class_heads_and_chunked_batch():
    """70 images = one full pass of 64 + a ragged one (different split-K plan); 5-class output [S, M, N, R(3), 3C(5)]."""
    m = _check("Doukhan", 30, 68, 70, n_classes=5)
    assert m.out_dim == 11 and m.output_names == ["S", "M", "N", "R", "3C"]


def test_reference_constructors_and_initialisers():
    from sm_hpss_mtl_amd.lib import proposed_architectures as pa
    P = {"Model": "Doukhan_et_al_MTL", "input_shape": {"Doukhan_et_al_MTL": (240, 68, 1), "Jang_et_al_MTL": (514, 20, 1),
                                                       "Papakostas_et_al_MTL": (402, 68, 1)},
         "n_fft": {"Jang_et_al_MTL": 512}}
    model, lr = pa.get_Doukhan_MTL_model(P)
    assert lr == 0.0001 and model.count_params() == sum(v.size for v in oc.init_doukhan(0, 240, 68).values())
    outs = model.predict(np.random.default_rng(0).standard_normal((2, 240, 68, 1)).astype(np.float32))
    assert [o.shape for o in outs] == [(2, 1), (2, 1), (2, 2), (2, 3)] and np.allclose(outs[3].sum(1), 1, atol=1e-5)
    assert model.optimizer.kind == "adam" and model.learning_rate() == 0.0001  # trained in tests/test_cnn_train_gpu.py
    P["Model"] = "Jang_et_al_MTL"
    jang, lr = pa.get_Jang_MTL_model(P)
    assert lr == 0.001
    assert jang.optimizer.kind == "adam" and jang.learning_rate() == 0.001
    ref = oc.init_jang(0, 20, mel_init=True, randomize=False)
    for i in (0, 57, 119):  # Constant(mel weights) initialiser of the mel-scale kernels
        assert np.allclose(jang.weights["harm_melCl%d/kernel" % i], ref["harm_melCl%d/kernel" % i], rtol=1e-6, atol=1e-9)
    P["Model"] = "Papakostas_et_al_MTL"
    pap, lr = pa.get_Papakostas_MTL_model(P)
    assert lr == 0.001 and pap.weights["fc1/kernel"].shape == (13312, 4096) and float(pap.weights["conv1/bias"][0]) == np.float32(0.1)
    assert abs(float(pap.weights["fc2/kernel"].std()) - 0.01) < 1e-3  # RandomNormal(stddev=0.01)


def test_error_behaviour():
    from sm_hpss_mtl_amd.cnn_models import CnnMTL
    with pytest.raises(ValueError):
        CnnMTL("Doukhan", (10, 20, 1))  # too small for the valid convolutions
    with pytest.raises(ValueError):
        CnnMTL("Jang", (500, 68, 1))  # height must be 2 * (n_fft/2 + 1)
    m = _model("Doukhan", 40, 68)
    with pytest.raises(ValueError):
        m.forward_device(torch.zeros((1, 41, 68), device="cuda"))
    with pytest.raises(ValueError):
        m.set_weights(m.get_weights()[:-1])
    assert m.forward_device(torch.zeros((0, 40, 68), device="cuda")).shape == (0, 7)


@pytest.mark.parametrize("kind,H,W,N,fc", [("Doukhan", 240, 68, 70, 0), ("Doukhan", 40, 68, 3, 0), ("Papakostas", 402, 68, 3, 128),
                                           ("Jang", 514, 20, 3, 0)])
def test_bf16_operand_variant(kind, H, W, N, fc):
    """smh_cnn_forward_bf16: bf16 GEMM operands (activations and kernels rounded to 8 mantissa bits), f32 accumulation.
    Not the parity path -- the distance to the f32 path is what operand rounding through 6-9 layers gives: features
    within 3 % of their largest entry, outputs within 5e-2, and the cache follows weight changes."""
    w, _ = _oracle(kind, H, W, 3, fc or 64)
    m = _model(kind, H, W, 3, fc_width=fc)
    m.set_weights_dict(w)
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((N, H, W)).astype(np.float32)).cuda()
    f32f, bff = torch.empty((N, m.feat_dim), device="cuda"), torch.empty((N, m.feat_dim), device="cuda")
    o32 = m.forward_device(x, features=f32f).cpu().numpy()
    o16 = m.forward_device(x, features=bff, dtype="bf16").cpu().numpy()
    fe32, fe16 = f32f.cpu().numpy(), bff.cpu().numpy()
    assert np.isfinite(o16).all()
    ferr = np.abs(fe16 - fe32).max() / max(np.abs(fe32).max(), 1e-6)
    oerr = np.abs(o16 - o32).max()
    print("%s %dx%d: bf16 vs f32 features %.2e of max, outputs %.2e" % (kind, H, W, ferr, oerr))
    assert 0 < ferr < 3e-2 and oerr < 5e-2
    # the operand cache is rebuilt when the weights change
    w2 = {k: (v * 0.5 if k == "conv1/kernel" else v) for k, v in w.items()}
    m.set_weights_dict(w2)
    o16b = m.forward_device(x, dtype="bf16").cpu().numpy()
    o32b = m.forward_device(x).cpu().numpy()
    assert np.abs(o16b - o32b).max() < 5e-2 and np.abs(o32b - o32).max() > 0
    with pytest.raises(ValueError):
        m.forward_device(x, dtype="fp8")
