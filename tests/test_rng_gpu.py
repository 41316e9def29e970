"""csrc/smh_rng.hip through the C ABI: the device generator bit for bit against its numpy restatement (tests/philox_ref.py, itself
pinned to Random123's vectors), the distributions, reproducibility, and the two call sites (batching.noise_augmentation, the
training step's dropout masks)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import philox_ref

pytestmark = pytest.mark.gpu


def _lib():
    from sm_hpss_mtl_amd import _lib
    return _lib, _lib.load()


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("n_a,n_b", [(0, 5), (7, 0), (1001, 333), (510 * 24 * 32, 510 * 3 * 16)])
def test_masks_equal_the_restatement_bit_for_bit(n_a, n_b):
    L, lib = _lib()
    seed, offset = (0x1234 << 32) | 0xABCDEF, (5 << 32) | 17
    out = torch.full((n_a + n_b + 8,), -1.0, device="cuda")
    L.check(lib.smh_dropout_masks_f32(C.c_void_p(out.data_ptr()), n_a, 0.8, n_b, 0.6, seed, offset, _st()))
    got = out.cpu().numpy()
    assert np.array_equal(got[:n_a + n_b], philox_ref.masks(n_a, 0.8, n_b, 0.6, seed, offset))
    assert np.all(got[n_a + n_b:] == -1.0)  # nothing written past the end
    if n_a > 10000:
        keep = (got[:n_a] > 0).mean()
        assert abs(keep - 0.8) < 4 * np.sqrt(0.16 / n_a)


@pytest.mark.parametrize("n", [1, 3, 4, 1027, 510 * 68 * 240])
def test_noise_follows_the_restatement_and_is_normal(n):
    L, lib = _lib()
    rng = np.random.default_rng(n)
    x = torch.from_numpy(rng.standard_normal(n + 4).astype(np.float32)).cuda()
    out = torch.full((n + 4,), 7.0, device="cuda")
    seed, offset, scale = 987654321987, 3, 5e-3
    L.check(lib.smh_noise_augment_f32(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), n, scale, seed, offset, _st()))
    got = out.cpu().numpy()
    assert np.all(got[n:] == 7.0)
    z = (got[:n].astype(np.float64) - x.cpu().numpy()[:n].astype(np.float64)) / scale
    want = philox_ref.normals(n, seed, offset)
    # v_log_f32 / v_sin_f32 / v_cos_f32 against float64, and the float32 rounding of x + scale z (|x| up to ~5: 2.4e-7 / 5e-3 = 1e-4)
    assert np.max(np.abs(z - want)) < 5e-4
    if n > 100000:
        from scipy import stats
        assert abs(z.mean()) < 4 / np.sqrt(n) + 1e-4 and abs(z.std() - 1) < 2e-3
        assert abs(stats.kurtosis(z)) < 0.02 and abs(stats.skew(z)) < 0.01
        assert stats.kstest(z[:200000], "norm").pvalue > 1e-3
        assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 4 / np.sqrt(n)
    # in place gives the same values; another offset another stream; the same call the same values
    y = x.clone()
    L.check(lib.smh_noise_augment_f32(C.c_void_p(y.data_ptr()), C.c_void_p(y.data_ptr()), n, scale, seed, offset, _st()))
    assert torch.equal(y[:n], out[:n])
    L.check(lib.smh_noise_augment_f32(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, scale, seed, offset + 1, _st()))
    assert not torch.equal(y[:n], out[:n]) or n < 2


def test_bad_arguments_are_errors():
    L, lib = _lib()
    x = torch.zeros(64, device="cuda")
    with pytest.raises(ValueError):
        L.check(lib.smh_noise_augment_f32(C.c_void_p(x.data_ptr() + 4), C.c_void_p(x.data_ptr()), 8, 1e-3, 1, 0, _st()))
    with pytest.raises(ValueError):
        L.check(lib.smh_noise_augment_f32(C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 8, -1.0, 1, 0, _st()))
    with pytest.raises(ValueError):
        L.check(lib.smh_dropout_masks_f32(C.c_void_p(x.data_ptr()), 8, 0.0, 8, 0.5, 1, 0, _st()))
    with pytest.raises(ValueError):
        L.check(lib.smh_noise_augment_f32(None, None, 8, 1e-3, 1, 0, _st()))


def test_noise_augmentation_call_site_follows_torch_manual_seed():
    """batching.noise_augmentation on a device batch: scale from the numpy generator (as the reference draws it), noise from the
    HIP kernel seeded through torch's generator -- torch.manual_seed makes it repeatable, and the input is left alone."""
    from sm_hpss_mtl_amd import batching
    x = torch.randn((6, 68, 240), device="cuda")
    x0 = x.clone()
    outs = []
    for _ in range(2):
        torch.manual_seed(3)
        outs.append(batching.noise_augmentation(x, np.random.default_rng(1)))
    assert torch.equal(x, x0) and outs[0].shape == x.shape
    assert torch.equal(outs[0], outs[1])               # same torch seed, same noise
    scale = float(np.random.default_rng(1).choice(batching.NOISE_SCALES))
    assert abs(float((outs[0] - x).std()) / scale - 1) < 0.02
    again = batching.noise_augmentation(x, np.random.default_rng(1))   # torch's generator has moved on: another draw of the same law
    assert not torch.equal(again, outs[0]) and abs(float((again - x).std()) / scale - 1) < 0.02
    with pytest.raises(TypeError):
        batching.noise_augmentation(x.double(), np.random.default_rng(1))


def test_training_step_masks_come_from_the_kernel_and_differ_per_step():
    from sm_hpss_mtl_amd.device_rng import dropout_masks
    from sm_hpss_mtl_amd.model import B3MTL
    m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
    assert not hasattr(m, "_rng") and m._mask_seed == 1234
    x = np.random.default_rng(0).standard_normal((6, 68, 240)).astype(np.float32)
    y = {"S": np.array([0, 1, 0, 1, 0, 1.]), "M": np.array([1, 0, 1, 0, 1, 0.]), "R": np.tile([[1., 0.]], (6, 1)),
         "3C": np.eye(3, dtype=np.float32)[[0, 1, 2, 0, 1, 2]]}
    l0 = m.train_on_batch(x, y)
    assert m._mask_calls == 1 and np.all(np.isfinite(l0))
    m.train_on_batch(x, y)
    assert m._mask_calls == 2
    a = dropout_masks(6 * 24 * 32, 1 - m.dropout_rate, 6 * 3 * 16, 0.6, 1234, 0)
    b = dropout_masks(6 * 24 * 32, 1 - m.dropout_rate, 6 * 3 * 16, 0.6, 1234, 1)
    assert not torch.equal(a, b)
    assert np.array_equal(a.cpu().numpy(), philox_ref.masks(6 * 24 * 32, 1 - m.dropout_rate, 6 * 3 * 16, 0.6, 1234, 0))
