"""tools.scale_data / tools.get_data_statistics on the device against the oracle (tools.pyx:138-215)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(42, 68), (1, 1), (160, 1001), (3, 0)])
def test_scale_data_bit_exact(dt, shape):
    from oracle import tools_stats
    from sm_hpss_mtl_amd.lib.cython_impl import tools
    rng = np.random.default_rng(5)
    FV = (rng.normal(size=shape) * 11 + 2).astype(dt)
    mean = rng.normal(size=shape[0])
    std = np.abs(rng.normal(size=shape[0]))
    std[0] = 0.0  # the 1e-10 guard
    got = tools.scale_data(FV, mean, std)
    ref = tools_stats.scale_data(FV, mean, std)
    assert got.dtype == np.float64 and got.shape == ref.shape
    assert np.array_equal(got, ref)  # IEEE subtract / add / divide: bit-exact


@pytest.mark.parametrize("stat", ["mean", "variance", "skew", "kurtosis"])
@pytest.mark.parametrize("axis", [0, 1])
def test_data_statistics(stat, axis):
    from oracle import tools_stats
    from sm_hpss_mtl_amd.lib.cython_impl import tools
    rng = np.random.default_rng(17)
    FV = (rng.gamma(2.0, size=(37, 42, 68)) - rng.normal(size=(37, 42, 68)) ** 2).astype(np.float32)
    FV[3, :, 5] = 2.5  # constant column (axis 0)
    FV[4, 7, :] = -1.0  # constant row (axis 1)
    got = tools.get_data_statistics(FV, stat_type=stat, axis=axis)
    ref = tools_stats.get_data_statistics(FV, stat, axis)
    assert got.shape == ref.shape == (37, 68 if axis == 0 else 42) and got.dtype == np.float64
    # float64 sums in index order here, pairwise in numpy
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12)
    got4 = tools.get_data_statistics(FV[..., None], stat_type=stat, axis=axis)  # (N, f, t, 1) as the reference squeezes
    assert np.array_equal(got4, got)


def test_tools_errors():
    from sm_hpss_mtl_amd.lib.cython_impl import tools
    with pytest.raises(ValueError):
        tools.get_data_statistics(np.zeros((2, 3, 4)), stat_type="median")
    with pytest.raises(ValueError):
        tools.get_data_statistics(np.zeros((2, 3, 4)), axis=2)
    with pytest.raises(ValueError):
        tools.scale_data(np.zeros((3, 4)), np.zeros(2), np.zeros(3))
    assert tools.get_data_statistics(np.zeros((0, 3, 4)), axis=0).shape == (0, 4)
