"""Counterpart of the reference's only native module, lib/cython_impl/tools.pyx.

`extract_patches` (tools.pyx:21-38) is on the hot path and runs as a HIP gather through the C ABI
(`smh_extract_patches_f32`); it returns float64 (nP, F, W) exactly like the Cython function.
`removeSilence` (tools.pyx:42-134) runs through `smh_remove_silence_f32`; `scale_data` /
`get_data_statistics` (:138-215; off by default in the reference: frame_level_scaling False, skewness_vector None)
are float64 device kernels (`smh_scale_data_f64`, `smh_data_statistics_f64`).
"""
from __future__ import annotations

import numpy as np
import torch

from ... import frontend as _fe

_ctx = None


def _frontend():
    global _ctx
    if _ctx is None:
        _ctx = _fe.Frontend(_fe.FrontendConfig())
    return _ctx


def extract_patches(FV, shape, patch_size, patch_shift):
    """FV (F, T) -> float64 (nP, F, W); `shape` is np.shape(FV) as in the reference call sites."""
    FV = np.asarray(FV)
    if tuple(shape) != FV.shape or FV.ndim != 2:
        raise ValueError("extract_patches: shape %s does not match FV %s" % (tuple(shape), FV.shape))
    fe = _frontend()
    F, T = FV.shape
    # tools.pyx works on the array as given (tiling is the caller's job): no modulo wrap here
    half = int(patch_size / 2)
    nP = len(range(half, T - half, patch_shift))
    if nP == 0:
        return np.zeros((0, F, patch_size))
    d = torch.from_numpy(np.ascontiguousarray(FV, dtype=np.float32)).cuda()[None]
    out = fe.extract_patches(d, patch_size, patch_shift, time_major=False)
    return out.cpu().numpy().astype(np.float64)


def removeSilence(Xin, nSamples, energy, nFrames, fs, Tw, Ts, alpha=0.025, beta=0.075):
    """tools.pyx:42-134 -> (Xin_silrem, sample_silMarker, frame_silMarker, totalSilDuration) with the reference's
    dtypes (float32 / int64 / int64 / int) and its quirks: nothing is removed unless two runs qualify, the output
    keeps the input length with a tail of 1.0, and with fewer than two runs `Xin` itself is returned."""
    from ... import silence as _sil
    Xin_np = np.asarray(Xin)
    energy_np = np.asarray(energy)
    if Xin_np.ndim != 1 or len(Xin_np) != nSamples:
        raise ValueError("removeSilence: Xin has shape %s, nSamples=%r" % (Xin_np.shape, nSamples))
    if energy_np.ndim != 1 or len(energy_np) != nFrames:
        raise ValueError("removeSilence: energy has shape %s, nFrames=%r" % (energy_np.shape, nFrames))
    dx = torch.from_numpy(np.ascontiguousarray(Xin_np, dtype=np.float32)).cuda()
    de = torch.from_numpy(np.ascontiguousarray(energy_np, dtype=np.float32)).cuda()
    out, n_keep, sm, fm = _sil.remove_silence(dx, de, fs, Tw, Ts, alpha, beta, markers=True)
    sample_silMarker = sm[0].cpu().numpy().astype(np.int64)
    frame_silMarker = fm[0].cpu().numpy().astype(np.int64)
    # the removed runs are the zero stretches of the sample marker (consecutive runs never touch)
    d = np.diff(np.concatenate([[1], sample_silMarker, [1]]))
    starts, ends = np.where(d == -1)[0], np.where(d == 1)[0]
    totalSilDuration = 0
    for k, l in zip(starts, ends):
        totalSilDuration = int(totalSilDuration + (l - k) / fs)  # `cdef int` accumulator, tools.pyx:85,121
    if len(starts) > 1:
        return out[0].cpu().numpy(), sample_silMarker, frame_silMarker, totalSilDuration
    return Xin, sample_silMarker, frame_silMarker, totalSilDuration


def _stream():
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def scale_data(FV, mean, stdev):
    """tools.pyx:138-165 -> float64 (F, T): (FV - mean[f]) / (stdev[f] + 1e-10)."""
    import ctypes as C
    from ... import _lib
    lib = _lib.require_gpu()
    FV = np.asarray(FV)
    if FV.ndim != 2:
        raise ValueError("scale_data: FV should be (num_features, num_frames), got %s" % (FV.shape,))
    F, T = FV.shape
    mean, stdev = np.asarray(mean, np.float64).ravel(), np.asarray(stdev, np.float64).ravel()
    if mean.size != F or stdev.size != F:
        raise ValueError("scale_data: mean / stdev need one value per feature row (%d), got %d / %d" % (F, mean.size, stdev.size))
    d = torch.from_numpy(np.ascontiguousarray(FV, dtype=np.float64)).cuda()
    dm, ds = torch.from_numpy(mean).cuda(), torch.from_numpy(stdev).cuda()
    out = torch.empty_like(d)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(lib.smh_scale_data_f64(p(d), F, T, p(dm), p(ds), p(out), _stream()), "smh_scale_data_f64")
    return out.cpu().numpy()


def get_data_statistics(FV, stat_type='skew', axis=0):
    """tools.pyx:169-215: FV (N, f, t) -> (N, t) for axis=0 (along the columns) or (N, f) for axis=1."""
    import ctypes as C
    from ... import _lib
    lib = _lib.require_gpu()
    stats = {'mean': 0, 'variance': 1, 'skew': 2, 'kurtosis': 3}
    if stat_type not in stats:
        raise ValueError("get_data_statistics: stat_type must be one of %s" % sorted(stats))
    if axis not in (0, 1):
        raise ValueError("get_data_statistics: axis must be 0 or 1")
    FV = np.asarray(FV)
    if FV.ndim == 4 and FV.shape[3] == 1:
        FV = FV[..., 0]
    if FV.ndim != 3:
        raise ValueError("get_data_statistics: FV should be (N, f, t), got %s" % (FV.shape,))
    N, F, T = FV.shape
    d = torch.from_numpy(np.ascontiguousarray(FV, dtype=np.float64)).cuda()
    out = torch.empty((N, T if axis == 0 else F), dtype=torch.float64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(lib.smh_data_statistics_f64(p(d), N, F, T, stats[stat_type], axis, p(out), _stream()), "smh_data_statistics_f64")
    return out.cpu().numpy()
