"""Counterpart of the reference's only native module, lib/cython_impl/tools.pyx.

`extract_patches` (tools.pyx:21-38) is on the hot path and runs as a HIP gather through the C ABI
(`smh_extract_patches_f32`); it returns float64 (nP, F, W) exactly like the Cython function.
`removeSilence` is SURVEY 8(f) rank 1 ("next"); `scale_data` / `get_data_statistics` are out of scope
(off by default in the reference: frame_level_scaling False, skewness_vector None).
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


def removeSilence(*args, **kwargs):
    raise NotImplementedError("tools.removeSilence is a 'next' row (SURVEY 8f rank 1), not built in this round")


def scale_data(*args, **kwargs):
    raise NotImplementedError("tools.scale_data is out of scope (frame_level_scaling is False on the hot path)")


def get_data_statistics(*args, **kwargs):
    raise NotImplementedError("tools.get_data_statistics is out of scope (skewness_vector is None on the hot path)")
