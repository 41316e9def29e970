"""The training batch's random draws on the device (csrc/smh_rng.hip, C ABI smh_noise_augment_f32 / smh_dropout_masks_f32).

Seeds come from torch's CPU generator, one per call, so `torch.manual_seed(s)` makes a run reproducible exactly as it did
when these were torch kernels.  A caller that fixes the seed gets a fresh Philox stream per call from a process-wide call
counter (the stream offset) unless it passes the offset too."""
from __future__ import annotations

import ctypes as C
import itertools

import torch

from . import _lib

_calls = itertools.count()


def fresh_seed() -> int:
    """63 random bits from torch's default CPU generator (follows torch.manual_seed; no device work, no sync)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def _stream():
    return _lib.current_stream()


def add_normal_noise(x: "torch.Tensor", scale: float, seed=None, offset=None, out=None) -> "torch.Tensor":
    """x + N(0, scale) in one pass (Proposed_Work_Results.py:239-242).  `out=x` works in place."""
    if seed is None:
        seed, offset = fresh_seed(), 0 if offset is None else offset
    elif offset is None:
        offset = next(_calls)
    if not (x.is_cuda and x.dtype == torch.float32):
        raise TypeError("add_normal_noise: a float32 CUDA tensor is required, got %s on %s" % (x.dtype, x.device))
    xc = x.contiguous()
    if out is None:
        out = torch.empty_like(xc)
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == xc.numel()):
        raise ValueError("add_normal_noise: out must be a contiguous float32 CUDA tensor of the input's size")
    lib = _lib.load()
    _lib.check(lib.smh_noise_augment_f32(C.c_void_p(xc.data_ptr()), C.c_void_p(out.data_ptr()), xc.numel(), float(scale),
                                         int(seed), int(offset), _stream()),
               "smh_noise_augment_f32")
    return out


def dropout_masks(n_a: int, keep_a: float, n_b: int, keep_b: float, seed: int, offset: int) -> "torch.Tensor":
    """One float32 CUDA vector [n_a masks at keep_a | n_b masks at keep_b], values 0 or 1 / keep."""
    out = torch.empty(int(n_a) + int(n_b), dtype=torch.float32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.smh_dropout_masks_f32(C.c_void_p(out.data_ptr()), int(n_a), float(keep_a), int(n_b), float(keep_b),
                                         int(seed), int(offset), _stream()), "smh_dropout_masks_f32")
    return out
