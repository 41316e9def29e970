"""Signal conditioning in front of the featuregram: host wrapper over the C ABI.

  normalize          lib/preprocessing.py:332-333, 348-349
  rms                librosa.feature.rms as called at lib/preprocessing.py:338
  remove_silence     lib/cython_impl/tools.pyx:42-134
  preprocess_signal  lib/preprocessing.py:332-349 in one call
  mix_signals        lib/preprocessing.py:297-325

All functions take float32 CUDA(=HIP) tensors of shape (B, N) -- B equal-length clips -- and return device
tensors; the arithmetic happens in libsmh.so (smh_silence.hip).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .frontend import _f32c, _ptr, _stream


def _batch(x, name):
    x = _f32c(x, name)
    if x.dim() == 1:
        x = x[None]
    if x.dim() != 2:
        raise ValueError("%s must be (B, N) or (N,), got shape %s" % (name, tuple(x.shape)))
    return x


def _work(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def normalize(x, out=None):
    """x (B, N) -> (x - mean) / max|x - mean| per clip."""
    lib = _lib.require_gpu()
    x = _batch(x, "x")
    B, N = x.shape
    out = torch.empty_like(x) if out is None else out
    work = _work(lib.smh_normalize_workspace_bytes(B, N), x.device)
    _lib.check(lib.smh_normalize_f32(_ptr(x), B, N, _ptr(out), _ptr(work), work.numel(), _stream()), "smh_normalize_f32")
    return out


def rms(y, frame_length, hop_length):
    """y (B, N) -> energy (B, 1 + N // hop_length) = librosa.feature.rms(y=, frame_length=, hop_length=)[0]."""
    lib = _lib.require_gpu()
    y = _batch(y, "y")
    B, N = y.shape
    nF = _lib.check(lib.smh_rms_f32(None, 0, N, int(frame_length), int(hop_length), None, None), "smh_rms_f32")
    e = torch.empty((B, nF), dtype=torch.float32, device=y.device)
    _lib.check(lib.smh_rms_f32(_ptr(y), B, N, int(frame_length), int(hop_length), _ptr(e), _stream()), "smh_rms_f32")
    return e


def remove_silence(x, energy, fs, Tw, Ts, alpha=0.025, beta=0.075, markers=False):
    """tools.removeSilence on a batch.  Returns (out (B, N), n_keep (B,) int32[, sample_marker (B, N) uint8,
    frame_marker (B, nFrames) int32]); out keeps the input length (retained samples, then the reference's 1.0 tail)."""
    lib = _lib.require_gpu()
    x = _batch(x, "x")
    energy = _batch(energy, "energy")
    B, N = x.shape
    if energy.shape[0] != B:
        raise ValueError("energy has %d rows for %d clips" % (energy.shape[0], B))
    nF = energy.shape[1]
    hop = int((Ts * fs) / 1000)
    if hop < 1:
        raise ValueError("frame shift Ts=%r ms is shorter than one sample at fs=%r" % (Ts, fs))
    out = torch.empty_like(x)
    n_keep = torch.empty((B,), dtype=torch.int32, device=x.device)
    sm = torch.empty((B, N), dtype=torch.uint8, device=x.device) if markers else None
    fm = torch.empty((B, nF), dtype=torch.int32, device=x.device) if markers else None
    work = _work(lib.smh_silence_workspace_bytes(B, N, hop), x.device)
    _lib.check(lib.smh_remove_silence_f32(_ptr(x), B, N, _ptr(energy), nF, int(fs), int(Tw), int(Ts), float(alpha),
                                          float(beta), _ptr(out), _ptr(sm), _ptr(fm), _ptr(n_keep), _ptr(work),
                                          work.numel(), _stream()), "smh_remove_silence_f32")
    return (out, n_keep, sm, fm) if markers else (out, n_keep)


def preprocess_signal(x, fs, Tw, Ts):
    """lib/preprocessing.py:332-349 for a batch: normalise -> rms -> removeSilence -> normalise.
    Returns (out (B, N), n_keep (B,) int32).  The caller applies the '< 0.1 s: duplicate' rule (depends on N only)."""
    lib = _lib.require_gpu()
    x = _batch(x, "x")
    B, N = x.shape
    hop = int((Ts * fs) / 1000)
    if hop < 1:
        raise ValueError("frame shift Ts=%r ms is shorter than one sample at fs=%r" % (Ts, fs))
    out = torch.empty_like(x)
    n_keep = torch.empty((B,), dtype=torch.int32, device=x.device)
    work = _work(lib.smh_silence_workspace_bytes(B, N, hop), x.device)
    _lib.check(lib.smh_preprocess_signal_f32(_ptr(x), B, N, int(fs), int(Tw), int(Ts), _ptr(out), _ptr(n_keep),
                                             _ptr(work), work.numel(), _stream()), "smh_preprocess_signal_f32")
    return out, n_keep


def mix_signals(sp, mu, target_db):
    """lib/preprocessing.py:297-325 for B pairs: sp (B, N), mu (B, N_mu) float32 CUDA tensors, target_db scalar or (B,)
    -> (B, N) normalised mixtures (music looped to the speech length)."""
    lib = _lib.require_gpu()
    sp, mu = _batch(sp, "sp"), _batch(mu, "mu")
    B, N = sp.shape
    if mu.shape[0] != B:
        raise ValueError("mu has %d rows for %d speech clips" % (mu.shape[0], B))
    db = torch.as_tensor(target_db, dtype=torch.float32, device=sp.device).reshape(-1)
    if db.numel() == 1:
        db = db.expand(B)
    if db.numel() != B:
        raise ValueError("target_db has %d entries for %d clips" % (db.numel(), B))
    db = db.contiguous()
    out = torch.empty_like(sp)
    work = _work(lib.smh_normalize_workspace_bytes(B, N), sp.device)
    _lib.check(lib.smh_mix_signals_f32(_ptr(sp), _ptr(mu), B, N, mu.shape[1], _ptr(db), _ptr(out), _ptr(work),
                                       work.numel(), _stream()), "smh_mix_signals_f32")
    return out
