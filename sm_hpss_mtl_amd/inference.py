"""Dense file-level inference and smoothing: device counterpart of
DAFx12_Speech_Music_Detection_B3_MTL_v2.py:594-706 (`patch_probability_generator`) and :94-98 (`smooth_labels`).

  patch_probabilities   standardise the file's featuregram (:612-626), walk it in 10 000-frame batches (:634-642),
                        get_feature_patches with shift W_shift_test = 1 (:647), predict one head (:519-523, 661-665)
  medfilt / smooth_labels   scipy.signal.medfilt(Predictions, 501) and the 0.5 threshold (:96-97)

Everything numeric runs in libsmh.so: StandardScaler rows, the hop-1 patch gather (time-major, straight into the TCN
layout), the B3_MTL forward, the 501-wide zero-padded median.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .frontend import Frontend, FrontendConfig, _ptr, _stream

_fe = None


def _frontend():
    global _fe
    if _fe is None:
        _fe = Frontend(FrontendConfig())
    return _fe


def _dev(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return x.float().cuda().contiguous()


def medfilt(x, kernel_size):
    """scipy.signal.medfilt(x, kernel_size) for a 1-D track (or B tracks as rows); numpy in -> numpy out, CUDA
    tensor in -> CUDA tensor out.  Zero padded, odd window, bit-exact."""
    lib = _lib.require_gpu()
    was_np = isinstance(x, np.ndarray)
    d = _dev(x)
    one = d.dim() == 1
    if one:
        d = d[None]
    if d.dim() != 2:
        raise ValueError("medfilt expects a 1-D track or (B, n) tracks, got shape %s" % (tuple(d.shape),))
    y = torch.empty_like(d)
    _lib.check(lib.smh_medfilt1d_f32(_ptr(d), d.shape[0], d.shape[1], int(kernel_size), _ptr(y), _stream()),
               "smh_medfilt1d_f32")
    y = y[0] if one else y
    return y.cpu().numpy() if was_np else y


def mode_filtering(X, win_size):
    """DAFx12...:81-89 (host code, label post-processing): the most frequent label of the window
    X[i - h : i + h], h = odd(win_size) // 2 -- note the window holds 2h elements, the element at i + h is not in it --
    the smallest label on a tie (np.unique sorts), the first and last h entries kept as they are."""
    X = np.asarray(X)
    if win_size % 2 == 0:
        win_size += 1
    h = int(win_size / 2)
    out = X.copy()
    n = len(X)
    if h == 0 or n <= 2 * h:
        return out
    labels = np.unique(X)
    # counts of every label in the window by prefix sums: window of i covers [i - h, i + h)
    best = np.full(n - 2 * h, -1, np.int64)
    arg = np.zeros(n - 2 * h, X.dtype)
    for lab in labels:  # ascending: a later label only wins with a strictly larger count
        cs = np.concatenate([[0], np.cumsum(X == lab)])
        cnt = cs[2 * h:n] - cs[0:n - 2 * h]
        win = cnt > best
        best = np.where(win, cnt, best)
        arg = np.where(win, lab, arg)
    out[h:n - h] = arg
    return out


def smooth_labels(Predictions, PtdLabels, win_size, smooth_type="prediction"):
    """DAFx12...:94-103 -> (Predictions_smooth, PtdLabels_smooth)."""
    if smooth_type == "label":
        return Predictions, mode_filtering(PtdLabels, win_size)
    if smooth_type != "prediction":
        raise ValueError("smooth_type must be 'prediction' or 'label'")
    sm = medfilt(Predictions, win_size)
    lab = (sm > 0.5).astype(int) if isinstance(sm, np.ndarray) else (sm > 0.5).to(torch.int64)
    return sm, lab


def patch_probabilities(fv, model, W, W_shift=1, output="M", batch_frames=10000):
    """fv (2R, nFrames) HarmPerc featuregram of one file -> 1-D float32 numpy track of the chosen head's
    probability, one value per patch, batches concatenated (DAFx12...:612-676)."""
    fe = _frontend()
    d = _dev(fv)
    if d.dim() != 2 or d.shape[0] % 2:
        raise ValueError("fv must be (2R, nFrames), got %s" % (tuple(d.shape),))
    names = model.output_names
    if output not in names:
        raise ValueError("output %r is not one of %s" % (output, names))
    col = 0
    for n, o in zip(names, model.split_outputs(torch.zeros((1, model.out_dim)))):
        if n == output:
            break
        col += o.shape[1]
    R = d.shape[0] // 2
    # batches longer than a patch skip the (nP, W, 2R) patch tensor (648 MB per 10 000 frames at W = 68, hop 1) when the model has
    # the entry for it; SMH_DENSE_PATCHES=1 keeps the patch path (A/B, tests)
    dense = (hasattr(model, "forward_dense") and getattr(model, "block_variant", 1) == 0 and d.shape[0] % 8 == 0
             and getattr(model, "patch_size", None) == W and not os.environ.get("SMH_DENSE_PATCHES"))
    d = fe.standardize_rows(d)  # :612-626, whole file; the scaler works row by row, so the two halves are one call
    T = d.shape[1]
    preds = []
    for s in range(0, T, batch_frames):
        e = min(s + batch_frames, T)
        chunk = d[:, s:e].contiguous()
        if dense and e - s > W:
            # the same patches without building them (smh_model_forward_dense_f32): each half standardised over the batch as
            # get_feature_patches does (:647), layer 0 once per frame, every hop-W_shift patch a window of it
            o = model.forward_dense(fe.standardize_rows(chunk), W_shift)
            if o.shape[0]:
                preds.append(o[:, col])
            continue
        # get_feature_patches on the batch (:647): tile if short, standardise each half over the batch, hop-W_shift
        # patches; written time-major = the transposed TCN input of :660
        h = fe.extract_patches(fe.standardize_rows(chunk[:R])[None], W, W_shift, time_major=True)
        p = fe.extract_patches(fe.standardize_rows(chunk[R:])[None], W, W_shift, time_major=True)
        if h.shape[0] == 0:
            continue
        x = torch.cat([h, p], dim=2)
        preds.append(model.forward_device(x)[:, col])
    if not preds:
        return np.zeros((0,), np.float32)
    out = torch.cat(preds)
    if hasattr(model, "check_status"):
        model.check_status()  # the forwards above only enqueued work: a device-side give-up raises here, before the track leaves
    return out.cpu().numpy()
