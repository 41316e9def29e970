"""numpy restatement of the dense file-level inference -- TEST INFRASTRUCTURE.

  patch_probability_generator   DAFx12_Speech_Music_Detection_B3_MTL_v2.py:594-706 (compute part :612-676)
  smooth_labels                 DAFx12_Speech_Music_Detection_B3_MTL_v2.py:94-98
  scipy.signal.medfilt          zero padding, odd kernel (pinned against scipy in tests/test_oracle_pins.py)
"""
from __future__ import annotations

import numpy as np

from . import b3_mtl
from . import frontend as ofe


def medfilt(x, kernel_size):
    x = np.asarray(x)
    h = kernel_size // 2
    xp = np.concatenate([np.zeros(h, x.dtype), x, np.zeros(h, x.dtype)])
    win = np.lib.stride_tricks.sliding_window_view(xp, kernel_size)
    return np.sort(win, axis=1)[:, h].astype(x.dtype)


def smooth_labels(pred, win_size):
    sm = medfilt(pred, win_size)
    return sm, (sm > 0.5).astype(int)


def patch_probabilities(fv, w, W, W_shift=1, output="M", n_classes=3, batch_frames=10000):
    fv = np.asarray(fv, np.float32)
    R = fv.shape[0] // 2
    fv = np.append(ofe.standardize_rows(fv[:R]), ofe.standardize_rows(fv[R:]), axis=0)
    names = [n for n, _, _ in b3_mtl.head_spec(n_classes)] + ["3C"]
    preds = []
    for s in range(0, fv.shape[1], batch_frames):
        chunk = fv[:, s:min(s + batch_frames, fv.shape[1])]
        patches = ofe.feature_patches(chunk, W, W_shift, "LogMelHarmPercSpec")
        if patches.shape[0] == 0:
            continue
        outs = b3_mtl.forward(ofe.tcn_input(patches), w, n_classes)
        preds.append(outs[names.index(output)][:, 0])
    return np.concatenate(preds) if preds else np.zeros((0,), np.float32)
