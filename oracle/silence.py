"""numpy restatement of the step in front of the hot path -- TEST INFRASTRUCTURE.

  load_and_preprocess_signal   lib/preprocessing.py:330-350
  librosa.feature.rms(y, frame_length, hop_length)  [librosa 0.8: center=True, pad_mode='reflect']
  tools.removeSilence          lib/cython_impl/tools.pyx:42-134

Quirks of the reference that are REPLICATED (tests pin them against the compiled tools.pyx):
  * the threshold is stored in a C float: energyThresh = float32(alpha * max(energy)), the product taken in
    float64 (the reference's pinned numpy 1.19 promotes float32-scalar * Python-float to float64; numpy 2 would
    multiply in float32 -- the two differ by at most one float32 ulp of the threshold);
  * totalSilDuration is a C int accumulator: each `+= (l-k)/fs` truncates;
  * scipy.signal.medfilt(marker, 5) zero-pads the ends;
  * run detection: the two inner while-loops stop at the LAST frame without consuming it, so a signal that
    ends in silence (or in sound) is handled asymmetrically; k = max(hop*(i-1)+win, 1), l = min(hop*(j-1)+win, N);
  * a run is removed only if (l-k)/fs > beta, and NOTHING is removed unless at least TWO runs qualify (nSil > 1);
  * when samples are removed the output keeps the input length: non-silent samples first, then a TAIL OF 1.0
    (np.ones initialisation), float32; otherwise the input array is returned untouched.
"""
from __future__ import annotations

import numpy as np


def rms(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """librosa.feature.rms(y=y, frame_length=, hop_length=)[0]: reflect-pad by frame_length//2 (no edge
    repeat), frames of frame_length at hop_length, sqrt(mean(|x|^2)).  float32 in, float32 out."""
    y = np.asarray(y)
    pad = frame_length // 2
    yp = np.pad(y, pad, mode="reflect")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    x = yp[idx]
    return np.sqrt(np.mean(np.abs(x) ** 2, axis=0))


def medfilt5_zero_pad(v: np.ndarray) -> np.ndarray:
    """scipy.signal.medfilt(v, 5): zero padding at both ends."""
    vp = np.concatenate([np.zeros(2), np.asarray(v, np.float64), np.zeros(2)])
    return np.median(np.stack([vp[i:i + len(v)] for i in range(5)]), axis=0)


def silence_runs(frame_marker: np.ndarray, n_samples: int, fs: int, Tw: int, Ts: int, beta: float = 0.075):
    """The run-detection loop of tools.pyx:101-123, literally.  Returns the list of removed [k, l)."""
    frameSize = int((Tw * fs) / 1000)
    frameShift = int((Ts * fs) / 1000)
    nFrames = len(frame_marker)
    runs = []
    i = 0
    while i < nFrames:
        while frame_marker[i] == 1:
            if i == nFrames - 1:
                break
            i += 1
        j = i
        while frame_marker[j] == 0:
            if j == nFrames - 1:
                break
            j += 1
        k = max(frameShift * (i - 1) + frameSize, 1)
        l = min(frameShift * (j - 1) + frameSize, n_samples)
        if (l - k) / fs > beta:
            runs.append((k, l))
        i = j + 1
    return runs


def remove_silence(x: np.ndarray, energy: np.ndarray, fs: int, Tw: int, Ts: int, alpha: float = 0.025, beta: float = 0.075):
    """tools.removeSilence(Xin, nSamples, energy, nFrames, fs, Tw, Ts).  Returns (Xin_silrem, sample_marker,
    frame_marker, totalSilDuration)."""
    n = len(x)
    thresh = np.float32(float(alpha) * float(np.max(energy)))  # cdef float <- float64 product
    marker = (np.asarray(energy) >= thresh).astype(np.int64)
    marker = (medfilt5_zero_pad(marker) > 0.5).astype(np.int64)
    runs = silence_runs(marker, n, fs, Tw, Ts, beta)
    sample_marker = np.ones(n, dtype=np.int64)
    total = 0
    for k, l in runs:
        sample_marker[k:l] = 0
        total = int(total + (l - k) / fs)  # cdef int totalSilDuration
    if len(runs) > 1:
        out = np.ones(n, dtype=np.float32)
        keep = np.asarray(x)[sample_marker == 1]
        out[: len(keep)] = keep
    else:
        out = x
    return out, sample_marker, marker, total


def normalize_signal(x):
    x = x - np.mean(x)
    return x / np.max(np.abs(x))


def load_and_preprocess_from_samples(x: np.ndarray, fs: int = 16000, Tw: int = 25, Ts: int = 10):
    """lib/preprocessing.py:332-350 starting from the decoded samples (librosa.load is file I/O)."""
    x = normalize_signal(np.asarray(x, np.float32))
    frameSize = int((Tw * fs) / 1000)
    frameShift = int((Ts * fs) / 1000)
    energy = rms(x, frameSize, frameShift)
    xs = remove_silence(x, energy, fs, Tw, Ts)[0]
    xs = xs.copy()
    if len(xs) / fs < 0.1:
        while len(xs) / fs < 0.1:
            xs = np.append(xs, xs)
    return normalize_signal(xs)
