"""Seeded synthetic 16 kHz clips: the inputs every parity test and bench run uses.

clip = 0.5*N(0,1) white noise (percussive-ish) + 4 sinusoids f~U(100,4000) Hz, a~U(0.1,1)
(harmonic-ish) + a 20 ms noise burst every 250 ms, then the reference normalisation
`x -= mean; x /= max|x|` (lib/preprocessing.py:130-131).  float32, shape (B, n_samples).
numpy only: identical arrays feed the CPU oracle and the HIP path.
"""
from __future__ import annotations

import numpy as np


def synth_clips(batch: int, seed: int = 0, n_samples: int = 16000, fs: int = 16000) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / fs
    out = np.empty((batch, n_samples), dtype=np.float32)
    burst_len = int(0.020 * fs)
    burst_hop = int(0.250 * fs)
    for i in range(batch):
        x = 0.5 * rng.standard_normal(n_samples)
        f = rng.uniform(100.0, 4000.0, size=4)
        a = rng.uniform(0.1, 1.0, size=4)
        ph = rng.uniform(0.0, 2 * np.pi, size=4)
        x += (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None])).sum(axis=0)
        for s in range(0, n_samples, burst_hop):
            e = min(s + burst_len, n_samples)
            x[s:e] += 2.0 * rng.standard_normal(e - s)
        x = x - x.mean()
        x = x / np.max(np.abs(x))
        out[i] = x.astype(np.float32)
    return out
