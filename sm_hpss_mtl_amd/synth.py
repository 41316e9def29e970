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


def bench_clips(batch: int, rank: int = 0, n_samples: int = 16000) -> np.ndarray:
    """The batch bench.py times on rank `rank`: `batch` DISTINCT clips -- rows 0..63 = synth_clips(64, seed=1000 + rank) (the
    clips tests/golden/bench_golden.npz holds oracle logits for), rows 64.. = synth_clips(batch - 64, seed=5000 + rank) (the
    golden file also holds rows 64 and 65).  Nothing is tiled: every step reads batch x 64 000 bytes of different audio."""
    head = synth_clips(min(batch, 64), seed=1000 + rank, n_samples=n_samples)
    if batch <= 64:
        return head
    return np.concatenate([head, synth_clips(batch - 64, seed=5000 + rank, n_samples=n_samples)], axis=0)


# (n_samples, [(start_s, end_s) of near-silent stretches]) -- the cases of tests/golden/silence_golden.npz
SILENCE_CASES = (
    (32000, ((0.2, 0.5), (1.0, 1.3))),              # two runs -> removal
    (32000, ((0.4, 0.9),)),                          # one run -> untouched
    (32000, ()),                                     # no silence
    (32000, ((0.0, 0.3), (0.8, 1.1), (1.7, 2.0))),   # silence at both ends
    (32000, ((0.1, 0.15), (0.5, 0.9), (1.2, 1.26))), # short stretches below beta
    (64000, ((0.2, 1.6), (2.0, 3.3))),               # runs longer than one second (integer duration accumulator)
    (16000, ((0.3, 0.45), (0.6, 0.8))),              # the bench clip length
)


def gappy_clip(case: int, fs: int = 16000) -> np.ndarray:
    """Seeded noise with near-silent stretches (float32, not normalised): input of the silence-removal tests."""
    n, gaps = SILENCE_CASES[case]
    rng = np.random.default_rng(1000 + case)
    x = (0.3 * rng.standard_normal(n)).astype(np.float32)
    env = (0.75 + 0.5 * ((np.arange(n) % 4000) / 4000.0)).astype(np.float32)  # sawtooth: exactly reproducible
    x *= env
    for a, b in gaps:
        x[int(a * fs):int(b * fs)] *= np.float32(1e-4)
    return x
