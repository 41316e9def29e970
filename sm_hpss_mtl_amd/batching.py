"""Batch and label assembly of the reference's training generators.

3-class: Proposed_Work_Results.py:161-262 -- batch = [bs music | bs speech | bs speech+music]; labels
  3C one-hot of (0,1,2); S = [0,1,0]; M = [1,0,0] (sic: the speech+music rows get S = 0 and M = 0, :249-260);
  R rows: music [1,0], speech [0,1], mixture: SMR >= 0 -> [10^(-SMR/10), 1] else [1, 10^(SMR/10)] (:172-174,227-231).
5-class: 5_class_classification.py:602-671 -- classes (music, speech, speech+music, noise, speech+noise);
  S = [0,1,1,0,1]; M = [1,0,1,0,0]; N = [0,0,0,1,1]; R is 3-dim (music, speech, noise shares).
Gaussian-noise augmentation: scale drawn from {5e-3, 1e-3, 5e-4, 1e-4} (Proposed_Work_Results.py:239-242).
The labels are host-side integer/float bookkeeping; the patches they describe come from the HIP front end.
"""
from __future__ import annotations

import numpy as np

NOISE_SCALES = (5e-3, 1e-3, 5e-4, 1e-4)


def _ratio_pair(smr_db):
    """(weaker, 1) ordering of the reference: SMR >= 0 -> [10^(-SMR/10), 1] else [1, 10^(SMR/10)]."""
    smr_db = np.asarray(smr_db, dtype=np.float64)
    a = np.where(smr_db >= 0, 1.0 / np.power(10.0, smr_db / 10.0), 1.0)
    b = np.where(smr_db >= 0, 1.0, np.power(10.0, smr_db / 10.0))
    return a, b


def make_labels_3class(bs: int, smr_db):
    """Labels of one 3-class batch of 3*bs patches; smr_db: the bs target SMRs of the mixture rows."""
    cls = np.repeat(np.arange(3), bs)
    R = np.ones((3 * bs, 2))
    R[:bs] = [1, 0]
    R[bs:2 * bs] = [0, 1]
    a, b = _ratio_pair(smr_db)
    R[2 * bs:, 0], R[2 * bs:, 1] = a, b
    S = (cls == 1).astype(np.int64)   # speech+music rows stay 0, as in the reference
    M = (cls == 0).astype(np.int64)
    return {"R": R, "S": S, "M": M, "3C": np.eye(3, dtype=np.float32)[cls]}


def make_labels_5class(bs: int, smr_spmu_db, smr_spno_db):
    cls = np.repeat(np.arange(5), bs)
    R = np.ones((5 * bs, 3))
    R[:bs] = [1, 0, 0]
    R[bs:2 * bs] = [0, 1, 0]
    a, b = _ratio_pair(smr_spmu_db)
    R[2 * bs:3 * bs] = np.stack([a, b, np.zeros(bs)], 1)
    R[3 * bs:4 * bs] = [0, 0, 1]
    a, b = _ratio_pair(smr_spno_db)
    R[4 * bs:] = np.stack([np.zeros(bs), a, b], 1)
    S = np.isin(cls, (1, 2, 4)).astype(np.int64)
    M = np.isin(cls, (0, 2)).astype(np.int64)
    N = np.isin(cls, (3, 4)).astype(np.int64)
    return {"R": R, "S": S, "M": M, "N": N, "3C": np.eye(5, dtype=np.float32)[cls]}


def noise_augmentation(batch, rng):
    """batch + N(0, scale), scale drawn once per batch from `rng` (Proposed_Work_Results.py:239-242).  Host batches (numpy) draw
    the noise from `rng` too; device batches (float32 CUDA tensors) get it from the HIP kernel in one pass over the patches
    (device_rng.add_normal_noise: Philox, seeded from torch's generator) -- there is no torch fallback for a device batch."""
    scale = float(rng.choice(NOISE_SCALES))
    if not isinstance(batch, np.ndarray):
        from .device_rng import add_normal_noise
        return add_normal_noise(batch, scale)
    return batch + rng.normal(0.0, scale, size=batch.shape)


def synthetic_batch(frontend, bs, W, shift, rng, smr_cycle=(-5, 0, 5, 10, 15, 20), n_samples=16000, augment=True):
    """One 3-class training batch on the device from synthetic audio (no MUSAN in this environment):
    'music' = tonal clips, 'speech' = noisy/bursty clips, mixtures via the reference's mix_signals at SMRs cycling
    -5..20 dB.  Returns (patches (3*bs*nP, W, F) CUDA tensor, labels dict)."""
    import torch
    from .lib.preprocessing import mix_signals, normalize_signal
    t = np.arange(n_samples) / 16000.0

    def music():
        f = rng.uniform(100, 4000, 4)
        a = rng.uniform(0.2, 1, 4)
        return normalize_signal(((a[:, None] * np.sin(2 * np.pi * f[:, None] * t)).sum(0) + 0.02 * rng.standard_normal(n_samples)).astype(np.float32))

    def speech():
        x = 0.3 * rng.standard_normal(n_samples)
        for s in range(0, n_samples, 4000):
            x[s:s + 320] += 2.0 * rng.standard_normal(min(320, n_samples - s))
        return normalize_signal(x.astype(np.float32))

    smr = np.array([smr_cycle[i % len(smr_cycle)] for i in range(bs)])
    mu = [music() for _ in range(bs)]
    sp = [speech() for _ in range(bs)]
    mix = [mix_signals(speech(), music(), float(d)) for d in smr]
    audio = torch.from_numpy(np.stack(mu + sp + mix).astype(np.float32)).cuda()
    res = frontend.run(audio, W=W, shift=shift)
    nP = res["n_patches"]
    x = res["patches"]
    if augment:
        x = noise_augmentation(x, rng)
    lab = make_labels_3class(bs, smr)
    lab = {k: np.repeat(v, nP, axis=0) for k, v in lab.items()} if nP != 1 else lab
    return x, lab
