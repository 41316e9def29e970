"""numpy restatement of the two optional functions of lib/cython_impl/tools.pyx -- TEST INFRASTRUCTURE.

  scale_data(FV, mean, stdev)               tools.pyx:138-165
  get_data_statistics(FV, stat_type, axis)  tools.pyx:169-215   (scipy.stats.skew / kurtosis: bias=True, Fisher)

Pinned in tests/test_oracle_pins.py against the compiled reference module (oracle/_ref) and scipy.stats.
A constant vector has skew 0 and kurtosis -3 here, as in the scipy the reference pins (1.5: `np.where(zero, 0, ...)`);
scipy >= 1.9 returns nan for it.
"""
from __future__ import annotations

import numpy as np


def scale_data(FV, mean, stdev):
    FV = np.asarray(FV).astype(np.float64)
    M = np.asarray(mean, np.float64).reshape(-1, 1)
    S = np.asarray(stdev, np.float64).reshape(-1, 1)
    return np.divide(np.subtract(FV, M), S + 1e-10)


def get_data_statistics(FV, stat_type="skew", axis=0):
    FV = np.asarray(FV, np.float64)
    ax = 1 + axis  # per patch: axis 0 = over the rows -> (N, t); axis 1 = over the frames -> (N, f)
    mean = FV.mean(axis=ax, keepdims=True)
    if stat_type == "mean":
        return mean.squeeze(ax)
    d = FV - mean
    m2 = (d ** 2).mean(axis=ax)
    if stat_type == "variance":
        return m2
    zero = m2 == 0
    safe = np.where(zero, 1.0, m2)
    if stat_type == "skew":
        return np.where(zero, 0.0, (d ** 3).mean(axis=ax) / safe ** 1.5)
    if stat_type == "kurtosis":
        return np.where(zero, 0.0, (d ** 4).mean(axis=ax) / safe ** 2) - 3.0
    raise ValueError(stat_type)
