"""The numpy restatement of the device generator against Random123's known-answer vectors (CPU)."""
import numpy as np
import pytest

from tests.philox_ref import masks, normals, philox4x32_10, uniform24, words

KAT = [  # counter (4 words), key (2 words) -> output: Random123 kat_vectors, philox4x32 10 rounds
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(ctr, key, want):
    got = philox4x32_10(*ctr, *key)
    assert tuple(int(v) for v in got) == want


def test_word_layout_and_value_maps():
    w = words(10, seed=(7 << 32) | 5, offset=3)
    g = philox4x32_10(np.array([0, 1, 2]), 0, 3, 0, 5, 7)
    assert list(w[:4]) == [int(g[j][0]) for j in range(4)] and list(w[8:10]) == [int(g[0][2]), int(g[1][2])]
    u = uniform24(w)
    assert u.min() >= 0 and u.max() < 1
    z = normals(200001, seed=11, offset=0)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    m = masks(1000, 0.8, 600, 0.6, seed=2, offset=9)
    assert set(np.unique(m[:1000])) <= {np.float32(0), np.float32(1) / np.float32(0.8)}
    assert abs((m[:1000] > 0).mean() - 0.8) < 0.05 and abs((m[1000:] > 0).mean() - 0.6) < 0.08
