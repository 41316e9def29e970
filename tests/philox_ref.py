"""numpy restatement of csrc/smh_rng.hip's generator (test infrastructure): Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random
numbers: as easy as 1, 2, 3", SC'11) with counter = (group lo, group hi, offset lo, offset hi), key = (seed lo, seed hi), and the
kernels' word -> value maps.  Pinned by the Random123 known-answer vectors in tests/test_rng.py."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """uint32 arrays (or scalars) -> four uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def words(n, seed, offset):
    """The n 32-bit words the kernels draw for elements 0 .. n-1: element i uses word i % 4 of group i // 4."""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    r = philox4x32_10(g & MASK, g >> np.uint64(32), offset & 0xFFFFFFFF, (offset >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF,
                      (seed >> 32) & 0xFFFFFFFF)
    return np.stack(r, axis=1).reshape(-1)[:n]


def uniform24(w):
    """[0, 1) on 24 bits, exactly as the kernels form it (float32-exact)."""
    return (w >> np.uint32(8)).astype(np.float64) * 2.0 ** -24


def normals(n, seed, offset):
    """Box-Muller on word pairs (0, 1) and (2, 3) of every group, in float64."""
    w = words(4 * ((n + 3) // 4), seed, offset).reshape(-1, 2)
    u1 = ((w[:, 0] >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
    u2 = (w[:, 1] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    rad = np.sqrt(-2.0 * np.log(u1))
    return np.stack([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)], axis=1).reshape(-1)[:n]


def masks(n_a, keep_a, n_b, keep_b, seed, offset):
    u = uniform24(words(n_a + n_b, seed, offset)).astype(np.float32)
    ka, kb = np.float32(keep_a), np.float32(keep_b)
    out = np.zeros(n_a + n_b, dtype=np.float32)
    out[:n_a] = np.where(u[:n_a] < ka, np.float32(1.0) / ka, np.float32(0.0))
    out[n_a:] = np.where(u[n_a:] < kb, np.float32(1.0) / kb, np.float32(0.0))
    return out
