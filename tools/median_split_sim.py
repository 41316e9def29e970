"""Scalar simulation of the block-split sliding median (the algorithm of smh_median_split.h) against
scipy.ndimage.median_filter(mode='reflect'): checks the index ranges that are kept of every sorted run.

Window W = 2H+1.  The padded sequence is cut into blocks of W elements; the window that ENDS at offset p of block k is
  (suffix of block k-1 of size a = W-1-p)  U  (prefix of block k of size b = p+1).
Both runs only ever GROW by insertion (1 med3 per kept slot): suffix runs are built walking block k-1 backwards (all
sizes are kept: the 'history'), prefix runs walking block k forwards.  The median is the H-th smallest of the union:
  r = min_i max(A[i-1], B[H-i]),  i in [max(0, H+1-b), min(a, H+1)]   (A[-1] = B[-1] = -inf)
so of a run of size s only the indices [max(0, s-H-1), min(s-1, H)] are ever read, by the selection and by the
insertion that builds the next size: that is what is kept in registers."""
import sys
import numpy as np
from scipy.ndimage import median_filter

NEG, POS = -np.inf, np.inf


def kept(s, H):
    return max(0, s - H - 1), min(s - 1, H)


def grow(run, lo_hi, x, s_new, H, counter):
    """run: dict index->value of the sorted run of size s_new-1 (kept range lo_hi); returns the run of size s_new."""
    lo, hi = kept(s_new, H)
    old_lo, old_hi = lo_hi
    new = {}
    for i in range(lo, hi + 1):
        below = NEG if i - 1 < 0 else run[i - 1]          # must be kept unless it is the virtual -inf
        above = POS if i > s_new - 2 else run[i]          # virtual +inf above the old top
        assert i - 1 < 0 or old_lo <= i - 1 <= old_hi, (s_new, i)
        assert i > s_new - 2 or old_lo <= i <= old_hi, (s_new, i)
        new[i] = sorted((below, x, above))[1]
        counter[0] += 1
    return new, (lo, hi)


def select(A, a, B, b, H, counter):
    best = POS
    for i in range(max(0, H + 1 - b), min(a, H + 1) + 1):
        u = NEG if i == 0 else A[i - 1]
        v = NEG if H - i < 0 else B[H - i]
        best = min(best, max(u, v))
        counter[1] += 1
    return best


def split_median(x, W):
    H = W // 2
    n = len(x)
    pad = np.concatenate([x[:H][::-1], x, x[::-1][:H]])   # scipy 'reflect' == numpy 'symmetric'
    E = len(pad)
    out = np.empty(n, x.dtype)
    counter = [0, 0]
    nb = (E + W - 1) // W
    hist = None
    for k in range(nb):
        blk = pad[k * W:(k + 1) * W]
        if k > 0:
            B, rng = {}, (0, -1)
            for p in range(len(blk)):
                B, rng = grow(B, rng, blk[p], p + 1, H, counter)
                a = W - 1 - p
                A = hist[a][0] if a > 0 else {}
                out[k * W + p - (W - 1)] = select(A, a, B, p + 1, H, counter)
        if len(blk) == W and k + 1 < nb or k == 0:
            hist = {}
            A, rng = {}, (0, -1)
            top = W if k == 0 else W - 1
            for s in range(1, top + 1):
                A, rng = grow(A, rng, blk[W - s], s, H, counter)
                hist[s] = (A, rng)
            if k == 0:
                out[0] = hist[W][0][H]
    return out, counter


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for W in (3, 5, 7, 11, 17, 21, 31):
        for n in (W // 2 + 5, 49, 98, 201):
            if n <= W // 2:
                continue
            x = rng.standard_normal(n).astype(np.float32)
            x[rng.integers(0, n, n // 4)] = x[0]           # ties
            ref = median_filter(x, size=W, mode="reflect")
            got, cnt = split_median(x, W)
            assert np.array_equal(ref, got), (W, n)
        print("W=%d ok: %.1f med3 + %.1f max per output at n=201" % (W, cnt[0] / 201, cnt[1] / 201))
