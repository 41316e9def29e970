/* Plain-C restatement of the selection / indexing parts of the hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as a second, independent
 * checker next to the numpy restatement, and by bench.py's cpu_baseline leg.
 *
 *  - orc_median_time / orc_median_freq : what librosa.decompose.hpss delegates to,
 *      scipy.ndimage.median_filter(S, size=(1,l_harm)|(l_perc,1), mode='reflect')
 *      (call sites /root/reference/lib/preprocessing.py:408,418,430,440).
 *  - orc_extract_patches              : /root/reference/lib/cython_impl/tools.pyx:21-38.
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC).
 */
#include <stddef.h>
#include <string.h>

static int reflect_idx(int i, int n) {
    /* scipy 'reflect' (d c b a | a b c d | d c b a): period 2n */
    int p = 2 * n;
    int j = i % p;
    if (j < 0) j += p;
    return j >= n ? p - 1 - j : j;
}

/* median of w floats by insertion sort on a scratch copy (w <= 255) */
static float median_small(const float *v, int w) {
    float s[256];
    int n = 0;
    for (int i = 0; i < w; ++i) {
        float x = v[i];
        int j = n++;
        while (j > 0 && s[j - 1] > x) { s[j] = s[j - 1]; --j; }
        s[j] = x;
    }
    return s[w / 2];
}

/* S, out: (K, T) row-major float32.  Median of odd size w along t. */
int orc_median_time(const float *S, float *out, int K, int T, int w) {
    if (w < 1 || w > 255 || (w & 1) == 0 || K < 0 || T < 1) return -1;
    int h = w / 2;
    float win[256];
    for (int k = 0; k < K; ++k) {
        const float *row = S + (size_t)k * T;
        for (int t = 0; t < T; ++t) {
            for (int j = -h; j <= h; ++j) win[j + h] = row[reflect_idx(t + j, T)];
            out[(size_t)k * T + t] = median_small(win, w);
        }
    }
    return 0;
}

/* Median of odd size w along k. */
int orc_median_freq(const float *S, float *out, int K, int T, int w) {
    if (w < 1 || w > 255 || (w & 1) == 0 || K < 1 || T < 0) return -1;
    int h = w / 2;
    float win[256];
    for (int t = 0; t < T; ++t) {
        for (int k = 0; k < K; ++k) {
            for (int j = -h; j <= h; ++j) win[j + h] = S[(size_t)reflect_idx(k + j, K) * T + t];
            out[(size_t)k * T + t] = median_small(win, w);
        }
    }
    return 0;
}

/* number of patches: len(range(half, T-half, shift)) */
int orc_num_patches(int T, int W, int shift) {
    int half = W / 2;
    int lo = half, hi = T - half;
    if (shift <= 0) return -1;
    if (hi <= lo) return 0;
    return (hi - lo + shift - 1) / shift;
}

/* FV: (F, T) float32 -> out (nP, F, W) float64, zero-initialised like np.zeros */
int orc_extract_patches(const float *FV, double *out, int F, int T, int W, int shift) {
    int nP = orc_num_patches(T, W, shift);
    if (nP < 0) return -1;
    int half = W / 2;
    memset(out, 0, sizeof(double) * (size_t)nP * F * W);
    int p = 0;
    for (int i = half; i < T - half; i += shift, ++p) {
        int s = i - half;
        int e = s + W < T ? s + W : T;
        if (e - s < W) s = e - W;
        for (int f = 0; f < F; ++f)
            for (int j = 0; j < W; ++j)
                out[((size_t)p * F + f) * W + j] = (double)FV[(size_t)f * T + s + j];
    }
    return nP;
}
