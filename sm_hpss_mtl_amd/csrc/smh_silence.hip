// The step in front of the hot path: lib/preprocessing.py:330-350 load_and_preprocess_signal
// after the decode -- normalise, librosa.feature.rms, tools.removeSilence (lib/cython_impl/tools.pyx:42-134),
// normalise again -- batched over B equal-length clips that stay resident in HBM.
//
// All of it is streaming / integer work (a few passes over N samples per clip): the kernels are written for
// coalesced float loads and deterministic reductions, not for arithmetic.  The run detection is the reference's
// literal while-loop executed by one thread per clip (nFrames iterations of byte reads); everything around it is
// parallel over samples.
#include "smh_common.h"

#include <cstdlib>

namespace {

constexpr int kChunk = 8192;   // samples per workgroup in the streaming passes
constexpr int kThreads = 256;

struct SilWork {  // carved out of the caller's workspace
    double *psum;          // (B, nchunk) ordered partial sums
    float *mean;           // (B)
    unsigned *absmax;      // (B) float bits of max|x-mean|   (max is order independent -> atomics are deterministic)
    unsigned *emax;        // (B) float bits of max(energy)
    int *nrun;             // (B)
    int *nkeep;            // (B)
    int *runs;             // (B, maxrun, 3): k, l, samples removed before this run
    unsigned char *raw;    // (B, nF) energy >= threshold
    unsigned char *fmark;  // (B, nF) after medfilt(., 5)
    float *energy;         // (B, nF)
    float *xn;             // (B, N) normalised signal of the fused entry point
    int nchunk, nF, maxrun;
    size_t bytes;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

SilWork carve(void *base, int B, int N, int hop, bool with_signal = true) {
    SilWork w{};
    w.nchunk = (N + kChunk - 1) / kChunk;
    if (w.nchunk < 1) w.nchunk = 1;
    w.nF = hop > 0 ? 1 + N / hop : 1;
    w.maxrun = w.nF / 2 + 2;
    size_t off = 0;
    char *p = (char *)base;
    auto take = [&](size_t n) {
        char *q = p ? p + off : nullptr;
        off += align256(n);
        return (void *)q;
    };
    const size_t b = (size_t)(B > 0 ? B : 1);
    w.psum = (double *)take(2 * b * w.nchunk * sizeof(double));  // two rows per clip for mix_signals
    w.mean = (float *)take(b * sizeof(float));
    w.absmax = (unsigned *)take(b * sizeof(unsigned));
    w.emax = (unsigned *)take(b * sizeof(unsigned));
    w.nrun = (int *)take(b * sizeof(int));
    w.nkeep = (int *)take(b * sizeof(int));
    w.runs = (int *)take(b * w.maxrun * 3 * sizeof(int));
    w.raw = (unsigned char *)take(b * w.nF);
    w.fmark = (unsigned char *)take(b * w.nF);
    w.energy = (float *)take(b * w.nF * sizeof(float));
    w.xn = with_signal ? (float *)take(b * (size_t)N * sizeof(float)) : nullptr;
    w.bytes = off;
    return w;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- normalise: x -= mean(x); x /= max|x|  (preprocessing.py:332-333, 348-349) ---------------------------------
// pass 1: ordered f64 partial sums per chunk (fixed tree -> bit-reproducible from run to run)
__global__ __launch_bounds__(kThreads) void sum_chunks_kernel(const float *__restrict__ x, int N, int nchunk,
                                                              double *__restrict__ psum) {
    const int b = blockIdx.y, c = blockIdx.x;
    const float *xb = x + (size_t)b * N;
    const int lo = c * kChunk, hi = min(N, lo + kChunk);
    double s = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += kThreads) s += (double)xb[i];
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) t += sh[w];
        psum[(size_t)b * nchunk + c] = t;
    }
}

// pass 1b: one wave per clip folds the partials in a fixed order; also resets the max accumulators
__global__ __launch_bounds__(64) void mean_kernel(const double *__restrict__ psum, int N, int nchunk,
                                                  float *__restrict__ mean, unsigned *__restrict__ absmax) {
    const int b = blockIdx.x;
    double s = 0.0;
    for (int c = threadIdx.x; c < nchunk; c += 64) s += psum[(size_t)b * nchunk + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        mean[b] = (float)(s / (double)N);
        absmax[b] = 0u;
    }
}

// pass 2: max |x - mean| (float bits of a non-negative float order like unsigned integers)
__global__ __launch_bounds__(kThreads) void absmax_kernel(const float *__restrict__ x, int N,
                                                          const float *__restrict__ mean,
                                                          unsigned *__restrict__ absmax) {
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * N;
    const float mu = mean[b];
    const int lo = blockIdx.x * kChunk, hi = min(N, lo + kChunk);
    float m = 0.f;
    bool nan = false;
    for (int i = lo + threadIdx.x; i < hi; i += kThreads) {
        const float d = fabsf(xb[i] - mu);
        nan |= (d != d);
        m = fmaxf(m, d);
    }
    m = wave_max(m);
    if (__any(nan)) m = __uint_as_float(0x7fc00000u);  // np.max propagates NaN
    if ((threadIdx.x & 63) == 0) atomicMax(&absmax[b], __float_as_uint(m));
}

// pass 3: write (x - mean) / max  (IEEE division, like numpy)
__global__ __launch_bounds__(kThreads) void normalize_write_kernel(const float *__restrict__ x, int N,
                                                                   const float *__restrict__ mean,
                                                                   const unsigned *__restrict__ absmax,
                                                                   float *__restrict__ out) {
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * N;
    float *ob = out + (size_t)b * N;
    const float mu = mean[b];
    const float m = __uint_as_float(absmax[b]);
    const int lo = blockIdx.x * kChunk, hi = min(N, lo + kChunk);
    for (int i = lo + threadIdx.x; i < hi; i += kThreads) ob[i] = __fdiv_rn(xb[i] - mu, m);
}

// ---- librosa.feature.rms(y, frame_length, hop_length): centre=True, reflect padding ----------------------------
// One wave per frame; the clip maximum of the energy (tools.pyx:93) is folded in.
__global__ __launch_bounds__(kThreads) void rms_kernel(const float *__restrict__ y, int N, int frame_length, int hop,
                                                       int nF, float *__restrict__ energy,
                                                       unsigned *__restrict__ emax) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (t >= nF) return;
    const int lane = threadIdx.x & 63;
    const float *yb = y + (size_t)b * N;
    const int pad = frame_length / 2;
    const int base = t * hop - pad;
    float s = 0.f;
    for (int i = lane; i < frame_length; i += 64) {
        int src = base + i;
        if (src < 0) src = -src;
        if (src >= N) src = 2 * (N - 1) - src;
        const float v = yb[src];
        s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float e = sqrtf(s / (float)frame_length);
        energy[(size_t)b * nF + t] = e;
        if (emax) atomicMax(&emax[b], __float_as_uint(e));
    }
}

__global__ void zero_u32_kernel(unsigned *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

__global__ __launch_bounds__(kThreads) void energy_max_kernel(const float *__restrict__ energy, int nF,
                                                              unsigned *__restrict__ emax) {
    const int b = blockIdx.x;
    unsigned m = 0u;  // bit pattern order == value order for the non-negative energies; NaN sorts highest (np.max)
    for (int i = threadIdx.x; i < nF; i += kThreads) m = max(m, __float_as_uint(energy[(size_t)b * nF + i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&emax[b], m);
}

// ---- tools.removeSilence ------------------------------------------------------------------------------------
// One workgroup per clip: threshold marker, medfilt(marker, 5) with zero padding, then the reference's run loop.
__global__ __launch_bounds__(kThreads) void silence_runs_kernel(const float *__restrict__ energy, int nF, int N,
                                                                const unsigned *__restrict__ emax, double alpha,
                                                                double beta, int fs, int frameSize, int frameShift,
                                                                int maxrun, unsigned char *__restrict__ raw,
                                                                unsigned char *__restrict__ fmark,
                                                                int *__restrict__ frame_marker_out,
                                                                int *__restrict__ runs, int *__restrict__ nrun,
                                                                int *__restrict__ nkeep) {
    const int b = blockIdx.x;
    const float *e = energy + (size_t)b * nF;
    unsigned char *rw = raw + (size_t)b * nF;
    unsigned char *fm = fmark + (size_t)b * nF;
    // tools.pyx:93 `cdef float energyThresh = alpha * np.max(energy)`: double product (numpy 1.19 promotes the
    // float32 scalar times a Python float to float64) stored into a C float
    const float thresh = (float)(alpha * (double)__uint_as_float(emax[b]));
    for (int i = threadIdx.x; i < nF; i += kThreads) rw[i] = e[i] >= thresh ? 1 : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nF; i += kThreads) {
        int c = 0;
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
            const int j = i + d;
            c += (j >= 0 && j < nF) ? rw[j] : 0;
        }
        const unsigned char m = c >= 3 ? 1 : 0;  // median of five 0/1 values
        fm[i] = m;
        if (frame_marker_out) frame_marker_out[(size_t)b * nF + i] = m;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int *rb = runs + (size_t)b * maxrun * 3;
    int n = 0, removed = 0;
    int i = 0;
    while (i < nF) {  // tools.pyx:101-123, literally
        while (fm[i] == 1) {
            if (i == nF - 1) break;
            ++i;
        }
        int j = i;
        while (fm[j] == 0) {
            if (j == nF - 1) break;
            ++j;
        }
        const int k = max(frameShift * (i - 1) + frameSize, 1);
        const int l = min(frameShift * (j - 1) + frameSize, N);
        if ((double)(l - k) / (double)fs > beta && n < maxrun) {
            rb[3 * n] = k;
            rb[3 * n + 1] = l;
            rb[3 * n + 2] = removed;
            removed += l - k;
            ++n;
        }
        i = j + 1;
    }
    nrun[b] = n;
    nkeep[b] = n > 1 ? N - removed : N;
}

// Compaction: kept samples move left by the number of removed samples before them; the tail keeps the 1.0 of the
// reference's np.ones() initialisation (tools.pyx:126-131).  With fewer than two runs the input is returned as is.
__global__ __launch_bounds__(kThreads) void silence_compact_kernel(const float *__restrict__ x, int N, int maxrun,
                                                                   const int *__restrict__ runs,
                                                                   const int *__restrict__ nrun,
                                                                   const int *__restrict__ nkeep,
                                                                   float *__restrict__ out,
                                                                   unsigned char *__restrict__ sample_marker) {
    const int b = blockIdx.y;
    const int *rb = runs + (size_t)b * maxrun * 3;
    const int n = nrun[b];
    const int keep = nkeep[b];
    const float *xb = x + (size_t)b * N;
    float *ob = out + (size_t)b * N;
    const int lo = blockIdx.x * kChunk, hi = min(N, lo + kChunk);
    for (int s = lo + threadIdx.x; s < hi; s += kThreads) {
        // last run with k <= s
        int a = -1, z = n - 1;
        while (a < z) {
            const int mid = (a + z + 1) >> 1;
            if (rb[3 * mid] <= s) a = mid; else z = mid - 1;
        }
        bool in_run = false;
        int before = 0;
        if (a >= 0) {
            const int k = rb[3 * a], l = rb[3 * a + 1];
            in_run = s < l;
            before = rb[3 * a + 2] + (in_run ? 0 : l - k);
        }
        if (sample_marker) sample_marker[(size_t)b * N + s] = in_run ? 0 : 1;
        if (n > 1) {
            if (!in_run) ob[s - before] = xb[s];
            if (s >= keep) ob[s] = 1.0f;
        } else {
            ob[s] = xb[s];
        }
    }
}

// ---- fused form of preprocessing.py:332-349 for clips that fit in LDS (N <= kFusedMaxN) ------------------------
// One workgroup per clip: the clip is read from HBM once into LDS, every later pass (mean, max, rms frames, marker,
// run loop, compaction, second normalisation) works out of LDS, and the result is written once: 8 bytes of HBM
// traffic per sample instead of the ~40 of the multi-pass path.  Same arithmetic as the stand-alone kernels.
constexpr int kFusedThreads = 1024;
constexpr int kFusedWaves = kFusedThreads / 64;
constexpr int kFusedMaxN = 36000;  // 144 KB of samples + tables within the 160 KB of one CU

struct BlockScratch {
    double d[kFusedWaves];
    float f[kFusedWaves];
    int flag[kFusedWaves];
    float bcast_f[2];
    double bcast_d;
    int nrun, nkeep;
};

__device__ __forceinline__ double block_sum(double v, BlockScratch &sc) {
    v = wave_sum(v);
    __syncthreads();  // scratch reuse
    if ((threadIdx.x & 63) == 0) sc.d[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kFusedWaves; ++w) t += sc.d[w];  // every thread folds in the same fixed order
    return t;
}

// max with NaN propagation (np.max): returns NaN if any lane saw one
__device__ __forceinline__ float block_absmax(float m, bool nan, BlockScratch &sc) {
    m = wave_max(m);
    const int anynan = __any(nan) ? 1 : 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sc.f[threadIdx.x >> 6] = m;
        sc.flag[threadIdx.x >> 6] = anynan;
    }
    __syncthreads();
    float t = 0.f;
    int fl = 0;
#pragma unroll
    for (int w = 0; w < kFusedWaves; ++w) {
        t = fmaxf(t, sc.f[w]);
        fl |= sc.flag[w];
    }
    return fl ? __uint_as_float(0x7fc00000u) : t;
}

__global__ __launch_bounds__(kFusedThreads) void preprocess_fused_kernel(const float *__restrict__ x, int N,
                                                                         int frameSize, int frameShift, int nF,
                                                                         int maxrun, int fs, double alpha, double beta,
                                                                         float *__restrict__ out,
                                                                         int *__restrict__ n_keep) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float *xs = (float *)lds_raw;                       // (N) rounded up to a multiple of 4
    const int Np = (N + 3) & ~3;
    float *en = xs + Np;                                // (nF)
    int *runs = (int *)(en + nF);                       // (maxrun, 3)
    unsigned char *raw = (unsigned char *)(runs + 3 * maxrun);  // (nF)
    unsigned char *fm = raw + nF;                       // (nF)
    __shared__ BlockScratch sc;

    const int b = blockIdx.x, tid = threadIdx.x;
    const float *xb = x + (size_t)b * N;
    float *ob = out + (size_t)b * N;

    // 1. HBM -> LDS, f64 sum
    double s = 0.0;
    if ((N & 3) == 0) {
        const float4 *x4 = (const float4 *)xb;
        for (int i = tid; i < N / 4; i += kFusedThreads) {
            const float4 v = x4[i];
            ((float4 *)xs)[i] = v;
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
    } else {
        for (int i = tid; i < N; i += kFusedThreads) {
            const float v = xb[i];
            xs[i] = v;
            s += (double)v;
        }
    }
    const float mu = (float)(block_sum(s, sc) / (double)N);
    // 2. x - mean, max |.|
    float m = 0.f;
    bool nan = false;
    for (int i = tid; i < N; i += kFusedThreads) {
        const float d = xs[i] - mu;
        xs[i] = d;
        const float a = fabsf(d);
        nan |= (a != a);
        m = fmaxf(m, a);
    }
    const float mx = block_absmax(m, nan, sc);
    for (int i = tid; i < N; i += kFusedThreads) xs[i] = __fdiv_rn(xs[i], mx);
    __syncthreads();
    // 3. rms frames (one wave per frame, same lane order as rms_kernel)
    const int lane = tid & 63, wave = tid >> 6;
    const int pad = frameSize / 2;
    unsigned emax_bits = 0u;
    for (int t = wave; t < nF; t += kFusedWaves) {
        const int base = t * frameShift - pad;
        float q = 0.f;
        for (int i = lane; i < frameSize; i += 64) {
            int src = base + i;
            if (src < 0) src = -src;
            if (src >= N) src = 2 * (N - 1) - src;
            const float v = xs[src];
            q = fmaf(v, v, q);
        }
        q = wave_sum(q);
        const float e = sqrtf(q / (float)frameSize);
        if (lane == 0) en[t] = e;
        emax_bits = max(emax_bits, __float_as_uint(e) & 0x7fffffffu);
    }
    __syncthreads();
    if (lane == 0) sc.flag[wave] = (int)emax_bits;
    __syncthreads();
    unsigned eb = 0u;
#pragma unroll
    for (int w = 0; w < kFusedWaves; ++w) eb = max(eb, (unsigned)sc.flag[w]);
    const float thresh = (float)(alpha * (double)__uint_as_float(eb));
    // 4. marker, medfilt(., 5) with zero padding, the reference's run loop
    for (int i = tid; i < nF; i += kFusedThreads) raw[i] = en[i] >= thresh ? 1 : 0;
    __syncthreads();
    for (int i = tid; i < nF; i += kFusedThreads) {
        int c = 0;
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
            const int j = i + d;
            c += (j >= 0 && j < nF) ? raw[j] : 0;
        }
        fm[i] = c >= 3 ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0, removed = 0, i = 0;
        while (i < nF) {  // tools.pyx:101-123, literally
            while (fm[i] == 1) {
                if (i == nF - 1) break;
                ++i;
            }
            int j = i;
            while (fm[j] == 0) {
                if (j == nF - 1) break;
                ++j;
            }
            const int k = max(frameShift * (i - 1) + frameSize, 1);
            const int l = min(frameShift * (j - 1) + frameSize, N);
            if ((double)(l - k) / (double)fs > beta && n < maxrun) {
                runs[3 * n] = k;
                runs[3 * n + 1] = l;
                runs[3 * n + 2] = removed;
                removed += l - k;
                ++n;
            }
            i = j + 1;
        }
        sc.nrun = n;
        sc.nkeep = n > 1 ? N - removed : N;
    }
    __syncthreads();
    const int n = sc.nrun > 1 ? sc.nrun : 0;  // fewer than two runs: nothing is removed
    const int keep = sc.nkeep;
    if (tid == 0 && n_keep) n_keep[b] = keep;
    // destination of sample s after compaction, or -1 if it lies in a removed run
    auto dest = [&](int sidx) {
        int a = -1, z = n - 1;
        while (a < z) {
            const int mid = (a + z + 1) >> 1;
            if (runs[3 * mid] <= sidx) a = mid; else z = mid - 1;
        }
        if (a < 0) return sidx;
        const int k = runs[3 * a], l = runs[3 * a + 1];
        return sidx < l ? -1 : sidx - (runs[3 * a + 2] + (l - k));
    };
    // 5. second normalisation over [retained samples..., 1.0 tail]
    double s2 = 0.0;
    for (int i = tid; i < N; i += kFusedThreads)
        if (dest(i) >= 0) s2 += (double)xs[i];
    const double tail = (double)(N - keep);
    const float mu2 = (float)((block_sum(s2, sc) + tail) / (double)N);
    float m2 = 0.f;
    bool nan2 = false;
    for (int i = tid; i < N; i += kFusedThreads)
        if (dest(i) >= 0) {
            const float a = fabsf(xs[i] - mu2);
            nan2 |= (a != a);
            m2 = fmaxf(m2, a);
        }
    if (keep < N) m2 = fmaxf(m2, fabsf(1.0f - mu2));
    const float mx2 = block_absmax(m2, nan2, sc);
    const float tailv = __fdiv_rn(1.0f - mu2, mx2);
    for (int i = tid; i < N; i += kFusedThreads) {
        const int dsti = dest(i);
        if (dsti >= 0) ob[dsti] = __fdiv_rn(xs[i] - mu2, mx2);
        if (i >= keep) ob[i] = tailv;
    }
}

// ---- mix_signals (preprocessing.py:297-325): music looped to the speech length, scaled to the target SMR --------
// pass 1: ordered f64 partial sums of sp^2 and (looped mu)^2 per chunk
__global__ __launch_bounds__(kThreads) void mix_energy_kernel(const float *__restrict__ sp, const float *__restrict__ mu,
                                                              int N, int Nmu, int nchunk, double *__restrict__ psum) {
    const int b = blockIdx.y, c = blockIdx.x;
    const float *s = sp + (size_t)b * N, *m = mu + (size_t)b * Nmu;
    const int lo = c * kChunk, hi = min(N, lo + kChunk);
    double es = 0.0, em = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += kThreads) {
        const double a = s[i], v = m[i % Nmu];
        es += a * a;
        em += v * v;
    }
    __shared__ double sh[2][kThreads / 64];
    es = wave_sum(es), em = wave_sum(em);
    if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = es, sh[1][threadIdx.x >> 6] = em;
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) t += sh[threadIdx.x][w];
        psum[((size_t)b * 2 + threadIdx.x) * nchunk + c] = t;
    }
}

// pass 2: the two mixing factors per clip (float32, like the float32 audio of the reference)
__global__ __launch_bounds__(64) void mix_factor_kernel(const double *__restrict__ psum, int N, int nchunk,
                                                        const float *__restrict__ target_db, float *__restrict__ fac) {
    const int b = blockIdx.x;
    double es = 0.0, em = 0.0;
    for (int c = threadIdx.x; c < nchunk; c += 64) {
        es += psum[((size_t)b * 2) * nchunk + c];
        em += psum[((size_t)b * 2 + 1) * nchunk + c];
    }
    es = wave_sum(es), em = wave_sum(em);
    if (threadIdx.x == 0) {
        const double e_sp = es / N, e_mu = em / N;
        const double req = e_sp / pow(10.0, (double)target_db[b] / 10.0);
        double f_mu = sqrt(req / e_mu), f_sp = 1.0;
        const double sum = f_mu + f_sp;
        f_mu /= sum, f_sp /= sum;
        fac[2 * b] = (float)f_sp;
        fac[2 * b + 1] = (float)f_mu;
    }
}

// pass 3: mix = f_sp * sp + f_mu * mu  (two roundings, as numpy evaluates it)
__global__ __launch_bounds__(kThreads) void mix_write_kernel(const float *__restrict__ sp, const float *__restrict__ mu,
                                                             int N, int Nmu, const float *__restrict__ fac,
                                                             float *__restrict__ out) {
    const int b = blockIdx.y;
    const float *s = sp + (size_t)b * N, *m = mu + (size_t)b * Nmu;
    float *o = out + (size_t)b * N;
    const float fs = fac[2 * b], fm = fac[2 * b + 1];
    const int lo = blockIdx.x * kChunk, hi = min(N, lo + kChunk);
    for (int i = lo + threadIdx.x; i < hi; i += kThreads) o[i] = __fadd_rn(__fmul_rn(fs, s[i]), __fmul_rn(fm, m[i % Nmu]));
}

int check_common(const char *fn, const void *d_x, int B, int N) {
    SMH_REQUIRE(B >= 0 && B <= 65535, "%s: B must be in [0, 65535]", fn);
    SMH_REQUIRE(N >= 1, "%s: N must be >= 1", fn);
    SMH_REQUIRE(d_x || B == 0, "%s: null input", fn);
    return SMH_OK;
}

int launch_normalize(const float *d_x, int B, int N, float *d_out, const SilWork &w, hipStream_t st) {
    const dim3 grid(w.nchunk, B);
    hipLaunchKernelGGL(sum_chunks_kernel, grid, dim3(kThreads), 0, st, d_x, N, w.nchunk, w.psum);
    hipLaunchKernelGGL(mean_kernel, dim3(B), dim3(64), 0, st, w.psum, N, w.nchunk, w.mean, w.absmax);
    hipLaunchKernelGGL(absmax_kernel, grid, dim3(kThreads), 0, st, d_x, N, w.mean, w.absmax);
    hipLaunchKernelGGL(normalize_write_kernel, grid, dim3(kThreads), 0, st, d_x, N, w.mean, w.absmax, d_out);
    return smh::launch_status("normalize kernels");
}

int launch_rms(const float *d_y, int B, int N, int frame_length, int hop, int nF, float *d_energy, unsigned *emax,
               hipStream_t st) {
    if (emax) hipLaunchKernelGGL(zero_u32_kernel, dim3((B + 255) / 256), dim3(256), 0, st, emax, B);
    hipLaunchKernelGGL(rms_kernel, dim3((nF + 3) / 4, B), dim3(kThreads), 0, st, d_y, N, frame_length, hop, nF,
                       d_energy, emax);
    return smh::launch_status("rms_kernel");
}

int launch_remove(const float *d_x, int B, int N, const float *d_energy, int nF, int fs, int Tw, int Ts, double alpha,
                  double beta, float *d_out, unsigned char *d_sample_marker, int *d_frame_marker, int *d_n_keep,
                  const SilWork &w, bool emax_ready, hipStream_t st) {
    // tools.pyx:88-89: int((Tw*fs)/1000) -- true division, truncated
    const int frameSize = (int)((double)Tw * fs / 1000.0);
    const int frameShift = (int)((double)Ts * fs / 1000.0);
    if (!emax_ready) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.emax, B);
        hipLaunchKernelGGL(energy_max_kernel, dim3(B), dim3(kThreads), 0, st, d_energy, nF, w.emax);
    }
    hipLaunchKernelGGL(silence_runs_kernel, dim3(B), dim3(kThreads), 0, st, d_energy, nF, N, w.emax, alpha, beta, fs,
                       frameSize, frameShift, w.maxrun, w.raw, w.fmark, d_frame_marker, w.runs, w.nrun, w.nkeep);
    hipLaunchKernelGGL(silence_compact_kernel, dim3(w.nchunk, B), dim3(kThreads), 0, st, d_x, N, w.maxrun, w.runs,
                       w.nrun, w.nkeep, d_out, d_sample_marker);
    if (d_n_keep)
        (void)hipMemcpyAsync(d_n_keep, w.nkeep, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, st);
    return smh::launch_status("remove-silence kernels");
}

}  // namespace

extern "C" size_t smh_silence_workspace_bytes(int B, int N, int hop) {
    if (B < 0 || N < 1 || hop < 1) return 0;
    return carve(nullptr, B, N, hop).bytes;
}

extern "C" size_t smh_normalize_workspace_bytes(int B, int N) {
    if (B < 0 || N < 1) return 0;
    return carve(nullptr, B, N, N, false).bytes;
}

extern "C" int smh_normalize_f32(const float *d_x, int B, int N, float *d_out, void *d_work, size_t work_bytes,
                                 void *stream) {
    if (int rc = check_common("smh_normalize_f32", d_x, B, N)) return rc;
    if (B == 0) return SMH_OK;
    SMH_REQUIRE(d_out && d_work, "smh_normalize_f32: null output or workspace");
    const SilWork w = carve(d_work, B, N, N, false);
    if (work_bytes < w.bytes)
        return smh::set_error(SMH_E_WORKSPACE, "smh_normalize_f32: workspace too small (%zu < %zu bytes)", work_bytes,
                              w.bytes);
    return launch_normalize(d_x, B, N, d_out, w, (hipStream_t)stream);
}

extern "C" int smh_rms_f32(const float *d_y, int B, int N, int frame_length, int hop, float *d_energy, void *stream) {
    if (int rc = check_common("smh_rms_f32", d_y, B, N)) return rc;
    SMH_REQUIRE(frame_length >= 1 && hop >= 1, "smh_rms_f32: frame_length and hop must be >= 1");
    SMH_REQUIRE(N > frame_length / 2, "smh_rms_f32: reflect padding needs N > frame_length/2 (N=%d)", N);
    const int nF = 1 + (N + 2 * (frame_length / 2) - frame_length) / hop;  // util.frame of the padded signal
    if (B == 0) return nF;
    SMH_REQUIRE(d_energy, "smh_rms_f32: null output");
    if (int rc = launch_rms(d_y, B, N, frame_length, hop, nF, d_energy, nullptr, (hipStream_t)stream)) return rc;
    return nF;
}

extern "C" int smh_remove_silence_f32(const float *d_x, int B, int N, const float *d_energy, int nFrames, int fs,
                                      int Tw, int Ts, double alpha, double beta, float *d_out,
                                      unsigned char *d_sample_marker, int *d_frame_marker, int *d_n_keep,
                                      void *d_work, size_t work_bytes, void *stream) {
    if (int rc = check_common("smh_remove_silence_f32", d_x, B, N)) return rc;
    SMH_REQUIRE(nFrames >= 1 && fs >= 1 && Tw >= 1 && Ts >= 1, "smh_remove_silence_f32: bad frame parameters");
    if (B == 0) return SMH_OK;
    SMH_REQUIRE(d_energy && d_out && d_work, "smh_remove_silence_f32: null argument");
    SMH_REQUIRE(d_out != d_x, "smh_remove_silence_f32: in-place operation is not supported");
    const int hop = (int)((double)Ts * fs / 1000.0);
    SMH_REQUIRE(hop >= 1, "smh_remove_silence_f32: frame shift rounds to zero samples");
    const SilWork w = carve(d_work, B, N, hop);
    SMH_REQUIRE(nFrames <= w.nF, "smh_remove_silence_f32: nFrames=%d exceeds 1 + N/hop = %d", nFrames, w.nF);
    if (work_bytes < w.bytes)
        return smh::set_error(SMH_E_WORKSPACE, "smh_remove_silence_f32: workspace too small (%zu < %zu bytes)",
                              work_bytes, w.bytes);
    return launch_remove(d_x, B, N, d_energy, nFrames, fs, Tw, Ts, alpha, beta, d_out, d_sample_marker, d_frame_marker,
                         d_n_keep, w, false, (hipStream_t)stream);
}

extern "C" int smh_preprocess_signal_f32(const float *d_x, int B, int N, int fs, int Tw, int Ts, float *d_out,
                                         int *d_n_keep, void *d_work, size_t work_bytes, void *stream) {
    if (int rc = check_common("smh_preprocess_signal_f32", d_x, B, N)) return rc;
    SMH_REQUIRE(fs >= 1 && Tw >= 1 && Ts >= 1, "smh_preprocess_signal_f32: bad frame parameters");
    const int frameSize = (int)((double)Tw * fs / 1000.0);
    const int hop = (int)((double)Ts * fs / 1000.0);
    SMH_REQUIRE(frameSize >= 1 && hop >= 1, "smh_preprocess_signal_f32: frame size/shift round to zero samples");
    SMH_REQUIRE(N > frameSize / 2, "smh_preprocess_signal_f32: reflect padding needs N > frameSize/2 (N=%d)", N);
    if (B == 0) return SMH_OK;
    SMH_REQUIRE(d_out && d_work, "smh_preprocess_signal_f32: null output or workspace");
    SMH_REQUIRE(d_out != d_x, "smh_preprocess_signal_f32: in-place operation is not supported");
    hipStream_t st = (hipStream_t)stream;
    const int nF = 1 + (N + 2 * (frameSize / 2) - frameSize) / hop;
    const bool multipass = getenv("SMH_SILENCE_MULTIPASS") != nullptr;  // test hook: force the general path
    if (N <= kFusedMaxN && !multipass) {  // clip fits in LDS: one read, one write
        const int maxrun = nF / 2 + 2;
        const size_t lds = (size_t)((N + 3) & ~3) * 4 + (size_t)nF * 4 + (size_t)maxrun * 12 + 2 * (size_t)nF;
        if (lds <= 155 * 1024) {
            SMH_CHECK_HIP(hipFuncSetAttribute((const void *)preprocess_fused_kernel,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(preprocess_fused_kernel, dim3(B), dim3(kFusedThreads), lds, st, d_x, N, frameSize, hop, nF,
                               maxrun, fs, 0.025, 0.075, d_out, d_n_keep);
            return smh::launch_status("preprocess_fused_kernel");
        }
    }
    const SilWork w = carve(d_work, B, N, hop);
    if (work_bytes < w.bytes)
        return smh::set_error(SMH_E_WORKSPACE, "smh_preprocess_signal_f32: workspace too small (%zu < %zu bytes)",
                              work_bytes, w.bytes);
    SMH_REQUIRE(nF <= w.nF, "smh_preprocess_signal_f32: internal frame count mismatch");
    if (int rc = launch_normalize(d_x, B, N, w.xn, w, st)) return rc;                       // :332-333
    if (int rc = launch_rms(w.xn, B, N, frameSize, hop, nF, w.energy, w.emax, st)) return rc;  // :338
    if (int rc = launch_remove(w.xn, B, N, w.energy, nF, fs, Tw, Ts, 0.025, 0.075, d_out, nullptr, nullptr, d_n_keep, w,
                               true, st))
        return rc;                                                                          // :339
    return launch_normalize(d_out, B, N, d_out, w, st);                                     // :348-349
}

extern "C" int smh_mix_signals_f32(const float *d_sp, const float *d_mu, int B, int N, int N_mu, const float *d_target_db,
                                   float *d_out, void *d_work, size_t work_bytes, void *stream) {
    if (int rc = check_common("smh_mix_signals_f32", d_sp, B, N)) return rc;
    SMH_REQUIRE(N_mu >= 1, "smh_mix_signals_f32: empty music signal");
    if (B == 0) return SMH_OK;
    SMH_REQUIRE(d_mu && d_target_db && d_out && d_work, "smh_mix_signals_f32: null argument");
    const SilWork w = carve(d_work, B, N, N, false);
    if (work_bytes < w.bytes)
        return smh::set_error(SMH_E_WORKSPACE, "smh_mix_signals_f32: workspace too small (%zu < %zu bytes)", work_bytes,
                              w.bytes);
    hipStream_t st = (hipStream_t)stream;
    float *fac = reinterpret_cast<float *>(w.runs);  // 2 floats per clip; the run table is unused here
    const dim3 grid(w.nchunk, B);
    hipLaunchKernelGGL(mix_energy_kernel, grid, dim3(kThreads), 0, st, d_sp, d_mu, N, N_mu, w.nchunk, w.psum);
    hipLaunchKernelGGL(mix_factor_kernel, dim3(B), dim3(64), 0, st, w.psum, N, w.nchunk, d_target_db, fac);
    hipLaunchKernelGGL(mix_write_kernel, grid, dim3(kThreads), 0, st, d_sp, d_mu, N, N_mu, fac, d_out);
    if (int rc = smh::launch_status("mix kernels")) return rc;
    return launch_normalize(d_out, B, N, d_out, w, st);  // preprocessing.py:323 normalize_signal(Xin_mix)
}
