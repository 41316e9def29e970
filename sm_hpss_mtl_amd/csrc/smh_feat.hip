// a3-a8: soft masks, mel projection, power_to_db, per-row standardisation, patch extraction.
// Stand-alone entry points mirror the reference's individual calls (parity API); the two fused kernels
// at the bottom (hp_feat_kernel, std_patch_kernel) are what smh_frontend_f32 runs.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <type_traits>

#include "smh_common.h"
#include "smh_feat.h"
#include "smh_rag.h"

namespace {

constexpr float kAmin = 1e-10f;   // librosa.power_to_db amin
constexpr float kTopDb = 80.0f;   // librosa.power_to_db top_db

// librosa.util.softmask(X, X_ref, power=2, split_zeros=True) for both orientations at once
// (called from librosa.decompose.hpss with margin 1; lib/preprocessing.py:408,418,430,440).
// Same float32 operation order as numpy: Z=max; bad=Z<tiny -> Z=1; (X/Z)^2; m/(m+r); bad -> 0.5.
__device__ __forceinline__ void hpss_masks(float s, float h, float p, float &H, float &P) {
    float Z = fmaxf(h, p);
    const bool bad = Z < FLT_MIN;
    Z = bad ? 1.0f : Z;
    const float a = h / Z, b = p / Z;
    const float m = __fmul_rn(a, a), r = __fmul_rn(b, b);  // no FMA contraction: numpy rounds the squares
    const float den = __fadd_rn(m, r);
    float mh = m / den, mp = r / den;
    mh = bad ? 0.5f : mh;
    mp = bad ? 0.5f : mp;
    H = __fmul_rn(s, mh);
    P = __fmul_rn(s, mp);
}

// Same masks for the FUSED path: hardware reciprocals (1 ulp) instead of IEEE divisions.  H and P are not
// outputs there; the dB features they feed are compared at 1e-3 dB (SURVEY 8d').
__device__ __forceinline__ void hpss_masks_fast(float s, float h, float p, float &H, float &P) {
    float Z = fmaxf(h, p);
    const bool bad = Z < FLT_MIN;
    Z = bad ? 1.0f : Z;
    const float iz = __builtin_amdgcn_rcpf(Z);
    const float a = h * iz, b = p * iz;
    const float m = a * a, r = b * b;
    const float id = __builtin_amdgcn_rcpf(m + r);
    float mh = m * id, mp = r * id;
    mh = bad ? 0.5f : mh;
    mp = bad ? 0.5f : mp;
    H = s * mh;
    P = s * mp;
}

__global__ void softmask_kernel(const float *__restrict__ S, const float *__restrict__ harm,
                                const float *__restrict__ perc, size_t n, float *__restrict__ H,
                                float *__restrict__ P) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float hv, pv;
        hpss_masks(S[i], harm[i], perc[i], hv, pv);
        H[i] = hv;
        P[i] = pv;
    }
}

// librosa.feature.melspectrogram(S=X): mel_basis @ X with the sparse (CSR-by-row) filterbank.
__global__ void mel_kernel(smh_feat::MelTable mel, const float *__restrict__ X, int K, int T, float *__restrict__ Y) {
    const int b = blockIdx.y;
    const int n = mel.n_mels * T;
    const float *Xb = X + (size_t)b * K * T;
    float *Yb = Y + (size_t)b * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int m = i / T, t = i - m * T;
        const int k0 = mel.start[m], cnt = mel.count[m];
        const float *w = mel.w + mel.off[m];
        float acc = 0.f;
        for (int j = 0; j < cnt; ++j) acc = fmaf(w[j], Xb[(size_t)(k0 + j) * T + t], acc);
        Yb[i] = acc;
    }
}

__device__ __forceinline__ float db_of_sq(float x) {
    const float p = __fmul_rn(x, x);
    return 10.0f * log10f(fmaxf(kAmin, p));
}
// fused path: 10*log10(p) = 3.0103*log2(p) on the hardware log (abs error ~1e-6 dB)
__device__ __forceinline__ float db_of_sq_fast(float x) {
    return 3.0102999566398120f * __builtin_amdgcn_logf(fmaxf(kAmin, x * x));
}

template <typename Tv>
__device__ __forceinline__ Tv block_reduce_max(Tv v, Tv *scratch) {
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    Tv r = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = max(r, scratch[w]);
    return r;
}

// librosa.core.power_to_db(X**2): one workgroup per array (the top_db reference max is per array).
__global__ void power_to_db_kernel(const float *__restrict__ X, int elems, float *__restrict__ Y) {
    __shared__ float scratch[16];
    const float *x = X + (size_t)blockIdx.x * elems;
    float *y = Y + (size_t)blockIdx.x * elems;
    float lmax = -FLT_MAX;
    for (int i = threadIdx.x; i < elems; i += blockDim.x) {
        const float d = db_of_sq(x[i]);
        y[i] = d;
        lmax = fmaxf(lmax, d);
    }
    const float thr = block_reduce_max(lmax, scratch) - kTopDb;
    // each thread revisits exactly the elements it wrote itself
    for (int i = threadIdx.x; i < elems; i += blockDim.x) y[i] = fmaxf(y[i], thr);
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// sklearn StandardScaler statistics of one row held by one wave: float64 mean / population variance,
// (near-)constant rows are left unscaled (sklearn _is_constant_feature / _handle_zeros_in_scale).
template <typename Load>
__device__ __forceinline__ void row_stats(int T, int lane, Load &&load, double &mean, double &scale) {
    double s = 0.0;
    for (int t = lane; t < T; t += 64) s += (double)load(t);
    mean = wave_sum(s) / (double)T;
    double q = 0.0;
    for (int t = lane; t < T; t += 64) {
        const double d = (double)load(t) - mean;
        q += d * d;
    }
    const double var = wave_sum(q) / (double)T;
    const double eps = 2.220446049250313e-16;
    const double nm = (double)T * mean * eps;
    const bool constant = var <= (double)T * eps * var + nm * nm;
    scale = sqrt(var);
    if (constant || scale == 0.0) scale = 1.0;
}

// `X -= mean; X /= scale` applied to a float32 array with float64 statistics: two roundings.
__device__ __forceinline__ float standardize(float x, double mean, double scale) {
    const float c = (float)((double)x - mean);
    return (float)((double)c / scale);
}

__global__ void standardize_rows_kernel(const float *__restrict__ X, int n_rows, int T, float *__restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float *x = X + (size_t)row * T;
    float *y = Y + (size_t)row * T;
    double mean, scale;
    row_stats(T, lane, [&](int t) { return x[t]; }, mean, scale);
    for (int t = lane; t < T; t += 64) y[t] = standardize(x[t], mean, scale);
}

// tools.extract_patches on the (virtually) tiled featuregram: frame index taken modulo T.
__global__ void extract_patches_kernel(const float *__restrict__ FV, int F, int T, int Ttiled, int W, int shift, int nP,
                                       int layout, float *__restrict__ out) {
    const int b = blockIdx.y;
    const size_t per_clip = (size_t)nP * F * W;
    const float *fv = FV + (size_t)b * F * T;
    float *o = out + (size_t)b * per_clip;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += (size_t)gridDim.x * blockDim.x) {
        int p, f, j;
        if (layout == 0) {  // (nP, F, W)
            j = (int)(i % W);
            f = (int)((i / W) % F);
            p = (int)(i / ((size_t)W * F));
        } else {  // (nP, W, F)
            f = (int)(i % F);
            j = (int)((i / F) % W);
            p = (int)(i / ((size_t)W * F));
        }
        const int half = W / 2;
        int s = p * shift;  // centre half + p*shift, start = centre - half
        const int e = min(s + W, Ttiled);
        if (e - s < W) s = e - W;
        (void)half;
        o[i] = fv[(size_t)f * T + (s + j) % T];
    }
}

// ---------------------------------------------------------------------------------------------------
// fused fast path, kernel 1: (S, harm, perc) -> soft masks -> mel -> dB (un-clipped), grid (frame slabs, B).
// One thread per output (mel row m, frame t), frames fastest, so S / perc rows are read coalesced and
// each thread recomputes the masks of the <= 11 bins of its filter (every bin is shared by ~2 filters:
// the re-reads are L1/L2 hits, HBM traffic stays 1x).  `harm` may arrive time-major (B,T,K) from the
// median kernel (coalesced harmonic stores); the slab is then transposed through LDS (odd row stride).
// The per-array top_db maximum is accumulated with one atomicMax per workgroup and applied by
// std_patch_kernel, which owns whole featuregram halves.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ordered_key(float f) {
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float key_to_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

constexpr int kFeatThreads = 512;

__global__ void __launch_bounds__(kFeatThreads)
hp_feat_kernel(smh_feat::MelTable mel, int log_db, const float *__restrict__ S, const float *__restrict__ harm,
               const float *__restrict__ perc, int harm_tmajor, int K, int T, int TS, int rows, float *__restrict__ fv,
               int *__restrict__ maxkeys) {
    extern __shared__ __attribute__((aligned(16))) float hs[];  // [TS][KP] harmonic slab (time-major input only)
    __shared__ int smax[2 * (kFeatThreads / 64)];
    __shared__ int s_start[smh_feat::kMaxMels], s_cnt[smh_feat::kMaxMels], s_off[smh_feat::kMaxMels];
    __shared__ float s_w[smh_feat::kMaxMelNnz];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TS;
    const int nt = min(TS, T - t0);
    const int KP = K | 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const size_t cb = (size_t)b * K * T;
    for (int i = threadIdx.x; i < mel.n_mels; i += blockDim.x) s_start[i] = mel.start[i], s_cnt[i] = mel.count[i], s_off[i] = mel.off[i];
    for (int i = threadIdx.x; i < mel.nnz; i += blockDim.x) s_w[i] = mel.w[i];
    if (!harm_tmajor) __syncthreads();
    if (harm_tmajor) {
        constexpr int kB = 4;
        for (int c0 = wave; c0 < nt; c0 += nw * kB) {
            for (int kb = 0; kb < K; kb += 64) {
                float v[kB];
                const int k = kb + lane;
#pragma unroll
                for (int r = 0; r < kB; ++r) {
                    const int c = min(c0 + r * nw, nt - 1);
                    v[r] = harm[cb + (size_t)(t0 + c) * K + min(k, K - 1)];
                }
#pragma unroll
                for (int r = 0; r < kB; ++r) {
                    const int c = c0 + r * nw;
                    if (c < nt && k < K) hs[c * KP + k] = v[r];
                }
            }
        }
        __syncthreads();
    }
    float mxH = -FLT_MAX, mxP = -FLT_MAX;
    float *fvH = fv + (size_t)b * 2 * rows * T + t0;
    float *fvP = fvH + (size_t)rows * T;
    const float *Sb = S + cb + t0, *Pb = perc + cb + t0, *Hb = harm + cb + t0;
    for (int i = threadIdx.x; i < rows * nt; i += blockDim.x) {
        const int m = i / nt, c = i - m * nt;
        int k0 = m, cnt = 1;
        const float *w = nullptr;
        if (mel.n_mels > 0) {
            k0 = s_start[m], cnt = s_cnt[m];
            w = s_w + s_off[m];
        }
        float aH = 0.f, aP = 0.f;
        for (int j = 0; j < cnt; j += 4) {
            float sv[4], pv[4], hv[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // four taps of loads in flight
                const int jj = min(j + u, cnt - 1);
                const size_t g = (size_t)(k0 + jj) * T + c;
                sv[u] = Sb[g];
                pv[u] = Pb[g];
                hv[u] = harm_tmajor ? hs[c * KP + k0 + jj] : Hb[g];
                wv[u] = (j + u < cnt) ? (w ? w[jj] : 1.0f) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float H, P;
                hpss_masks_fast(sv[u], hv[u], pv[u], H, P);
                aH = fmaf(wv[u], H, aH);
                aP = fmaf(wv[u], P, aP);
            }
        }
        if (log_db) {
            aH = db_of_sq_fast(aH);
            aP = db_of_sq_fast(aP);
            mxH = fmaxf(mxH, aH);
            mxP = fmaxf(mxP, aP);
        }
        fvH[(size_t)m * T + c] = aH;
        fvP[(size_t)m * T + c] = aP;
    }
    if (log_db) {
        int kH = ordered_key(mxH), kP = ordered_key(mxP);
        for (int off = 32; off > 0; off >>= 1) {
            kH = max(kH, __shfl_xor(kH, off));
            kP = max(kP, __shfl_xor(kP, off));
        }
        if (lane == 0) smax[wave] = kH, smax[nw + wave] = kP;
        __syncthreads();
        if (threadIdx.x == 0) {
            int a = smax[0], c = smax[nw];
            for (int q = 1; q < nw; ++q) a = max(a, smax[q]), c = max(c, smax[nw + q]);
            atomicMax(&maxkeys[2 * b], a);
            atomicMax(&maxkeys[2 * b + 1], c);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fused fast path, kernel 1 in its bin-walk form (the default): one workgroup per clip, lane <-> frame.  A wave
// walks the bins of its row segment ONCE: the soft masks of a bin are evaluated once (not once per filter tap) and
// added into the accumulators of the at most four filters pending at that bin; a filter is emitted (dB, store,
// running maximum) when the walk has passed its last bin.  Which filters are pending, their weights and the emit
// points are the same for every lane, so they are precomputed per context as a per-bin plan
// {w0, w1, w2, w3, n_emit} (smh_ctx.hip) that the wave reads with scalar loads, eight bins ahead together with the
// vector loads of S and perc -- nothing in the loop depends on a table lookup.  S and perc rows are read coalesced
// along frames, every element once; the time-major harm clip is copied to LDS (odd row stride) and read column-wise.
// ---------------------------------------------------------------------------------------------------
using smh_feat::FeatPlan;
constexpr int kWalkBatch = 8;  // bins of loads in flight per lane
constexpr int kHalfBatch = 8;   // the same in features_half_kernel (lane = frame pair)
constexpr int kL0Steps = 30;    // k steps of the layer-0 products (four rows each) that features_half_kernel's per-M-tile form is built for: 120 mel rows

__global__ void __launch_bounds__(1024)
hp_feat_walk_kernel(FeatPlan fp, int log_db, const float *__restrict__ S, const float *__restrict__ harm,
                    const float *__restrict__ perc, int harm_tmajor, int K, int T, int rows, float *__restrict__ fv,
                    int *__restrict__ maxkeys) {
    // LDS: [harm clip T x KP (time-major input only)] [32 ints]: 98 x 201 x 4 + 128 = 78 920 B, two workgroups per CU
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int b = blockIdx.x;
    const int KP = K | 1;
    int *smax = reinterpret_cast<int *>(hs + (harm_tmajor ? (size_t)T * KP : 0));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const size_t cb = (size_t)b * K * T;
    if (harm_tmajor) {  // straight copy, rows of K floats -> rows of KP
        constexpr int kB = 4;
        for (int c0 = wave; c0 < T; c0 += nw * kB) {
            for (int kb = 0; kb < K; kb += 64) {
                float v[kB];
                const int k = kb + lane;
#pragma unroll
                for (int r = 0; r < kB; ++r) v[r] = harm[cb + (size_t)min(c0 + r * nw, T - 1) * K + min(k, K - 1)];
#pragma unroll
                for (int r = 0; r < kB; ++r) {
                    const int c = c0 + r * nw;
                    if (c < T && k < K) hs[c * KP + k] = v[r];
                }
            }
        }
        __syncthreads();
    }

    const int nwt = (T + 63) >> 6;  // waves per segment
    float mxH = -FLT_MAX, mxP = -FLT_MAX;
    for (int task = wave; task < fp.nseg * nwt; task += nw) {
        const int seg = __builtin_amdgcn_readfirstlane(task / nwt);
        const int tw = task - seg * nwt;
        const int t = tw * 64 + lane;
        const bool active = t < T;
        const int tc = min(t, T - 1);
        const int m1 = fp.m1[seg], kbeg = fp.kbeg[seg], kend = fp.kend[seg];
        int mcur = fp.m0[seg];
        const float *plan = fp.plan + fp.off[seg];
        float aH[4] = {0.f, 0.f, 0.f, 0.f}, aP[4] = {0.f, 0.f, 0.f, 0.f};
        float *fvH = fv + (size_t)b * 2 * rows * T + tc;
        float *fvP = fvH + (size_t)rows * T;
        auto emit_first = [&]() {
            float vH = aH[0], vP = aP[0];
            if (log_db) {
                vH = db_of_sq_fast(vH), vP = db_of_sq_fast(vP);
                mxH = fmaxf(mxH, vH), mxP = fmaxf(mxP, vP);
            }
            if (active) {
                fvH[(size_t)mcur * T] = vH;
                fvP[(size_t)mcur * T] = vP;
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) aH[e] = aH[e + 1], aP[e] = aP[e + 1];
            aH[3] = aP[3] = 0.f;
            ++mcur;
        };
        const float *Sb = S + cb + tc, *Pb = perc + cb + tc, *Hb = harm + cb + tc;
        const float *hrow = hs + tc * KP;
        for (int k0 = kbeg; k0 < kend; k0 += kWalkBatch) {
            float sv[kWalkBatch], pv[kWalkBatch], hv[kWalkBatch];
            float4 wq[kWalkBatch];
            int ne[kWalkBatch];
#pragma unroll
            for (int u = 0; u < kWalkBatch; ++u) {
                const int kk = min(k0 + u, K - 1);
                sv[u] = Sb[(size_t)kk * T];
                pv[u] = Pb[(size_t)kk * T];
                hv[u] = harm_tmajor ? hrow[kk] : Hb[(size_t)kk * T];
                const int pi = min(k0 + u, kend - 1) - kbeg;  // wave-uniform: scalar loads
                wq[u] = *reinterpret_cast<const float4 *>(plan + (size_t)pi * 8);
                ne[u] = __float_as_int(plan[(size_t)pi * 8 + 4]);
            }
#pragma unroll
            for (int u = 0; u < kWalkBatch; ++u) {
                if (k0 + u >= kend) break;
                for (int i = 0; i < ne[u]; ++i) emit_first();  // wave-uniform trip count
                float H, P;
                hpss_masks_fast(sv[u], hv[u], pv[u], H, P);
                aH[0] = fmaf(wq[u].x, H, aH[0]), aP[0] = fmaf(wq[u].x, P, aP[0]);
                aH[1] = fmaf(wq[u].y, H, aH[1]), aP[1] = fmaf(wq[u].y, P, aP[1]);
                aH[2] = fmaf(wq[u].z, H, aH[2]), aP[2] = fmaf(wq[u].z, P, aP[2]);
                aH[3] = fmaf(wq[u].w, H, aH[3]), aP[3] = fmaf(wq[u].w, P, aP[3]);
            }
        }
        while (mcur < m1) emit_first();
    }
    if (log_db) {
        int kH = ordered_key(mxH), kP = ordered_key(mxP);
        for (int off = 32; off > 0; off >>= 1) {
            kH = max(kH, __shfl_xor(kH, off));
            kP = max(kP, __shfl_xor(kP, off));
        }
        if (lane == 0) smax[wave] = kH, smax[16 + wave] = kP;
        __syncthreads();
        if (threadIdx.x == 0) {
            int a = smax[0], c = smax[16];
            for (int q = 1; q < nw; ++q) a = max(a, smax[q]), c = max(c, smax[16 + q]);
            atomicMax(&maxkeys[2 * b], a);
            atomicMax(&maxkeys[2 * b + 1], c);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fused fast path, kernel 2: one workgroup per (clip, half): top_db clip (written back: the final
// featuregram) -> StandardScaler per row -> time-major patches (B*nP, W, 2*rows) for the TCN.
// ---------------------------------------------------------------------------------------------------
constexpr int kPatchThreads = 1024;

template <bool L0>
__global__ void __launch_bounds__(kPatchThreads)
std_patch_kernel(int log_db, float *__restrict__ fv, const int *__restrict__ maxkeys, int rows, int T, int Ttiled, int W,
                 int shift, int nP, float *__restrict__ patches, const float *__restrict__ w0, float *__restrict__ x0p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.y, half = blockIdx.x;
    const int ld = T | 1;
    float *g = fv + ((size_t)b * 2 + half) * rows * T;
    const float thr = log_db ? key_to_float(maxkeys[2 * b + half]) - kTopDb : -FLT_MAX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // rows in batches: loads first, then clip / write-back / LDS
    constexpr int kB = 4;
    for (int r0 = wave; r0 < rows; r0 += nw * kB) {
        for (int tb = 0; tb < T; tb += 64) {
            float v[kB];
            const int t = tb + lane;
#pragma unroll
            for (int q = 0; q < kB; ++q) v[q] = g[(size_t)min(r0 + q * nw, rows - 1) * T + min(t, T - 1)];
#pragma unroll
            for (int q = 0; q < kB; ++q) {
                const int r = r0 + q * nw;
                if (r < rows && t < T) {
                    float x = v[q];
                    if (log_db) {
                        x = fmaxf(x, thr);
                        g[(size_t)r * T + t] = x;
                    }
                    tile[r * ld + t] = x;
                }
            }
        }
    }
    __syncthreads();
    if (!patches && !(L0 && x0p)) return;
    // StandardScaler statistics: ONE THREAD PER ROW (rows are short; a wave-wide f64 reduction per row
    // costs ~20x more instructions).  LDS reads are conflict-free: consecutive rows, odd row stride.
    float *s_mean = tile + (size_t)rows * ld, *s_inv = s_mean + rows;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const float *row = tile + r * ld;
        double sum = 0.0;
        for (int t = 0; t < T; ++t) sum += (double)row[t];
        const double mean = sum / (double)T;
        double q = 0.0;
        for (int t = 0; t < T; ++t) {
            const double dlt = (double)row[t] - mean;
            q += dlt * dlt;
        }
        const double var = q / (double)T;
        const double eps = 2.220446049250313e-16;
        const double nm = (double)T * mean * eps;
        const bool constant = var <= (double)T * eps * var + nm * nm;  // sklearn _is_constant_feature
        double scale = sqrt(var);
        if (constant || scale == 0.0) scale = 1.0;
        s_mean[r] = (float)mean;
        // keep the f64 mean exactly: store the low part too (mean = hi + lo)
        s_inv[r] = (float)(1.0 / scale);
        s_mean[rows + rows + r] = (float)(mean - (double)(float)mean);
    }
    __syncthreads();
    const float *s_lo = s_mean + 2 * rows;
    const int F = 2 * rows;
    if constexpr (L0) {
        // Layer 0 of B3_MTL fused in (smh_features_l0_f32): this half's share of Conv1D(32, 1) on the standardised
        // patch, D[channel][frame] = sum_r W0[half*rows + r][channel] * xstd[r][frame], exact-f32 MFMA straight from
        // the LDS tile.  One (patch, 16-frame tile, channel half) per wave; the two halves of a clip write separate
        // partial images (B*nP, 2, W, 32) that the network kernel adds (no atomics, no zero fill).  A separate
        // instantiation: the plain kernel keeps its register count.
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const int q = lane >> 4, j = lane & 15;
        const int ut = (W + 15) >> 4;
        constexpr int kG = 8;  // weight loads in flight per group (8 measured best of 8 / 15 / 32: registers cost occupancy)
        for (int task = wave; task < nP * ut * 2; task += nw) {
            const int mt = task & 1, pu = task >> 1;
            const int p = pu / ut, u = pu - p * ut;
            int s = p * shift;
            const int e = min(s + W, Ttiled);
            if (e - s < W) s = e - W;
            const int jt = 16 * u + j;                   // frame inside the patch
            int tt = s + min(jt, W - 1);
            tt -= (tt / T) * T;
            const float *wr = w0 + ((size_t)half * rows + q) * 32 + 16 * mt + j;  // W0[f = half*rows + 4 st + q][c]
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
            const int nst = rows / 4;
            for (int s0 = 0; s0 < nst; s0 += kG) {
                float wa[kG];
#pragma unroll
                for (int g = 0; g < kG; ++g) wa[g] = wr[(size_t)(4 * min(s0 + g, nst - 1)) * 32];
#pragma unroll
                for (int g = 0; g < kG; ++g) {
                    if (s0 + g < nst) {
                        const int r = 4 * (s0 + g) + q;
                        const float c = (float)((double)tile[r * ld + tt] - ((double)s_mean[r] + (double)s_lo[r]));
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g], c * s_inv[r], c0, 0, 0, 0);
                    }
                }
            }
            if (jt < W)
                *reinterpret_cast<f32x4 *>(x0p + ((((size_t)b * nP + p) * 2 + half) * W + jt) * 32 + 16 * mt + 4 * q) = c0;
        }
    }
    if (!patches) return;
    for (int p = 0; p < nP; ++p) {
        int s = p * shift;
        const int e = min(s + W, Ttiled);
        if (e - s < W) s = e - W;
        float *o = patches + ((size_t)b * nP + p) * W * F + (size_t)half * rows;
        // one wave per frame j: lanes run over the feature rows (coalesced 480-byte stores)
        for (int j = wave; j < W; j += nw) {
            int tt = s + j;
            tt -= (tt / T) * T;
            for (int f = lane; f < rows; f += 64) {
                // (x - mean) rounded to f32 as sklearn does (mean carried as hi + lo), then * 1/scale
                const float c = (float)((double)tile[f * ld + tt] - ((double)s_mean[f] + (double)s_lo[f]));
                o[(size_t)j * F + f] = c * s_inv[f];
            }
        }
    }
}

// long clips (featuregram half larger than an LDS tile): top_db clip in place, one workgroup column per array
__global__ void clip_fv_kernel(float *__restrict__ fv, const int *__restrict__ maxkeys, size_t elems) {
    const int arr = blockIdx.y;  // 2 * b + half
    const float thr = key_to_float(maxkeys[arr]) - kTopDb;
    float *g = fv + (size_t)arr * elems;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x)
        g[i] = fmaxf(g[i], thr);
}

// ---------------------------------------------------------------------------------------------------
// fused fast path as ONE kernel per clip (harm in the 16-frame blocked layout written by the block-split median
// kernels, clips whose whole featuregram fits in LDS).  The bin walk of hp_feat_walk_kernel -- 8 row segments x 2
// waves = 16 waves, every lane a frame -- emits the un-clipped dB rows into an LDS image of the featuregram; the
// per-array maximum is then known inside the workgroup, so the top-dB clip, the StandardScaler statistics, the final
// featuregram (its only trip to HBM), the standardised patches and / or the network's first layer (section 12 of
// DESIGN.md) follow from LDS.  No harm tile: lane t reads harm[t/16][k][t%16], 64-byte runs per bin.
// HBM traffic per clip: S + harm + perc in, featuregram + layer-0 partials out = 360 KB instead of 566 KB.
// LDS: image [2*rows][T|1] + 3 floats per row + 32 ints  (98 frames, 240 rows: 98 KB, one workgroup per CU).
// ---------------------------------------------------------------------------------------------------
// RAG (smh_rag.h): workgroup i takes clip list[i] of a ragged call -- T, the tiled length, the patch count and every buffer offset
// come from that clip's descriptor; the arithmetic is the equal-length instantiation's, so a clip gets the same bits in both.
template <bool RAG>
__global__ void __launch_bounds__(1024)
features_clip_kernel(FeatPlan fp, int log_db, int stop_after /* tuning: phase probe */, const float *__restrict__ S, const float *__restrict__ harmb,
                     const float *__restrict__ perc, int K, int T, int rows, int Ttiled, int W, int shift, int nP,
                     float *__restrict__ fv, float *__restrict__ patches, const float *__restrict__ w0,
                     float *__restrict__ x0p, const smh_rag::Clip *__restrict__ rag, const int *__restrict__ list) {
    extern __shared__ __attribute__((aligned(16))) float img[];  // [R2][ld]
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int b = blockIdx.x;
    size_t cb, hb, fvb, pb;  // this clip's S / perc, blocked harm, featuregram (floats) and first patch
    if constexpr (RAG) {
        const smh_rag::Clip &c = rag[list[b]];
        T = c.T, Ttiled = c.Ttiled, nP = c.nP;
        cb = (size_t)c.spec_off, hb = (size_t)c.harm_off, fvb = (size_t)c.fv_off, pb = (size_t)c.patch_off;
    } else {
        cb = (size_t)b * K * T, hb = (size_t)b * ((T + 15) >> 4) * K * 16, fvb = (size_t)b * 2 * rows * T, pb = (size_t)b * nP;
    }
    const int ld = T | 1, R2 = 2 * rows;
    float *s_mean = img + (size_t)R2 * ld;  // mean hi [R2], 1/scale [R2], mean lo [R2]
    float *s_inv = s_mean + R2, *s_lo = s_mean + 2 * R2;
    int *smax = reinterpret_cast<int *>(s_mean + 3 * (size_t)R2);  // 32 ints
    float *w0s = s_mean + 3 * (size_t)R2 + 32;  // layer-0 weights [R2][32] (x0p only)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const float *hclip = harmb + hb;
    float mxH = 0.f, mxP = 0.f;  // maxima of the filter sums (sums of non-negative terms)

    // ---- the bin walk, lane = frame (odd T) ----
    const int nwt = (T + 63) >> 6;
    for (int task = wave; task < fp.nseg * nwt; task += nw) {
        const int seg = __builtin_amdgcn_readfirstlane(task / nwt);
        const int tw = task - seg * nwt;
        const int t = tw * 64 + lane;
        const bool active = t < T;
        const int tc = min(t, T - 1);
        const int m1 = fp.m1[seg], kbeg = fp.kbeg[seg], kend = fp.kend[seg];
        int mcur = fp.m0[seg];
        const float *plan = fp.plan + fp.off[seg];
        float aH[4] = {0.f, 0.f, 0.f, 0.f}, aP[4] = {0.f, 0.f, 0.f, 0.f};
        auto emit_first = [&]() {
            const float vH = aH[0], vP = aP[0];
            mxH = fmaxf(mxH, vH), mxP = fmaxf(mxP, vP);
            if (active) {
                img[mcur * ld + tc] = vH;
                img[(rows + mcur) * ld + tc] = vP;
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) aH[e] = aH[e + 1], aP[e] = aP[e + 1];
            aH[3] = aP[3] = 0.f;
            ++mcur;
        };
        const float *Sb = S + cb + tc, *Pb = perc + cb + tc;
        const float *Hb = hclip + (size_t)(tc >> 4) * K * 16 + (tc & 15);
        for (int k0 = kbeg; k0 < kend; k0 += kWalkBatch) {
            float sv[kWalkBatch], pv[kWalkBatch], hv[kWalkBatch];
            float4 wq[kWalkBatch];
            int ne[kWalkBatch];
#pragma unroll
            for (int u = 0; u < kWalkBatch; ++u) {
                const int kk = min(k0 + u, K - 1);
                sv[u] = Sb[(size_t)kk * T];
                pv[u] = Pb[(size_t)kk * T];
                hv[u] = Hb[(size_t)kk * 16];
                const int pi = min(k0 + u, kend - 1) - kbeg;
                wq[u] = *reinterpret_cast<const float4 *>(plan + (size_t)pi * 8);
                ne[u] = __float_as_int(plan[(size_t)pi * 8 + 4]);
            }
#pragma unroll
            for (int u = 0; u < kWalkBatch; ++u) {
                if (k0 + u >= kend) break;
                for (int i = 0; i < ne[u]; ++i) emit_first();
                float H, P;
                hpss_masks_fast(sv[u], hv[u], pv[u], H, P);
                aH[0] = fmaf(wq[u].x, H, aH[0]), aP[0] = fmaf(wq[u].x, P, aP[0]);
                aH[1] = fmaf(wq[u].y, H, aH[1]), aP[1] = fmaf(wq[u].y, P, aP[1]);
                aH[2] = fmaf(wq[u].z, H, aH[2]), aP[2] = fmaf(wq[u].z, P, aP[2]);
                aH[3] = fmaf(wq[u].w, H, aH[3]), aP[3] = fmaf(wq[u].w, P, aP[3]);
            }
        }
        while (mcur < m1) emit_first();
    }
    if (x0p)  // layer-0 weights -> LDS; consumed after several barriers, fetched behind the walk's own loads
        for (int i = threadIdx.x; i < R2 * 32; i += blockDim.x) w0s[i] = w0[i];
    // per-array maximum of the filter sums -> the top_db floor in the power domain:
    //   max(10 log10(max(amin, x^2)), dBmax - 80) = 10 log10(max(amin, x^2, max(amin, xmax^2) * 1e-8))       (log is monotone)
    float limH = 0.f, limP = 0.f;
    {
        int kH = ordered_key(mxH), kP = ordered_key(mxP);
        for (int off = 32; off > 0; off >>= 1) {
            kH = max(kH, __shfl_xor(kH, off));
            kP = max(kP, __shfl_xor(kP, off));
        }
        if (lane == 0) smax[wave] = kH, smax[16 + wave] = kP;
        __syncthreads();  // also: the image is complete
        if (stop_after == 1) return;  // tuning probes (SMH_FEAT_STOP): time the kernel phase by phase
        if (log_db) {
            int a = smax[0], c = smax[16];
            for (int q = 1; q < nw; ++q) a = max(a, smax[q]), c = max(c, smax[16 + q]);
            const float xh = key_to_float(a), xp = key_to_float(c);
            limH = fmaxf(kAmin, fmaxf(kAmin, xh * xh) * 1e-8f), limP = fmaxf(kAmin, fmaxf(kAmin, xp * xp) * 1e-8f);
        }
    }
    // dB + clip in LDS (the image becomes the final featuregram), write it out (coalesced rows)
    auto final_value = [&](float x, float lim) { return log_db ? 3.0102999566398120f * __builtin_amdgcn_logf(fmaxf(x * x, lim)) : x; };
    float *g = fv + fvb;
    if ((T & 1) == 0) {  // rows start on 8-byte boundaries: one float2 per lane, a 98-frame row is one instruction
        for (int r = wave; r < R2; r += nw) {
            const float lim = r < rows ? limH : limP;
            for (int t2 = lane; t2 < T / 2; t2 += 64) {
                const float x0 = final_value(img[r * ld + 2 * t2], lim), x1 = final_value(img[r * ld + 2 * t2 + 1], lim);
                img[r * ld + 2 * t2] = x0;
                img[r * ld + 2 * t2 + 1] = x1;
                f32x2 v = {x0, x1};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x2 *>(g + (size_t)r * T) + t2);
            }
        }
    } else {
        for (int r = wave; r < R2; r += nw) {
            const float lim = r < rows ? limH : limP;
            for (int t = lane; t < T; t += 64) {
                const float x = final_value(img[r * ld + t], lim);
                img[r * ld + t] = x;
                g[(size_t)r * T + t] = x;
            }
        }
    }
    __syncthreads();
    if ((!patches && !x0p) || nP <= 0 || stop_after == 2) return;
    // StandardScaler statistics: one thread per row, f64 (see std_patch_kernel)
    for (int r0 = 0; r0 < R2; r0 += (int)(blockDim.x >> 2)) {  // four lanes per row, f64 partial sums
        const int r = r0 + (int)(threadIdx.x >> 2), sub = threadIdx.x & 3;
        const bool on = r < R2;
        const float *row = img + (on ? r : 0) * ld;
        double sum = 0.0;
        for (int t = sub; t < T; t += 4) sum += (double)row[t];
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const double mean = sum / (double)T;
        double qv = 0.0;
        for (int t = sub; t < T; t += 4) {
            const double dlt = (double)row[t] - mean;
            qv += dlt * dlt;
        }
        qv += __shfl_xor(qv, 1);
        qv += __shfl_xor(qv, 2);
        if (!on || sub != 0) continue;
        const double var = qv / (double)T;
        const double eps = 2.220446049250313e-16;
        const double nm = (double)T * mean * eps;
        const bool constant = var <= (double)T * eps * var + nm * nm;
        double scale = sqrt(var);
        if (constant || scale == 0.0) scale = 1.0;
        s_mean[r] = (float)mean;
        s_inv[r] = (float)(1.0 / scale);
        s_lo[r] = (float)(mean - (double)(float)mean);
    }
    __syncthreads();
    if (stop_after == 3) return;
    if (x0p) {
        // The network's first layer, per clip half (std_patch_kernel<true> has the derivation):
        // x0p[half][t][c] = sum_r W0[half*rows + r][c] * standardised(img[half*rows + r][t]).  One task = one 16-frame tile
        // of one half, BOTH 16-channel M-tiles (they share the B operand, the standardised value); the layer's
        // weights were copied to LDS at kernel start.  Steps run in branch-free groups of 8 (steps past the last row
        // multiply a zero), so the LDS reads of a group are all in flight before its first product.
        const int q = lane >> 4, j = lane & 15;
        const int ut = (W + 15) >> 4;
        const int nst = rows / 4;
        for (int task = wave; task < nP * ut * 2; task += nw) {
            const int half = task & 1, pu = task >> 1;
            const int p = pu / ut, u = pu - p * ut;
            int s = p * shift;
            const int e = min(s + W, Ttiled);
            if (e - s < W) s = e - W;
            const int jt = 16 * u + j;
            int tt = s + min(jt, W - 1);
            tt -= (tt / T) * T;
            const float *wr = w0s + ((size_t)half * rows + q) * 32 + j;
            const float *tl = img + (size_t)half * rows * ld + tt;
            const float *mh = s_mean + half * rows, *ml = s_lo + half * rows, *iv = s_inv + half * rows;
            f32x4 c0a = {0.f, 0.f, 0.f, 0.f}, c0b = c0a, c1a = c0a, c1b = c0a;  // two chains per M-tile
            for (int s0 = 0; s0 < nst; s0 += 8) {
                float xs[8], is[8], wa0[8], wa1[8], hs[8], ls[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int r = 4 * min(s0 + g, nst - 1) + q;
                    xs[g] = tl[r * ld];
                    hs[g] = mh[r], ls[g] = ml[r];
                    is[g] = s0 + g < nst ? iv[r] : 0.f;
                    wa0[g] = wr[(size_t)(r - q) * 32];
                    wa1[g] = wr[(size_t)(r - q) * 32 + 16];
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    // x - mean with the f64 mean held as hi + lo floats: two f32 subtractions (the f64 form of the patch
                    // writer costs four quarter-rate instructions per value and made this phase VALU-bound)
                    const float c = __fsub_rn(__fsub_rn(xs[g], hs[g]), ls[g]) * is[g];
                    if (g & 1) {
                        c0b = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[g], c, c0b, 0, 0, 0);
                        c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[g], c, c1b, 0, 0, 0);
                    } else {
                        c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[g], c, c0a, 0, 0, 0);
                        c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[g], c, c1a, 0, 0, 0);
                    }
                }
            }
            c0a += c0b, c1a += c1b;
            if (jt < W) {
                float *o = x0p + (((pb + p) * 2 + half) * W + jt) * 32 + 4 * q;
                *reinterpret_cast<f32x4 *>(o) = c0a;
                *reinterpret_cast<f32x4 *>(o + 16) = c1a;
            }
        }
    }
    if (!patches) return;
    for (int p = 0; p < nP; ++p) {
        int s0 = p * shift;
        const int e = min(s0 + W, Ttiled);
        if (e - s0 < W) s0 = e - W;
        float *o = patches + (pb + p) * W * R2;
        for (int j = wave; j < W; j += nw) {  // one wave per frame: lanes over the 2*rows features (960-byte rows)
            int tt = s0 + j;
            tt -= (tt / T) * T;
            for (int f = lane; f < R2; f += 64) {
                const float c = (float)((double)img[f * ld + tt] - ((double)s_mean[f] + (double)s_lo[f]));
                o[(size_t)j * R2 + f] = c * s_inv[f];
            }
        }
    }
}

__global__ void fill_int_kernel(int *p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}


// ---------------------------------------------------------------------------------------------------
// features_half_kernel: the single-kernel feature path with ONE WORKGROUP PER (clip, half) -- harmonic rows or
// percussive rows -- instead of one per clip.  Everything behind the masks is per half anyway (the top-dB maximum is
// taken per H / P array, lib/preprocessing.py:420,422; StandardScaler runs per half, :211-224; the layer-0 partials
// are per half), so the two halves need nothing from each other.  What the split buys: the LDS image is 47 KB instead
// of 94 KB, so TWO workgroups share a CU and one's memory phases (the walk streams 236 KB, the featuregram write) run
// beside the other's arithmetic phases (statistics, layer 0) -- with one workgroup per CU the kernel was the plain sum of
// its phases.  What it costs: both halves stream S / harm / perc (the second read comes from L2: the two
// workgroups of a clip sit on the same XCD, 8 dispatch slots apart) and evaluate the mask denominator.
// Even T only; odd T keeps features_clip_kernel (lane = frame).
// The bin walk here runs with lane = PAIR of frames: every load is 8 bytes, the masks and the filter sums use packed f32
// instructions (v_pk_mul / v_pk_add / v_pk_fma: two frames per VALU slot).  The soft mask is evaluated as
//   out = S own^2 / (own^2 + other^2)          (own = harm for the H half, perc for the P half; one reciprocal per frame)
// which is librosa's m / (m + r) with m = (h/Z)^2, r = (p/Z)^2, Z = max(h, p) without the normalisation by Z; where
// own^2 + other^2 would leave the normal range (both medians below ~1e-15: digital silence) the wave takes the normalised
// form with its split_zeros rule instead (wave-uniform branch, practically never taken).
// The image receives the filter sums themselves (magnitudes); the dB conversion waits for the write phase, where the VALU
// idles behind the stores.  NP = pending filters per bin: 2 for every Slaney bank whose filters are at least as wide as they
// are apart (the reference's 120 mels over 201 bins: a frequency lies in exactly two triangles), 4 in general.
// ---------------------------------------------------------------------------------------------------
// TRACE: tools/trace_features.py only.  The stamps cost registers (93 instead of 78, i.e. the third workgroup per CU); the
// TRACE build is therefore held to 80 VGPRs and spills ten of them: its timeline is indicative, not the plain build's.
// PROBE: tools/perc_in_walk_bound.py only (SMH_FEAT_PROBE_PERC): a separate instantiation, the plain kernel's registers stay as they are
// RAG (smh_rag.h): the B entries of `list` are clips of a ragged call (all of even T); T, the tiled length, the patch count and every
// buffer offset come from the clip's descriptor; the arithmetic is the equal-length instantiation's, so a clip gets the same bits.
template <int NP, bool TRACE = false, bool PROBE = false, bool RAG = false>
__global__ void __launch_bounds__(512, (TRACE ? 6 : 1))
features_half_kernel(FeatPlan fp, int log_db, int stop_after, const float *__restrict__ S, const float *__restrict__ harmb,
                     const float *__restrict__ perc, int B, int K, int T, int rows, int Ttiled, int W, int shift, int nP,
                     float *__restrict__ fv, float *__restrict__ patches, const float *__restrict__ w0,
                     float *__restrict__ x0p, const smh_rag::Clip *__restrict__ rag, const int *__restrict__ list) {
    extern __shared__ __attribute__((aligned(16))) float img[];  // [rows][ld]
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // blocks i and i + 8 share an XCD (round-robin dispatch): clip 8 g + r, half h  <->  block 16 g + 8 h + r
    const int grp = blockIdx.x >> 4, rr = blockIdx.x & 15;
    const int half = rr >> 3, b = grp * 8 + (rr & 7);
    if (b >= B) return;
    size_t cb, hb, fvb, pb;  // this clip's S / perc, blocked harm, featuregram (floats) and first patch
    if constexpr (RAG) {
        const smh_rag::Clip &c = rag[list[b]];
        T = c.T, Ttiled = c.Ttiled, nP = c.nP;
        cb = (size_t)c.spec_off, hb = (size_t)c.harm_off, fvb = (size_t)c.fv_off, pb = (size_t)c.patch_off;
    } else {
        cb = (size_t)b * K * T, hb = (size_t)b * ((T + 15) >> 4) * K * 16, fvb = (size_t)b * 2 * rows * T, pb = (size_t)b * nP;
    }
    const bool w0_lds = !(stop_after & 16);  // bit 4 of the probe argument: layer-0 weights from L2 instead of an LDS copy
    // bits 8..15 (timing probe SMH_FEAT_PROBE_PERC = n, outputs invalid): what "the percussive median inside this walk" would
    // cost -- perc is NOT read (S stands in for it), every bin step issues 2 n selection instructions (n per frame of the lane's
    // pair: the block-split scheme needs 17.6 per output at window 17) and every segment walks l_perc - 1 = 16 more bins of S
    // in front (the window's warm-up).  tools/perc_in_walk_bound.py.
    const int probe_n = PROBE ? (stop_after >> 8) & 255 : 0;
    stop_after &= 15;
    const int ld = T | 1, R2 = 2 * rows;
    float *s_mean = img + (size_t)rows * ld;  // mean hi [rows], 1/scale [rows], mean lo [rows]
    float *s_inv = s_mean + rows, *s_lo = s_mean + 2 * rows;
    int *smax = reinterpret_cast<int *>(s_mean + 3 * (size_t)rows);  // 16 ints
    float *w0s = s_mean + 3 * (size_t)rows + 16;  // this half's layer-0 weights [rows][32] (x0p only)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    auto stamp = [&](int i) {  // (tools only) phase i of this wave: 100 MHz ticks + where the wave runs
        if constexpr (!TRACE) return;
        if (fp.trace) {
            const int ws = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            unsigned long long *r = fp.trace + ((size_t)blockIdx.x * 8 + ws) * 8;
            const unsigned long long tk = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                r[i] = tk;
                if (i == 0) r[7] = __builtin_amdgcn_s_getreg((3 << 11) | 4 /* HW_REG_HW_ID */);
            }
        }
    };
    stamp(0);
    const float *hclip = harmb + hb;
    float mx = 0.f;  // maximum of this half's filter sums (sums of non-negative terms)

    const int npair = T >> 1;
    const int nwp = (npair + 63) >> 6;
    for (int task = wave; task < fp.nseg * nwp; task += nw) {
        const int seg = __builtin_amdgcn_readfirstlane(task / nwp);
        const int tw = task - seg * nwp;
        const int l = tw * 64 + lane;
        const bool active = l < npair;
        const int t0 = 2 * min(l, npair - 1);
        const int m1 = fp.m1[seg], kbeg = fp.kbeg[seg], kend = fp.kend[seg];
        int mcur = fp.m0[seg];
        const float *plan = fp.plan + fp.off[seg];
        f32x2 acc[NP];
#pragma unroll
        for (int e = 0; e < NP; ++e) acc[e] = f32x2{0.f, 0.f};
        auto emit_first = [&]() {
            const f32x2 v = acc[0];
            mx = fmaxf(mx, fmaxf(v.x, v.y));
            if (active) {
                float *o = img + mcur * ld + t0;
                o[0] = v.x, o[1] = v.y;
            }
#pragma unroll
            for (int e = 0; e + 1 < NP; ++e) acc[e] = acc[e + 1];
            acc[NP - 1] = f32x2{0.f, 0.f};
            ++mcur;
        };
        // "own" = the median this half's mask favours (harm for H, perc for P): out = S own^2 / (own^2 + other^2)
        const float *Sb = S + cb + t0, *Pb = ((PROBE && probe_n) ? S : perc) + cb + t0;
        const float *Hb = hclip + (size_t)(t0 >> 4) * K * 16 + (t0 & 15);
        // own / other as (base, bin stride) pairs chosen once per workgroup: the loads land in the registers the mask reads
        // (selecting own / other per bin cost four v_cndmask per bin step, a quarter of the walk's VALU instructions)
        const float *OwnB = half ? Pb : Hb, *OthB = half ? Hb : Pb;
        const int own_st = half ? T : 16, oth_st = half ? 16 : T;
        float pd0 = 0.f, pd1 = 1.f;  // (probe) the dummy window the selection instructions work on
        auto probe_select = [&](f32x2 v) {
            for (int i = 0; i < probe_n; i += 6) {
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(pd0) : "v"(v.x), "v"(pd1));
                    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(pd1) : "v"(v.y), "v"(pd0));
                }
            }
        };
        if (PROBE && probe_n) {  // the window's warm-up: 16 more bins of S in front of the segment (reflected at the clip's edge)
            f32x2 hv16[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) hv16[u] = *reinterpret_cast<const f32x2 *>(Sb + (size_t)max(kbeg - 16 + u, 0) * T);
#pragma unroll
            for (int u = 0; u < 16; ++u) probe_select(hv16[u]);
        }
        for (int k0 = kbeg; k0 < kend; k0 += kHalfBatch) {
            f32x2 sv[kHalfBatch], ov[kHalfBatch], tv[kHalfBatch];  // S, own median, other median
            float4 wq[kHalfBatch];
            int ne[kHalfBatch];
#pragma unroll
            for (int u = 0; u < kHalfBatch; ++u) {
                const int kk = min(k0 + u, K - 1);
                sv[u] = *reinterpret_cast<const f32x2 *>(Sb + (size_t)kk * T);
                ov[u] = *reinterpret_cast<const f32x2 *>(OwnB + (size_t)kk * own_st);
                tv[u] = *reinterpret_cast<const f32x2 *>(OthB + (size_t)kk * oth_st);
                const int pi = min(k0 + u, kend - 1) - kbeg;
                wq[u] = *reinterpret_cast<const float4 *>(plan + (size_t)pi * 8);
                ne[u] = __float_as_int(plan[(size_t)pi * 8 + 4]);
            }
#pragma unroll
            for (int u = 0; u < kHalfBatch; ++u) {
                if (k0 + u >= kend) break;
                for (int i = 0; i < ne[u]; ++i) emit_first();
                if (PROBE && probe_n) probe_select(sv[u]);
                const f32x2 own = ov[u], oth = tv[u];
                const f32x2 o2 = own * own;
                const f32x2 den = o2 + oth * oth;
                f32x2 X;
                constexpr float kDenMin = 7.8886091e-31f;  // 2^-100
                if (__builtin_expect(__any(den.x < kDenMin || den.y < kDenMin), 0)) {
                    float Hx, Px, Hy, Py;  // (rare: digital silence) the normalised form, harmonic median first
                    hpss_masks_fast(sv[u].x, half ? oth.x : own.x, half ? own.x : oth.x, Hx, Px);
                    hpss_masks_fast(sv[u].y, half ? oth.y : own.y, half ? own.y : oth.y, Hy, Py);
                    X = half ? f32x2{Px, Py} : f32x2{Hx, Hy};
                } else {
                    X = o2 * (sv[u] * f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)});
                }
                acc[0] += wq[u].x * X;
                acc[1] += wq[u].y * X;
                if constexpr (NP > 2) {
                    acc[2] += wq[u].z * X;
                    acc[3] += wq[u].w * X;
                }
            }
        }
        while (mcur < m1) emit_first();
    }
    if (x0p && w0_lds)  // this half's layer-0 weights -> LDS; consumed after several barriers
        for (int i = threadIdx.x; i < rows * 32; i += blockDim.x) w0s[i] = w0[(size_t)half * rows * 32 + i];
    // maximum of the array -> the top_db floor in the power domain (see features_clip_kernel)
    float lim = 0.f;
    {
        int kx = ordered_key(mx);
        for (int off = 32; off > 0; off >>= 1) kx = max(kx, __shfl_xor(kx, off));
        if (lane == 0) smax[wave] = kx;
        stamp(1);  // this wave's walk is done
        __syncthreads();  // also: the image is complete
        if (stop_after == 1) return;
        if (log_db) {
            int a = smax[0];
            for (int q = 1; q < nw; ++q) a = max(a, smax[q]);
            const float xm = key_to_float(a);
            lim = fmaxf(kAmin, fmaxf(kAmin, xm * xm) * 1e-8f);
        }
    }
    // dB + clip in LDS (the image becomes this half of the final featuregram), write it out (coalesced rows)
    float *g = fv + fvb + (size_t)half * rows * T;
    for (int r = wave; r < rows; r += nw) {
        for (int t2 = lane; t2 < T / 2; t2 += 64) {
            float x0 = img[r * ld + 2 * t2], x1 = img[r * ld + 2 * t2 + 1];
            if (log_db) {
                x0 = 3.0102999566398120f * __builtin_amdgcn_logf(fmaxf(x0 * x0, lim));
                x1 = 3.0102999566398120f * __builtin_amdgcn_logf(fmaxf(x1 * x1, lim));
            }
            img[r * ld + 2 * t2] = x0;
            img[r * ld + 2 * t2 + 1] = x1;
            f32x2 v = {x0, x1};
            // nontemporal on purpose (tools/gpu/r3_feat_nt.sh): as plain stores the rows merge into whole lines in L2 (WRITE_SIZE 121.7 ->
            // 114.2 MB per 1024 clips, the bytes of fv + x0p) but push out the S / harm / perc lines the other half's workgroup is
            // about to read (FETCH_SIZE 309 -> 321 MB) and the kernel takes 118.3 us instead of 111.7
#ifdef SMH_PLAIN_FV_STORES  // (A/B build only)
            reinterpret_cast<f32x2 *>(g + (size_t)r * T)[t2] = v;
#else
            __builtin_nontemporal_store(v, reinterpret_cast<f32x2 *>(g + (size_t)r * T) + t2);
#endif
        }
    }
    stamp(2);  // dB + clip + write
    __syncthreads();
    if ((!patches && !x0p) || nP <= 0 || stop_after == 2) return;
    // StandardScaler statistics: four lanes per row, f64 partial sums (see std_patch_kernel)
    for (int r0 = 0; r0 < rows; r0 += (int)(blockDim.x >> 2)) {
        const int r = r0 + (int)(threadIdx.x >> 2), sub = threadIdx.x & 3;
        const bool on = r < rows;
        const float *row = img + (on ? r : 0) * ld;
        double sum = 0.0;
        for (int t = sub; t < T; t += 4) sum += (double)row[t];
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const double mean = sum / (double)T;
        double qv = 0.0;
        for (int t = sub; t < T; t += 4) {
            const double dlt = (double)row[t] - mean;
            qv += dlt * dlt;
        }
        qv += __shfl_xor(qv, 1);
        qv += __shfl_xor(qv, 2);
        if (!on || sub != 0) continue;
        const double var = qv / (double)T;
        const double eps = 2.220446049250313e-16;
        const double nm = (double)T * mean * eps;
        const bool constant = var <= (double)T * eps * var + nm * nm;
        double scale = sqrt(var);
        if (constant || scale == 0.0) scale = 1.0;
        s_mean[r] = (float)mean;
        s_inv[r] = (float)(1.0 / scale);
        s_lo[r] = (float)(mean - (double)(float)mean);
    }
    stamp(3);  // statistics
    __syncthreads();
    stamp(4);
    if (stop_after == 3) return;
    auto layer0 = [&](auto wbase) {
        // this half's share of the network's first layer (features_clip_kernel has the derivation); one task = one 16-frame
        // tile with both 16-channel M-tiles
        const int q = lane >> 4, j = lane & 15;
        const int ut = (W + 15) >> 4;
        const int nst = rows / 4;
        for (int task = wave; task < nP * ut; task += nw) {
            const int p = task / ut, u = task - p * ut;
            int s = p * shift;
            const int e = min(s + W, Ttiled);
            if (e - s < W) s = e - W;
            const int jt = 16 * u + j;
            int tt = s + min(jt, W - 1);
            tt -= (tt / T) * T;
            const auto wr = wbase + (size_t)q * 32 + j;
            const float *tl = img + tt;
            f32x4 c0a = {0.f, 0.f, 0.f, 0.f}, c0b = c0a, c1a = c0a, c1b = c0a;  // two chains per M-tile
            for (int s0 = 0; s0 < nst; s0 += 8) {
                float xs[8], is[8], wa0[8], wa1[8], hs[8], ls[8];
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) {
                    const int r = 4 * min(s0 + g8, nst - 1) + q;
                    xs[g8] = tl[r * ld];
                    hs[g8] = s_mean[r], ls[g8] = s_lo[r];
                    is[g8] = s0 + g8 < nst ? s_inv[r] : 0.f;
                    wa0[g8] = wr[(size_t)(r - q) * 32];
                    wa1[g8] = wr[(size_t)(r - q) * 32 + 16];
                }
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) {
                    const float c = __fsub_rn(__fsub_rn(xs[g8], hs[g8]), ls[g8]) * is[g8];
                    if (g8 & 1) {
                        c0b = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[g8], c, c0b, 0, 0, 0);
                        c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[g8], c, c1b, 0, 0, 0);
                    } else {
                        c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[g8], c, c0a, 0, 0, 0);
                        c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[g8], c, c1a, 0, 0, 0);
                    }
                }
            }
            c0a += c0b, c1a += c1b;
            if (jt < W) {
                float *o = x0p + (((pb + p) * 2 + half) * W + jt) * 32 + 4 * q;
                *reinterpret_cast<f32x4 *>(o) = c0a;
                *reinterpret_cast<f32x4 *>(o + 16) = c1a;
            }
        }
    };
    // The same products with ONE TASK PER (16-frame tile, 16-channel M-tile), for the reference's 120 mel rows (kL0Steps k steps) and
    // the weights in L2.  Waves 2 i and 2 i + 1 take the two M-tiles of tiles i, i + nw/2, ...: they read the same standardised
    // values (the B operand), each multiplies its own 16 output channels, and a wave keeps ITS M-tile's weights of every k step in
    // 30 registers for all its tiles -- requested in one batch, one L2 round trip per wave, where the task above runs a chain of
    // four [16 loads -> wait -> 16 products] groups.  Per M-tile the sums are the ones of the undivided task -- even k steps on one
    // accumulator chain, odd steps on a second, added at the end -- so x0p keeps its bits.  (Requesting the weights in front of the
    // statistics phase, to hide that round trip as well, takes the kernel from 79 to 109 VGPRs: one workgroup less per CU.)
    auto layer0_mtile = [&](const float *wu) {
        const int q = lane >> 4, j = lane & 15, mt = wave & 1;
        const int ut = (W + 15) >> 4;
        float wa[kL0Steps];
        {
            const float *wl = wu + (q * 32 + j + 16 * mt);
#pragma unroll
            for (int s = 0; s < kL0Steps; ++s) wa[s] = wl[s * 128];
        }
        const float *mq = s_mean + q, *lq = s_lo + q, *iq = s_inv + q;
        for (int pu = wave >> 1; pu < nP * ut; pu += nw >> 1) {
            // (the statistics are the same for every tile: without this fence the compiler keeps all 90 of them in registers)
            asm volatile("" ::: "memory");
            const int p = pu / ut, u = pu - p * ut;
            int s = p * shift;
            const int e = min(s + W, Ttiled);
            if (e - s < W) s = e - W;
            const int jt = 16 * u + j;
            int tt = s + min(jt, W - 1);
            tt -= (tt / T) * T;
            const float *tl = img + tt + q * ld;
            f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = ca;
#pragma unroll
            for (int s0 = 0; s0 < kL0Steps; s0 += 6) {
                float xs[6], is[6], hs[6], ls[6];
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    xs[g] = tl[(s0 + g) * 4 * ld];
                    hs[g] = mq[(s0 + g) * 4], ls[g] = lq[(s0 + g) * 4], is[g] = iq[(s0 + g) * 4];
                }
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    const float c = __fsub_rn(__fsub_rn(xs[g], hs[g]), ls[g]) * is[g];
                    if ((s0 + g) & 1) cb = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s0 + g], c, cb, 0, 0, 0);
                    else ca = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s0 + g], c, ca, 0, 0, 0);
                }
            }
            ca += cb;
            if (jt < W)
                *reinterpret_cast<f32x4 *>(x0p + (((pb + p) * 2 + half) * W + jt) * 32 + 4 * q + 16 * mt) = ca;
        }
    };
    if (x0p) {
        if (w0_lds) layer0(w0s);
        else if (rows == 4 * kL0Steps && !(nw & 1)) layer0_mtile(w0 + (size_t)half * rows * 32);
        else layer0(w0 + (size_t)half * rows * 32);  // weights straight from L2: 15 KB less LDS, three workgroups per CU
    }
    stamp(5);  // layer 0
    if (!patches) return;
    for (int p = 0; p < nP; ++p) {
        int s0 = p * shift;
        const int e = min(s0 + W, Ttiled);
        if (e - s0 < W) s0 = e - W;
        float *o = patches + (pb + p) * W * R2 + (size_t)half * rows;
        for (int j = wave; j < W; j += nw) {  // one wave per frame: lanes over this half's features
            int tt = s0 + j;
            tt -= (tt / T) * T;
            for (int f = lane; f < rows; f += 64) {
                const float c = (float)((double)img[f * ld + tt] - ((double)s_mean[f] + (double)s_lo[f]));
                o[(size_t)j * R2 + f] = c * s_inv[f];
            }
        }
    }
}

}  // namespace

// tools/trace_features.py: phase stamps of features_half_kernel (8 words per wave, 8 waves per workgroup)
static unsigned long long *g_feat_trace = nullptr;
constexpr size_t kFeatTraceWords = (size_t)4096 * 8 * 8;
extern "C" int smh_internal_feat_trace(int enable, unsigned long long *host, size_t words) {
    if (enable && !g_feat_trace) {
        if (hipMalloc((void **)&g_feat_trace, kFeatTraceWords * 8) != hipSuccess) return -1;
        (void)hipMemset(g_feat_trace, 0, kFeatTraceWords * 8);
    }
    if (host && g_feat_trace) {
        (void)hipDeviceSynchronize();
        if (hipMemcpy(host, g_feat_trace, std::min(words, kFeatTraceWords) * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    if (!enable && g_feat_trace) {
        (void)hipFree(g_feat_trace);
        g_feat_trace = nullptr;
    }
    return 0;
}

// Residency of the bench path's feature kernel (tests/test_bench_path_gpu.py): workgroups of features_half_kernel<2> per CU
// for a clip of T frames and `rows` mel rows per half with the layer-0 weights read from L2.  Three is what the measured
// 113-118 us rest on (78 VGPRs, 49 KB of LDS); a build that needs more than 80 VGPRs silently drops to two and ~130 us.
extern "C" int smh_internal_feat_residency(int rows, int T) {
    const size_t ldh = sizeof(float) * ((size_t)rows * (T | 1) + 3 * (size_t)rows) + 64;
    if (hipFuncSetAttribute((const void *)features_half_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldh) != hipSuccess)
        return -1;
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)features_half_kernel<2, false>, 512, ldh) != hipSuccess) return -1;
    return nb;
}

namespace smh_feat {

FeatPlan feat_plan(const smh_ctx *c, int which) {
    FeatPlan fp;
    fp.nseg = c->feat_nseg[which], fp.pend = c->feat_pend;
    for (int i = 0; i < smh_ctx::kMaxFeatSegs; ++i)
        fp.m0[i] = c->feat_m0[which][i], fp.m1[i] = c->feat_m1[which][i], fp.kbeg[i] = c->feat_kbeg[which][i],
        fp.kend[i] = c->feat_kend[which][i], fp.off[i] = c->feat_off[which][i];
    fp.plan = c->d_feat_plan;
    fp.trace = nullptr;
    return fp;
}

MelTable mel_table(const smh_ctx *c) {
    MelTable m;
    m.n_mels = c->n_mels;
    m.start = c->d_mel_start, m.count = c->d_mel_count, m.off = c->d_mel_off, m.w = c->d_mel_w;
    m.nnz = c->mel_nnz;
    return m;
}

int launch_hp_feat(const smh_ctx *c, const float *S, const float *harm, const float *perc, int harm_tmajor, int B, int T,
                   float *fv, int *maxkeys, hipStream_t st) {
    const int K = c->K, rows = c->feat_rows;
    if (c->n_mels > smh_feat::kMaxMels || c->mel_nnz > smh_feat::kMaxMelNnz)
        return smh::set_error(SMH_E_INVALID, "mel filterbank too large for the fused kernel (n_mels=%d nnz=%d)", c->n_mels, c->mel_nnz);
    if (c->cfg.log_db) {
        hipLaunchKernelGGL(fill_int_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, st, maxkeys, 2 * B, (int)0x80000000);
        int rc = smh::launch_status("fill_int_kernel");
        if (rc) return rc;
    }
    // bin-walk kernel: one workgroup per clip (SMH_FEAT_TAPS=1 forces the per-tap kernel below)
    const size_t lds_walk = sizeof(float) * (harm_tmajor ? (size_t)T * (K | 1) : 0) + 128;
    const int walk_waves = c->feat_nseg[0] * ((T + 63) / 64);
    if (c->feat_walk_ok && lds_walk <= 150 * 1024 && walk_waves >= 1 && !smh::lab_env("SMH_FEAT_TAPS")) {
        const FeatPlan fp = feat_plan(c, 0);
        const int nwaves = std::min(16, walk_waves);
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)hp_feat_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_walk));
        hipLaunchKernelGGL(hp_feat_walk_kernel, dim3(B), dim3(64 * nwaves), lds_walk, st, fp, c->cfg.log_db, S, harm, perc,
                           harm_tmajor, K, T, rows, fv, maxkeys);
        return smh::launch_status("hp_feat_walk_kernel");
    }
    // frame slabs: split T evenly into pieces of <= 64 frames (T=98 -> 2 x 49)
    const int nslab = (T + 63) / 64;
    const int TS = (T + nslab - 1) / nslab;
    const size_t lds = harm_tmajor ? sizeof(float) * (size_t)TS * (K | 1) : 0;
    if (lds > 150 * 1024) return smh::set_error(SMH_E_INVALID, "K=%d too large for the feature kernel", K);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)hp_feat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(hp_feat_kernel, dim3((T + TS - 1) / TS, B), dim3(kFeatThreads), lds, st, mel_table(c), c->cfg.log_db, S,
                       harm, perc, harm_tmajor, K, T, TS, rows, fv, maxkeys);
    return smh::launch_status("hp_feat_kernel");
}

// single-kernel path (harm in layout 2): returns 1 if it ran, 0 if the shape does not qualify, < 0 on error
int launch_features_clip(const smh_ctx *c, const float *S, const float *harmb, const float *perc, int B, int T, int W,
                         int shift, int nP, float *fv, float *patches, const float *w0, float *x0p, hipStream_t st) {
    const int K = c->K, rows = c->feat_rows;
    if (!c->feat_walk_ok || smh::lab_env("SMH_FEAT_TAPS")) return 0;
    if (x0p && (rows % 4 != 0 || rows > 128)) return 0;
    size_t lds = sizeof(float) * ((size_t)2 * rows * (T | 1) + 3 * (size_t)2 * rows) + 128;
    if (x0p) lds += sizeof(float) * 2 * rows * 32;  // the layer's weights
    if (lds > 158 * 1024) return 0;
    // even T: one workgroup per (clip, half), lane = frame pair (features_half_kernel); odd T: one workgroup per clip, lane = frame
    const int pair = (T % 2 == 0 && !getenv("SMH_FEAT_NOPAIR")) ? 1 : 0;
    FeatPlan fp = feat_plan(c, 1);
    fp.trace = g_feat_trace;
    const char *stop_ev = smh::probe_env("SMH_FEAT_STOP");  // timing probe: the kernel returns early
    int stop = stop_ev ? atoi(stop_ev) & 15 : 0;
    if (const char *pe = smh::probe_env("SMH_FEAT_PROBE_PERC")) stop |= (std::max(0, std::min(atoi(pe), 255)) << 8);  // see the kernel
    if (pair) {
        // half the LDS per workgroup: two share a CU.  The 8 waves take the 8-segment plan, one segment each: every segment
        // boundary costs a re-read of the bins its filters straddle (8 segments: 280 bin reads for 201 bins; 16 segments,
        // two per wave: 340 and 5 % slower).  Tried and dropped: disjoint bin ranges with the straddling filters' partial sums
        // combined by ds_add_f32 in a zeroed image (every bin read once, but 1.7x slower: LDS float atomics); software
        // pipelining of the walk's half-batches (slower: the loads already overlap across the two workgroups of a CU).
        size_t ldh = sizeof(float) * ((size_t)rows * (T | 1) + 3 * (size_t)rows) + 64;
        // the layer-0 weights come straight from L2 (15 KB per half, shared by every workgroup): without an LDS copy a workgroup
        // needs 49 KB and THREE share a CU (77 VGPRs: 6 waves per SIMD) -- 127-130 -> 113-115 us; SMH_FEAT_W0LDS=1: the copy
        const bool w0_l2 = smh::lab_env("SMH_FEAT_W0LDS") == nullptr;
        if (x0p && !w0_l2) ldh += sizeof(float) * rows * 32;
        const int probe = (stop & ~16) | (w0_l2 ? 16 : 0);
        // (the two halves of a clip back to back in their XCD's dispatch order: consecutive workgroups land on the same CU and the
        // second finds the first one's S / harm / perc lines in that CU's vector cache -- n halves apart the kernel takes 134 us
        // instead of 111, profiles/r03_store_policies.txt)
        const unsigned grid = 16u * (unsigned)((B + 7) / 8);
        if (getenv("SMH_FEAT_OCC")) {  // tools only: what the runtime says about residency
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)features_half_kernel<2, false>, 512, ldh);
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, (const void *)features_half_kernel<2, false>);
            fprintf(stderr, "features_half_kernel<2>: dynamic LDS %zu B, regs %d, occupancy %d workgroups per CU\n", ldh, fa.numRegs, nb);
        }
#define SMH_LAUNCH_HALF(NPV, TR, ...)                                                                                     \
    do {                                                                                                                \
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)features_half_kernel<NPV, TR, ##__VA_ARGS__>,                   \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldh));                      \
        hipLaunchKernelGGL((features_half_kernel<NPV, TR, ##__VA_ARGS__>), dim3(grid), dim3(512), ldh, st, fp, c->cfg.log_db, probe, S, \
                           harmb, perc, B, K, T, rows, smh_tiled_frames(T, W), W, shift, nP, fv, patches, w0, x0p, nullptr, nullptr); \
    } while (0)
        if (probe >> 8) {  // SMH_FEAT_PROBE_PERC: the instantiation with the probe compiled in (outputs invalid)
            SMH_LAUNCH_HALF(2, false, true);
        } else if (fp.trace) {
            if (fp.pend <= 2) SMH_LAUNCH_HALF(2, true);
            else SMH_LAUNCH_HALF(4, true);
        } else {
            if (fp.pend <= 2) SMH_LAUNCH_HALF(2, false);
            else SMH_LAUNCH_HALF(4, false);
        }
#undef SMH_LAUNCH_HALF
        int rch = smh::launch_status("features_half_kernel");
        return rch ? rch : 1;
    }
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)features_clip_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(features_clip_kernel<false>, dim3(B), dim3(1024), lds, st, fp, c->cfg.log_db, stop & 15, S, harmb, perc, K, T, rows,
                       smh_tiled_frames(T, W), W, shift, nP, fv, patches, w0, x0p, nullptr, nullptr);
    int rc = smh::launch_status("features_clip_kernel");
    return rc ? rc : 1;
}

// Clips of different lengths that each fit the LDS image (smh_rag.h): the same two kernels, one workgroup (pair) per list entry.
int launch_features_rag(const smh_ctx *c, const float *S, const float *harmb, const float *perc, const smh_rag::Clip *d_clips,
                        const int *d_list, int n, int max_T, int even_T, int W, int shift, float *fv, float *patches, hipStream_t st) {
    if (n <= 0) return SMH_OK;
    const int K = c->K, rows = c->feat_rows;
    const FeatPlan fp = feat_plan(c, 1);
    if (even_T) {
        const size_t ldh = sizeof(float) * ((size_t)rows * (max_T | 1) + 3 * (size_t)rows) + 64;
        const unsigned grid = 16u * (unsigned)((n + 7) / 8);
        const int probe = 16;  // (layer-0 weights are not used here)
        if (fp.pend <= 2) {
            SMH_CHECK_HIP(hipFuncSetAttribute((const void *)features_half_kernel<2, false, false, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldh));
            hipLaunchKernelGGL((features_half_kernel<2, false, false, true>), dim3(grid), dim3(512), ldh, st, fp, c->cfg.log_db, probe, S,
                               harmb, perc, n, K, 0, rows, 0, W, shift, 0, fv, patches, nullptr, nullptr, d_clips, d_list);
        } else {
            SMH_CHECK_HIP(hipFuncSetAttribute((const void *)features_half_kernel<4, false, false, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldh));
            hipLaunchKernelGGL((features_half_kernel<4, false, false, true>), dim3(grid), dim3(512), ldh, st, fp, c->cfg.log_db, probe, S,
                               harmb, perc, n, K, 0, rows, 0, W, shift, 0, fv, patches, nullptr, nullptr, d_clips, d_list);
        }
        return smh::launch_status("features_half_kernel (ragged)");
    }
    const size_t lds = sizeof(float) * ((size_t)2 * rows * (max_T | 1) + 3 * (size_t)2 * rows) + 128;
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)features_clip_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(features_clip_kernel<true>, dim3(n), dim3(1024), lds, st, fp, c->cfg.log_db, 0, S, harmb, perc, K, 0, rows, 0, W,
                       shift, 0, fv, patches, nullptr, nullptr, d_clips, d_list);
    return smh::launch_status("features_clip_kernel (ragged)");
}

int launch_std_patch(const smh_ctx *c, float *fv, const int *maxkeys, int B, int T, int W, int shift, int nP,
                     float *patches, hipStream_t st, const float *w0, float *x0p, void *scratch, size_t scratch_bytes) {
    const int rows = c->feat_rows;
    const size_t lds = sizeof(float) * ((size_t)rows * (T | 1) + 3 * (size_t)rows);
    if (x0p && (lds > 150 * 1024 || rows % 4 != 0 || rows > 128))
        return smh::set_error(SMH_E_INVALID, "smh_features_l0_f32: needs a featuregram half that fits one LDS tile "
                              "(T=%d) and a row count divisible by 4, at most 128 (rows=%d)", T, rows);
    if (lds > 150 * 1024) {
        // long clips: the same three steps as separate streaming kernels over a stream-ordered scratch copy
        if (B > 32767) return smh::set_error(SMH_E_INVALID, "B=%d too large for the long-clip path; split the batch", B);
        const size_t half = (size_t)rows * T;
        if (c->cfg.log_db) {
            size_t nb = (half + 255) / 256;
            if (nb > 1024) nb = 1024;
            hipLaunchKernelGGL(clip_fv_kernel, dim3((unsigned)nb, 2 * B), dim3(256), 0, st, fv, maxkeys, half);
            int rc = smh::launch_status("clip_fv_kernel");
            if (rc) return rc;
        }
        if (!patches || nP <= 0) return SMH_OK;
        float *tmp = nullptr;
        const size_t n_rows = (size_t)2 * B * rows;
        const bool own = !(scratch && scratch_bytes >= n_rows * T * sizeof(float));  // no caller scratch: a stream-ordered allocation
        if (own) SMH_CHECK_HIP(hipMallocAsync((void **)&tmp, n_rows * T * sizeof(float), st));
        else tmp = static_cast<float *>(scratch);
        hipLaunchKernelGGL(standardize_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, (const float *)fv,
                           (int)n_rows, T, tmp);
        const size_t per_clip = (size_t)nP * 2 * rows * W;
        size_t nb = (per_clip + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(extract_patches_kernel, dim3((unsigned)nb, B), dim3(256), 0, st, (const float *)tmp, 2 * rows, T,
                           smh_tiled_frames(T, W), W, shift, nP, 1, patches);
        int rc = smh::launch_status("long-clip standardise / patch kernels");
        if (own) SMH_CHECK_HIP(hipFreeAsync(tmp, st));
        return rc;
    }
    if (x0p) {
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)std_patch_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(std_patch_kernel<true>, dim3(2, B), dim3(kPatchThreads), lds, st, c->cfg.log_db, fv, maxkeys, rows,
                           T, smh_tiled_frames(T, W), W, shift, nP, patches, w0, x0p);
        return smh::launch_status("std_patch_kernel<l0>");
    }
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)std_patch_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(std_patch_kernel<false>, dim3(2, B), dim3(kPatchThreads), lds, st, c->cfg.log_db, fv, maxkeys, rows, T,
                       smh_tiled_frames(T, W), W, shift, nP, patches, w0, x0p);
    return smh::launch_status("std_patch_kernel");
}

}  // namespace smh_feat

// ---------------------------------------------------------------------------------------------------
extern "C" int smh_softmask_f32(const smh_ctx *, const float *d_S, const float *d_harm, const float *d_perc, size_t n,
                                float *d_H, float *d_P, void *stream) {
    SMH_REQUIRE(d_S && d_harm && d_perc && d_H && d_P, "smh_softmask_f32: null argument");
    if (n == 0) return SMH_OK;
    const int bs = 256;
    size_t nb = (n + bs - 1) / bs;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(softmask_kernel, dim3((unsigned)nb), dim3(bs), 0, (hipStream_t)stream, d_S, d_harm, d_perc, n, d_H,
                       d_P);
    return smh::launch_status("softmask_kernel");
}

extern "C" int smh_mel_f32(const smh_ctx *ctx, const float *d_X, int B, int T, float *d_Y, void *stream) {
    SMH_REQUIRE(ctx && d_X && d_Y, "smh_mel_f32: null argument");
    SMH_REQUIRE(ctx->n_mels > 0, "smh_mel_f32: context built without a mel filterbank (n_mels <= 0)");
    SMH_REQUIRE(B >= 0 && B <= 65535 && T >= 1, "smh_mel_f32: bad shape B=%d T=%d", B, T);
    if (B == 0) return SMH_OK;
    const int n = ctx->n_mels * T;
    hipLaunchKernelGGL(mel_kernel, dim3((n + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, smh_feat::mel_table(ctx),
                       d_X, ctx->K, T, d_Y);
    return smh::launch_status("mel_kernel");
}

extern "C" int smh_power_to_db_sq_f32(const smh_ctx *, const float *d_X, int n_arrays, int elems, float *d_Y,
                                      void *stream) {
    SMH_REQUIRE(d_X && d_Y, "smh_power_to_db_sq_f32: null argument");
    SMH_REQUIRE(n_arrays >= 0 && elems >= 1, "smh_power_to_db_sq_f32: bad shape");
    if (n_arrays == 0) return SMH_OK;
    hipLaunchKernelGGL(power_to_db_kernel, dim3(n_arrays), dim3(256), 0, (hipStream_t)stream, d_X, elems, d_Y);
    return smh::launch_status("power_to_db_kernel");
}

extern "C" int smh_standardize_rows_f32(const smh_ctx *, const float *d_X, int n_rows, int T, float *d_Y, void *stream) {
    SMH_REQUIRE(d_X && d_Y, "smh_standardize_rows_f32: null argument");
    SMH_REQUIRE(n_rows >= 0 && T >= 1, "smh_standardize_rows_f32: bad shape");
    if (n_rows == 0) return SMH_OK;
    hipLaunchKernelGGL(standardize_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_X, n_rows, T,
                       d_Y);
    return smh::launch_status("standardize_rows_kernel");
}

extern "C" int smh_extract_patches_f32(const smh_ctx *, const float *d_FV, int B, int F, int T, int W, int shift,
                                       int layout, float *d_out, void *stream) {
    SMH_REQUIRE(d_FV, "smh_extract_patches_f32: null input");
    SMH_REQUIRE(B >= 0 && B <= 65535 && F >= 1 && T >= 1 && W >= 1 && shift >= 1, "smh_extract_patches_f32: bad shape");
    SMH_REQUIRE(layout == 0 || layout == 1, "smh_extract_patches_f32: layout must be 0 or 1");
    const int Ttiled = smh_tiled_frames(T, W);
    const int nP = smh_num_patches(Ttiled, W, shift);
    if (nP <= 0 || B == 0) return nP < 0 ? SMH_E_INVALID : nP;
    SMH_REQUIRE(d_out, "smh_extract_patches_f32: null output");
    const size_t per_clip = (size_t)nP * F * W;
    size_t nb = (per_clip + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(extract_patches_kernel, dim3((unsigned)nb, B), dim3(256), 0, (hipStream_t)stream, d_FV, F, T, Ttiled,
                       W, shift, nP, layout, d_out);
    int rc = smh::launch_status("extract_patches_kernel");
    return rc ? rc : nP;
}
