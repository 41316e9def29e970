// HPSS median-filter kernel template.  See smh_median.hip for the entry points.
//
// harm = median along frames, perc = median along bins, boundary 'reflect' (edge sample repeated),
// bit-exact selection: replaces the two scipy.ndimage.median_filter calls inside
// librosa.decompose.hpss (call sites /root/reference/lib/preprocessing.py:408,418,430,440).
//
// Design (gfx950): one workgroup per (clip, frame-tile).  The (K x ncols) spectrogram tile is staged
// once into LDS with an ODD row stride, so that both walks are bank-conflict free:
//   harmonic lanes  : lane <-> bin k,   walks frames t  (LDS address k*stride + t, stride odd)
//   percussive lanes: lane <-> frame t, walks bins k    (LDS address k*stride + t, consecutive t)
// Each lane keeps its window SORTED IN REGISTERS (template size W) and advances with one fused
// delete-outgoing / insert-incoming pass: 3 VALU per slot (v_cmp, v_cndmask, v_med3) -- no sorting
// network per output and no cross-lane traffic.  The kernel is VALU-bound by construction; the tile
// is read from HBM once, harm/perc are written once (236,376 algorithmic bytes per 201x98 clip).
#pragma once
#include "smh_common.h"

namespace smh_median {

__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// 'reflect' for -n <= i < 2n (one fold per side); the host guarantees window/2 < n for this kernel.
__device__ __forceinline__ int reflect_lo(int i) { return i ^ (i >> 31); }               // i<0 -> -i-1
__device__ __forceinline__ int reflect_hi(int i, int n) { return min(i, 2 * n - 1 - i); }  // i>=n -> 2n-1-i

template <int W>
struct SortedWindow {
    float s[W];
    float ninf, pinf;  // run-time sentinels: keeps every slot update a single v_med3_f32

    __device__ __forceinline__ void clear(float ni, float pi) {
        ninf = ni, pinf = pi;
#pragma unroll
        for (int i = 0; i < W; ++i) s[i] = pi;
    }
    // insert x when exactly N slots are occupied (the rest hold +inf)
    template <int N>
    __device__ __forceinline__ void insert(float x) {
        float prev = ninf;
#pragma unroll
        for (int i = 0; i <= N; ++i) {
            const float cur = s[i];
            s[i] = med3(prev, x, cur);
            prev = cur;
        }
    }
    // remove one instance of `out_v` (must be present) and insert `in_v`
    __device__ __forceinline__ void replace(float out_v, float in_v) {
        float rprev = ninf;
#pragma unroll
        for (int i = 0; i < W - 1; ++i) {
            const float ri = (s[i] >= out_v) ? s[i + 1] : s[i];
            s[i] = med3(rprev, in_v, ri);
            rprev = ri;
        }
        s[W - 1] = med3(rprev, in_v, pinf);
    }
    __device__ __forceinline__ float median() const { return s[W / 2]; }
};

template <int W, int N>
struct Filler {
    template <typename F>
    static __device__ __forceinline__ void run(SortedWindow<W> &w, F &&fetch) {
        w.template insert<N>(fetch(N));
        if constexpr (N + 1 < W) Filler<W, N + 1>::run(w, fetch);
    }
};

// Walk n_out outputs starting at position p0 along an axis of extent n (W/2 < n).
// fetch(pos): sample at pos in [0,n).  emit4(pos, v0..v3): four consecutive outputs; emit1(pos, v): one.
template <int W, typename Fetch, typename Emit4, typename Emit1>
__device__ __forceinline__ void sliding_median(int p0, int n_out, int n, float ninf, float pinf, Fetch &&fetch,
                                               Emit4 &&emit4, Emit1 &&emit1) {
    constexpr int H = W / 2;
    SortedWindow<W> win;
    win.clear(ninf, pinf);
    Filler<W, 0>::run(win, [&](int j) { return fetch(reflect_hi(reflect_lo(p0 - H + j), n)); });
    const int p_end = p0 + n_out;
    int p = p0;
    // full groups of four: every advance is needed because output p+4 exists
    for (; p + 4 < p_end; p += 4) {
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[u] = win.median();
            win.replace(fetch(reflect_lo(p + u - H)), fetch(reflect_hi(p + u + H + 1, n)));
        }
        emit4(p, o[0], o[1], o[2], o[3]);
    }
    // last 1..4 outputs one at a time (no advance past the end: indices stay inside the axis)
    for (; p < p_end; ++p) {
        emit1(p, win.median());
        if (p + 1 < p_end) win.replace(fetch(reflect_lo(p - H)), fetch(reflect_hi(p + H + 1, n)));
    }
}

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

// LH / LP = 0 disables that role (single-filter entry points).
template <int LH, int LP>
__global__ void __launch_bounds__(1024)
hpss_median_kernel(const float *__restrict__ S, float *__restrict__ harm, float *__restrict__ perc, int K, int T,
                   int TT, int stride, int nsh, int nsp, int nwh, float ninf, float pinf) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int HH = LH / 2;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TT;
    const int t1 = min(T, t0 + TT);
    const int c0 = max(0, t0 - HH), c1 = min(T, t1 + HH);
    const int ncols = c1 - c0;
    const float *Sb = S + (size_t)b * K * T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwaves = blockDim.x >> 6;

    // stage the tile: one wave per row, coalesced along frames (8-byte loads when rows allow it)
    if (((T | c0 | ncols) & 1) == 0) {
        const int n2 = ncols >> 1;
        for (int k = wave; k < K; k += nwaves) {
            const float2v *src = reinterpret_cast<const float2v *>(Sb + (size_t)k * T + c0);
            float *dst = tile + k * stride;
            for (int c = lane; c < n2; c += 64) {
                const float2v v = __builtin_nontemporal_load(src + c);
                dst[2 * c] = v.x;
                dst[2 * c + 1] = v.y;
            }
        }
    } else {
        for (int k = wave; k < K; k += nwaves) {
            const float *src = Sb + (size_t)k * T + c0;
            float *dst = tile + k * stride;
            for (int c = lane; c < ncols; c += 64) dst[c] = __builtin_nontemporal_load(src + c);
        }
    }
    __syncthreads();

    const int nt = t1 - t0;
    if (wave < nwh) {
        if constexpr (LH > 0) {
            // harmonic: task = (segment, bin); the window runs over frames of the WHOLE clip
            const int id = wave * 64 + lane;
            if (id < K * nsh) {
                const int sg = id / K, k = id - sg * K;
                const int seglen = (nt + nsh - 1) / nsh;
                const int ts = t0 + sg * seglen;
                const int te = min(t1, ts + seglen);
                if (ts < te) {
                    const float *row = tile + k * stride - c0;
                    float *orow = harm + ((size_t)b * K + k) * T;
                    sliding_median<LH>(
                        ts, te - ts, T, ninf, pinf, [&](int t) { return row[t]; },
                        [&](int t, float v0, float v1, float v2, float v3) {
                            float4u v = {v0, v1, v2, v3};
                            *reinterpret_cast<float4u *>(orow + t) = v;
                        },
                        [&](int t, float v) { orow[t] = v; });
                }
            }
        }
    } else {
        if constexpr (LP > 0) {
            // percussive: task = (segment, frame); the window runs over bins
            const int id = (wave - nwh) * 64 + lane;
            if (id < nt * nsp) {
                const int sg = id / nt, tt = id - sg * nt;
                const int seglen = (K + nsp - 1) / nsp;
                const int ks = sg * seglen;
                const int ke = min(K, ks + seglen);
                if (ks < ke) {
                    const float *col = tile + (t0 + tt - c0);
                    float *ocol = perc + (size_t)b * K * T + t0 + tt;
                    sliding_median<LP>(
                        ks, ke - ks, K, ninf, pinf, [&](int k) { return col[k * stride]; },
                        [&](int k, float v0, float v1, float v2, float v3) {
                            __builtin_nontemporal_store(v0, ocol + (size_t)k * T);
                            __builtin_nontemporal_store(v1, ocol + (size_t)(k + 1) * T);
                            __builtin_nontemporal_store(v2, ocol + (size_t)(k + 2) * T);
                            __builtin_nontemporal_store(v3, ocol + (size_t)(k + 3) * T);
                        },
                        [&](int k, float v) { __builtin_nontemporal_store(v, ocol + (size_t)k * T); });
                }
            }
        }
    }
}

using KernelFn = void (*)(const float *, float *, float *, int, int, int, int, int, int, int, float, float);

struct Entry {
    int lh, lp;
    KernelFn fn;
};

#define SMH_MEDIAN_E2(a, b) {a, b, smh_median::hpss_median_kernel<a, b>},
#define SMH_MEDIAN_SINGLE(w) SMH_MEDIAN_E2(w, 0) SMH_MEDIAN_E2(0, w)

KernelFn find_pair_kernel(int lh, int lp);    // smh_median.hip
KernelFn find_single_kernel(int lh, int lp);  // smh_median_singles.hip

}  // namespace smh_median
