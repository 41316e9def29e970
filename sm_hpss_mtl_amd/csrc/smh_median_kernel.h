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
// network per output and no cross-lane traffic.  Measured on MI355X (tools/ubench_valu.hip): every one
// of these VALU ops occupies a SIMD for ~4.1 cycles per wave64, so the kernel is VALU-bound by
// construction (~50 VALU per output for 17x17); the tile is read from HBM once, harm/perc are written
// once (236,376 algorithmic bytes per 201x98 clip).
//
// Store shapes (tools/median_lab.hip): percussive outputs are coalesced along frames (streaming,
// non-temporal).  Harmonic outputs in the reference (B,K,T) layout are 16-byte pieces of 64 different
// rows per wave instruction and cost ~30 us per 1024 clips; the fused pipeline therefore asks for the
// time-major (B,T,K) layout, where the 64 bins of a wave are contiguous.
#pragma once
#include "smh_common.h"

namespace smh_median {

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// 'reflect' for -n <= i < 2n (one fold per side); the host guarantees window/2 + 4 < n for this kernel.
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
    // remove one instance of `out_v` (must be present) and insert `in_v`: slot by slot
    __device__ __forceinline__ void replace_chain(float out_v, float in_v) {
        float rprev = ninf;
#pragma unroll
        for (int i = 0; i < W - 1; ++i) {
            const float ri = (s[i] >= out_v) ? s[i + 1] : s[i];
            s[i] = med3(rprev, in_v, ri);
            rprev = ri;
        }
        s[W - 1] = med3(rprev, in_v, pinf);
    }
    // same update with all compare masks formed first, selects in place, insertion top-down in place
    __device__ __forceinline__ void replace_masks(float out_v, float in_v) {
        bool f[W - 1];
#pragma unroll
        for (int i = 0; i < W - 1; ++i) f[i] = s[i] >= out_v;
#pragma unroll
        for (int i = 0; i < W - 1; ++i) s[i] = f[i] ? s[i + 1] : s[i];
        s[W - 1] = med3(s[W - 2], in_v, pinf);
#pragma unroll
        for (int i = W - 2; i >= 1; --i) s[i] = med3(s[i - 1], in_v, s[i]);
        s[0] = med3(ninf, in_v, s[0]);
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

enum StoreMode { kRowVec4 = 0, kStridedPlain = 1, kStridedStream = 2 };

// Walk n_out outputs starting at p0 along an axis of extent n (W/2 + 4 < n) that lives in LDS:
// element(pos) = line[pos * es] (ES > 0: compile-time element stride).  Output `pos` goes to byte offset
// boff0 + (pos - p0) * ostep from the wave-uniform `obase`.
//   PHASES = false: one uniform loop, 4 outputs per iteration, index folding in every step
//   PHASES = true : H folded steps, pointer-increment groups of four, folded steps to the end; the phase
//                   boundaries depend on the step index only, so a wave never diverges.
template <int W, int ES, int STORE, bool PHASES>
__device__ __forceinline__ void sliding_median(const float *line, int es_rt, int p0, int n_out, int n, float ninf,
                                               float pinf, char *obase, unsigned boff, unsigned ostep) {
    constexpr int H = W / 2;
    const int es = ES > 0 ? ES : es_rt;
    auto at = [&](int pos) { return line[pos * es]; };
    auto emit1 = [&](float v) {
        if constexpr (STORE == kStridedStream) __builtin_nontemporal_store(v, reinterpret_cast<float *>(obase + boff));
        else *reinterpret_cast<float *>(obase + boff) = v;
        boff += ostep;
    };
    auto emit4 = [&](const float *o) {
        if constexpr (STORE == kRowVec4) {
            float4u v = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<float4u *>(obase + boff) = v;
            boff += 4 * ostep;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) emit1(o[u]);
        }
    };
    SortedWindow<W> win;
    win.clear(ninf, pinf);
    Filler<W, 0>::run(win, [&](int j) { return at(reflect_hi(reflect_lo(p0 - H + j), n)); });
    const int p_end = p0 + n_out;
    int p = p0;
    if constexpr (!PHASES) {
        for (; p + 4 < p_end; p += 4) {
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                o[u] = win.median();
                win.replace_masks(at(reflect_lo(p + u - H)), at(reflect_hi(p + u + H + 1, n)));
            }
            emit4(o);
        }
    } else {
        const int head_end = min(p_end, p0 + H);
        for (; p < head_end; ++p) {
            emit1(win.median());
            if (p + 1 < p_end) win.replace_chain(at(reflect_lo(p - H)), at(reflect_hi(p + H + 1, n)));
        }
        if (p < p_end) {
            const float *pin = line + (p + H + 1) * es;
            const float *pout = line + (p - H) * es;
            const int steady_end = p_end - (H + 4);
            for (; p + 4 <= steady_end; p += 4) {
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    o[u] = win.median();
                    win.replace_chain(pout[u * es], pin[u * es]);
                }
                pin += 4 * es;
                pout += 4 * es;
                emit4(o);
            }
        }
    }
    for (; p < p_end; ++p) {
        emit1(win.median());
        if (p + 1 < p_end) win.replace_chain(at(reflect_lo(p - H)), at(reflect_hi(p + H + 1, n)));
    }
}

// LH / LP = 0 disables that role (single-filter entry points).
// harm_tmajor != 0: harm is written as (B, T, K) instead of (B, K, T).
template <int LH, int LP>
__global__ void __launch_bounds__(1024)
hpss_median_kernel(const float *__restrict__ S, float *__restrict__ harm, float *__restrict__ perc, int K, int T,
                   int TT, int stride, int nsh, int nsp, int nwh, int harm_tmajor, float ninf, float pinf) {
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

    // stage the tile: one wave per row (no index division), kRowBatch rows of loads in flight per wave
    // before the first LDS write (the tile load is latency-, not bandwidth-limited).
    constexpr int kRowBatch = 6;
    if (((T | c0 | ncols) & 1) == 0 && ncols <= 128) {
        const int n2 = ncols >> 1;
        for (int k0 = wave; k0 < K; k0 += nwaves * kRowBatch) {
            float2v v[kRowBatch];
#pragma unroll
            for (int r = 0; r < kRowBatch; ++r) {
                const int k = min(k0 + r * nwaves, K - 1);
                v[r] = __builtin_nontemporal_load(reinterpret_cast<const float2v *>(Sb + (size_t)k * T + c0) +
                                                  min(lane, n2 - 1));
            }
#pragma unroll
            for (int r = 0; r < kRowBatch; ++r) {
                const int k = k0 + r * nwaves;
                if (k < K && lane < n2) {
                    tile[k * stride + 2 * lane] = v[r].x;
                    tile[k * stride + 2 * lane + 1] = v[r].y;
                }
            }
        }
    } else {
        for (int k0 = wave; k0 < K; k0 += nwaves * kRowBatch) {
            for (int cb = 0; cb < ncols; cb += 64) {
                float v[kRowBatch];
                const int c = cb + lane;
#pragma unroll
                for (int r = 0; r < kRowBatch; ++r) {
                    const int k = min(k0 + r * nwaves, K - 1);
                    v[r] = __builtin_nontemporal_load(Sb + (size_t)k * T + c0 + min(c, ncols - 1));
                }
#pragma unroll
                for (int r = 0; r < kRowBatch; ++r) {
                    const int k = k0 + r * nwaves;
                    if (k < K && c < ncols) tile[k * stride + c] = v[r];
                }
            }
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
                    char *ob = reinterpret_cast<char *>(harm + (size_t)b * K * T);
                    if (harm_tmajor)
                        sliding_median<LH, 1, kStridedPlain, false>(row, 1, ts, te - ts, T, ninf, pinf, ob,
                                                                    (unsigned)(ts * K + k) * 4u, (unsigned)K * 4u);
                    else
                        sliding_median<LH, 1, kRowVec4, false>(row, 1, ts, te - ts, T, ninf, pinf, ob,
                                                               (unsigned)(k * T + ts) * 4u, 4u);
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
                    sliding_median<LP, 0, kStridedStream, true>(col, stride, ks, ke - ks, K, ninf, pinf,
                                                                reinterpret_cast<char *>(perc + (size_t)b * K * T),
                                                                (unsigned)(ks * T + t0 + tt) * 4u, (unsigned)T * 4u);
                }
            }
        }
    }
}

using KernelFn = void (*)(const float *, float *, float *, int, int, int, int, int, int, int, int, float, float);

struct Entry {
    int lh, lp;
    KernelFn fn;
};

#define SMH_MEDIAN_E2(a, b) {a, b, smh_median::hpss_median_kernel<a, b>},
#define SMH_MEDIAN_SINGLE(w) SMH_MEDIAN_E2(w, 0) SMH_MEDIAN_E2(0, w)

KernelFn find_pair_kernel(int lh, int lp);    // smh_median.hip
KernelFn find_single_kernel(int lh, int lp);  // smh_median_singles.hip

}  // namespace smh_median
