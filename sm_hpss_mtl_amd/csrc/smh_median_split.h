// Block-split sliding median: the HPSS median kernel for windows up to 21.
//
// Same contract and same LDS tile as smh_median_kernel.h (harmonic lanes walk frames, percussive lanes walk bins,
// 'reflect' boundary, bit-exact selection), but the per-lane window is no longer one sorted array that loses its
// oldest element at every step (delete = compare + select per slot).  The padded line is cut into blocks of W
// elements; the window ending at offset p of block k is
//        (suffix of block k-1 of size a = W-1-p)  U  (prefix of block k of size b = p+1)
// and both sorted runs only ever GROW, by insertion = one v_med3_f32 per kept slot:
//   * suffix runs: block k-1 walked backwards once, every size kept in registers (the "history");
//   * prefix runs: block k walked forwards, in place.
// The median is the H-th smallest of the union of two sorted runs,
//        r = min_i max(A[i-1], B[H-i]),   i in [max(0, H+1-b), min(a, H+1)],
// so of a run of size s only the indices [max(0, s-H-1), min(s-1, H)] are ever read -- by the selection and by the
// insertion that builds the next size -- and only those are computed and stored.  For W = 17 this is 9.6 med3 +
// 5.3 max + 2.7 min3 per output instead of the 51 VALU of the delete/insert window (tools/median_split_sim.py
// checks the scheme and these index ranges against scipy.ndimage.median_filter).
//
// Compiled with -fno-honor-nans (build.py): spectrogram magnitudes are finite, and without the flag every
// fminf/fmaxf on a value loaded from LDS costs an extra canonicalising v_max_f32.
#pragma once
#include <cstdint>
#include <type_traits>
#include <utility>

#include "smh_median_kernel.h"
#include "smh_rag.h"

namespace smh_median {

template <int W>
struct SplitTraits {
    static constexpr int H = W / 2;
    static constexpr int lo(int s) { return s - H - 1 > 0 ? s - H - 1 : 0; }
    static constexpr int hi(int s) { return s - 1 < H ? s - 1 : H; }
    static constexpr int off(int s) {  // first register of the run of size s inside the history
        int o = 0;
        for (int r = 1; r < s; ++r) o += hi(r) - lo(r) + 1;
        return o;
    }
    static constexpr int HIST = off(W);  // runs of size 1 .. W-1
};

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (ascending) -- keeps every register-array
// index a constant expression
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
template <int B, int E, typename F>
__device__ __forceinline__ void static_for_down(F &&f) {  // I = E-1 .. B
    if constexpr (B < E) {
        f(std::integral_constant<int, E - 1>{});
        static_for_down<B, E - 1>(f);
    }
}

// one slot of "insert x into a sorted run": below/above are the old neighbours (absent at the ends)
template <bool HasBelow, bool HasAbove>
__device__ __forceinline__ float insert_slot(float below, float x, float above) {
    if constexpr (HasBelow && HasAbove) return med3(below, x, above);
    else if constexpr (HasBelow) return fmaxf(below, x);
    else if constexpr (HasAbove) return fminf(x, above);
    else return x;
}

// History of block e[0..W-1]: suffix runs of size 1..W-1 (run s = sorted e[W-s .. W-1], kept indices only).
template <int W>
__device__ __forceinline__ void build_history(const float (&e)[W], float (&h)[SplitTraits<W>::HIST]) {
    using Tr = SplitTraits<W>;
    static_for<1, W>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const float x = e[W - s];
        static_for<Tr::lo(s), Tr::hi(s) + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool has_below = i - 1 >= 0;
            constexpr bool has_above = i <= s - 2;
            constexpr int po = Tr::off(s > 1 ? s - 1 : 1), pl = Tr::lo(s > 1 ? s - 1 : 1);
            const float below = has_below ? h[has_below ? po + (i - 1) - pl : 0] : 0.f;
            const float above = has_above ? h[has_above ? po + i - pl : 0] : 0.f;
            h[Tr::off(s) + i - Tr::lo(s)] = insert_slot<has_below, has_above>(below, x, above);
        });
    });
}

// Walk one line that lives in LDS: element(pos) = line[pos * es], outputs p0 .. p0+n_out-1 of an axis of extent n
// (W/2 + 4 < n).  `n_steps` >= n_out is wave-uniform: every lane of the wave runs the same blocks and lanes with
// fewer outputs only skip their stores.  Output j goes to obase + boff + j * ostep.
template <int W, int ES, int STORE>
__device__ __forceinline__ void split_median_walk(const float *line, int es_rt, int p0, int n_out, int n_steps, int n,
                                                  char *obase, unsigned boff, unsigned ostep) {
    using Tr = SplitTraits<W>;
    constexpr int H = W / 2;
    const int es = ES > 0 ? ES : es_rt;
    float e[W];
    float h[Tr::HIST];
    float pre[H + 1];

    // block of padded elements q0 .. q0+W-1  (padded index q <-> axis position p0 - H + q).  Positions past the
    // last one the lane's own outputs need (plast) only feed outputs that are never stored: they are clamped so
    // that no lane reads outside the part of the tile that was staged for it.
    const int plast = p0 + n_out - 1 + H;
    auto load_block = [&](int q0) {
        const int lo = p0 - H + q0;
        const bool interior = (lo >= 0) & (lo + W <= n) & (lo + W - 1 <= plast);
        if (__all(interior)) {
            const float *src = line + lo * es;
            static_for<0, W>([&](auto uc) { e[decltype(uc)::value] = src[decltype(uc)::value * es]; });
        } else {
            static_for<0, W>([&](auto uc) {
                const int pos = min(lo + decltype(uc)::value, plast);
                e[decltype(uc)::value] = line[reflect_hi(reflect_lo(pos), n) * es];
            });
        }
    };
    // Stores: the byte offset advances by ostep per output (no multiply).  Every block runs all W steps as
    // straight-line code (elements past a lane's range are clamped duplicates); only the blocks in which some lane
    // of the wave runs past its own n_out (!FAST) pay for the per-lane bound check on the store.
    unsigned off = boff;
    auto emit = [&](int j, float v, bool guard) {
        if (!guard || j < n_out) {
            float *dst = reinterpret_cast<float *>(obase + off);
            // Plain stores also for the percussive rows (kStridedStream): a wave's 256-byte piece of a 392-byte row shares its
            // first and last line with the neighbouring pieces, which arrive a bin step later from this or another wave of the
            // workgroup -- in L2 they merge into whole lines; as nontemporal stores the partial lines left for HBM at once
            // (WRITE_SIZE 184 MB per 1024 clips against 160 MB now = the bytes of harm + perc; tools/gpu/r3_median_nt.sh).
#ifdef SMH_NT_STORES  // (A/B build only)
            if constexpr (STORE == kStridedStream) __builtin_nontemporal_store(v, dst);
            else *dst = v;
#else
            *dst = v;
#endif
        }
        off += ostep;
    };
    int n_min = n_out;  // smallest n_out among the lanes of this wave
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n_min = min(n_min, __shfl_xor(n_min, o, 64));

    auto forward = [&](auto fast_c, int j0) {
        constexpr bool FAST = decltype(fast_c)::value;
        static_for<0, W>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            constexpr int b = p + 1, a = W - 1 - p;
            {
                // grow the prefix run to size b (in place, top slot first: old neighbours are still intact)
                static_for_down<Tr::lo(b), Tr::hi(b) + 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr bool has_below = i - 1 >= 0;
                    constexpr bool has_above = i <= b - 2;
                    const float below = has_below ? pre[has_below ? i - 1 : 0] : 0.f;
                    const float above = has_above ? pre[i] : 0.f;
                    pre[i] = insert_slot<has_below, has_above>(below, e[p], above);
                });
                // H-th smallest of suffix run (size a) U prefix run (size b)
                constexpr int i_lo = (H + 1 - b) > 0 ? (H + 1 - b) : 0;
                constexpr int i_hi = a < H + 1 ? a : H + 1;
                constexpr int ao = Tr::off(a > 0 ? a : 1), al = Tr::lo(a > 0 ? a : 1);
                float r = 0.f;
                static_for<i_lo, i_hi + 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    float t;
                    if constexpr (i == 0) t = pre[H];
                    else if constexpr (H - i < 0) t = h[ao + (i - 1) - al];
                    else t = fmaxf(h[ao + (i - 1) - al], pre[H - i]);
                    r = (i == i_lo) ? t : fminf(r, t);
                });
                emit(j0 + p, r, !FAST);
            }
        });
    };

    // block 0: only its last window is an output (the whole block); its history serves block 1
    load_block(0);
    build_history<W>(e, h);
    {
        constexpr int po = Tr::off(W - 1), pl = Tr::lo(W - 1);
        emit(0, med3(h[po + (H - 1) - pl], e[0], h[po + H - pl]), false);  // n_out >= 1
    }
    for (int j0 = 1; j0 < n_steps; j0 += W) {
        load_block(j0 + W - 1);  // windows ending in this block: outputs j0 .. j0+W-1
        if (j0 + W <= n_min) forward(std::true_type{}, j0);
        else forward(std::false_type{}, j0);
        if (j0 + W < n_steps) build_history<W>(e, h);
    }
}

// History of a block of W-1 elements e[0..W-2]: suffix runs of size 1..W-1 (run s = sorted e[W-1-s .. W-2]).
template <int W>
__device__ __forceinline__ void build_history_b(const float (&e)[W - 1], float (&h)[SplitTraits<W>::HIST]) {
    using Tr = SplitTraits<W>;
    static_for<1, W>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const float x = e[W - 1 - s];
        static_for<Tr::lo(s), Tr::hi(s) + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool has_below = i - 1 >= 0;
            constexpr bool has_above = i <= s - 2;
            constexpr int po = Tr::off(s > 1 ? s - 1 : 1), pl = Tr::lo(s > 1 ? s - 1 : 1);
            const float below = has_below ? h[has_below ? po + (i - 1) - pl : 0] : 0.f;
            const float above = has_above ? h[has_above ? po + i - pl : 0] : 0.f;
            h[Tr::off(s) + i - Tr::lo(s)] = insert_slot<has_below, has_above>(below, x, above);
        });
    });
}

// Variant with blocks of W-1 elements: every window is (suffix of the previous block, size W-1-p) U (prefix of the
// current one, size p+1) for p = 0 .. W-2, so block 0 only provides history and every later block yields exactly
// W-1 outputs (16 for W = 17) -- whole float4 groups for the harmonic rows.  Harmonic outputs go to the 16-frame
// blocked layout harm[b][t/16][k][t%16] (lanes of the feature kernel are frames: 64-byte runs per bin), written as
// one float4 per four outputs.  `gk16` = bytes between frame groups (K*64), `kofs` = k*64.
template <int W, int ES>
__device__ __forceinline__ void split_median_walk_blocked(const float *line, int p0, int n_out, int n_steps, int n,
                                                          char *obase, unsigned gk16, unsigned kofs) {
    using Tr = SplitTraits<W>;
    constexpr int H = W / 2, BK = W - 1;
    float e[BK];
    float h[Tr::HIST];
    float pre[H + 1];
    const int plast = p0 + n_out - 1 + H;
    auto load_block = [&](int q0) {
        const int lo = p0 - H + q0;
        const bool interior = (lo >= 0) & (lo + BK <= n) & (lo + BK - 1 <= plast);
        if (__all(interior)) {
            const float *src = line + lo * ES;
            static_for<0, BK>([&](auto uc) { e[decltype(uc)::value] = src[decltype(uc)::value * ES]; });
        } else {
            static_for<0, BK>([&](auto uc) {
                const int pos = min(lo + decltype(uc)::value, plast);
                e[decltype(uc)::value] = line[reflect_hi(reflect_lo(pos), n) * ES];
            });
        }
    };
    load_block(0);
    build_history_b<W>(e, h);
    for (int j0 = 0; j0 < n_steps; j0 += BK) {
        load_block(j0 + BK);
        float o4[4];
        static_for<0, BK>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            constexpr int b = p + 1, a = BK - p;
            static_for_down<Tr::lo(b), Tr::hi(b) + 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr bool has_below = i - 1 >= 0;
                constexpr bool has_above = i <= b - 2;
                const float below = has_below ? pre[has_below ? i - 1 : 0] : 0.f;
                const float above = has_above ? pre[i] : 0.f;
                pre[i] = insert_slot<has_below, has_above>(below, e[p], above);
            });
            constexpr int i_lo = (H + 1 - b) > 0 ? (H + 1 - b) : 0;
            constexpr int i_hi = a < H + 1 ? a : H + 1;
            constexpr int ao = Tr::off(a), al = Tr::lo(a);
            float r = 0.f;
            static_for<i_lo, i_hi + 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float t;
                if constexpr (i == 0) t = pre[H];
                else if constexpr (H - i < 0) t = h[ao + (i - 1) - al];
                else t = fmaxf(h[ao + (i - 1) - al], pre[H - i]);
                r = (i == i_lo) ? t : fminf(r, t);
            });
            o4[p & 3] = r;
            if constexpr ((p & 3) == 3 || p == BK - 1) {
                constexpr int cnt = (p & 3) + 1, pf = p - (p & 3);
                const int j = j0 + pf, t = p0 + j;  // first frame of the chunk
                char *dst = obase + (unsigned)(t >> 4) * gk16 + kofs + (unsigned)(t & 15) * 4u;
                if (cnt == 4 && (t & 3) == 0 && j + 3 < n_out) {
                    float4u v = {o4[0], o4[1], o4[2], o4[3]};
                    *reinterpret_cast<float4u *>(dst) = v;
                } else {
#pragma unroll
                    for (int c = 0; c < cnt; ++c) {
                        const int tc = t + c;
                        if (j + c < n_out)
                            *reinterpret_cast<float *>(obase + (unsigned)(tc >> 4) * gk16 + kofs + (unsigned)(tc & 15) * 4u) = o4[c];
                    }
                }
            }
        });
        if (j0 + BK < n_steps) build_history_b<W>(e, h);
    }
}

// Register budget: the history is H*(H+1)-1 registers (80 for W = 17, 120 for W = 21), so the workgroup size is
// chosen for 4 waves per SIMD (128 VGPRs) up to W = 17 and 3 waves per SIMD (168 VGPRs) above; two workgroups share
// a CU (LDS: 2 x 80 KB), one staging its tile while the other computes.
template <int LH, int LP>
struct SplitCfg {
    static constexpr int kMaxW = LH > LP ? LH : LP;
    static constexpr int kThreads = kMaxW <= 17 ? 512 : 384;
    static constexpr int kWavesPerSimd = kMaxW <= 17 ? 4 : 3;
    // float2 loads in flight per lane in the flat tile load: batch x threads x 2 >= 201 x 98 (the reference clip)
    static constexpr int kFlatBatch = kMaxW <= 17 ? 20 : 27;
};

// Same arguments and tile as hpss_median_kernel; LH / LP = 0 disables that role.
// rag != nullptr (smh_rag.h): clips of DIFFERENT lengths in one launch -- n_items (clip, frame tile) items, in clip-major order cut
// into 8 contiguous ranges (one per XCD: neighbouring tiles share their halo columns in that L2), per-clip T and buffer offsets from
// the descriptor table, harm in the 16-frame blocked layout.  Selection only: a clip's medians do not depend on how it was tiled.
template <int LH, int LP>
__global__ void __launch_bounds__((SplitCfg<LH, LP>::kThreads), (SplitCfg<LH, LP>::kWavesPerSimd))
hpss_median_split_kernel(const float *__restrict__ S, float *__restrict__ harm, float *__restrict__ perc, int K, int T,
                         int TT, int stride, int nsh, int nsp, int nwh, int harm_tmajor, float probe, float,
                         const smh_rag::Clip *__restrict__ rag, const smh_rag::Item *__restrict__ items, int n_items) {
    // probe > 0 (tools/gpu/r2_fusion_bound.sh, SMH_MEDIAN_PROBE_NOLOAD): the tile is NOT staged -- what the medians cost when
    // their input is already in LDS, i.e. the upper bound of fusing this kernel behind the STFT; the outputs are garbage
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int HH = LH / 2;
    int b = blockIdx.y, tile_i = blockIdx.x;
    size_t spec_off, harm_off;
    if (rag) {
        const unsigned total = (unsigned)n_items, per_xcd = (total + 7u) >> 3;
        const unsigned j = blockIdx.x >> 3, n = (blockIdx.x & 7u) * per_xcd + j;
        if (j >= per_xcd || n >= total) return;
        const smh_rag::Item item = items[n];
        b = item.clip, tile_i = item.tile;
        T = rag[b].T;
        spec_off = (size_t)rag[b].spec_off, harm_off = (size_t)rag[b].harm_off;
    } else {
        spec_off = (size_t)b * K * T;
        harm_off = harm_tmajor == 2 ? (size_t)b * ((T + 15) >> 4) * K * 16 : spec_off;
    }
    const int t0 = tile_i * TT;
    const int t1 = min(T, t0 + TT);
    const int c0 = max(0, t0 - HH), c1 = min(T, t1 + HH);
    const int ncols = c1 - c0;
    const float *Sb = S + spec_off;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwaves = blockDim.x >> 6;

    // stage the tile.  Whole clip in one tile (the common case): the K x T block is contiguous in HBM, so it is read
    // as one flat float2 stream with every load of a lane in flight at once (a single HBM round trip per workgroup)
    // and scattered to the odd-stride tile: element g = k*T + t lands at g + k*(stride - T).
    constexpr int kFlatBatch = SplitCfg<LH, LP>::kFlatBatch;
    if (probe > 0.f) {
        for (int i = threadIdx.x; i < K * stride; i += blockDim.x) tile[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.f);
    } else if (TT >= T && ((K * T) & 1) == 0 && K * T <= kFlatBatch * 2 * (int)blockDim.x) {
        const int n2 = (K * T) >> 1;
        const float2v *src = reinterpret_cast<const float2v *>(Sb);
        const unsigned magic = 0xFFFFFFFFu / (unsigned)T + 1u;  // g / T == umulhi(g, magic) for g * T < 2^32
        const int extra = stride - T;
        float2v v[kFlatBatch];
#pragma unroll
        for (int r = 0; r < kFlatBatch; ++r) {
            const int idx = r * (int)blockDim.x + (int)threadIdx.x;
            v[r] = __builtin_nontemporal_load(src + min(idx, n2 - 1));
        }
#pragma unroll
        for (int r = 0; r < kFlatBatch; ++r) {
            const int idx = r * (int)blockDim.x + (int)threadIdx.x;
            if (idx < n2) {
                const unsigned g = 2u * (unsigned)idx;
                const unsigned k0 = __umulhi(g, magic), k1 = __umulhi(g + 1u, magic);
                tile[g + k0 * extra] = v[r].x;
                tile[g + 1u + k1 * extra] = v[r].y;
            }
        }
    } else if (((T | c0 | ncols) & 1) == 0 && ncols <= 128) {
        constexpr int kRowBatch = 6;
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
        constexpr int kRowBatch = 6;
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
                    char *ob = reinterpret_cast<char *>(harm + harm_off);
                    if (harm_tmajor == 2)  // (B, ceil(T/16), K, 16): the clip's image is ceil(T/16)*K*16 floats
                        split_median_walk_blocked<LH, 1>(row, ts, te - ts, seglen, T, ob, (unsigned)K * 64u, (unsigned)k * 64u);
                    else if (harm_tmajor)
                        split_median_walk<LH, 1, kStridedPlain>(row, 1, ts, te - ts, seglen, T, ob,
                                                                (unsigned)(ts * K + k) * 4u, (unsigned)K * 4u);
                    else
                        split_median_walk<LH, 1, kStridedPlain>(row, 1, ts, te - ts, seglen, T, ob,
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
                    split_median_walk<LP, 0, kStridedStream>(col, stride, ks, ke - ks, seglen, K,
                                                             reinterpret_cast<char *>(perc + spec_off),
                                                             (unsigned)(ks * T + t0 + tt) * 4u, (unsigned)T * 4u);
                }
            }
        }
    }
}

// ---- persistent form for whole-clip tiles (the fused pipeline at batch >= 2 x CUs) --------------------------------
// One workgroup per CU loops over clips b = blockIdx.x, + gridDim.x, ... with TWO tiles in LDS: while the 8 waves
// compute on one, the next clip streams into the other by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write).
// The DMA writes LDS linearly, so the tile is the flat K x T image (row stride T): conflict-free for the harmonic
// lanes (lane <-> bin) whenever gcd(T, 64) <= 2 -- T = 98 -- because ds_read_b32 is serviced in 32-lane halves.
// A clip need not start on a 16-byte boundary (K*T*4 = 78 792 B): the copy starts at the aligned address below it
// and the tile pointer is moved up by the same `mis` bytes.
__host__ __device__ constexpr int persist_tile_bytes(int K, int T) { return ((K * T * 4 + 16) + 1023) & ~1023; }

// THREADS = 512 / 768 / 1024: 2 / 3 / 4 waves per SIMD and 256 / 168 / 128 VGPRs per lane.
template <int LH, int LP, int THREADS>
__global__ void __launch_bounds__(THREADS, (THREADS / 256))
hpss_median_persist_kernel(const float *__restrict__ S, float *__restrict__ harm, float *__restrict__ perc, int B, int K,
                           int T, int nsh, int nsp, int nwh, int harm_tmajor) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tile_bytes = persist_tile_bytes(K, T);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwaves = blockDim.x >> 6;
    const size_t clip_bytes = (size_t)K * T * 4;

    auto issue_load = [&](int b, int buf) {
        const char *src = reinterpret_cast<const char *>(S) + (size_t)b * clip_bytes;
        const int mis = (int)(reinterpret_cast<uintptr_t>(src) & 15);
        src -= mis;
        const int nchunks = ((int)clip_bytes + mis + 15) >> 4;
        char *dst = lds + buf * tile_bytes;
        for (int i = wave; i * 64 < nchunks; i += nwaves) {
            const int c = i * 64 + lane;
            if (c < nchunks)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + (size_t)c * 16),
                    (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
        }
    };

    int b = blockIdx.x;
    if (b >= B) return;
    issue_load(b, 0);
    for (int it = 0; b < B; b += gridDim.x, ++it) {
        const int cur = it & 1;
        // tile `cur` has landed (each wave drains its own DMAs) and every wave is done with tile cur^1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (b + (int)gridDim.x < B) issue_load(b + gridDim.x, cur ^ 1);
        const int mis = (int)((reinterpret_cast<uintptr_t>(S) + (size_t)b * clip_bytes) & 15);
        const float *tile = reinterpret_cast<const float *>(lds + cur * tile_bytes + mis);
        if (wave < nwh) {
            const int id = wave * 64 + lane;
            if (id < K * nsh) {
                const int sg = id / K, k = id - sg * K;
                const int seglen = (T + nsh - 1) / nsh;
                const int ts = sg * seglen;
                const int te = min(T, ts + seglen);
                if (ts < te) {
                    const float *row = tile + k * T;
                    char *ob = reinterpret_cast<char *>(harm + (size_t)b * K * T);
                    if (harm_tmajor == 2)  // (B, ceil(T/16), K, 16): the clip's image is ceil(T/16)*K*16 floats
                        split_median_walk_blocked<LH, 1>(row, ts, te - ts, seglen, T,
                                                         reinterpret_cast<char *>(harm + (size_t)b * ((T + 15) >> 4) * K * 16),
                                                         (unsigned)K * 64u, (unsigned)k * 64u);
                    else if (harm_tmajor)
                        split_median_walk<LH, 1, kStridedPlain>(row, 1, ts, te - ts, seglen, T, ob,
                                                                (unsigned)(ts * K + k) * 4u, (unsigned)K * 4u);
                    else
                        split_median_walk<LH, 1, kStridedPlain>(row, 1, ts, te - ts, seglen, T, ob,
                                                                (unsigned)(k * T + ts) * 4u, 4u);
                }
            }
        } else {
            const int id = (wave - nwh) * 64 + lane;
            if (id < T * nsp) {
                const int sg = id / T, tt = id - sg * T;
                const int seglen = (K + nsp - 1) / nsp;
                const int ks = sg * seglen;
                const int ke = min(K, ks + seglen);
                if (ks < ke)
                    split_median_walk<LP, 0, kStridedStream>(tile + tt, T, ks, ke - ks, seglen, K,
                                                             reinterpret_cast<char *>(perc + (size_t)b * K * T),
                                                             (unsigned)(ks * T + tt) * 4u, (unsigned)T * 4u);
            }
        }
    }
}

using PersistFn = void (*)(const float *, float *, float *, int, int, int, int, int, int, int);
struct PersistEntry {
    int lh, lp, threads;
    PersistFn fn;
};
const PersistEntry *find_persist_kernel(int lh, int lp, int threads);  // smh_median_split.hip

constexpr int kSplitMaxWindow = 21;  // 120 history registers; larger windows keep the delete/insert kernel

using SplitFn = void (*)(const float *, float *, float *, int, int, int, int, int, int, int, int, float, float,
                         const smh_rag::Clip *, const smh_rag::Item *, int);
struct SplitEntry {
    int lh, lp, threads;
    SplitFn fn;
};
const SplitEntry *find_split_kernel(int lh, int lp);  // smh_median_split.hip

}  // namespace smh_median
