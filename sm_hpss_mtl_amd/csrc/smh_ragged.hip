// Ragged batches: clips of DIFFERENT lengths in one call, one launch per stage.
//
// The reference's generators process whole files of any length one by one (Proposed_Work_Results.py:92-95, 131-134, 189-192,
// 465-474 -> get_featuregram, lib/preprocessing.py:355-457, and get_feature_patches, :137-292).  Round 3 ran such a batch as a
// chain of seven B = 1 launches per file; here the host decides, from the lengths alone, a descriptor table (smh_rag::Clip) and a
// flat (clip, tile) work list per stage, uploads both in one copy, and every stage is ONE launch over all files:
//
//   stft400_kernel / stft_mag_kernel      items (clip, 20- or 16-frame tile)          audio -> S            (workspace)
//   hpss_median_split_kernel              items (clip, 76-frame tile + halo)          S -> harm (16-frame blocked), perc
//   clips whose featuregram fits an LDS image (T <= 161 for 240 rows; smh_features_blocked_ok):
//     features_half_kernel<RAG> (even T) / features_clip_kernel<RAG> (odd T): the equal-length path's kernels, shapes per clip
//   longer clips, three streaming kernels of this file:
//     rag_walk_kernel     items (clip, 128-frame chunk): soft masks + mel sums -> fv (magnitudes), per-array maximum (atomicMax)
//     rag_stats_kernel    one wave per featuregram row: StandardScaler statistics of the dB row
//     rag_final_kernel    items (clip, 64-frame chunk): dB + top-dB floor -> fv (final), standardised time-major patches
//
// Bits: medians are selections and an STFT frame does not depend on its tile, so those stages give every clip what the equal-length
// entry points give it; the LDS-image clips run the same kernel code as smh_frontend_f32; the streaming kernels ARE what
// smh_frontend_f32 runs for clips beyond the LDS image (it builds the same tables for its B equal clips).  Clips no ragged kernel
// covers (a handful of frames, misaligned starts, window pairs without a block-split kernel) go through smh_frontend_f32 one by one.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "smh_common.h"
#include "smh_feat.h"
#include "smh_rag.h"

namespace smh_rag {

struct Staging {
    static constexpr int kSlots = 4;
    struct Slot {
        void *host = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool in_flight = false;
    } slot[kSlots];
    int next = 0;
};

void destroy_staging(Staging *s) {
    if (!s) return;
    for (auto &sl : s->slot) {
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.host) (void)hipHostFree(sl.host);
    }
    delete s;
}

}  // namespace smh_rag

namespace {

using smh_feat::FeatPlan;
using smh_rag::Clip;
using smh_rag::Item;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kAmin = 1e-10f;  // librosa.power_to_db amin
constexpr int kWalkFrames = 128;  // frames per walk item: 64 lanes x a pair of frames
constexpr int kFinalFrames = 64;  // frames per finalisation item

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// host -> device copy of a call's tables through a pinned slot of the context (ring of four; a slot is reused only after its
// previous copy has completed, which in practice it long has)
int stage_upload(const smh_ctx *ctx, const void *src, size_t bytes, void *d_dst, hipStream_t st) {
    std::lock_guard<std::mutex> lock(ctx->rag_mu);
    if (!ctx->rag_staging) ctx->rag_staging = new smh_rag::Staging();
    smh_rag::Staging &sg = *ctx->rag_staging;
    smh_rag::Staging::Slot &sl = sg.slot[sg.next];
    sg.next = (sg.next + 1) % smh_rag::Staging::kSlots;
    if (!sl.ev) SMH_CHECK_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.in_flight) {
        SMH_CHECK_HIP(hipEventSynchronize(sl.ev));
        sl.in_flight = false;
    }
    if (sl.cap < bytes) {
        if (sl.host) SMH_CHECK_HIP(hipHostFree(sl.host));
        sl.host = nullptr, sl.cap = 0;
        const size_t cap = align_up(std::max<size_t>(bytes, 64 * 1024), 64 * 1024);
        SMH_CHECK_HIP(hipHostMalloc(&sl.host, cap, hipHostMallocDefault));
        sl.cap = cap;
    }
    memcpy(sl.host, src, bytes);
    SMH_CHECK_HIP(hipMemcpyAsync(d_dst, sl.host, bytes, hipMemcpyHostToDevice, st));
    SMH_CHECK_HIP(hipEventRecord(sl.ev, st));
    sl.in_flight = true;
    return SMH_OK;
}

// item n of a 1-D grid whose workgroup i runs on XCD i % 8: the list in 8 contiguous ranges, one per XCD (smh_stft.hip has the reasoning)
__device__ __forceinline__ bool xcd_item(int n_items, unsigned &n) {
    const unsigned total = (unsigned)n_items, per_xcd = (total + 7u) >> 3;
    const unsigned j = blockIdx.x >> 3;
    n = (blockIdx.x & 7u) * per_xcd + j;
    return j < per_xcd && n < total;
}

// ---------------------------------------------------------------------------------------------------------------------------
// rag_walk_kernel: the bin walk of features_half_kernel (smh_feat.hip) for clips beyond the LDS image.  One workgroup per
// (clip, 128-frame chunk, half): nseg waves = row segments, lane = PAIR of frames; a wave walks the bins of its segment once,
// evaluates this half's soft mask  S own^2 / (own^2 + other^2)  per bin (lib/preprocessing.py:418; librosa.util.softmask with
// power 2) and adds it into the pending mel filters (:419-422); a finished filter's SUM goes to the featuregram row (the dB
// conversion needs the array's maximum, which only exists after the launch) and into the running maximum of its (clip, half) array.
// Even T: 8-byte loads and stores; odd T: two 4-byte accesses per pair, the lone last frame paired with itself.
// ---------------------------------------------------------------------------------------------------------------------------
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Addresses in the walk are a buffer descriptor of the clip's array (four SGPRs, built from wave-uniform values), a wave-uniform
// byte offset of the row (soffset) and a 32-bit lane offset in bytes (voffset): buffer_load_dwordx2 v, v_off, s[rsrc], s_row offen.
// Written as plain pointers hipcc keeps a 64-bit pointer per lane and array and adds every row offset to it with a
// v_lshl_add_u64 (140 of them in the first version of this kernel, 13 spilled registers at 80 VGPRs).
template <bool EVEN>
__device__ __forceinline__ f32x2 load_pair(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned d1) {
    if constexpr (EVEN) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        return f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
    } else {
        return f32x2{__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)),
                     __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff + 4u * d1, soff, 0))};
    }
}

template <int NP, bool EVEN>
__device__ __forceinline__ void walk_pairs(const FeatPlan &fp, int seg, int half, int lane, int chunk, const float *S,
                                           const float *harmc, const float *perc, int K, int T, int rows, float *fvc, float &mx) {
    const int t0 = chunk * kWalkFrames + 2 * lane;
    const bool active = t0 < T;
    const int tp = min(t0, EVEN ? T - 2 : T - 1);  // this lane's pair starts here (an even frame)
    const unsigned d1 = (EVEN || tp + 1 < T) ? 1u : 0u;   // the second frame of the pair, or the first again
    const int m1 = fp.m1[seg], kbeg = fp.kbeg[seg], kend = fp.kend[seg];
    int mcur = fp.m0[seg];
    const float *plan = fp.plan + fp.off[seg];
    f32x2 acc[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) acc[e] = f32x2{0.f, 0.f};
    const unsigned spec_bytes = 4u * (unsigned)K * (unsigned)T, harm_bytes = 64u * (unsigned)K * (unsigned)((T + 15) >> 4);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S), 0, spec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(perc), 0, spec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(harmc), 0, harm_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(fvc, 0, 8u * (unsigned)rows * (unsigned)T, 0x00020000);
    // lane offsets: lo = the pair in a (K, T) row, ho = in bin 0 of the blocked harm image (T / 16, K, 16)
    const unsigned lo = 4u * (unsigned)tp, ho = 4u * (unsigned)((tp >> 4) * K * 16 + (tp & 15));
    // "own" = the median this half's mask favours (harm for H, perc for P), chosen once: descriptor, bin stride and lane offset
    const __amdgpu_buffer_rsrc_t r_own = half ? rP : rH, r_oth = half ? rH : rP;
    const unsigned own_st = half ? 4u * (unsigned)T : 64u, oth_st = half ? 64u : 4u * (unsigned)T;
    const unsigned own_lo = half ? lo : ho, oth_lo = half ? ho : lo;
    const unsigned row_bytes = 4u * (unsigned)T, fv_half = (unsigned)half * (unsigned)rows * row_bytes;
    auto emit_first = [&]() {
        const f32x2 v = acc[0];
        mx = fmaxf(mx, fmaxf(v.x, v.y));
        if (active) {
            const unsigned soff = fv_half + (unsigned)mcur * row_bytes;  // wave-uniform row
            if constexpr (EVEN) {
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v.x), __float_as_uint(v.y)}, rF, lo, soff, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), rF, lo, soff, 0);
                if (d1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), rF, lo + 4u, soff, 0);
            }
        }
#pragma unroll
        for (int e = 0; e + 1 < NP; ++e) acc[e] = acc[e + 1];
        acc[NP - 1] = f32x2{0.f, 0.f};
        ++mcur;
    };
    constexpr int kBatch = 8;  // bins of loads in flight per lane
    for (int k0 = kbeg; k0 < kend; k0 += kBatch) {
        f32x2 sv[kBatch], ov[kBatch], tv[kBatch];
        float4 wq[kBatch];
        int ne[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const unsigned kk = (unsigned)min(k0 + u, K - 1);
            sv[u] = load_pair<EVEN>(rS, lo, kk * row_bytes, d1);
            ov[u] = load_pair<EVEN>(r_own, own_lo, kk * own_st, d1);
            tv[u] = load_pair<EVEN>(r_oth, oth_lo, kk * oth_st, d1);
            const int pi = min(k0 + u, kend - 1) - kbeg;  // wave-uniform: scalar loads
            wq[u] = *reinterpret_cast<const float4 *>(plan + (size_t)pi * 8);
            ne[u] = __float_as_int(plan[(size_t)pi * 8 + 4]);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (k0 + u >= kend) break;
            for (int i = 0; i < ne[u]; ++i) emit_first();
            const f32x2 own = ov[u], oth = tv[u];
            const f32x2 o2 = own * own;
            const f32x2 den = o2 + oth * oth;
            f32x2 X;
            constexpr float kDenMin = 7.8886091e-31f;  // 2^-100: below it (digital silence) the normalised form with its split_zeros rule
            if (__builtin_expect(__any(den.x < kDenMin || den.y < kDenMin), 0)) {
                auto one = [&](float s, float h, float p) {  // librosa.util.softmask as smh_feat.hip's hpss_masks_fast evaluates it
                    float Z = fmaxf(h, p);
                    const bool bad = Z < FLT_MIN;
                    Z = bad ? 1.0f : Z;
                    const float iz = __builtin_amdgcn_rcpf(Z);
                    const float a = h * iz, b = p * iz;
                    const float m = a * a, r = b * b;
                    const float id = __builtin_amdgcn_rcpf(m + r);
                    const float mo = bad ? 0.5f : (half ? r : m) * id;
                    return s * mo;
                };
                X = f32x2{one(sv[u].x, half ? oth.x : own.x, half ? own.x : oth.x), one(sv[u].y, half ? oth.y : own.y, half ? own.y : oth.y)};
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

// (80 VGPRs: three 8-wave workgroups per CU, like features_half_kernel -- the walk lives on loads in flight)
template <int NP>
__global__ void __launch_bounds__(512, 6)
rag_walk_kernel(FeatPlan fp, const float *__restrict__ S, const float *__restrict__ harmb, const float *__restrict__ perc, int K, int rows,
                float *__restrict__ fv, int *__restrict__ maxkeys, const Clip *__restrict__ clips, const Item *__restrict__ items,
                int n_items) {
    // item list entry n / 2, half n & 1: the two halves of a chunk read the same S / harm / perc and sit next to each other in
    // their XCD's dispatch order (the second one's loads come from that L2)
    unsigned n;
    if (!xcd_item(2 * n_items, n)) return;
    const Item item = items[n >> 1];
    const int half = n & 1;
    const Clip &c = clips[item.clip];
    const int T = c.T;
    const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float mx = 0.f;  // sums of non-negative terms
    if (T & 1)
        walk_pairs<NP, false>(fp, seg, half, lane, item.tile, S + c.spec_off, harmb + c.harm_off, perc + c.spec_off, K, T, rows,
                              fv + c.fv_off, mx);
    else
        walk_pairs<NP, true>(fp, seg, half, lane, item.tile, S + c.spec_off, harmb + c.harm_off, perc + c.spec_off, K, T, rows,
                             fv + c.fv_off, mx);
    int key = __float_as_int(mx);  // non-negative floats order like their bit patterns
    for (int off = 32; off > 0; off >>= 1) key = max(key, __shfl_xor(key, off));
    if (lane == 0) atomicMax(&maxkeys[2 * item.clip + half], key);
}

// top-dB floor in the power domain (features_clip_kernel has the derivation):
//   max(10 log10(max(amin, x^2)), dBmax - 80) = 10 log10(max(x^2, lim)),  lim = max(amin, max(amin, xmax^2) * 1e-8)
__device__ __forceinline__ float floor_of_max(int key) {
    const float xm = __int_as_float(key);
    return fmaxf(kAmin, fmaxf(kAmin, xm * xm) * 1e-8f);
}
__device__ __forceinline__ float final_value(float x, float lim, int log_db) {
    return log_db ? 3.0102999566398120f * __builtin_amdgcn_logf(fmaxf(x * x, lim)) : x;
}

// ---------------------------------------------------------------------------------------------------------------------------
// rag_stats_kernel: StandardScaler statistics of one featuregram row per wave (lib/preprocessing.py:211-214, 221-224: per row over
// the frames, population variance, constant rows left unscaled -- sklearn's _is_constant_feature / _handle_zeros_in_scale), in
// float64, on the final dB values, which are formed on the fly from the walk's sums.  Rows of all long clips of the call are numbered consecutively: clip list[i] owns rows i * R2 .. (i + 1) * R2 - 1.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rag_stats_kernel(const float *__restrict__ fv, const int *__restrict__ maxkeys, int log_db, int rows, const Clip *__restrict__ clips,
                 const int *__restrict__ list, int n_rows, float4 *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int rg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rg >= n_rows) return;
    const int R2 = 2 * rows;
    const int i = rg / R2, r = rg - i * R2;
    const int b = list[i];
    const Clip &c = clips[b];
    const int T = c.T;
    const float *x = fv + c.fv_off + (size_t)r * T;
    const float lim = floor_of_max(maxkeys[2 * b + (r >= rows ? 1 : 0)]);
    // one pass: sums of d = value - (the row's first value) and of d^2 in float64 (|d| <= 80 dB over at most 2^31 frames: the
    // shifted second moment loses nothing that a float32 patch could show)
    const double x0 = (double)final_value(x[0], lim, log_db);
    double s = 0.0, q = 0.0;
    for (int t = lane; t < T; t += 64) {
        const double d = (double)final_value(x[t], lim, log_db) - x0;
        s += d;
        q += d * d;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off), q += __shfl_xor(q, off);
    const double md = s / (double)T;
    const double mean = x0 + md;
    const double var = fmax(q / (double)T - md * md, 0.0);
    const double eps = 2.220446049250313e-16;
    const double nm = (double)T * mean * eps;
    const bool constant = var <= (double)T * eps * var + nm * nm;
    double scale = sqrt(var);
    if (constant || scale == 0.0) scale = 1.0;
    // the f64 mean as hi + lo floats, 1 / scale as a float: what the LDS-image kernels keep per row as well
    if (lane == 0) stats[rg] = make_float4((float)mean, (float)(mean - (double)(float)mean), (float)(1.0 / scale), 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------------
// rag_final_kernel: one workgroup per (clip, 64-frame chunk).  Phase 1, a wave per row, lanes = frames: the walk's sum -> dB with the
// array's top-dB floor (librosa.power_to_db(x**2), lib/preprocessing.py:420,422) -> written back: the FINAL featuregram; with patches
// asked for, the standardised value goes to an LDS tile [row][frame] (odd stride).  Phase 2, a wave per frame, lanes = rows: the
// frame's column leaves the tile as one contiguous (2*rows)-float row of every patch that holds the frame -- tools.extract_patches'
// grid (lib/cython_impl/tools.pyx:21-38: patch p = frames p*shift .. p*shift + W - 1 of the tiled featuregram, frame index modulo T
// for clips shorter than a patch, lib/preprocessing.py:139-142), transposed to time-major (Proposed_Work_Results.py:235-236).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
rag_final_kernel(float *__restrict__ fv, const int *__restrict__ maxkeys, int log_db, int rows, int W, int shift,
                 float *__restrict__ patches, const float4 *__restrict__ stats, const Clip *__restrict__ clips,
                 const Item *__restrict__ items, int n_items) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [R2][kFinalFrames + 1]
    unsigned n;
    if (!xcd_item(n_items, n)) return;
    const Item item = items[n];
    const Clip &c = clips[item.clip];
    const int T = c.T, R2 = 2 * rows;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int t0 = item.tile * kFinalFrames;
    const int nt = min(kFinalFrames, T - t0);
    constexpr int ld = kFinalFrames + 1;
    const bool want = patches != nullptr && c.nP > 0;
    const float limH = floor_of_max(maxkeys[2 * item.clip]), limP = floor_of_max(maxkeys[2 * item.clip + 1]);
    float *g = fv + c.fv_off + t0;
    constexpr int kB = 6;  // rows of loads in flight per wave
    for (int r0 = wave; r0 < R2; r0 += nw * kB) {
        float v[kB];
#pragma unroll
        for (int q = 0; q < kB; ++q) v[q] = g[(size_t)min(r0 + q * nw, R2 - 1) * T + min(lane, nt - 1)];
#pragma unroll
        for (int q = 0; q < kB; ++q) {
            const int r = r0 + q * nw;
            if (r < R2) {
                const float d = final_value(v[q], r < rows ? limH : limP, log_db);
                if (lane < nt) g[(size_t)r * T + lane] = d;
                if (want) {
                    const float4 st = stats[c.row0 + r];  // wave-uniform
                    // (x - mean) rounded to f32 as sklearn does (mean carried as hi + lo), then * 1/scale: std_patch_kernel's arithmetic
                    const float cv = (float)((double)d - ((double)st.x + (double)st.y));
                    tile[r * ld + lane] = cv * st.z;
                }
            }
        }
    }
    if (!want) return;
    __syncthreads();
    const int nP = c.nP, Tt = c.Ttiled;
    for (int tl = wave; tl < nt; tl += nw) {
        for (int v = t0 + tl; v < Tt; v += T) {  // the frame's positions in the tiled featuregram (one unless T < W)
            int p_hi = v / shift;
            if (p_hi > nP - 1) p_hi = nP - 1;
            int p_lo = v - W + 1 <= 0 ? 0 : (v - W + shift) / shift;  // ceil((v - W + 1) / shift)
            for (int p = p_lo; p <= p_hi; ++p) {
                const int j = v - p * shift;  // 0 <= j < W: patch starts are never clamped (their centres end W/2 before the last frame)
                float *o = patches + (((size_t)c.patch_off + p) * W + j) * R2;
                for (int f = lane; f < R2; f += 64) o[f] = tile[f * ld + tl];
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
struct HostClip {
    long long audio_off, fv_off, patch_off;
    int T, Ttiled, nP;
    int cls;  // 0: LDS image, even T; 1: LDS image, odd T; 2: streaming kernels
};

struct RagGeom {
    int K, rows, stft_frames, med_frames;
};

// device bytes one clip adds to a sub-batch (tables, statistics, S, perc, harm)
size_t clip_bytes(const RagGeom &g, const HostClip &c) {
    const size_t spec = align_up((size_t)g.K * c.T, 4) * sizeof(float);
    const size_t harm = (size_t)((c.T + 15) / 16) * 16 * g.K * sizeof(float);
    size_t items = (size_t)(c.T + g.stft_frames - 1) / g.stft_frames + (size_t)(c.T + g.med_frames - 1) / g.med_frames;
    size_t extra = sizeof(Clip) + 2 * sizeof(int) /* max keys */ + sizeof(int) /* list entry */;
    if (c.cls == 2) {
        items += (size_t)(c.T + kWalkFrames - 1) / kWalkFrames + (size_t)(c.T + kFinalFrames - 1) / kFinalFrames;
        extra += (size_t)2 * g.rows * sizeof(float4);
    }
    return 2 * spec + harm + items * sizeof(Item) + extra;
}
constexpr size_t kFixedBytes = 8 * 256;  // alignment slack between the regions of a sub-batch

bool rag_context_ok(const smh_ctx *ctx) {
    if (!ctx->feat_walk_ok || smh::lab_env("SMH_FEAT_TAPS") || getenv("SMH_FEAT_TWO_KERNELS") || getenv("SMH_RAGGED_PERFILE")) return false;
    if (ctx->feat_nseg[1] < 1 || ctx->feat_nseg[1] > 8) return false;
    const size_t final_lds = sizeof(float) * (size_t)2 * ctx->feat_rows * (kFinalFrames + 1);
    if (final_lds > 150 * 1024) return false;
    return smh_median::rag_tile_frames(ctx->K, ctx->cfg.l_harm, ctx->cfg.l_perc, nullptr) > 0;
}
// the block-split walkers fold once per side: window / 2 + 4 < axis length (smh_median.hip: fast_ok)
bool rag_clip_ok(const smh_ctx *ctx, int T) {
    // (the streaming kernels address a clip's rows with 32-bit offsets: K * T and 2 * rows * T below 2^29 elements -- a day of audio)
    if ((long long)(ctx->K + 16) * T >= (1ll << 29) || (long long)2 * ctx->feat_rows * T >= (1ll << 29)) return false;
    return ctx->cfg.l_harm / 2 + 4 < T && ctx->cfg.l_perc / 2 + 4 < ctx->K && T >= 2;
}

// one sub-batch: tables -> one upload -> one launch per stage
int run_sub_batch(const smh_ctx *ctx, const RagGeom &g, const float *d_audio, const HostClip *hc, int n, int W, int shift, float *d_fv,
                  float *d_patches, char *d_work, bool stft_aligned8, hipStream_t st) {
    std::vector<Clip> clips(n);
    std::vector<Item> it_stft, it_med, it_walk, it_final;
    std::vector<int> list[3];
    size_t spec = 0, harm = 0;
    int max_T[2] = {0, 0};
    for (int b = 0; b < n; ++b) {
        const HostClip &h = hc[b];
        Clip &c = clips[b];
        memset(&c, 0, sizeof(c));
        c.audio_off = h.audio_off, c.fv_off = h.fv_off, c.patch_off = h.patch_off;
        c.spec_off = (long long)spec, c.harm_off = (long long)harm;
        c.T = h.T, c.Ttiled = h.Ttiled, c.nP = d_patches ? h.nP : 0;
        c.row0 = h.cls == 2 ? (int)list[2].size() * 2 * g.rows : 0;
        spec += align_up((size_t)g.K * h.T, 4);
        harm += (size_t)((h.T + 15) / 16) * 16 * g.K;
        for (int t = 0, i = 0; t < h.T; t += g.stft_frames, ++i) it_stft.push_back({b, i});
        for (int t = 0, i = 0; t < h.T; t += g.med_frames, ++i) it_med.push_back({b, i});
        if (h.cls == 2) {
            for (int t = 0, i = 0; t < h.T; t += kWalkFrames, ++i) it_walk.push_back({b, i});
            for (int t = 0, i = 0; t < h.T; t += kFinalFrames, ++i) it_final.push_back({b, i});
        } else {
            max_T[h.cls] = std::max(max_T[h.cls], h.T);
        }
        list[h.cls].push_back(b);
    }
    // the tables as one blob: [clips][stft items][median items][walk items][final items][lists 0, 1, 2][max keys = 0]
    size_t off = 0;
    auto place = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 16);
        return o;
    };
    const size_t o_clips = place(clips.size() * sizeof(Clip));
    const size_t o_stft = place(it_stft.size() * sizeof(Item)), o_med = place(it_med.size() * sizeof(Item));
    const size_t o_walk = place(it_walk.size() * sizeof(Item)), o_final = place(it_final.size() * sizeof(Item));
    const size_t o_l0 = place(list[0].size() * sizeof(int)), o_l1 = place(list[1].size() * sizeof(int)),
                 o_l2 = place(list[2].size() * sizeof(int));
    const size_t o_keys = place((size_t)2 * n * sizeof(int));
    std::vector<char> blob(off, 0);
    auto put = [&](size_t o, const void *p, size_t bytes) {
        if (bytes) memcpy(blob.data() + o, p, bytes);
    };
    put(o_clips, clips.data(), clips.size() * sizeof(Clip));
    put(o_stft, it_stft.data(), it_stft.size() * sizeof(Item));
    put(o_med, it_med.data(), it_med.size() * sizeof(Item));
    put(o_walk, it_walk.data(), it_walk.size() * sizeof(Item));
    put(o_final, it_final.data(), it_final.size() * sizeof(Item));
    put(o_l0, list[0].data(), list[0].size() * sizeof(int));
    put(o_l1, list[1].data(), list[1].size() * sizeof(int));
    put(o_l2, list[2].data(), list[2].size() * sizeof(int));
    int rc = stage_upload(ctx, blob.data(), blob.size(), d_work, st);
    if (rc) return rc;
    // device regions behind the tables
    size_t w = align_up(blob.size(), 256);
    float4 *d_stats = reinterpret_cast<float4 *>(d_work + w);
    w = align_up(w + list[2].size() * (size_t)2 * g.rows * sizeof(float4), 256);
    float *d_S = reinterpret_cast<float *>(d_work + w);
    w = align_up(w + spec * sizeof(float), 256);
    float *d_perc = reinterpret_cast<float *>(d_work + w);
    w = align_up(w + spec * sizeof(float), 256);
    float *d_harm = reinterpret_cast<float *>(d_work + w);
    const Clip *d_clips = reinterpret_cast<const Clip *>(d_work + o_clips);
    auto items_at = [&](size_t o) { return reinterpret_cast<const Item *>(d_work + o); };
    auto list_at = [&](size_t o) { return reinterpret_cast<const int *>(d_work + o); };
    int *d_keys = reinterpret_cast<int *>(d_work + o_keys);

    rc = smh_stft::launch_rag(ctx, d_audio, d_S, d_clips, items_at(o_stft), (int)it_stft.size(), stft_aligned8, st);
    if (rc) return rc;
    rc = smh_median::launch_rag(d_S, d_harm, d_perc, g.K, ctx->cfg.l_harm, ctx->cfg.l_perc, d_clips, items_at(o_med), (int)it_med.size(), st);
    if (rc) return rc;
    // (The LDS-image clips' small grids were tried on a side stream beside the streaming kernels: the fork / join events cost the host
    // 0.4 ms per call and the call got slower, 0.74 -> 1.10 ms per 256 files; everything stays on the caller's stream.)
    rc = smh_feat::launch_features_rag(ctx, d_S, d_harm, d_perc, d_clips, list_at(o_l0), (int)list[0].size(), max_T[0], 1, W > 0 ? W : 1,
                                       shift > 0 ? shift : 1, d_fv, d_patches, st);
    if (rc) return rc;
    rc = smh_feat::launch_features_rag(ctx, d_S, d_harm, d_perc, d_clips, list_at(o_l1), (int)list[1].size(), max_T[1], 0, W > 0 ? W : 1,
                                       shift > 0 ? shift : 1, d_fv, d_patches, st);
    if (rc) return rc;
    if (!list[2].empty()) {
        const FeatPlan fp = smh_feat::feat_plan(ctx, 1);
        const unsigned gw = (unsigned)(8 * ((2 * (long long)it_walk.size() + 7) / 8));
        if (fp.pend <= 2)
            hipLaunchKernelGGL(rag_walk_kernel<2>, dim3(gw), dim3(64 * fp.nseg), 0, st, fp, d_S, d_harm, d_perc, g.K, g.rows, d_fv, d_keys,
                               d_clips, items_at(o_walk), (int)it_walk.size());
        else
            hipLaunchKernelGGL(rag_walk_kernel<4>, dim3(gw), dim3(64 * fp.nseg), 0, st, fp, d_S, d_harm, d_perc, g.K, g.rows, d_fv, d_keys,
                               d_clips, items_at(o_walk), (int)it_walk.size());
        rc = smh::launch_status("rag_walk_kernel");
        if (rc) return rc;
        if (d_patches) {
            const int n_rows = (int)list[2].size() * 2 * g.rows;
            hipLaunchKernelGGL(rag_stats_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, st, (const float *)d_fv, (const int *)d_keys,
                               ctx->cfg.log_db, g.rows, d_clips, list_at(o_l2), n_rows, d_stats);
            rc = smh::launch_status("rag_stats_kernel");
            if (rc) return rc;
        }
        const size_t lds = sizeof(float) * (size_t)2 * g.rows * (kFinalFrames + 1);
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)rag_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const unsigned gf = (unsigned)(8 * (((long long)it_final.size() + 7) / 8));
        hipLaunchKernelGGL(rag_final_kernel, dim3(gf), dim3(512), lds, st, d_fv, (const int *)d_keys, ctx->cfg.log_db, g.rows, W > 0 ? W : 1,
                           shift > 0 ? shift : 1, d_patches, (const float4 *)d_stats, d_clips, items_at(o_final), (int)it_final.size());
        rc = smh::launch_status("rag_final_kernel");
        if (rc) return rc;
    }
    return SMH_OK;
}

// all clips of `hc` through the ragged kernels, in as few sub-batches as the workspace allows
int run_rag(const smh_ctx *ctx, const float *d_audio, const std::vector<HostClip> &hc, int W, int shift, float *d_fv, float *d_patches,
            void *d_work, size_t work_bytes, bool stft_aligned8, hipStream_t st) {
    if (hc.empty()) return SMH_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return smh::set_error(SMH_E_INVALID, "the ragged front end uploads its tables from a staging buffer: it cannot be captured in a graph");
    RagGeom g;
    g.K = ctx->K, g.rows = ctx->feat_rows;
    g.stft_frames = smh_stft::rag_frames(ctx, stft_aligned8);
    g.med_frames = smh_median::rag_tile_frames(ctx->K, ctx->cfg.l_harm, ctx->cfg.l_perc, nullptr);
    SMH_REQUIRE(g.med_frames > 0, "ragged front end: no median kernel for this context");
    size_t b0 = 0;
    while (b0 < hc.size()) {
        size_t need = kFixedBytes, b1 = b0;
        while (b1 < hc.size()) {
            const size_t cb = clip_bytes(g, hc[b1]);
            if (b1 > b0 && need + cb > work_bytes) break;
            need += cb;
            ++b1;
        }
        if (need > work_bytes)
            return smh::set_error(SMH_E_WORKSPACE, "ragged front end: workspace %zu < %zu needed by a single clip", work_bytes, need);
        int rc = run_sub_batch(ctx, g, d_audio, hc.data() + b0, (int)(b1 - b0), W, shift, d_fv, d_patches, (char *)d_work, stft_aligned8, st);
        if (rc) return rc;
        b0 = b1;
    }
    return SMH_OK;
}

constexpr size_t kRagWorkCap = (size_t)8 << 30;  // what smh_frontend_ragged_sizes asks for at most (a larger batch runs in sub-batches)

}  // namespace

namespace smh_rag {

// smh_frontend_f32's route for B equal clips beyond the LDS image: the streaming kernels above (returns 1 if it ran, 0 if this
// context or shape has no ragged kernels, < 0 on error)
size_t equal_overhead_bytes(const smh_ctx *ctx, int B, int T) {
    RagGeom g;
    g.K = ctx->K, g.rows = ctx->feat_rows, g.stft_frames = 16, g.med_frames = 16;  // (upper bounds of the item counts)
    HostClip h;
    h.T = T, h.cls = 2;
    const size_t spec = align_up((size_t)g.K * T, 4) * sizeof(float);
    const size_t harm = (size_t)((T + 15) / 16) * 16 * g.K * sizeof(float);
    return (size_t)B * (clip_bytes(g, h) - 2 * spec - harm) + kFixedBytes + 4 * 256 + (size_t)B * 16;
}

int run_equal(const smh_ctx *ctx, const float *d_audio, int B, int n_samples, int T, int W, int shift, int nP, float *d_fv,
              float *d_patches, void *d_work, size_t work_bytes, bool stft_aligned8, hipStream_t st) {
    if (!rag_context_ok(ctx) || !rag_clip_ok(ctx, T) || (reinterpret_cast<uintptr_t>(d_work) % 16) != 0) return 0;
    const int rows2 = 2 * ctx->feat_rows;
    std::vector<HostClip> hc(B);
    for (int b = 0; b < B; ++b) {
        hc[b].audio_off = (long long)b * n_samples, hc[b].fv_off = (long long)b * rows2 * T, hc[b].patch_off = (long long)b * nP;
        hc[b].T = T, hc[b].Ttiled = smh_tiled_frames(T, W > 0 ? W : 1), hc[b].nP = nP, hc[b].cls = 2;
    }
    int rc = run_rag(ctx, d_audio, hc, W, shift, d_fv, d_patches, d_work, work_bytes, stft_aligned8, st);
    return rc ? rc : 1;
}

}  // namespace smh_rag

// ---- the C ABI -------------------------------------------------------------------------------------------------------------
namespace {
struct RaggedPlan {
    std::vector<int> T, nP;
    std::vector<long long> fv_off, patch_off;  // floats / patches in front of clip b
};
int plan_ragged(const smh_ctx *ctx, const long long *off, const int *len, int B, int W, int shift, bool patches, RaggedPlan &p) {
    p.T.assign(B, 0), p.nP.assign(B, 0), p.fv_off.assign(B + 1, 0), p.patch_off.assign(B + 1, 0);
    const int rows2 = 2 * ctx->feat_rows;
    for (int b = 0; b < B; ++b) {
        SMH_REQUIRE(off[b] >= 0 && len[b] >= 0, "ragged: clip %d has a negative offset or length", b);
        const int T = smh_num_frames(len[b], ctx->cfg.n_fft, ctx->cfg.hop);
        SMH_REQUIRE(T >= 1, "ragged: clip %d of %d samples is shorter than n_fft=%d", b, len[b], ctx->cfg.n_fft);
        p.T[b] = T;
        p.nP[b] = patches ? smh_num_patches(smh_tiled_frames(T, W), W, shift) : 0;
        p.fv_off[b + 1] = p.fv_off[b] + (long long)rows2 * T;
        p.patch_off[b + 1] = p.patch_off[b] + p.nP[b];
    }
    return SMH_OK;
}
// which clips the ragged kernels take (cls 0 / 1 / 2) and which go through smh_frontend_f32 alone (-1)
void classify(const smh_ctx *ctx, const float *d_audio, const long long *off, const RaggedPlan &p, int B, std::vector<int> &cls) {
    cls.assign(B, -1);
    if (!rag_context_ok(ctx)) return;
    // the specialised STFT needs every frame on an 8-byte boundary: a clip that starts elsewhere takes the generic kernel when it is
    // processed alone, so it is processed alone here too (same bits either way)
    const bool need8 = smh_stft::rag_frames(ctx, true) == smh_stft::kRagFrames;
    for (int b = 0; b < B; ++b) {
        if (!rag_clip_ok(ctx, p.T[b])) continue;
        if (need8 && ((reinterpret_cast<uintptr_t>(d_audio) + (uintptr_t)off[b] * 4) % 8) != 0) continue;
        // (even T: one workgroup per clip half; odd T -- or SMH_FEAT_NOPAIR, which makes smh_frontend_f32 choose the same -- one per clip)
        if (smh_features_blocked_ok(ctx, p.T[b], 0)) cls[b] = ((p.T[b] & 1) || getenv("SMH_FEAT_NOPAIR")) ? 1 : 0;
        else cls[b] = 2;
    }
}
}  // namespace

extern "C" int smh_frontend_ragged_sizes(const smh_ctx *ctx, const long long *h_offsets, const int *h_lengths, int B, int W,
                                         int shift, long long *h_fv_off, long long *h_patch_off, int *h_T, int *h_nP,
                                         size_t *work_bytes) {
    SMH_REQUIRE(ctx && (B == 0 || (h_offsets && h_lengths)) && B >= 0, "smh_frontend_ragged_sizes: bad argument");
    SMH_REQUIRE(W <= 0 || shift >= 1, "smh_frontend_ragged_sizes: bad patch geometry W=%d shift=%d", W, shift);
    RaggedPlan p;
    int rc = plan_ragged(ctx, h_offsets, h_lengths, B, W, shift, W > 0, p);
    if (rc) return rc;
    for (int b = 0; b <= B; ++b) {
        if (h_fv_off) h_fv_off[b] = p.fv_off[b];
        if (h_patch_off) h_patch_off[b] = p.patch_off[b];
    }
    for (int b = 0; b < B; ++b) {
        if (h_T) h_T[b] = p.T[b];
        if (h_nP) h_nP[b] = p.nP[b];
    }
    if (work_bytes) {
        // every clip the ragged kernels take, at once -- up to kRagWorkCap, beyond that the call runs in sub-batches -- and never less
        // than the largest single clip needs (alone in a sub-batch, or through smh_frontend_f32)
        std::vector<int> cls;
        classify(ctx, nullptr, h_offsets, p, B, cls);  // (alignment is the call's business: sized as if every clip qualified)
        RagGeom g;
        g.K = ctx->K, g.rows = ctx->feat_rows, g.stft_frames = 16;
        g.med_frames = std::max(16, smh_median::rag_tile_frames(ctx->K, ctx->cfg.l_harm, ctx->cfg.l_perc, nullptr));
        size_t total = kFixedBytes, single = 0;
        for (int b = 0; b < B; ++b) {
            single = std::max(single, smh_frontend_workspace_bytes(ctx, 1, h_lengths[b]));
            if (cls[b] < 0) continue;
            HostClip h;
            h.T = p.T[b], h.cls = cls[b];
            const size_t cb = clip_bytes(g, h);
            total += cb;
            single = std::max(single, cb + kFixedBytes);
        }
        *work_bytes = B == 0 ? 0 : align_up(std::max(single, std::min(total, kRagWorkCap)), 256);
    }
    return SMH_OK;
}

extern "C" int smh_frontend_ragged_f32(const smh_ctx *ctx, const float *d_audio, const long long *h_offsets,
                                       const int *h_lengths, int B, int W, int shift, float *d_fv, float *d_patches,
                                       void *d_work, size_t work_bytes, void *stream) {
    SMH_REQUIRE(ctx && d_audio && d_fv && d_work && h_offsets && h_lengths && B >= 0, "smh_frontend_ragged_f32: bad argument");
    const bool patches = d_patches != nullptr;
    SMH_REQUIRE(!patches || (W >= 1 && shift >= 1), "smh_frontend_ragged_f32: bad patch geometry W=%d shift=%d", W, shift);
    SMH_REQUIRE((reinterpret_cast<uintptr_t>(d_work) % 16) == 0, "smh_frontend_ragged_f32: the workspace must start on a 16-byte boundary");
    RaggedPlan p;
    int rc = plan_ragged(ctx, h_offsets, h_lengths, B, W, shift, patches, p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::vector<int> cls;
    classify(ctx, d_audio, h_offsets, p, B, cls);
    std::vector<HostClip> hc;
    hc.reserve(B);
    for (int b = 0; b < B; ++b) {
        if (cls[b] < 0) continue;
        HostClip h;
        h.audio_off = h_offsets[b], h.fv_off = p.fv_off[b], h.patch_off = p.patch_off[b];
        h.T = p.T[b], h.Ttiled = smh_tiled_frames(p.T[b], W > 0 ? W : 1), h.nP = p.nP[b], h.cls = cls[b];
        hc.push_back(h);
    }
    rc = run_rag(ctx, d_audio, hc, W, shift, d_fv, d_patches, d_work, work_bytes, true, st);
    if (rc) return rc;
    // the clips no ragged kernel covers, one by one on the same stream (the workspace is free again in stream order)
    const size_t prow = (size_t)(W > 0 ? W : 0) * 2 * ctx->feat_rows;
    for (int b = 0; b < B; ++b) {
        if (cls[b] >= 0) continue;
        float *pt = patches && p.nP[b] > 0 ? d_patches + (size_t)p.patch_off[b] * prow : nullptr;
        rc = smh_frontend_f32(ctx, d_audio + h_offsets[b], 1, h_lengths[b], pt ? W : 0, pt ? shift : 0, d_fv + p.fv_off[b], pt, d_work,
                              work_bytes, nullptr, nullptr, nullptr, stream);
        if (rc < 0) return rc;
    }
    return SMH_OK;
}
