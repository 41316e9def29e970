// B3_MTL backward pass of the residual blocks on the bf16 matrix pipe -- the trainer's dtype 1 (smh_trainer_set_dtype): every matrix
// product takes SPLIT operands (x = hi + lo, both bf16; hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16, f32 accumulators), so the
// products are f32-grade (~1e-5 relative) at a fifth of the exact-f32 pipe's time (three 16-cycle instructions for k = 32 against
// eight 32-cycle ones).  Same inputs and outputs as tcn_backward_mfma_kernel (smh_train.hip): the activations and the dilated-conv
// outputs the training forward saved (TrainIO::acts / upre), the SpatialDropout1D masks, d loss / d pre of the Dense-on-trunk
// outputs; the weight gradients are added to `grad` (float atomics, or the deterministic fixed-point accumulators).
// Mirrors the reference's model.fit backward of lib/proposed_architectures.py:127-154 (keras-tcn residual blocks); arithmetic
// restated in oracle/b3_mtl_train.py.
//
// Layouts.  "C layout": a lane (q = lane / 16, j = lane % 16) of a tile's wave holds, for frame 16 u + j, the channels 4 q + r and
// 16 + 4 q + r (the MFMA accumulator layout with channels as M) -- as eight values in the order k' = 8 q + e they are at once the
// B operand of a product over channels and the A operand of a product that TRANSPOSES them: D[frame][channel] = sum_k' v[frame][k']
// sel[k'][channel] with a 0 / 1 selection matrix is exact for bf16 values and leaves lane (q, j) with frames 16 u + 4 q + r of
// channel j ("T layout"): four consecutive frames, one ds_write_b64 into a frame-contiguous image.  The weight gradients are
// products over FRAMES (dW[c][co] = sum_t a[t][c] b[t][co]), whose operands must hold eight consecutive frames per lane: they read
// those images.
//
// One workgroup = G patches (2 at the reference's W = 68: 2 x 65 KB of images + one 21 KB operand slot), 8 waves per patch.  What a
// lane keeps in registers over the whole kernel, for its row of its 16-frame tile: g (d loss / d block output -- it starts as
// dtrunk_kernel's product) and x (the residual stream, walked backwards: x_b = x_b+1 - W2 . y_b - b2, so the forward saves the TCN
// output alone and neither side moves the 111 MB of block inputs).  Per block:
//   A  (tile waves) y = norm(relu(u)) from the saved conv output u, x_b (6 products), dyn = W2 . g (6 products), the relu /
//      channel-max backward in registers -> du; frame-contiguous images xT, yT, gT, duT (hi and lo) and du rows (hi, lo);
//      the next block's u is requested from HBM as soon as this one's is dead;
//   -- barrier --
//   B  (four waves that own no tile at W = 68) one weight-gradient item each -- dW2, dW1 tap 0 / 1 / 2 -- as 2 x 2 accumulator tiles
//      over the frames of all the workgroup's patches (K = three steps of 32 frames per patch at W = 68; eight operand reads per
//      twelve products).  A side tap reads xT at frame t + off: whole 8-frame chunks when the dilation is a multiple of 8 (a chunk
//      outside the patch is zero), otherwise five dwords and a funnel shift inside an 8-frame halo.  Bias gradients: two more
//      products per step with an all-ones A operand.  The item's 18 atomic instructions go out at the top of the NEXT block, under
//      the tile waves' phase A (the CU's texture path takes ~a word per cycle for them: 2 us per block);
//   C  (tile waves, beside B) g += sum_tap W1[tap] . du[t - off] (18 products);
//   -- barrier --
// The block's kernels come as split A operands from a packed copy (pack_bwd_kernel, once per step) by LDS-DMA into ONE slot: what
// phase A reads (both orientations of the 1x1 kernel, its bias) is replaced behind the first barrier, the dilated kernel (phase C)
// behind the second.  No barrier is a __syncthreads(): its fence is an s_waitcnt vmcnt(0), which on a wave with atomics or an HBM
// prefetch in flight is a wait for their round trips; the waves that request DMA pieces count their waits instead.
// Measured (510 patches, W = 68, one MI355X): 169 us + 8 us dtrunk_kernel + 3 us pack, against 280 us for the exact-f32 kernel.
#include <algorithm>
#include <cstdlib>

#include "smh_train_bwd.h"

using namespace smh_tcn;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// two f32 -> packed hi = bf16(x), lo = bf16(x - hi) (smh_tcn_bf16.hip: split2)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo) {
    const bf16x2s h = __builtin_convertvector(f32x2s{x0, x1}, bf16x2s);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2s hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    const bf16x2s l = __builtin_convertvector(f32x2s{x0, x1} - hf, bf16x2s);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split8(f32x4 a, f32x4 b, bf16x8 &hi, bf16x8 &lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2(a[0], a[1], h0, l0);
    split2(a[2], a[3], h1, l1);
    split2(b[0], b[1], h2, l2);
    split2(b[2], b[3], h3, l3);
    hi = __builtin_bit_cast(bf16x8, u32x4s{h0, h1, h2, h3});
    lo = __builtin_bit_cast(bf16x8, u32x4s{l0, l1, l2, l3});
}
__device__ __forceinline__ f32x4 product3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
    c = mfma_bf16(al, bh, c);  // small terms first
    c = mfma_bf16(ah, bl, c);
    return mfma_bf16(ah, bh, c);
}
// combine over the four lanes that hold the same frame (l, l ^ 16, l ^ 32, l ^ 48)
template <class F>
__device__ __forceinline__ float quad_reduce(float v, F f) {
    const unsigned u = __float_as_uint(v);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = f(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    const unsigned ua = __float_as_uint(a);
    const auto r16 = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
    return f(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
}

constexpr int kDuRow = 80;       // bytes per row of the du images: 32 bf16 + 16 bytes of padding (smh_tcn_bf16.hip: kRS)
// The block's operand slot (and its image in the packed buffer), in pieces of 1 KiB = 64 lanes x 16 bytes: hi units 0..9, the 1x1
// bias b2 as 32 raw floats, lo units 0..9.  Units: 0, 1 = W2 [mt] for dyn (canonical rows), 2..7 = W1 [tap][mt], 8, 9 = W2 [mt] in the
// forward's orientation (rows = output channel), for x_b = x_b+1 - W2 . y_b - b2.
constexpr int kUnitsPerBlk = 10;
constexpr int kBiasPiece = kUnitsPerBlk, kLoPiece = kUnitsPerBlk + 1, kSlotPieces = 2 * kUnitsPerBlk + 1;
constexpr int kSlotLo = kLoPiece * 1024;          // byte offset of the lo units
constexpr int kSlotBytes = kSlotPieces * 1024;

// LDS plan (host: bwd_geo)
struct BwdGeo {
    int G;        // patches per workgroup
    int units;    // 16-frame tiles per patch (<= 8: one per wave)
    int K32;      // k steps of 32 frames in the weight-gradient products
    int halo;     // zero frames in front of / behind the xT rows: 8 -- only dilations below 8 read across a chunk boundary
    int sxt, st;  // row strides in bytes of xT and of yT / gT / duT (16 x odd)
    int o_x, o_y, o_g, o_du_t, o_du;  // byte offsets of the image pairs inside a patch's region (hi, then lo at + the pair's half)
    int h_x, h_t, h_du;               // the halves: 32 * sxt, 32 * st, (T + 1) * kDuRow
    int per_patch;                    // bytes per patch
};

// the part of the plan that follows from K32 alone, as compile-time constants (as kernel arguments every image offset became a
// per-lane address register of its own, hoisted out of the block loop and spilled)
template <int K32>
struct GeoK {
    static constexpr int halo = 8, sxt = 2 * (2 * halo + 32 * K32) + 16, st = 2 * 32 * K32 + 16;
    static constexpr int h_x = 32 * sxt, h_t = 32 * st;
    static constexpr int o_x = 0, o_y = o_x + 2 * h_x, o_g = o_y + 2 * h_t, o_du_t = o_g + 2 * h_t, o_du = o_du_t + 2 * h_t;
    static_assert((sxt / 16) % 2 == 1 && (st / 16) % 2 == 1, "row strides: odd multiples of 16 bytes");
};

// wave -> weight-gradient item of phase B, nibble w of the constant: 0 = dW2, 1..3 = dW1 tap 0..2, 15 = none.  An item runs over ALL the
// workgroup's patches before it sends its 18 atomic instructions: the atomics on the 4 176 gradient words of a block, which every
// workgroup of the grid adds to within the same microsecond, are what the kernel waits for (timing probe: 100 us of 263 with one item
// per kind and patch).  Two patches per workgroup (16 waves): waves 5..8, which own no 16-frame tile at W = 68.
constexpr unsigned long long kItemOfWave2 = 0xFFFFFFF3120FFFFFull;  // waves 5..8: dW2, tap 1, tap 0, tap 2
constexpr unsigned long long kItemOfWave1 = 0xFFFFFFFF3120FFFFull;  // one patch per workgroup (8 waves): waves 4..7

// canonical f32 kernels -> the slot's image (above), one thread per 16-byte unit of the hi half: lane (i, kg) of an operand unit holds row
// 16 mt + i of its matrix and the eight columns k' = 8 kg + e' in the accumulator's channel order; the lo half is written beside it
__global__ void pack_bwd_kernel(const float *__restrict__ flat, Offsets off, int n_blocks, bf16x8 *__restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_blocks * (kUnitsPerBlk + 1) * 64) return;
    const int blk = idx / ((kUnitsPerBlk + 1) * 64), e = (idx >> 6) % (kUnitsPerBlk + 1), lane = idx & 63, i = lane & 15, kg = lane >> 4;
    const size_t wo = off.blk0 + (size_t)blk * off.blk_stride;
    const float *k1 = flat + wo, *k2 = flat + wo + 3 * C * C + C, *b2 = k2 + C * C;
    bf16x8 *d = dst + (size_t)blk * kSlotPieces * 64;
    if (e == kBiasPiece) {  // 32 raw floats in the first 8 units of the piece
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (lane < 8) v = *reinterpret_cast<const f32x4 *>(b2 + 4 * lane);
        *reinterpret_cast<f32x4 *>(d + kBiasPiece * 64 + lane) = v;
        return;
    }
    float vf[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ch = k < 4 ? 4 * kg + k : 16 + 4 * kg + (k - 4);  // the accumulator's channel order k'
        if (e < 2) vf[k] = k2[(size_t)(16 * e + i) * C + ch];                                        // dyn[c] = sum_co k2[c][co] g[co]
        else if (e < 8) vf[k] = k1[((size_t)((e - 2) >> 1) * C + 16 * ((e - 2) & 1) + i) * C + ch];  // dx[c] += sum_co k1[tap][c][co] du[co]
        else vf[k] = k2[(size_t)ch * C + 16 * (e - 8) + i];                                          // o[co] = sum_c k2[c][co] y[c]
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) hi[k] = (__bf16)vf[k], lo[k] = (__bf16)(vf[k] - (float)hi[k]);
    d[e * 64 + lane] = hi;
    d[kLoPiece * 64 + e * 64 + lane] = lo;
}

// d loss / d (TCN output before its final relu), the gradient the block loop starts from:
//   gt[n][k] = relu'(x_last[n][k]) * sum_o dpre[n][o] Wh[k][o],   k = frame * 32 + channel, o over [3C | Dense(16) of every head]
// as its own small product (exact-f32 matrix instructions, M = 16 rows of k per workgroup, N = 16 patches per tile, K = the 51 / 69
// outputs): inside the backward kernel every workgroup read all of Wh for its two patches -- 444 KB from L2, 35 us of prologue.
// The order of the k index is free as long as both operands use the same one: lane group q takes outputs 4 q .. 4 q + 3 of a head's
// sixteen in four consecutive steps, so that its A operands (a row of the head's Dense kernel) and its B operands (a row of dpre)
// are one 16-byte load per head instead of four scalar ones; the 3 / 5 class outputs take one / two steps of scalar loads.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(256)
dtrunk_kernel(BwdArgs a, const float *__restrict__ flatw, const float *__restrict__ acts, const float *__restrict__ dpre,
              float *__restrict__ gt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const int k = 16 * blockIdx.x + j, nslot = a.n_blocks + 1, D = a.T * C;
    f32x4 ah[kMaxHeads];
    float ac[2];
#pragma unroll
    for (int h = 0; h < kMaxHeads; ++h)
        ah[h] = h < a.n_heads ? f32x4(*reinterpret_cast<const f32x4u *>(flatw + a.off.head[h] + (size_t)k * kHidden + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) ac[c] = 4 * c + q < a.n_classes ? flatw[a.off.c3_k + (size_t)k * a.n_classes + 4 * c + q] : 0.f;
    const int tiles = (a.N + 15) / 16;
    for (int pt = wave + 4 * blockIdx.y; pt < tiles; pt += 4 * gridDim.y) {
        const int n = 16 * pt + j;
        const bool valid = n < a.N;
        const float *dp = dpre + (size_t)(valid ? n : 0) * kPS;
        f32x4 bh[kMaxHeads];
        float bc[2];
#pragma unroll
        for (int h = 0; h < kMaxHeads; ++h)
            bh[h] = (valid && h < a.n_heads) ? f32x4(*reinterpret_cast<const f32x4u *>(dp + a.n_classes + h * kHidden + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) bc[c] = (valid && 4 * c + q < a.n_classes) ? dp[4 * c + q] : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) acc = mfma4(ac[c], bc[c], acc);
#pragma unroll
        for (int h = 0; h < kMaxHeads; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma4(ah[h][e], bh[h][e], acc);
        if (valid) {  // D[row 4 q + r of the k tile][patch j]
            const int kk = 16 * blockIdx.x + 4 * q;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(acts + ((size_t)n * nslot + a.n_blocks) * D + kk);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = x[r] > 0.f ? acc[r] : 0.f;
            *reinterpret_cast<f32x4 *>(gt + (size_t)n * D + kk) = acc;
        }
    }
}

// STAMPS: tools only (SMH_BWD_STAMPS) -- the phase counters cost 20 registers, so they are an instantiation of their own
template <int K32, int G, bool STAMPS>
__global__ void __launch_bounds__(512 * G)
tcn_backward_bf16_kernel(BwdArgs a, BwdGeo geo, const float *__restrict__ X, const float *__restrict__ flatw,
                         const bf16x8 *__restrict__ pk, const float *__restrict__ acts, const float *__restrict__ drop,
                         const float *__restrict__ gt, float *__restrict__ grad, const float *__restrict__ upre) {
    extern __shared__ __attribute__((aligned(16))) char smb[];
    using GH = GeoK<K32>;
    const int T = a.T, nslot = a.n_blocks + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int n0 = blockIdx.x * G;
    const int g_here = min(G, a.N - n0);
    // this wave's patch and 16-frame tile (patch 1's tiles sit one wave further on, so that the 5 + 5 tiles of two 68-frame patches
    // spread 3 / 3 / 2 / 2 over the four SIMDs)
    const int p = wave >> 3;
    const int u = ((wave & 7) - p) & 7;
    const bool has_tile = u < geo.units && p < g_here;
    const int t = 16 * u + j;
    const bool live = has_tile && t < T;
    char *img = smb + (size_t)p * geo.per_patch;
    char *wslot = smb + (size_t)G * geo.per_patch;

    // tools only (SMH_BWD_STAMPS): waves 0 (owns a tile) and 5 (owns none at W = 68) of workgroup 0 print their phases, summed over the blocks
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool stamping = STAMPS && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 5);
    if (stamping) tlast = __builtin_amdgcn_s_memrealtime();
    auto lap = [&](int i) {
        if constexpr (!STAMPS) return;
        if (stamping) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            tph[i] += now - tlast;
            tlast = now;
        }
    };
    for (int i = tid; i < G * geo.per_patch / 16; i += nt) reinterpret_cast<u32x4s *>(smb)[i] = u32x4s{0u, 0u, 0u, 0u};
    // the block's operands, piece by piece (kSlotPieces of 1 KiB = 64 lanes x 16 bytes, layout above) from the packed copy into the slot
    auto stage_piece = [&](int blk, int piece) {
        const char *src = reinterpret_cast<const char *>(pk + ((size_t)blk * kSlotPieces + piece) * 64 + lane);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(wslot + piece * 1024), 16, 0, 0);
    };
    // The waves that carry a weight-gradient item (phase B) request nothing: a wave has to wait for its pieces with vmcnt(0), and for
    // them that would be a wait for the round trips of their atomics -- on 4 176 words that every workgroup of the grid adds to
    // within the same microsecond (timing probe: 100 of the kernel's 260 us when every barrier waited for them).
    constexpr int kDmaWaves = G == 2 ? 12 : 4;
    const int didx = G == 2 ? (wave < 5 ? wave : (wave >= 9 ? wave - 4 : -1)) : (wave < 4 ? wave : -1);
    // what phase A reads: both orientations of the 1x1 kernel and its bias (nine pieces, waves 0..3) ...
    auto stage_w2 = [&](int blk) {
        if (wave < 4)
            for (int i = wave; i < 9; i += 4) {
                const int unit = i < 5 ? (i < 2 ? i : 6 + i) : (i < 7 ? i - 5 : i + 1);  // hi 0, 1, 8, 9, bias (= 10); lo 0, 1, 8, 9
                stage_piece(blk, i < 5 ? unit : kLoPiece + unit);
            }
    };
    // ... and what phase C reads: the dilated kernel (twelve pieces)
    auto stage_w1 = [&](int blk) {
        if (didx >= 0)
            for (int i = didx; i < 12; i += kDmaWaves) stage_piece(blk, i < 6 ? 2 + i : kLoPiece + 2 + (i - 6));
    };
    if (a.n_blocks > 0) stage_w2(a.n_blocks - 1), stage_w1(a.n_blocks - 1);

    // ---- d loss / d (TCN output), from dtrunk_kernel, in C layout: straight into the registers it lives in over the whole kernel ----
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float *gp = gt + ((size_t)(n0 + p) * T + t) * C + 4 * q;
        g0 = *reinterpret_cast<const f32x4 *>(gp), g1 = *reinterpret_cast<const f32x4 *>(gp + 16);
    }

    // x: the residual stream at this lane's row, walked backwards: it starts as the TCN output (slot n_blocks of `acts`, the only slot
    // the forward saves for this kernel) and becomes every block's input in turn, x_b = x_b+1 - (W2 . y_b + b2) -- y_b is rebuilt from the
    // saved conv output anyway, the six products cost less than reading (and the forward writing) 111 MB of saved block inputs, and
    // what the subtraction loses (an ulp of x per block) is far below the split products' own 1e-5
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float *xp = acts + (((size_t)(n0 + p) * nslot + a.n_blocks) * T + t) * C + 4 * q;
        x0 = *reinterpret_cast<const f32x4 *>(xp), x1 = *reinterpret_cast<const f32x4 *>(xp + 16);
    }
    // register prefetch of a block's inputs for this lane's row: saved dilated-conv output, dropout mask
    f32x4 pf_u0, pf_u1, pf_d0, pf_d1;
    auto prefetch = [&](int blk) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
        pf_u0 = pf_u1 = z;
        pf_d0 = pf_d1 = one;
        if (live) {
            const float *up = upre + (((size_t)(n0 + p) * a.n_blocks + blk) * T + t) * C + 4 * q;
            pf_u0 = *reinterpret_cast<const f32x4 *>(up), pf_u1 = *reinterpret_cast<const f32x4 *>(up + 16);
            if (drop) {
                const float *dq = drop + ((size_t)(n0 + p) * a.n_blocks + blk) * C + 4 * q;
                pf_d0 = *reinterpret_cast<const f32x4 *>(dq), pf_d1 = *reinterpret_cast<const f32x4 *>(dq + 16);
            }
        }
    };
    if (a.n_blocks > 0) prefetch(a.n_blocks - 1);

    // frame-contiguous images of eight values in C layout: two exact transposing products per half (hi, lo), packed back to bf16
    // (the upper halves of the f32 results), one 8-byte write each; e0 = element index of the tile's first frame in the row
    // the selection operands: channel 16 h + n of the k' order sits at k' = 8 (n / 4) + 4 h + n % 4 (built where they are used: eight
    // registers that would otherwise stay live across the weight-gradient phase)
    auto make_sel = [&](bf16x8 (&sel)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) sel[h][e] = (__bf16)((q == (j >> 2) && e == 4 * h + (j & 3)) ? 1.0f : 0.0f);
    };
    auto transpose_store = [&](const bf16x8 (&sel)[2], bf16x8 vh, bf16x8 vl, char *ih, char *il, int stride, int e0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 th = mfma_bf16(vh, sel[h], z), tl = mfma_bf16(vl, sel[h], z);
            u32x2s wh, wl;
            wh[0] = __builtin_amdgcn_perm(__float_as_uint(th[1]), __float_as_uint(th[0]), 0x07060302u);
            wh[1] = __builtin_amdgcn_perm(__float_as_uint(th[3]), __float_as_uint(th[2]), 0x07060302u);
            wl[0] = __builtin_amdgcn_perm(__float_as_uint(tl[1]), __float_as_uint(tl[0]), 0x07060302u);
            wl[1] = __builtin_amdgcn_perm(__float_as_uint(tl[3]), __float_as_uint(tl[2]), 0x07060302u);
            const int o = (16 * h + j) * stride + 2 * (e0 + 4 * q);
            *reinterpret_cast<u32x2s *>(__builtin_assume_aligned(ih + o, 8)) = wh;
            *reinterpret_cast<u32x2s *>(__builtin_assume_aligned(il + o, 8)) = wl;
        }
    };
    auto ld16 = [&](const char *ptr) { return *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(ptr, 16)); };

    // this wave's weight-gradient item (phase B), if any, and the accumulators it hands from one block to the next: an item's 18 atomic
    // instructions are SENT at the top of the next block, while the tile waves run phase A -- the CU's texture path takes about a
    // lane's word per cycle for them (4 176 words = 2 us per block), and sent inside phase B that time stood between two barriers
    const int my_kind = wave < 16 ? (int)(((G == 2 ? kItemOfWave2 : kItemOfWave1) >> (4 * wave)) & 15ull) : 15;
    f32x4 acc[2][2], accb[2];
    int pend_blk = -1;  // the block whose gradients acc / accb hold
    auto flush = [&]() {
        if (pend_blk < 0 || (a.split3 & 1)) return;  // (the mask: timing probe)
        const size_t wo = a.off.blk0 + (size_t)pend_blk * a.off.blk_stride;
        const bool w2 = my_kind == 0;
        const int tap = w2 ? 1 : my_kind - 1;
        const unsigned gbase = (unsigned)(w2 ? wo + 3 * C * C + C : wo + (size_t)tap * C * C);
        const unsigned gbias = (unsigned)(w2 ? wo + 3 * C * C + C + C * C : wo + 3 * C * C);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    gadd(grad, a.gq, gbase + (unsigned)((16 * mt + 4 * q + r) * C + 16 * n2 + j), acc[mt][n2][r]);
        if (tap == 1 && q == 0) {
            gadd(grad, a.gq, gbias + j, accb[0][0]);
            gadd(grad, a.gq, gbias + 16 + j, accb[1][0]);
        }
        pend_blk = -1;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // images zeroed, the last block's operands in the slot
    lap(0);

    for (int blk = a.n_blocks - 1; blk >= 0; --blk) {
        const int d = 1 << (blk % a.n_dil);
        if (blk < a.n_blocks - 1) stage_w1(blk);  // (phase C of the block before is behind the barrier)
        flush();  // the previous block's weight gradients (the waves that own an item own no tile at W = 68: phase A is their idle time)
        // ---- phase A ---------------------------------------------------------------------------------------------------------------
        if (has_tile && !(a.split3 & 4)) {  // (wave-uniform; the mask: timing probes)
            const f32x4 u0 = pf_u0, u1 = pf_u1, dm0 = pf_d0, dm1 = pf_d1;
            bf16x8 sel[2];
            make_sel(sel);
            bf16x8 vh, vl;
            float r0[4], r1[4], mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                r0[r] = fmaxf(u0[r], 0.f), r1[r] = fmaxf(u1[r], 0.f);
                mx = fmaxf(mx, fmaxf(r0[r], r1[r]));
            }
            mx = quad_reduce(mx, [](float x, float y) { return fmaxf(x, y); });
            const float inv_m = __builtin_amdgcn_rcpf(mx + kNormEps);  // the forward's own 1 / (max + eps)
            f32x4 y0, y1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y0[r] = live ? r0[r] * inv_m * dm0[r] : 0.f;
                y1[r] = live ? r1[r] * inv_m * dm1[r] : 0.f;
            }
            split8(y0, y1, vh, vl);
            {   // this block's input from its output: the forward's own 1x1 products on the same y
                const float *b2 = reinterpret_cast<const float *>(wslot + kBiasPiece * 1024);
                f32x4 o0 = *reinterpret_cast<const f32x4 *>(b2 + 4 * q), o1 = *reinterpret_cast<const f32x4 *>(b2 + 16 + 4 * q);
                o0 = product3(ld16(wslot + (8 * 64 + lane) * 16), ld16(wslot + kSlotLo + (8 * 64 + lane) * 16), vh, vl, o0);
                o1 = product3(ld16(wslot + (9 * 64 + lane) * 16), ld16(wslot + kSlotLo + (9 * 64 + lane) * 16), vh, vl, o1);
#pragma unroll
                for (int r = 0; r < 4; ++r) x0[r] = live ? x0[r] - o0[r] : 0.f, x1[r] = live ? x1[r] - o1[r] : 0.f;
            }
            transpose_store(sel, vh, vl, img + GH::o_y, img + GH::o_y + GH::h_t, GH::st, 16 * u);
            split8(x0, x1, vh, vl);
            transpose_store(sel, vh, vl, img + GH::o_x, img + GH::o_x + GH::h_x, GH::sxt, GH::halo + 16 * u);
            bf16x8 gh, gl;
            split8(g0, g1, gh, gl);
            transpose_store(sel, gh, gl, img + GH::o_g, img + GH::o_g + GH::h_t, GH::st, 16 * u);
            // dyn[c][frame] = sum_co W2[c][co] g[frame][co]
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
            d0 = product3(ld16(wslot + (0 * 64 + lane) * 16), ld16(wslot + kSlotLo + (0 * 64 + lane) * 16), gh, gl, d0);
            d1 = product3(ld16(wslot + (1 * 64 + lane) * 16), ld16(wslot + kSlotLo + (1 * 64 + lane) * 16), gh, gl, d1);
            float s1 = 0.f, cnt = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d0[r] *= dm0[r], d1[r] *= dm1[r];
                s1 = fmaf(d0[r], r0[r], s1);
                s1 = fmaf(d1[r], r1[r], s1);
                cnt += (r0[r] == mx ? 1.f : 0.f) + (r1[r] == mx ? 1.f : 0.f);
            }
            s1 = quad_reduce(s1, [](float x, float y) { return x + y; });
            cnt = quad_reduce(cnt, [](float x, float y) { return x + y; });
            const float corr = s1 * inv_m * inv_m / cnt;
            f32x4 du0, du1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a0 = d0[r] * inv_m, a1 = d1[r] * inv_m;
                if (r0[r] == mx && r0[r] > 0.f) a0 -= corr;
                if (r1[r] == mx && r1[r] > 0.f) a1 -= corr;
                du0[r] = (live && u0[r] > 0.f) ? a0 : 0.f;
                du1[r] = (live && u1[r] > 0.f) ? a1 : 0.f;
            }
            split8(du0, du1, vh, vl);
            if (live) {
                *reinterpret_cast<bf16x8 *>(__builtin_assume_aligned(img + GH::o_du + t * kDuRow + 16 * q, 16)) = vh;
                *reinterpret_cast<bf16x8 *>(__builtin_assume_aligned(img + GH::o_du + geo.h_du + t * kDuRow + 16 * q, 16)) = vl;
            }
            transpose_store(sel, vh, vl, img + GH::o_du_t, img + GH::o_du_t + GH::h_t, GH::st, 16 * u);
        }
        // the next block's saved conv outputs and masks, requested as soon as this block's are dead: they come from HBM, and requested
        // behind the barrier they had only phase C to arrive in (the tile waves then stood at the top of the next phase A)
        asm volatile("" ::: "memory");  // (the counted wait below relies on these loads being issued BEHIND the DMA requests of this block)
        if (blk > 0) prefetch(blk - 1);
        lap(1);
        // this wave's pieces of the dilated kernel have landed: everything but the two / four loads of the prefetch just issued (loads
        // return in order; a wave without a tile issued none)
        if (didx >= 0) {
            if (has_tile && blk > 0) {
                if (drop) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (not __syncthreads(): see the barrier at the end of the block)
        lap(2);
        if (blk > 0) stage_w2(blk - 1);  // phase A was the 1x1 kernel's last reader
        // ---- phase B: weight gradients ---------------------------------------------------------------------------------------------
        // item = dW2 | dW1 tap 0, 1, 2 over all the workgroup's patches: the whole 32 x 32 gradient as 2 x 2 accumulator tiles, so that a k step's eight
        // operand reads feed twelve products (as sixteen 16 x 16 jobs of four reads per three products the phase sat on LDS bandwidth:
        // 384 KB per block and CU)
        do {
            const int kind = my_kind;
            if (kind == 15 || (a.split3 & 2)) break;
            const bool w2 = kind == 0;
            const int tap = w2 ? 1 : kind - 1;
            const int off = w2 ? 0 : (tap - 1) * d;
            if (off != 0 && d >= T) break;  // a side tap that only ever sees the zero padding
            const bool bias = tap == 1;        // dW2: db2 = column sums of g; centre tap: db1 = column sums of du
            const int oA = w2 ? GH::o_y : GH::o_x, hA = w2 ? GH::h_t : GH::h_x, sA = w2 ? GH::st : GH::sxt;
            const int oB = w2 ? GH::o_g : GH::o_du_t;
            const int eA = (w2 ? 0 : GH::halo) + 8 * q + off;  // element index of this lane's first frame in its A row
            const int arow = oA + j * sA;                       // (its first 16 bytes: the zero halo chunk of an xT row)
            const int aoff = arow + 2 * (eA & ~1), boff = oB + j * GH::st + 16 * q;
            const unsigned sh = (eA & 1) ? 16u : 0u;
            const bool aligned = (off & 7) == 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][0] = acc[i][1] = accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
            auto run = [&](auto aligned_c) {
                constexpr bool AL = decltype(aligned_c)::value;
                auto lda = [&](const char *ptr) {
                    if constexpr (AL) {
                        return ld16(ptr);
                    } else {  // five dwords from a 4-byte aligned address, shifted down by 0 or 16 bits
                        const unsigned *w = reinterpret_cast<const unsigned *>(__builtin_assume_aligned(ptr, 4));
                        const unsigned w0 = w[0], w1 = w[1], w2_ = w[2], w3 = w[3], w4 = w[4];
                        return __builtin_bit_cast(bf16x8, u32x4s{__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2_, w1, sh),
                                                                 __builtin_amdgcn_alignbit(w3, w2_, sh), __builtin_amdgcn_alignbit(w4, w3, sh)});
                    }
                };
                // (one step's 32 operand registers at a time: unrolled, the steps' loads were hoisted together and spilled)
#pragma unroll 1
                for (int ps = 0; ps < g_here * K32; ++ps) {
                    const int pp = ps >= K32 ? 1 : 0, s = ps - pp * K32;
                    const char *ip = smb + (size_t)pp * geo.per_patch;
                    const char *pa = ip + aoff + 64 * s;
                    bool ok = true;
                    if constexpr (AL) {
                        // an aligned tap moves whole 8-frame chunks: one outside the patch's 32 K32 frames is zero (the halo chunk at the
                        // start of the row), a step with all four chunks outside is skipped (d = 64 at W = 68: two of three)
                        if (32 * s + off + 24 < 0 || 32 * s + off >= 32 * K32) continue;  // (uniform)
                        ok = (unsigned)(32 * s + 8 * q + off) < (unsigned)(32 * K32);
                    }
                    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const char *pm = ok ? pa + 16 * mt * sA : ip + arow + 16 * mt * sA;
                        ah[mt] = lda(pm), al[mt] = lda(pm + hA);
                    }
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) {
                        const char *pb = ip + boff + 16 * n2 * GH::st + 64 * s;
                        bh[n2] = ld16(pb), bl[n2] = ld16(pb + GH::h_t);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) acc[mt][n2] = product3(ah[mt], al[mt], bh[n2], bl[n2], acc[mt][n2]);
                    if (bias) {  // (uniform) column sums of B
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) {
                            accb[n2] = mfma_bf16(ones, bl[n2], accb[n2]);
                            accb[n2] = mfma_bf16(ones, bh[n2], accb[n2]);
                        }
                    }
                }
            };
            if (aligned) run(std::true_type{});
            else run(std::false_type{});
            pend_blk = blk;
        } while (false);
        lap(3);
        // ---- phase C: g[frame][c] += sum_tap sum_co W1[tap][c][co] du[frame - off][co] ------------------------------------------
        if (has_tile && !(a.split3 & 8)) {
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int off = (tap - 1) * d;
                const bool ok = (t - off >= 0) && (t - off < T);
                if (tap != 1 && !__any(ok)) continue;
                const int row = ok ? t - off : T;  // row T of the du images stays zero
                const bf16x8 bh = ld16(img + GH::o_du + row * kDuRow + 16 * q), bl = ld16(img + GH::o_du + geo.h_du + row * kDuRow + 16 * q);
                const int e0 = 2 + 2 * tap;
                g0 = product3(ld16(wslot + (e0 * 64 + lane) * 16), ld16(wslot + kSlotLo + (e0 * 64 + lane) * 16), bh, bl, g0);
                g1 = product3(ld16(wslot + ((e0 + 1) * 64 + lane) * 16), ld16(wslot + kSlotLo + ((e0 + 1) * 64 + lane) * 16), bh, bl, g1);
            }
            if (!live) g0 = g1 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // the pieces of the next block's 1x1 kernel requested at the top of phase B have landed (these waves send no atomics at two
        // patches per workgroup; at one, theirs went out a phase ago)
        if (wave < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(4);
        // (not __syncthreads(): its workgroup-scope fence is an s_waitcnt vmcnt(0), i.e. a wait for the round trips of the weight-gradient
        // atomics this wave has just sent; the barrier orders LDS only -- the images and the operand slot)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        lap(5);
    }

    flush();
    // ---- initial Conv1D(32, 1): dW0[f][c] = sum_t x[t][f] g[t][c], db0[c] = sum_t g[t][c] -- exact-f32 products (k = 4 steps over the
    // frames, X read from global memory as in tcn_backward_mfma_kernel); g goes to LDS as f32 rows of stride SX over the patch's images ----
    const int RPm = 16 * geo.units;
    if (has_tile) {
        float *Gs = reinterpret_cast<float *>(img) + (size_t)t * SX;
        *reinterpret_cast<f32x4 *>(Gs + 4 * q) = g0;
        *reinterpret_cast<f32x4 *>(Gs + 16 + 4 * q) = g1;
    }
    __syncthreads();
    const int fmt = (a.F + 15) >> 4;
    for (int job = wave; job < fmt * 2 + 2; job += nw) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        const bool bias = job >= fmt * 2;
        const int mt = bias ? 0 : job >> 1, nt_ = bias ? job - fmt * 2 : job & 1;
        const int f = 16 * mt + j;
        for (int pp = 0; pp < g_here; ++pp) {
            const float *Gs = reinterpret_cast<const float *>(smb + (size_t)pp * geo.per_patch);
            const float *xcol = X + (size_t)(n0 + pp) * T * a.F + (f < a.F ? f : 0);
            auto fetch = [&](int s, float (&av)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kr = 4 * s + e + 4 * q;
                    av[e] = bias ? 1.0f : ((kr < T && f < a.F) ? xcol[(size_t)kr * a.F] : 0.f);
                }
            };
            float av[4];
            fetch(0, av);
            for (int s = 0; s < RPm / 4; s += 4) {
                float an[4];
                fetch(min(s + 4, RPm / 4 - 4), an);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    acc = mfma4(av[e], Gs[(size_t)(4 * s + e + 4 * q) * SX + 16 * nt_ + j], acc);
                    acc2 = mfma4(av[e + 1], Gs[(size_t)(4 * s + e + 1 + 4 * q) * SX + 16 * nt_ + j], acc2);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = an[e];
            }
        }
        acc += acc2;
        if (bias) {
            if (q == 0) gadd(grad, a.gq, a.off.w0_b + 16 * nt_ + j, acc[0]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ff = 16 * mt + 4 * q + r;
                if (ff < a.F) gadd(grad, a.gq, a.off.w0_k + (size_t)ff * C + 16 * nt_ + j, acc[r]);
            }
        }
    }
    lap(6);
    if (stamping)
        printf("tcn_backward_bf16_kernel wg0 wave %d (x10 ns, summed over %d blocks): prologue %llu  A %llu  barrier1 %llu  B %llu  C %llu  barrier2 %llu  layer0 %llu\n",
               wave, a.n_blocks, tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[6]);
}

constexpr size_t kLdsMax = 160 * 1024;

bool bwd_geo(int T, int n_dil, BwdGeo *g) {
    if (T < 1 || T > 128) return false;
    g->units = (T + 15) / 16;
    g->K32 = (16 * g->units + 31) / 32;
    (void)n_dil;
    g->halo = 8;
    g->sxt = 2 * (2 * g->halo + 32 * g->K32) + 16;  // 16 x (3 + 4 K32): an odd multiple of 16 bytes
    g->st = 2 * 32 * g->K32 + 16;
    g->h_x = 32 * g->sxt, g->h_t = 32 * g->st, g->h_du = (T + 1) * kDuRow;
    g->o_x = 0;
    g->o_y = g->o_x + 2 * g->h_x;
    g->o_g = g->o_y + 2 * g->h_t;
    g->o_du_t = g->o_g + 2 * g->h_t;
    g->o_du = g->o_du_t + 2 * g->h_t;
    g->per_patch = g->o_du + 2 * g->h_du;
    g->per_patch = (g->per_patch + 15) / 16 * 16;
    if ((size_t)16 * g->units * SX * sizeof(float) > (size_t)g->per_patch) return false;  // the tail's f32 rows live on the patch's images
    g->G = (size_t)2 * g->per_patch + kSlotBytes <= kLdsMax ? 2 : 1;
    return (size_t)g->G * g->per_patch + kSlotBytes <= kLdsMax;
}

}  // namespace

namespace smh_tcn {

bool backward_bf16_supported(int T, int n_dil) {
    BwdGeo geo;
    return bwd_geo(T, n_dil, &geo) && ((geo.G == 2 && geo.K32 <= 3) || (geo.G == 1 && geo.K32 == 4));
}

int launch_dtrunk(const BwdArgs &ba, const float *d_flat, const float *d_acts, const float *d_dpre, float *d_gt, hipStream_t st) {
    SMH_REQUIRE(ba.NH <= kPS && ba.n_classes <= 8 && ba.n_heads <= kMaxHeads, "launch_dtrunk: %d Dense-on-trunk outputs", ba.NH);
    const int tiles = (ba.N + 15) / 16;
    hipLaunchKernelGGL(dtrunk_kernel, dim3(ba.T * C / 16, std::max(1, std::min(8, tiles / 16))), dim3(256), 0, st, ba, d_flat, d_acts, d_dpre, d_gt);
    return smh::launch_status("dtrunk_kernel");
}

int launch_backward_bf16(const BwdArgs &ba, void **d_pack, size_t *pack_cap, const float *d_x, const float *d_flat,
                         const float *d_acts, const float *d_drop_tcn, const float *d_dpre, float *d_grad, const float *d_upre,
                         hipStream_t st) {
    BwdGeo geo;
    if (!bwd_geo(ba.T, ba.n_dil, &geo)) return kBwdBf16Unsupported;
    BwdArgs bp = ba;
    bp.split3 = 0;  // this kernel's timing-probe mask (results invalid): 1 no atomics, 2 no phase B, 4 no phase A, 8 no phase C
    if (const char *ev = smh::probe_env("SMH_BWD_PROBE")) bp.split3 = atoi(ev);
    // workspace: [the blocks' kernels as split A operands | gt (N, T, 32)]
    const size_t pack_bytes = (size_t)ba.n_blocks * kSlotBytes;
    const size_t need = pack_bytes + (size_t)ba.N * ba.T * C * sizeof(float);
    if (*pack_cap < need) {
        if (*d_pack) SMH_CHECK_HIP(hipFree(*d_pack));
        *d_pack = nullptr, *pack_cap = 0;
        SMH_CHECK_HIP(hipMalloc(d_pack, need));
        *pack_cap = need;
    }
    float *d_gt = reinterpret_cast<float *>(static_cast<char *>(*d_pack) + pack_bytes);
    if (ba.n_blocks > 0) {
        const int n = ba.n_blocks * (kUnitsPerBlk + 1) * 64;
        hipLaunchKernelGGL(pack_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_flat, ba.off, ba.n_blocks, (bf16x8 *)*d_pack);
        int rc = smh::launch_status("pack_bwd_kernel");
        if (rc) return rc;
    }
    {
        int rc = launch_dtrunk(ba, d_flat, d_acts, d_dpre, d_gt, st);
        if (rc) return rc;
    }
    const size_t lds = (size_t)geo.G * geo.per_patch + kSlotBytes;
    auto go = [&](auto kern) -> int {
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((ba.N + geo.G - 1) / geo.G), dim3(512 * geo.G), lds, st, bp, geo, d_x, d_flat, (const bf16x8 *)*d_pack,
                           d_acts, d_drop_tcn, (const float *)d_gt, d_grad, d_upre);
        return SMH_OK;
    };
    int rc;
    if (geo.K32 == 3 && geo.G == 2) rc = ba.stamps ? go(tcn_backward_bf16_kernel<3, 2, true>) : go(tcn_backward_bf16_kernel<3, 2, false>);
    else if (geo.K32 == 1 && geo.G == 2) rc = go(tcn_backward_bf16_kernel<1, 2, false>);
    else if (geo.K32 == 2 && geo.G == 2) rc = go(tcn_backward_bf16_kernel<2, 2, false>);
    else if (geo.K32 == 4 && geo.G == 1) rc = go(tcn_backward_bf16_kernel<4, 1, false>);
    else return kBwdBf16Unsupported;
    if (rc) return rc;
    return smh::launch_status("tcn_backward_bf16_kernel");
}

}  // namespace smh_tcn
