// a10-a12: B3_MTL inference forward = get_Lemaire_MTL_model (lib/proposed_architectures.py:85-170),
// MTL heads (:25-80; 5-class variant 5_class_classification.py:150-215) and the keras-tcn 2.3 trunk
// (third party, restated in oracle/b3_mtl.py):
//   x = Conv1D(32,1)(in);  24 x { y = Conv1D(32, k=3, dilation d, 'same')(x); y = relu(y);
//   y = y / (max_c|y| + 1e-5);  x = x + Conv1D(32,1)(y) };  x = relu(x);  Flatten;
//   3C = softmax(Dense);  heads: Dense(16) -> BN -> relu -> Dense(1|2|3) [sigmoid|linear].
//
// gfx950 mapping (exact f32, v_mfma_f32_16x16x4_f32): every product is computed TRANSPOSED,
//   D[channel (32 rows = 2 M-tiles)][time (16 columns)] = W^T[channel][k] * X^T[k][time],
// so a lane owns one time step and 8 of its 32 channels in registers.  Consequences:
//   * the channel-max normalisation is 7 in-lane max + 2 cross-lane steps,
//   * the normalised activations ARE the B operand of the following 1x1 convolution (k order chosen
//     to match the accumulator layout) -- no LDS round trip between the two convolutions,
//   * bias and residual are folded into the accumulator initialisation.
// One workgroup owns G patches; the activations x (G*T rows x 32 ch, fp32) live in LDS for all
// 24 blocks (double buffered, one barrier per block).  Block weights are pre-packed in MFMA A-operand order; the
// workgroup copies block b+1 from L2 into an LDS slot by LDS-DMA while block b is computed, and every wave reads its
// 80 operand registers from that slot at the top of the block (one register set: 9..12 waves fit under 170 VGPRs).
// Waves per workgroup follow the tile count: 17 column tiles (4 patches of 68 frames) run on 9 waves -- 2 tiles each,
// one wave with 1 -- because with 8 the wave holding 3 tiles sets the time of every block.
// The Dense layers that read the flattened trunk (3C logits + the Dense(16) of every head) run in the
// same kernel as one more MFMA product D[output][patch] on the LDS-resident activations; BN / relu /
// output Dense / sigmoid / softmax finish in a few threads.  One launch per forward.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "smh_model.h"
#include "smh_tcn_heads.h"

using namespace smh_tcn;

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// max over the four lanes that hold the same time step (l, l^16, l^32, l^48): gfx950's VALU lane swaps instead of two
// ds_bpermute round trips
__device__ __forceinline__ float quad_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    const unsigned ua = __float_as_uint(a);
    const auto r16 = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
    return fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
}


struct BlockW {
    float wc[24][2], wp[8][2];
    f32x4 b1lo, b1hi, b2lo, b2hi;
};

// The skewed task body takes its A operands through one of two providers with the same interface -- conv(g), g < 12: the
// four operand slots {k-step 2g M-tile 0, 2g M-tile 1, 2g+1 M-tile 0, 2g+1 M-tile 1} of the dilated conv (k-step s = 8 tap + s8);
// pw(g), g < 4: the same for the 1x1 conv; b1lo() .. b2hi(): the biases in accumulator layout.
struct RegWeights {  // registers (kSkew: two BlockW sets per wave)
    const BlockW &w;
    __device__ __forceinline__ f32x4 conv(int g) const { return f32x4{w.wc[2 * g][0], w.wc[2 * g][1], w.wc[2 * g + 1][0], w.wc[2 * g + 1][1]}; }
    __device__ __forceinline__ f32x4 pw(int g) const { return f32x4{w.wp[2 * g][0], w.wp[2 * g][1], w.wp[2 * g + 1][0], w.wp[2 * g + 1][1]}; }
    __device__ __forceinline__ f32x4 b1lo() const { return w.b1lo; }
    __device__ __forceinline__ f32x4 b1hi() const { return w.b1hi; }
    __device__ __forceinline__ f32x4 b2lo() const { return w.b2lo; }
    __device__ __forceinline__ f32x4 b2hi() const { return w.b2hi; }
};
struct LdsWeights {  // an LDS slot holding the packed block (kSkew16): one ds_read_b128 per group, conflict-free (lane-contiguous)
    const float *blk;  // LDS address of the block: [16 groups][64 lanes][4] then b1[32], b2[32]
    int lane, q;
    __device__ __forceinline__ f32x4 at(int i) const { return *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(blk + i, 16)); }
    __device__ __forceinline__ f32x4 conv(int g) const { return at(g * 256 + 4 * lane); }
    __device__ __forceinline__ f32x4 pw(int g) const { return at((12 + g) * 256 + 4 * lane); }
    __device__ __forceinline__ f32x4 b1lo() const { return at(4096 + 4 * q); }
    __device__ __forceinline__ f32x4 b1hi() const { return at(4096 + 16 + 4 * q); }
    __device__ __forceinline__ f32x4 b2lo() const { return at(4096 + 32 + 4 * q); }
    __device__ __forceinline__ f32x4 b2hi() const { return at(4096 + 48 + 4 * q); }
};

// Block weights, LDS -> registers.  The workgroup's LDS copy of the block is staged by LDS-DMA one block ahead: every wave
// needs the whole block, so the 16.6 KB come from L2 once per workgroup instead of once per wave.  Packed order
// (pack_weights below): operand slot n = 2 s + mt (dilated conv, s < 24) or 48 + 2 e + mt (1x1 conv, e < 8) of lane l
// sits at [(n / 4) * 256 + 4 l + n % 4], so one ds_read_b128 per lane delivers four slots, conflict-free.
// the 20 ds_read_b128 of a block, numbered: 0..11 dilated-conv operand groups, 12..15 1x1-conv groups, 16..19 the biases
template <int G0, int G1>
__device__ __forceinline__ void load_block_part(BlockW &w, const float *wblk, int lane, int q) {
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(wblk) + lane;
    const float *b1 = wblk + 24 * 2 * 64 + 8 * 2 * 64;
#pragma unroll
    for (int g = G0; g < G1; ++g) {
        if (g < 12) {
            const f32x4 v = wv[g * 64];
            w.wc[2 * g][0] = v[0], w.wc[2 * g][1] = v[1], w.wc[2 * g + 1][0] = v[2], w.wc[2 * g + 1][1] = v[3];
        } else if (g < 16) {
            const f32x4 v = wv[g * 64];
            const int h = g - 12;
            w.wp[2 * h][0] = v[0], w.wp[2 * h][1] = v[1], w.wp[2 * h + 1][0] = v[2], w.wp[2 * h + 1][1] = v[3];
        } else if (g == 16) {
            w.b1lo = *reinterpret_cast<const f32x4 *>(b1 + 4 * q);
        } else if (g == 17) {
            w.b1hi = *reinterpret_cast<const f32x4 *>(b1 + 16 + 4 * q);
        } else if (g == 18) {
            w.b2lo = *reinterpret_cast<const f32x4 *>(b1 + 32 + 4 * q);
        } else {
            w.b2hi = *reinterpret_cast<const f32x4 *>(b1 + 48 + 4 * q);
        }
    }
}
__device__ __forceinline__ void load_block_lds(BlockW &w, const float *wblk, int lane, int q) {
    load_block_part<0, 20>(w, wblk, lane, q);
}

// Per-wave column tiles ("units" of 16 time steps), fixed for the whole kernel: row of this lane, its frame index inside
// the patch (the dilated taps are valid while 0 <= t + off < T) and its patch (dropout masks).
constexpr int kMaxTiles = 4;  // 32 tiles (512-frame patches) over 8 waves
struct TileInfo {
    int R[kMaxTiles];   // row 16 u + j (may lie past the last live row of a partial tile: then Rc = last row)
    int t[kMaxTiles];   // frame index of row min(R, GR - 1) inside its patch
    int g[kMaxTiles];   // its patch inside the workgroup (SpatialDropout1D masks, training only)
    int n;              // tiles of this wave
    int hrole;          // -1, or 0 / 1: this wave computes that half of the last tile (TcnArgs::split_last)
    int hR, ht, hg;     // ... whose row / frame / patch of this lane are these
};

// Prefetch of the NEXT block's weights (LDS -> the other register set) in four chunks of five ds_read_b128, issued at
// fixed points of the tile loop ("slots": 2 i = behind the operand reads of tile i, 2 i + 1 = behind its dilated-conv
// products).  LDS returns in order and the wait counter holds 15: with at most 8 operand reads + 5 weight reads in
// flight the compiler can wait for exactly the operands and let the weights stream in behind the products; all 20 reads
// in one place would stand between the barrier and the first product, like loading them at the top of the block.
struct NoPrefetch {
    template <int SLOT>
    __device__ __forceinline__ void slot() const {}
    __device__ __forceinline__ void rest(int) const {}
};
// One register set (9+ waves): the CURRENT block's weights are read from the LDS slot behind the first tile's operand reads.
// LDS returns in order, so the first product waits for its operands and the first weight chunk only; the rest of the
// 20 ds_read_b128 (180 KB per workgroup and block) arrive while the tile's products run instead of standing between the
// barrier and the first product.
struct WeightLoad {
    BlockW &w;
    const float *wblk;
    int lane, q;
    template <int SLOT>
    __device__ __forceinline__ void slot() const {
        if constexpr (SLOT == 0) load_block_lds(w, wblk, lane, q);
    }
    __device__ __forceinline__ void rest(int) const {}  // a wave without tiles needs no weights
};
struct WeightPrefetch {  // behind the last block the reads still run (they fetch the other slot's stale block, unused)
    BlockW &w;
    const float *wblk;
    int lane, q;
    template <int SLOT>
    __device__ __forceinline__ void slot() const {
        if constexpr (SLOT < 4) load_block_part<5 * SLOT, 5 * SLOT + 5>(w, wblk, lane, q);
    }
    // chunks whose slot this wave never reached (it owns fewer than two tiles)
    __device__ __forceinline__ void rest(int tiles) const {
        if (tiles < 2) {
            if (tiles < 1) load_block_part<0, 10>(w, wblk, lane, q);
            load_block_part<10, 20>(w, wblk, lane, q);
        }
    }
};

// One column tile (16 time steps x 32 channels) of one residual block: dilated conv -> relu -> channel-max normalisation ->
// 1x1 conv + bias + residual, xin rows -> xout row R.  R / t / g: this lane's row, its frame index inside its patch and its
// patch (TileInfo); I numbers the tile inside the wave's list (prefetch slots).
// drop: SpatialDropout1D masks of this block for the workgroup's first patch, (n, blk) stride dstride
// ZR: index of the all-zero row behind the activation rows (padding taps read it instead of masking every operand)
template <bool TRAIN, int I, class PF>
__device__ __forceinline__ void tile_compute(const BlockW &w, int d, bool side_taps, int T, int GR, int ZR, int R, int t, int g,
                                             int q, const float *__restrict__ xin, float *__restrict__ xout,
                                             const float *__restrict__ drop, int dstride, float *__restrict__ ub, int ustride,
                                             const PF &prefetch) {
    // ub (training): this block's slice of the saved dilated-conv outputs for the workgroup's first patch, patch stride ustride
    const int Rc = min(R, GR - 1);
    // all LDS operands of the tile up front: 3 taps x 8 channels of this lane's k slice (k index = tap*32 + c; MFMA
    // step s8 of a tap takes channel c = 8 q + s8 from lane group q, so a lane's eight B operands are contiguous in
    // its activation row: two ds_read_b128 per tap), then the residual row
    f32x4 b[3][2];
    bool any_tap[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int off = (tap - 1) * d;
        const bool ok = (tap == 1) || (side_taps && (t + off >= 0) && (t + off < T));
        any_tap[tap] = (tap == 1) || (side_taps && __any(ok));  // wave-uniform: no row of the tile is live through this tap
        // the reads are issued on every path (a dead tap reads the zero row): with a path-independent number of LDS
        // operations in flight the compiler can wait for exactly the operands a product needs
        const float *src = xin + (size_t)(ok ? Rc + off : ZR) * SX + 8 * q;
        b[tap][0] = *reinterpret_cast<const f32x4 *>(src);
        b[tap][1] = *reinterpret_cast<const f32x4 *>(src + 4);
    }
    const float *res = xin + (size_t)Rc * SX + 4 * q;
    f32x4 o0 = *reinterpret_cast<const f32x4 *>(res);
    f32x4 o1 = *reinterpret_cast<const f32x4 *>(res + 16);
    __builtin_amdgcn_sched_barrier(0);
    prefetch.template slot<2 * I>();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = w.b1lo, acc1 = w.b1hi;
    // centre tap first, as in the skew schedule's tasks: the two schedules then add the same products in the same order and
    // a patch's outputs do not depend on which of them its batch size selects (bit for bit)
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
        const int tap = ti == 0 ? 1 : (ti == 1 ? 0 : 2);
        if (!any_tap[tap]) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc0 = mfma4(w.wc[tap * 8 + 4 * h + s4][0], b[tap][h][s4], acc0);
                acc1 = mfma4(w.wc[tap * 8 + 4 * h + s4][1], b[tap][h][s4], acc1);
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    prefetch.template slot<2 * I + 1>();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TRAIN) {
        if (ub && R < GR) {
            const unsigned uo = (unsigned)(g * ustride + t * C + 4 * q);  // 32-bit offset from a uniform base
            *reinterpret_cast<f32x4 *>(ub + uo) = acc0;
            *reinterpret_cast<f32x4 *>(ub + uo + 16) = acc1;
        }
    }
    // relu + channel-max normalisation ('norm_relu'); 1 / (max + eps) by v_rcp_f32 (1 ulp)
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc0[r] = fmaxf(acc0[r], 0.f);
        acc1[r] = fmaxf(acc1[r], 0.f);
        mx = fmaxf(mx, fmaxf(acc0[r], acc1[r]));
    }
    mx = quad_max(mx);
    const float inv = __builtin_amdgcn_rcpf(mx + kNormEps);
    f32x4 dm0 = {1.f, 1.f, 1.f, 1.f}, dm1 = {1.f, 1.f, 1.f, 1.f};
    if constexpr (TRAIN) {
        if (drop) {
            const unsigned dofs = (unsigned)(g * dstride + 4 * q);  // 32-bit offset from a uniform base
            dm0 = *reinterpret_cast<const f32x4 *>(drop + dofs);
            dm1 = *reinterpret_cast<const f32x4 *>(drop + dofs + 16);
        }
    }
    // 1x1 conv on the normalised activations + bias + residual, all from registers
    o0 += w.b2lo, o1 += w.b2hi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float y0 = acc0[r] * inv * dm0[r];  // channel 4q + r
        o0 = mfma4(w.wp[r][0], y0, o0);
        o1 = mfma4(w.wp[r][1], y0, o1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float y1 = acc1[r] * inv * dm1[r];  // channel 16 + 4q + r
        o0 = mfma4(w.wp[4 + r][0], y1, o0);
        o1 = mfma4(w.wp[4 + r][1], y1, o1);
    }
    float *dst = xout + (size_t)R * SX + 4 * q;
    *reinterpret_cast<f32x4 *>(dst) = o0;
    *reinterpret_cast<f32x4 *>(dst + 16) = o1;
}

// The LAST column tile of a block shared by two waves on different SIMDs (TcnArgs::split_last): wave H = 0 / 1 computes output
// channels 16 H .. 16 H + 15 -- the accumulator chain acc0 / acc1 of tile_compute, the same products in the same order, so
// the results are bit-identical to the undivided tile.  With 5 tiles on 8 waves (one 68-frame patch per workgroup: every batch
// up to 256 patches, BASELINE config 3) the fifth tile was a second full tile on SIMD 0 while three waves idled; as halves on
// two otherwise idle waves the busiest SIMD runs 96 instead of 128 matrix instructions per block.  What the halves owe each
// other is the relu'd dilated-conv output (the channel maximum runs over all 32 channels, and the 1x1 conv's k runs over all 32):
// 16 bytes per lane each way through `xch`, ordered by one flag word per half that carries the block number.
// epoch = block + 1; xflag[0..1] zeroed before the first block; the per-block barrier keeps the epochs of the two waves in step.
template <bool TRAIN, int H>
__device__ __forceinline__ void half_tile_compute(const BlockW &w, int d, bool side_taps, int T, int GR, int ZR, int R, int t, int g,
                                                  int q, int lane, const float *__restrict__ xin, float *__restrict__ xout,
                                                  const float *__restrict__ drop, int dstride, float *__restrict__ ub, int ustride,
                                                  float *__restrict__ xch, int epoch, int spin_limit, int *status) {
    const int Rc = min(R, GR - 1);
    f32x4 b[3][2];
    bool any_tap[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int off = (tap - 1) * d;
        const bool ok = (tap == 1) || (side_taps && (t + off >= 0) && (t + off < T));
        any_tap[tap] = (tap == 1) || (side_taps && __any(ok));
        const float *src = xin + (size_t)(ok ? Rc + off : ZR) * SX + 8 * q;
        b[tap][0] = *reinterpret_cast<const f32x4 *>(src);
        b[tap][1] = *reinterpret_cast<const f32x4 *>(src + 4);
    }
    f32x4 o = *reinterpret_cast<const f32x4 *>(xin + (size_t)Rc * SX + 4 * q + 16 * H);
    f32x4 acc = H ? w.b1hi : w.b1lo;
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
        const int tap = ti == 0 ? 1 : (ti == 1 ? 0 : 2);
        if (!any_tap[tap]) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = mfma4(w.wc[tap * 8 + 4 * h + s4][H], b[tap][h][s4], acc);
    }
    if constexpr (TRAIN) {
        if (ub && R < GR) *reinterpret_cast<f32x4 *>(ub + (unsigned)(g * ustride + t * C + 4 * q) + 16 * H) = acc;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
    // hand this half over, take the other
    __attribute__((address_space(3))) volatile int *xflag = (__attribute__((address_space(3))) volatile int *)(xch + 2 * 64 * 4);
    *reinterpret_cast<f32x4 *>(xch + (size_t)(H * 64 + lane) * 4) = acc;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) xflag[H] = epoch;
    for (int spins = 0; xflag[1 - H] < epoch; ++spins) {
        if (spins > spin_limit) {  // cannot happen while both waves run the block; never hang the grid on it
            if (lane == 0 && status) atomicOr(status, 1);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    const f32x4 other = *reinterpret_cast<const f32x4 *>(xch + (size_t)((1 - H) * 64 + lane) * 4);
    const f32x4 a0 = H ? other : acc, a1 = H ? acc : other;  // channels 4q + r and 16 + 4q + r, as tile_compute holds them
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fmaxf(a0[r], a1[r]));
    mx = quad_max(mx);
    const float inv = __builtin_amdgcn_rcpf(mx + kNormEps);
    f32x4 dm0 = {1.f, 1.f, 1.f, 1.f}, dm1 = {1.f, 1.f, 1.f, 1.f};
    if constexpr (TRAIN) {
        if (drop) {
            const unsigned dofs = (unsigned)(g * dstride + 4 * q);
            dm0 = *reinterpret_cast<const f32x4 *>(drop + dofs);
            dm1 = *reinterpret_cast<const f32x4 *>(drop + dofs + 16);
        }
    }
    o += H ? w.b2hi : w.b2lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) o = mfma4(w.wp[r][H], a0[r] * inv * dm0[r], o);
#pragma unroll
    for (int r = 0; r < 4; ++r) o = mfma4(w.wp[4 + r][H], a1[r] * inv * dm1[r], o);
    *reinterpret_cast<f32x4 *>(xout + (size_t)R * SX + 4 * q + 16 * H) = o;
}

// one residual block for this wave's column tiles (barrier-per-block schedule); prefetch: see WeightPrefetch
template <bool TRAIN, class PF>
__device__ __forceinline__ void run_block(BlockW &w, int d, int T, int GR, int ZR, const TileInfo &ti, int q,
                                          const float *__restrict__ xin, float *__restrict__ xout,
                                          const float *__restrict__ drop, int dstride, float *__restrict__ ub, int ustride,
                                          PF prefetch, int lane = 0, float *__restrict__ xch = nullptr, int epoch = 0,
                                          int spin_limit = 0, int *status = nullptr) {
    const bool side_taps = d < T;  // |offset| >= T: the side taps only ever see zero padding
    auto tile = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        tile_compute<TRAIN, i>(w, d, side_taps, T, GR, ZR, ti.R[i], ti.t[i], ti.g[i], q, xin, xout, drop, dstride, ub, ustride,
                               prefetch);
    };
    static_assert(kMaxTiles == 4, "tile list below");
    if (ti.n > 0) tile(std::integral_constant<int, 0>{});  // wave-uniform
    if (ti.n > 1) tile(std::integral_constant<int, 1>{});
    if (ti.n > 2) tile(std::integral_constant<int, 2>{});
    if (ti.n > 3) tile(std::integral_constant<int, 3>{});
    prefetch.rest(ti.n);
    if (ti.hrole == 0)  // (wave-uniform)
        half_tile_compute<TRAIN, 0>(w, d, side_taps, T, GR, ZR, ti.hR, ti.ht, ti.hg, q, lane, xin, xout, drop, dstride, ub, ustride, xch,
                                    epoch, spin_limit, status);
    else if (ti.hrole == 1)
        half_tile_compute<TRAIN, 1>(w, d, side_taps, T, GR, ZR, ti.hR, ti.ht, ti.hg, q, lane, xin, xout, drop, dstride, ub, ustride, xch,
                                    epoch, spin_limit, status);
}

// MODE: how the 24 residual blocks are scheduled over the waves (launch_forward chooses)
constexpr int kOneSet = 0;   // barrier per block, one weight register set read from the LDS slot at the top of the block (9..12 waves)
constexpr int kPrefetch = 1; // barrier per block, two register sets (8 waves)
constexpr int kSkew = 2;     // no barrier: (block, tile) tasks taken from a counter by 8 waves, tile-level completion flags
// The same task schedule on SIXTEEN waves (four per SIMD, 128 VGPRs each): the block weights are not held in registers (two sets
// of 80 made the 8-wave build a 225-VGPR kernel, two waves per SIMD, whose product and non-product sections interleave at only
// two thirds of the matrix pipe's rate) but stay in LDS -- a ring of four block slots filled by LDS-DMA two blocks ahead -- and
// every product group reads its A operands from there (one ds_read_b128 per four operand slots).  Same instructions in the same
// order on the same values: outputs are bit-identical to the other schedules.  Inference only.
constexpr int kSkew16 = 3;
constexpr int kWeightRing = 4;   // LDS block slots of kSkew16 (blocks b - 1 .. b + 2 can be live at once)
constexpr int kSkewSpinLimit = 1 << 22;  // polls before a wave gives up on a dependency (never reached; the grid must drain)

// TRACE: tools/trace_model.py only -- s_memtime / s_memrealtime stamps into a.trace; every stamp compiles out otherwise
template <bool TRAIN, int MODE, bool TRACE = false>
__global__ void __launch_bounds__(MODE == kOneSet ? 768 : (MODE == kSkew16 ? 1024 : 512))
b3mtl_forward_kernel(TcnArgs a, const float *__restrict__ X, const float *__restrict__ W0, const float *__restrict__ Wb,
                     const float *__restrict__ WhA, const float *__restrict__ hp, float *__restrict__ trunk,
                     float *__restrict__ out, TrainIO tio) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (the wave index through readfirstlane: the compiler then keeps everything derived from it -- the task list of the
    // skewed schedule -- in scalar registers and branches on it without exec masks)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int n0 = blockIdx.x * a.G;
    const int g_here = min(a.G, a.N - n0);
    const int T = a.T;
    const int GR = g_here * T;
    const int units = (GR + 15) >> 4;
    const int ZR = a.GRP;  // all-zero row behind the GRP activation rows of each buffer (padding taps of the dilated conv)
    float *xa = lds, *xb = lds + (size_t)(a.GRP + 1) * SX;
    const bool tracing = TRACE && a.trace != nullptr;
    if (tracing && blockIdx.x < 256 && threadIdx.x == 0) a.trace[4 * 3000 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();

    // ---- initial Conv1D(32, 1): K order f = q*FQ + s so that every lane streams a contiguous run ----
    // Layer 0 is an HBM stream (261 KB of X per workgroup): every lane issues ALL loads of its column
    // tiles first (registers are free: the block weights are not loaded yet), so one HBM latency is
    // exposed instead of one per k-group.  A operands are staged once in LDS and shared by all tiles.
    // column tiles per wave whose rows are all in flight at once: 17 tiles over 8 waves -> 3; the 16-wave build has 128 VGPRs and
    // takes its (at most two) tiles one after the other, four waves per SIMD covering each other's latency
    constexpr int kMaxU = MODE == kSkew16 ? 1 : 3;
    constexpr int kRoundsU = MODE == kSkew16 ? 2 : 1;
    constexpr int kFQ4 = 15;    // float4 groups per lane for n_feat = 240
    {
        float *w0s = xb;  // layer-0 A operands live in the not-yet-used activation buffer
        const int nW0 = a.FQ * 2 * 64;
        const float *bias0 = W0 + (size_t)nW0;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(bias0 + 4 * q);
        const f32x4 bh = *reinterpret_cast<const f32x4 *>(bias0 + 16 + 4 * q);
        if (a.from_x0) {
            // layer 0 was computed by the feature kernel: sum its two per-half partials, add the bias
            // all loads of a thread in flight before the first sum (up to 4 rounds x 2 partials: one HBM latency, not four)
            constexpr int kX0R = 4;
            const int n4 = GR * (C / 4);
            for (int i0 = threadIdx.x; i0 < n4; i0 += kX0R * blockDim.x) {
                f32x4 pa[kX0R], pb[kX0R];
#pragma unroll
                for (int r = 0; r < kX0R; ++r) {
                    const int i = min(i0 + r * (int)blockDim.x, n4 - 1);
                    const int R = i >> 3, c4 = (i & 7) * 4;
                    const int g = R / T, t = R - g * T;
                    // packed partials (n, 2, T, 32), or windows of the per-frame partials (2, x0_T, 32) of a whole featuregram
                    const size_t first = a.x0_shift ? (size_t)min((n0 + g) * a.x0_shift, a.x0_T - T) : (size_t)(n0 + g) * 2 * T;
                    const float *p0 = X + ((first + t) * C + c4);
                    pa[r] = *reinterpret_cast<const f32x4 *>(p0);
                    pb[r] = *reinterpret_cast<const f32x4 *>(p0 + (size_t)(a.x0_shift ? a.x0_T : T) * C);
                }
#pragma unroll
                for (int r = 0; r < kX0R; ++r) {
                    const int i = i0 + r * (int)blockDim.x;
                    if (i < n4) {
                        const int R = i >> 3, c4 = (i & 7) * 4;
                        *reinterpret_cast<f32x4 *>(xa + (size_t)R * SX + c4) = pa[r] + pb[r] + *reinterpret_cast<const f32x4 *>(bias0 + c4);
                    }
                }
            }
        } else if (a.vec_ok && a.FQ == 4 * kFQ4 && units <= kRoundsU * kMaxU * nw) {
            for (int i = threadIdx.x; i < nW0; i += blockDim.x) w0s[i] = W0[i];
            for (int rnd = 0; rnd < kRoundsU; ++rnd) {
            f32x4 xr[kMaxU][kFQ4];
#pragma unroll
            for (int i = 0; i < kMaxU; ++i) {
                const int u = min(wave + (rnd * kMaxU + i) * nw, units - 1);
                const int Rc = min(16 * u + j, GR - 1);
                const float *xrow = X + ((size_t)n0 * T + Rc) * a.F + (size_t)q * a.FQ;
#pragma unroll
                for (int g = 0; g < kFQ4; ++g) xr[i][g] = *reinterpret_cast<const f32x4 *>(xrow + 4 * g);
            }
            if (rnd == 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < kMaxU; ++i) {
                const int u = wave + (rnd * kMaxU + i) * nw;
                if (u < units) {  // wave-uniform
                    f32x4 c0 = bl, c1 = bh;
#pragma unroll
                    for (int g = 0; g < kFQ4; ++g) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            c0 = mfma4(w0s[((4 * g + e) * 2 + 0) * 64 + lane], xr[i][g][e], c0);
                            c1 = mfma4(w0s[((4 * g + e) * 2 + 1) * 64 + lane], xr[i][g][e], c1);
                        }
                    }
                    float *dst = xa + (size_t)(16 * u + j) * SX + 4 * q;
                    *reinterpret_cast<f32x4 *>(dst) = c0;
                    *reinterpret_cast<f32x4 *>(dst + 16) = c1;
                }
            }
            }
        } else {
            for (int i = threadIdx.x; i < nW0; i += blockDim.x) w0s[i] = W0[i];
            __syncthreads();
            for (int u = wave; u < units; u += nw) {
                const int R = 16 * u + j;
                const int Rc = min(R, GR - 1);
                const float *xr = X + ((size_t)n0 * T + Rc) * a.F + (size_t)q * a.FQ;
                f32x4 c0 = bl, c1 = bh;
                for (int s = 0; s < a.FQ; ++s) {
                    const float xv = (q * a.FQ + s < a.F) ? xr[s] : 0.f;
                    c0 = mfma4(w0s[(s * 2 + 0) * 64 + lane], xv, c0);
                    c1 = mfma4(w0s[(s * 2 + 1) * 64 + lane], xv, c1);
                }
                float *dst = xa + (size_t)R * SX + 4 * q;
                *reinterpret_cast<f32x4 *>(dst) = c0;
                *reinterpret_cast<f32x4 *>(dst + 16) = c1;
            }
        }
    }

    // ---- residual blocks.  Block weights: global -> LDS by LDS-DMA one block ahead (two LDS slots), LDS -> registers
    // at the top of the block ----
    float *ws = lds + 2 * (size_t)(a.GRP + 1) * SX;
    auto stage = [&](int blk) {
        const char *src = reinterpret_cast<const char *>(Wb + (size_t)blk * kBlockFloats);
        char *dst = reinterpret_cast<char *>(ws + (size_t)(blk & 1) * kBlockFloats);
        constexpr int nch = kBlockFloats * 4 / 16;
        for (int i = wave; i * 64 < nch; i += nw) {
            const int c = i * 64 + lane;
            if (c < nch)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)c * 16),
                                                 (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
        }
    };
    // this wave's column tiles and the zero rows
    TileInfo ti;
    ti.n = 0;
    // (split_last: the last tile belongs to no wave's list; waves (units - 1) % nw and the next one take a half each)
    const bool split = MODE == kPrefetch && a.split_last && units >= 2;
#pragma unroll
    for (int i = 0; i < kMaxTiles; ++i) {
        const int u = wave + i * nw;
        const int R = 16 * u + j, Rc = min(R, GR - 1);
        ti.R[i] = R, ti.g[i] = Rc / T, ti.t[i] = Rc - ti.g[i] * T;
        if (u < units - (split ? 1 : 0)) ti.n = i + 1;
    }
    {
        const int hw0 = (units - 1) % nw, hw1 = (hw0 + 1) % nw;
        ti.hrole = split ? (wave == hw0 ? 0 : (wave == hw1 ? 1 : -1)) : -1;
        const int R = 16 * (units - 1) + j, Rc = min(R, GR - 1);
        ti.hR = R, ti.hg = Rc / T, ti.ht = Rc - ti.hg * T;
    }
    float *xch = lds + 2 * (size_t)(a.GRP + 1) * SX + 2 * (size_t)kBlockFloats;  // behind the weight slots: 2 x 64 x 16 bytes + the flags
    if (split && threadIdx.x < 2) reinterpret_cast<int *>(xch + 2 * 64 * 4)[threadIdx.x] = 0;
    if (threadIdx.x < SX) xa[(size_t)ZR * SX + threadIdx.x] = 0.f, xb[(size_t)ZR * SX + threadIdx.x] = 0.f;
    float *xin = xa, *xout = xb;
    const int nslot = a.n_blocks + 1;
    auto save_acts = [&](const float *src, int slot) {  // block input -> acts[n][slot][t][c]
        if constexpr (TRAIN) {
            for (int i = threadIdx.x; i < GR * (C / 4); i += blockDim.x) {
                const int R = i >> 3, c4 = (i & 7) * 4;
                const int g = R / T, t = R - g * T;
                *reinterpret_cast<f32x4 *>(tio.acts + (((size_t)(n0 + g) * nslot + slot) * T + t) * C + c4) =
                    *reinterpret_cast<const f32x4 *>(src + (size_t)R * SX + c4);
            }
        }
    };
    const float *drop0 = TRAIN && tio.drop_tcn ? tio.drop_tcn + (size_t)n0 * a.n_blocks * C : nullptr;
    const int dstride = a.n_blocks * C;
    const int ustride = a.n_blocks * T * C;
    float *ub0 = TRAIN && tio.upre ? tio.upre + (size_t)n0 * ustride : nullptr;
    auto one_block = [&](int blk, BlockW &w, auto prefetch) {
        save_acts(xin, blk);
        run_block<TRAIN>(w, 1 << (blk % a.n_dil), T, GR, ZR, ti, q, xin, xout, drop0 ? drop0 + (size_t)blk * C : nullptr, dstride,
                         ub0 ? ub0 + (size_t)blk * T * C : nullptr, ustride, prefetch, lane, xch, blk + 1, a.spin_limit, a.status);
        float *tmp = xin;
        xin = xout;
        xout = tmp;
    };
    if (MODE != kSkew && MODE != kSkew16 && tracing && blockIdx.x < 1024 && threadIdx.x == 0) {
        a.trace[4 * (2000 + blockIdx.x)] = __builtin_amdgcn_s_memtime();
        a.trace[4 * (2000 + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime();
    }
    static_assert(!(TRAIN && MODE == kSkew16), "the 16-wave schedule is built for inference");
    // completion flags of the skewed schedules: kSkew keeps them in the (unused) weight slots, kSkew16 behind its weight ring
    float *flagbase = ws + (MODE == kSkew16 ? (size_t)kWeightRing * kBlockFloats : 0);
    if constexpr (MODE == kSkew || MODE == kSkew16) {
        // Skewed schedule.  Task n = (block n / units, tile n % units) belongs to wave n % 8, i.e. SIMD n % 4: every SIMD gets a
        // quarter of the block x tile grid (17 tiles per block are 5-4-4-4 under a barrier per block, 4.25 each here).  A task
        // starts when the tiles it touches have finished the previous block -- done[x] counts the blocks tile x has completed:
        //   read-after-write   rows +-d(block) of buffer block % 2, written by the previous block's tasks,
        //   write-after-read   its output row replaces x(block - 1), which the previous block's tasks read at +-d(block - 1).
        // Every dependency is a lower-numbered task, every wave runs its tasks in rising order and a wave never waits while
        // it holds an unpublished flag, so the lowest unfinished task can always run: no deadlock.
        // Weights: each wave reads block b + 1 from L2 into its second register set while it works on block b (nothing
        // shared, nothing to synchronise).
        // The synchronisation is taken off the critical path of a task (it cost 18 % of the loop when every task polled,
        // read its operands, computed, drained its stores and set its flag in sequence):
        //   * the flag of task k is published inside task k + 1 of the wave, behind its first tap's products -- LDS returns
        //     in order, so by then the rows of task k have long landed and the drain is free;
        //   * the flags task k + 1 depends on are sampled there too and judged behind the dilated-conv products of task k
        //     + 1..., i.e. one task ahead: if they stand, the operands of the NEXT task are read into the registers the
        //     current one has just finished with, under its 1x1-conv products.  Only when they do not stand does the wave
        //     publish, poll and read in sequence after the task.
        // (an LDS-typed pointer: a generic volatile access would be a flat load that waits for the weight loads in flight)
        typedef __attribute__((address_space(3))) volatile int lds_vint;
        // [0..31] per tile, [56..59] (kSkew16) the block whose weights have landed in ring slot 0..3, [62] the task counter,
        // [63] = a wave gave up waiting, [64..127] per-lane scratch words (targets of the lanes that have nothing to say)
        lds_vint *done = (lds_vint *)(flagbase);
        if (threadIdx.x < 64) {
            int v = threadIdx.x == 62 ? nw : 0;
            if (MODE == kSkew16 && threadIdx.x >= 56 && threadIdx.x < 60)  // blocks 0 and 1 are staged below, before the barrier
                v = (threadIdx.x - 56 < 2 && threadIdx.x - 56 < a.n_blocks) ? threadIdx.x - 56 : -1;
            done[threadIdx.x] = v;
        }
        // kSkew16: block blk -> ring slot blk % 4 by LDS-DMA, 1 KiB per wave instruction (17 for the 16.25 KiB of a block),
        // shared out over waves w0, w0 + nwv, ...
        auto stage_ring = [&](int blk, int w0, int nwv) {
            const char *src = reinterpret_cast<const char *>(Wb + (size_t)blk * kBlockFloats);
            char *dst = reinterpret_cast<char *>(ws + (size_t)(blk & (kWeightRing - 1)) * kBlockFloats);
            constexpr int nch = kBlockFloats * 4 / 16;
            for (int i = w0; i * 64 < nch; i += nwv) {
                const int c = i * 64 + lane;
                if (c < nch)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)c * 16),
                                                     (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
            }
        };
        BlockW w0, w1;
        if constexpr (MODE == kSkew) {
            load_block_lds(w0, Wb, lane, q);
        } else {
            stage_ring(0, wave, nw);
            if (a.n_blocks > 1) stage_ring(1, wave, nw);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed; the barrier covers the others'
        }
        __syncthreads();  // x0, zero rows, flags
        save_acts(xin, 0);  // (training) the input of block 0; every later slot is written by the task that produces it
        // save_acts copies rows of xa striding over ALL tiles, and block-1 tasks write their outputs back into xa: the flags only
        // order tasks, not this copy, so every wave's share of it is finished before any task starts (one barrier per launch)
        if constexpr (TRAIN) __syncthreads();
        if (tracing && blockIdx.x < 1024 && threadIdx.x == 0) {
            a.trace[4 * (2000 + blockIdx.x)] = __builtin_amdgcn_s_memtime();
            a.trace[4 * (2000 + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime();
        }
        const int n_tasks = a.n_blocks * units;
        // Per-task arithmetic is kept short on purpose: a wave issues about one instruction per four cycles, so a hundred
        // instructions of index arithmetic per task are as long as sixteen products.  Divisions by multiplication:
        //   x / v == (x * ceil(2^16 / v)) >> 16 for x < 2048, v <= 32;   x / T == umulhi(x, ceil(2^32 / T)) for x, T < 2^16
        const int m_units = (65536 + units - 1) / units, m_dil = (65536 + a.n_dil - 1) / a.n_dil;
        const unsigned m_T = (unsigned)(((1ull << 32) + (unsigned)T - 1) / (unsigned)T);
        const unsigned all_tiles = units >= 32 ? ~0u : (1u << units) - 1;
        const int xb_off = (a.GRP + 1) * SX, zrow = ZR * SX + 8 * q;
        struct Ops {  // LDS operands of one task: the three taps' k slices, the residual row, which side taps are live
            f32x4 b[3][2], r0, r1;
            bool live0, live2;
            // training only: the SpatialDropout1D mask of (patch, block) for this lane's channels, and this lane's row of the
            // saved activations, ((n0 + g) * nslot) * T + t -- slot 0; -1 for the rows behind the last patch
            f32x4 dm0, dm1;
            int arow, urow;  // urow: ((n0 + g) * n_blocks) * T + t, this lane's row of the saved dilated-conv outputs (block 0)
        };
        auto dil = [&](int blk) { return 1 << (blk - ((blk * m_dil) >> 16) * a.n_dil); };
        // (every index below is a multiple of four floats; said explicitly so that the reads stay ds_read_b128)
        auto ld4 = [&](int i) { return *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(lds + i, 16)); };
        auto issue_ops = [&](Ops &o, int blk, int u) {
            const int d = dil(blk);
            const int base = 16 * u, R = base + j;
            int g = (int)__umulhi((unsigned)base, m_T);  // patch of the tile's first row (only the training build uses g)
            int t = base - T * g + j;                    // frame index of row R inside its patch
            {   // (T >= 16 -- launch_forward takes this schedule only then --: a 16-row tile crosses at most one patch boundary;
                // the loop that shorter patches needed was two branches in every task's operand fetch)
                const bool wr = t >= T;
                t -= wr ? T : 0, g += wr ? 1 : 0;
            }
            const bool past = R >= GR;  // rows behind the last patch repeat its last row (frame T - 1)
            const int Rc = past ? GR - 1 : R;
            t = past ? T - 1 : t;
            if constexpr (TRAIN) {
                g = past ? g_here - 1 : g;
                o.arow = past ? -1 : ((n0 + g) * nslot) * T + t;
                o.urow = ((n0 + g) * a.n_blocks) * T + t;
                if (drop0) {  // (uniform) L2-resident; read one task ahead like the LDS operands
                    const float *dp = drop0 + (size_t)blk * C + (size_t)g * dstride + 4 * q;
                    o.dm0 = *reinterpret_cast<const f32x4 *>(dp);
                    o.dm1 = *reinterpret_cast<const f32x4 *>(dp + 16);
                } else {
                    o.dm0 = f32x4{1.f, 1.f, 1.f, 1.f}, o.dm1 = o.dm0;
                }
            }
            const bool ok0 = t >= d, ok2 = t < T - d;  // a dilation >= T leaves both side taps in the zero padding
            o.live0 = __any(ok0), o.live2 = __any(ok2);
            const int i1 = __mul24(Rc, SX) + ((blk & 1) ? xb_off : 0) + 8 * q;
            const int zr = zrow + ((blk & 1) ? xb_off : 0);
            const int i0 = ok0 ? i1 - d * SX : zr, i2 = ok2 ? i1 + d * SX : zr;
            o.b[0][0] = ld4(i0), o.b[0][1] = ld4(i0 + 4);
            o.b[1][0] = ld4(i1), o.b[1][1] = ld4(i1 + 4);
            o.b[2][0] = ld4(i2), o.b[2][1] = ld4(i2 + 4);
            const int ir = i1 - 4 * q;
            o.r0 = ld4(ir), o.r1 = ld4(ir + 16);
        };
        // the tiles whose flags task (blk, u) needs, as a bit mask (units <= 32): every tile within ceil(max(d(blk), d(blk - 1)) /
        // 16) of u -- a superset of the tiles its rows +-d actually fall into (the ones in between are older still), built with
        // two 64-bit shifts instead of the eighty scalar instructions the exact set cost per task
        auto dep_mask = [&](int blk, int u) {
            // (a dilation >= T reaches nothing: its side taps lie in the zero padding of every row, see issue_ops)
            const int dr = dil(blk), dw = dil(blk > 0 ? blk - 1 : 0);
            const int dmax = max(dr < T ? dr : 0, dw < T ? dw : 0);
            const int w = min((dmax + 15) >> 4, 31);
            const unsigned long long win = ((2ull << (2 * w)) - 1ull) << u;  // 2 w + 1 ones from bit u
            return (unsigned)(win >> w) & all_tiles;                         // ... centred on u
        };
        auto stands = [&](int have, int blk, unsigned mask) {  // every tile of the mask has finished block blk - 1
            const unsigned ok = (unsigned)__ballot(have >= blk);
            return (ok & mask) == mask;
        };
        // (kSkew16) the weights of block blk have landed in their ring slot: flag word 56 + slot, sampled with the tile flags
        auto w_ready = [&](int have, int blk) {
            if constexpr (MODE != kSkew16) return true;
            return __builtin_amdgcn_readlane(have, 56 + (blk & (kWeightRing - 1))) >= blk;
        };
        auto wait_for = [&](int blk, int u) {
            const unsigned mask = blk > 0 ? dep_mask(blk, u) : 0u;
            for (int spins = 0;; ++spins) {
                const int have = done[lane];
                if (stands(have, blk, mask) && w_ready(have, blk)) break;
                if (__any(lane == 63 && have != 0)) break;
                if (spins > a.spin_limit) {
                    if (lane == 0) done[63] = 1;
                    break;
                }
                // (a.tune bits 9..12, tools/gpu/r3_poll.sh only: extra sleeps per poll -- measured, changes neither time nor count)
                for (int s = 0; s <= ((a.tune >> 9) & 15); ++s) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        };
        // the task counter (done[62], starts at the number of waves: wave w begins with task w).  A wave takes its next task
        // late -- behind the dilated-conv products of the current one -- so that at most the 8 running tasks and a few taken
        // ones are unfinished at any time: the tiles a taken task depends on lie 15..19 tasks back and have long finished.
        // (a.tune bits 13..16, probe: wave k starts k x that many s_sleep(4) = 256-cycle units late -- an even stagger of the eight
        // waves puts a task's dependencies 7/8 of a task time behind it and the two waves of a SIMD half a task apart)
        for (int s = wave * ((a.tune >> 13) & 15); s > 0; --s) __builtin_amdgcn_s_sleep(4);
        // (a.tune bits 17..18, probe: a static priority for the younger wave of every SIMD -- arbitration by age lets the older one
        // complete 55-57 tasks to the younger's 45-47; set once, outside the task loop)
        if (wave >= 4) {
            switch ((a.tune >> 17) & 3) {
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                case 3: __builtin_amdgcn_s_setprio(3); break;
                default: break;
            }
        }
        int n = wave, blk = (n * m_units) >> 16, u = n - blk * units;  // this wave's current task
        int pend_u = -1, pend_v = 0;                                   // a finished task whose flag is not published yet
        Ops cur;
        bool have_cur = n < n_tasks;
        if (have_cur) {
            if (blk > 0) wait_for(blk, u);  // fewer tiles than waves: a wave's first task may sit in a later block (kSkew16: + its weights)
            issue_ops(cur, blk, u);
        }
        // Branch-free on purpose (as is the task counter below): a lane-0-only LDS access compiles to an exec save, a skip branch and
        // an exec restore, and every branch ends a scheduling region of the task body -- two uniform branches more per task cost
        // this kernel 3 % (round 3).  All 64 lanes store; lane 0 hits the flag, lane l > 0 (and lane 0 when nothing is pending)
        // a scratch word of its own behind the flags: done[64 + l].
        auto publish = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the finished task's rows are in LDS before its flag moves
            // (a.tune & 256, test_skew_give_up_is_reported only: wave 1 never publishes -- its dependants must give up)
            const bool real = lane == 0 && pend_u >= 0 && !((a.tune & 256) && wave == 1);
            done[real ? pend_u : 64 + lane] = pend_v;
            pend_u = -1;
        };
        // (kSkew16) a block this wave has put on its way into the ring (stage_ring, below) and not yet named in its slot's flag word
        int staged = -1;
        auto publish_staged = [&]() {
            if constexpr (MODE == kSkew16) {
                if (staged >= 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA pieces have landed in LDS
                    if (lane == 0) done[56 + (staged & (kWeightRing - 1))] = staged;
                    staged = -1;
                }
            }
        };
        auto one_task = [&](const auto &w) {
            {
                const int dsti = __mul24(16 * u + j, SX) + ((blk & 1) ? 0 : xb_off) + 4 * q;  // this lane's output row
                const bool stamp = tracing && (a.tune & 128) && blockIdx.x == 0;
                unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
                if (stamp) st[0] = __builtin_amdgcn_s_memtime();
                f32x4 acc0 = w.b1lo(), acc1 = w.b1hi();
                auto half_tap = [&](int tp, int h) {  // k-steps 8 tp + 4 h + s4, s4 = 0..3 = operand groups 4 tp + 2 h, + 1
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const f32x4 wv = w.conv(4 * tp + 2 * h + g2);
                        acc0 = mfma4(wv[0], cur.b[tp][h][2 * g2], acc0);
                        acc1 = mfma4(wv[1], cur.b[tp][h][2 * g2], acc1);
                        acc0 = mfma4(wv[2], cur.b[tp][h][2 * g2 + 1], acc0);
                        acc1 = mfma4(wv[3], cur.b[tp][h][2 * g2 + 1], acc1);
                    }
                };
                half_tap(1, 0);
                publish();  // behind the first products: the previous task's flag
                if (stamp) st[1] = __builtin_amdgcn_s_memtime();
                half_tap(1, 1);
                if (cur.live0) half_tap(0, 0), half_tap(0, 1);
                if (cur.live2) half_tap(2, 0), half_tap(2, 1);
                if constexpr (TRAIN) {  // the dilated-conv output before the relu: the backward's gates
                    if (tio.upre && cur.arow >= 0) {
                        float *up = tio.upre + ((size_t)cur.urow + (size_t)blk * T) * C + 4 * q;
                        *reinterpret_cast<f32x4 *>(up) = acc0;
                        *reinterpret_cast<f32x4 *>(up + 16) = acc1;
                    }
                }
                // (kSkew16) the block staged at the top of this task has had the dilated-conv products' time to land: name it BEFORE this
                // wave can wait for anything -- with few tiles per block its own next task may need that very block
                publish_staged();
                // take the next task and sample the flags; both are judged behind the epilogue
                if (stamp) st[2] = __builtin_amdgcn_s_memtime();
                // (lane 0 counts on the task counter, every other lane on its scratch word: one ds_add_rtn, no exec branch)
                const int taken = __hip_atomic_fetch_add((int *)(done + (lane == 0 ? 62 : 64 + lane)), 1, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                const int have = done[lane];
                // relu + channel-max normalisation ('norm_relu'); 1 / (max + eps) by v_rcp_f32 (1 ulp)
                float mx = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0[r] = fmaxf(acc0[r], 0.f);
                    acc1[r] = fmaxf(acc1[r], 0.f);
                    mx = fmaxf(mx, fmaxf(acc0[r], acc1[r]));
                }
                mx = quad_max(mx);
                const float inv = __builtin_amdgcn_rcpf(mx + kNormEps);
                f32x4 o0 = cur.r0 + w.b2lo(), o1 = cur.r1 + w.b2hi();
                asm volatile("" : "+v"(o0), "+v"(o1));  // the sums now: the residual registers are about to be reloaded
                int arow = 0;
                if constexpr (TRAIN) {  // normalised and masked now: issue_ops below reloads cur (the next task's mask)
                    acc0 = acc0 * inv * cur.dm0, acc1 = acc1 * inv * cur.dm1;
                    arow = cur.arow;
                }
                // the 1x1 conv's operand groups now (LdsWeights: four ds_read_b128 in front of the next task's operand reads -- LDS
                // returns in order, the products must not queue behind those; RegWeights: nothing)
                const f32x4 pwv[4] = {w.pw(0), w.pw(1), w.pw(2), w.pw(3)};
                const int nn = __builtin_amdgcn_readfirstlane(taken);
                if (stamp) st[3] = __builtin_amdgcn_s_memtime();
                // The operand registers are free: the next task's operands are read under the 1x1-conv products.  Unconditionally
                // -- one straight-line region from here to the stores, which the compiler interleaves with the sixteen products;
                // whether the reads were allowed (the task exists and its tiles stand) only decides if they count.
                const bool have_next = nn < n_tasks;
                const int nnc = min(nn, n_tasks - 1);
                const int nblk = (nnc * m_units) >> 16, nu = nnc - nblk * units;
                // (no short circuits: every flag word is >= 0, so block 0 stands by itself, and three scalar ANDs cost less than the
                // branches a chain of && / || compiles to -- the task body should stay one scheduling region)
                const bool fetched = (int)have_next & (int)stands(have, nblk, dep_mask(nblk, nu)) & (int)w_ready(have, nblk);
                issue_ops(cur, nblk, nu);
                if (stamp) st[4] = __builtin_amdgcn_s_memtime();
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {  // 1x1-conv k-steps r = 2 g2, 2 g2 + 1 on acc0: channels 4q + r
                    const f32x4 wv = pwv[g2];
                    const float ya = TRAIN ? acc0[2 * g2] : acc0[2 * g2] * inv, yb = TRAIN ? acc0[2 * g2 + 1] : acc0[2 * g2 + 1] * inv;
                    o0 = mfma4(wv[0], ya, o0);
                    o1 = mfma4(wv[1], ya, o1);
                    o0 = mfma4(wv[2], yb, o0);
                    o1 = mfma4(wv[3], yb, o1);
                }
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {  // ... and on acc1: channels 16 + 4q + r
                    const f32x4 wv = pwv[2 + g2];
                    const float ya = TRAIN ? acc1[2 * g2] : acc1[2 * g2] * inv, yb = TRAIN ? acc1[2 * g2 + 1] : acc1[2 * g2 + 1] * inv;
                    o0 = mfma4(wv[0], ya, o0);
                    o1 = mfma4(wv[1], ya, o1);
                    o0 = mfma4(wv[2], yb, o0);
                    o1 = mfma4(wv[3], yb, o1);
                }
                *reinterpret_cast<f32x4 *>(__builtin_assume_aligned(lds + dsti, 16)) = o0;
                *reinterpret_cast<f32x4 *>(__builtin_assume_aligned(lds + dsti + 16, 16)) = o1;
                if constexpr (TRAIN) {  // the block's output is the next block's saved input: slot blk + 1, from registers
                    if (arow >= 0) {
                        float *ap = tio.acts + ((size_t)arow + (size_t)(blk + 1) * T) * C + 4 * q;
                        *reinterpret_cast<f32x4 *>(ap) = o0;
                        *reinterpret_cast<f32x4 *>(ap + 16) = o1;
                    }
                }
                pend_u = u, pend_v = blk + 1;
                if (stamp) {
                    st[5] = __builtin_amdgcn_s_memtime();
                    if (lane == 0) {
                        unsigned long long *r = a.trace + 8 * (size_t)n;
#pragma unroll
                        for (int i = 0; i < 6; ++i) r[i] = st[i];
                        r[6] = wave, r[7] = fetched;
                    }
                }
                if (have_next && !fetched) {  // a dependency was still running: publish, wait, read -- in sequence
                    if (tracing && (a.tune & 2) && lane == 0) atomicAdd(a.trace + 4 * 3900, 1ull);
                    publish();
                    wait_for(nblk, nu);
                    issue_ops(cur, nblk, nu);
                }
                have_cur = have_next;
                n = nn, blk = nblk, u = nu;
            }
        };
        if constexpr (MODE == kSkew) {
            // the weight loads are unconditional (past the last block: the last block again) so that the number of loads in flight
            // behind a register set is the same on every path -- the compiler then waits for exactly that set (vmcnt(20 + x))
            // instead of for the youngest loads of the shortest path, i.e. for the prefetch it has just issued
            const int last = a.n_blocks - 1;
            for (int b0 = 0; b0 < a.n_blocks; b0 += 2) {
                load_block_lds(w1, Wb + (size_t)min(b0 + 1, last) * kBlockFloats, lane, q);
                while (have_cur && blk == b0) one_task(RegWeights{w0});
                if (b0 + 1 >= a.n_blocks) break;
                load_block_lds(w0, Wb + (size_t)min(b0 + 2, last) * kBlockFloats, lane, q);
                while (have_cur && blk == b0 + 1) one_task(RegWeights{w1});
            }
        } else {
            // kSkew16.  The wave that runs tile 0 of block b brings block b + 2 into the ring: slot (b + 2) % 4 last held block
            // b - 2, whose readers are the tasks of block b - 2 -- all long finished (they lie 34+ tasks back), which is checked,
            // not assumed.  The DMA travels while the wave runs its task; behind the task the wave drains it and names the block
            // in the slot's flag word.  Consumers see that word in the flag sample they take anyway (w_ready).
            while (have_cur) {
                if (u == 0 && blk + 2 < a.n_blocks) {  // (wave-uniform)
                    if (blk >= 2) {
                        // "a wave never waits while it holds an unpublished flag": this wave's previous task may be one of block
                        // blk - 2 (a slow wave takes task (blk, 0) from as far back as that), and its own tile is in the set below
                        publish();
                        for (int spins = 0;; ++spins) {
                            const int have = done[lane];
                            if (stands(have, blk - 1, all_tiles)) break;
                            if (__any(lane == 63 && have != 0)) break;
                            if (spins > a.spin_limit) {
                                if (lane == 0) done[63] = 1;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    staged = blk + 2;
                    stage_ring(staged, 0, 1);
                }
                one_task(LdsWeights{ws + (size_t)(blk & (kWeightRing - 1)) * kBlockFloats, lane, q});
            }
        }
        publish();
        xin = (a.n_blocks & 1) ? xb : xa;
        xout = (a.n_blocks & 1) ? xa : xb;
        if (tracing && blockIdx.x < 1024 && threadIdx.x == 0) {
            a.trace[4 * (2000 + blockIdx.x) + 2] = __builtin_amdgcn_s_memtime();
            a.trace[4 * (2000 + blockIdx.x) + 3] = __builtin_amdgcn_s_memrealtime();
        }
    } else if (a.wlds && MODE == kPrefetch) {
        // Two register sets: while block b runs from one, the weights of block b + 1 travel LDS -> registers into the other
        // (the 20 ds_read_b128 per wave of a block -- 150 KB of LDS reads per workgroup -- no longer stand between the
        // barrier and the first product).  LDS slot (b & 1) is refilled by LDS-DMA with block b + 2 as soon as every wave
        // holds block b in registers, i.e. right behind the barrier at the top of block b.
        BlockW w0, w1;
        stage(0);
        if (a.n_blocks > 1) stage(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        load_block_lds(w0, ws, lane, q);
        for (int blk = 0; blk < a.n_blocks; blk += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of block blk + 1 has landed
            __syncthreads();                                  // xin complete; block blk + 1 complete in its slot; slot (blk & 1) consumed
            if (blk + 2 < a.n_blocks) stage(blk + 2);
            const bool more1 = blk + 1 < a.n_blocks;
            one_block(blk, w0, WeightPrefetch{w1, ws + kBlockFloats, lane, q});
            if (!more1) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (blk + 3 < a.n_blocks) stage(blk + 3);
            one_block(blk + 1, w1, WeightPrefetch{w0, ws, lane, q});
        }
    } else {
        if (a.wlds) stage(0);
        BlockW w;
        for (int blk = 0; blk < a.n_blocks; ++blk) {
            if (a.wlds) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of block blk's weights has landed
                __syncthreads();                                  // xin complete, every wave's share has landed
                // the other slot was read during block blk - 1 and consumed before the barrier above: free to refill
                if (blk + 1 < a.n_blocks) stage(blk + 1);
                one_block(blk, w, WeightLoad{w, ws + (size_t)(blk & 1) * kBlockFloats, lane, q});
            } else {  // very long patches: no LDS left for the weight slots, every wave reads the block from L2
                __syncthreads();
                load_block_lds(w, Wb + (size_t)blk * kBlockFloats, lane, q);
                one_block(blk, w, NoPrefetch());
            }
        }
    }
    if (MODE != kSkew && MODE != kSkew16 && tracing && blockIdx.x < 1024 && threadIdx.x == 0) {
        a.trace[4 * (2000 + blockIdx.x) + 2] = __builtin_amdgcn_s_memtime();
        a.trace[4 * (2000 + blockIdx.x) + 3] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if constexpr (MODE != kSkew && MODE != kSkew16) save_acts(xin, a.n_blocks);  // pre-relu TCN output (training)
    if constexpr (TRAIN) __syncthreads();
    // final relu in place (xin = TCN output); optional tap to global as (N, T, 32) == Keras Flatten order
    for (int i = threadIdx.x; i < GR * (C / 4); i += blockDim.x) {
        const int R = i >> 3, c4 = (i & 7) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(xin + (size_t)R * SX + c4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *reinterpret_cast<f32x4 *>(xin + (size_t)R * SX + c4) = v;
        if (trunk) *reinterpret_cast<f32x4 *>(trunk + ((size_t)n0 * T + R) * C + c4) = v;
    }
    __syncthreads();

    bool gave_up = false;
    if constexpr (MODE == kSkew || MODE == kSkew16) gave_up = ((__attribute__((address_space(3))) volatile int *)(flagbase))[63] != 0;  // the heads do not touch it
    dense_and_heads<TRAIN>(a, xin, xout, WhA, hp, out, tio, n0, g_here);
    if (tracing && blockIdx.x < 256 && threadIdx.x == 0) a.trace[4 * 3000 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    if (gave_up) {  // a dependency never arrived: the outputs are not results -- say so in the model's error word
        __syncthreads();
        for (int i = threadIdx.x; i < g_here * a.out_dim; i += blockDim.x) out[(size_t)n0 * a.out_dim + i] = 0.f;
        if (threadIdx.x == 0 && a.status) atomicOr(a.status, 1);
    }
}

// gather kernel: packed[i] = map[i] ? flat[map[i] - 1] : 0
__global__ void repack_kernel(const float *__restrict__ flat, const int *__restrict__ map, float *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int m = map[i];
        dst[i] = m ? flat[m - 1] : 0.f;
    }
}

// Host packing of canonical weights `h` into the four operand buffers (used once, on an iota vector,
// to build the device gather map).
// Canonical flat order (Keras array layouts):
//   initial_conv kernel (1,F,32), bias(32); per block [conv kernel (3,32,32), bias, conv1x1 kernel (1,32,32), bias];
//   3C kernel (D,ncls), bias; per head [dense kernel (D,16), bias, gamma, beta, moving_mean, moving_var,
//   out kernel (16,odim), out bias].
static void pack_host(const smh_model *m, const float *h, std::vector<float> &W0, std::vector<float> &Wb,
                      std::vector<float> &WhA, std::vector<float> &hp) {
    const int F = m->cfg.n_feat, FQ = m->FQ, D = m->D, NH = m->NH, ncls = m->cfg.n_classes;
    W0.assign(m->nW0, 0.f), Wb.assign(m->nWb, 0.f), WhA.assign(m->nWhA, 0.f), hp.assign(m->nhp, 0.f);
    std::vector<float> Wh((size_t)D * NH, 0.f), bhv((size_t)m->n_mt * 16, 0.f);
    const float *p = h;
    if (m->cfg.block_variant == 1) {  // smh_tcn_v2.hip reads the trunk from the canonical tensor: only the heads are packed
        p += (size_t)3 * F * C + C + 3 * C * C + C + (size_t)F * C + C + (size_t)(m->n_blocks - 1) * 2 * (3 * C * C + C);
    } else {
    for (int s = 0; s < FQ; ++s)
        for (int mt = 0; mt < 2; ++mt)
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, i = lane & 15, f = q * FQ + s;
                W0[((size_t)s * 2 + mt) * 64 + lane] = f < F ? p[(size_t)f * C + 16 * mt + i] : 0.f;
            }
    p += (size_t)F * C;
    std::memcpy(&W0[(size_t)FQ * 2 * 64], p, C * sizeof(float));
    p += C;
    for (int b = 0; b < m->n_blocks; ++b) {
        float *wb = &Wb[(size_t)b * kBlockFloats];
        const float *k1 = p;  // (3, 32, 32): [tap][cin][cout]
        p += 3 * C * C;
        const float *b1 = p;
        p += C;
        const float *k2 = p;  // (1, 32, 32): [cin][cout]
        p += C * C;
        const float *b2 = p;
        p += C;
        for (int s = 0; s < 24; ++s)
            for (int mt = 0; mt < 2; ++mt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, i = lane & 15;
                    const int tap = s / 8, c = 8 * q + s % 8;  // run_block: lane group q feeds channel 8 q + s8 at step s8
                    const int n = s * 2 + mt;
                    wb[(n / 4) * 256 + 4 * lane + n % 4] = k1[((size_t)tap * C + c) * C + 16 * mt + i];
                }
        for (int e = 0; e < 8; ++e)
            for (int mt = 0; mt < 2; ++mt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, i = lane & 15;
                    const int cin = 16 * (e / 4) + 4 * q + (e % 4);
                    const int n = 48 + e * 2 + mt;
                    wb[(n / 4) * 256 + 4 * lane + n % 4] = k2[(size_t)cin * C + 16 * mt + i];
                }
        std::memcpy(wb + 24 * 2 * 64 + 8 * 2 * 64, b1, C * sizeof(float));
        std::memcpy(wb + 24 * 2 * 64 + 8 * 2 * 64 + 32, b2, C * sizeof(float));
    }
    }
    const float *k3c = p;
    p += (size_t)D * ncls;
    const float *b3c = p;
    p += ncls;
    for (int i = 0; i < D; ++i)
        for (int c = 0; c < ncls; ++c) Wh[(size_t)i * NH + c] = k3c[(size_t)i * ncls + c];
    for (int c = 0; c < ncls; ++c) bhv[c] = b3c[c];
    float *php = hp.data();
    for (int hd = 0; hd < m->n_heads; ++hd) {
        const float *kd = p;
        p += (size_t)D * kHidden;
        const float *bd = p;
        p += kHidden;
        for (int i = 0; i < D; ++i)
            for (int c = 0; c < kHidden; ++c) Wh[(size_t)i * NH + ncls + hd * kHidden + c] = kd[(size_t)i * kHidden + c];
        for (int c = 0; c < kHidden; ++c) bhv[ncls + hd * kHidden + c] = bd[c];
        const int od = m->head_odim[hd];
        const size_t cnt = 4 * kHidden + (size_t)kHidden * od + od;
        std::memcpy(php, p, cnt * sizeof(float));  // gamma beta mean var Wout bout are contiguous in canonical order
        php += cnt;
        p += cnt;
    }
    // Wh[k / 4][ld][k % 4], ld = 64 * ceil(NH / 64): four consecutive inputs k (= four channels of one frame) of output o lie
    // together, so a lane fetches them with ONE 16-byte load (a wave: 1 KiB per instruction) where the [k][ld] layout took four
    // dword loads -- the Dense phase streams 557 KB per workgroup from L2 and was bound by the number of its load instructions
    const int ld = 64 * ((NH + 63) / 64);
    for (int k = 0; k < D; ++k)
        for (int o = 0; o < NH; ++o) WhA[(size_t)(k / 4) * ld * 4 + (size_t)o * 4 + (k % 4)] = Wh[(size_t)k * NH + o];
    std::memcpy(&WhA[(size_t)D * ld], bhv.data(), bhv.size() * sizeof(float));
}

}  // namespace

// tools/trace_model.py: timestamps (s_memtime) of every (block, tile) task of workgroup 0 under the skewed schedule
static unsigned long long *g_trace = nullptr;
static unsigned g_trace_launches = 0;
constexpr size_t kTraceWords = 4 * 4096;
extern "C" int smh_internal_tcn_trace(int enable, unsigned long long *host, size_t words) {
    if (enable && !g_trace) {
        if (hipMalloc((void **)&g_trace, 2 * kTraceWords * sizeof(unsigned long long)) != hipSuccess) return -1;
        (void)hipMemset(g_trace, 0, 2 * kTraceWords * sizeof(unsigned long long));
    }
    if (host && g_trace) {
        (void)hipDeviceSynchronize();
        if (hipMemcpy(host, g_trace, std::min(words, 2 * kTraceWords) * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    if (!enable && g_trace) {
        (void)hipFree(g_trace);
        g_trace = nullptr;
    }
    return 0;
}

namespace smh_tcn {

Offsets offsets(const smh_model *m) {
    Offsets o;
    const size_t F = m->cfg.n_feat, D = m->D, ncls = m->cfg.n_classes;
    size_t p = 0;
    o.w0_k = p, p += F * C;
    o.w0_b = p, p += C;
    o.blk0 = p, o.blk_stride = 3 * C * C + C + C * C + C;
    p += (size_t)m->n_blocks * o.blk_stride;
    o.c3_k = p, p += D * ncls;
    o.c3_b = p, p += ncls;
    for (int h = 0; h < m->n_heads; ++h) {
        o.head[h] = p;
        p += D * kHidden + kHidden + 4 * kHidden + (size_t)kHidden * m->head_odim[h] + m->head_odim[h];
    }
    return o;
}

int repack(smh_model *m, hipStream_t st) {
    m->version++;  // every change of the master weights passes through here
    const size_t n = m->nW0 + m->nWb + m->nWhA + m->nhp;
    // the four operand buffers are one allocation: d_W0 is its base
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->d_flat, m->d_map, m->d_W0, n);
    return smh::launch_status("repack_kernel");
}

void fill_args(const smh_model *m, int N, TcnArgs *pa, size_t *plds) {
    TcnArgs &a = *pa;
    const int T = m->cfg.patch_size;
    a.N = N, a.T = T, a.F = m->cfg.n_feat, a.FQ = m->FQ, a.n_blocks = m->n_blocks, a.n_dil = m->cfg.n_dilations;
    a.vec_ok = (a.F % 4 == 0) && (a.FQ % 4 == 0) && (a.FQ * 4 == a.F);
    a.skip_heads = 0;
    a.tune = 0;
    a.trace = nullptr;
    a.from_x0 = 0, a.x0_shift = 0, a.x0_T = 0;
    a.status = m->d_status, a.spin_limit = kSkewSpinLimit;
    a.D = m->D, a.NH = m->NH, a.n_mt = m->n_mt, a.n_classes = m->cfg.n_classes, a.n_heads = m->n_heads;
    a.out_dim = m->out_dim;
    for (int i = 0; i < kMaxHeads; ++i) a.head_odim[i] = m->head_odim[i], a.head_sigmoid[i] = m->head_sigmoid[i];
    // patches per workgroup: up to 272 rows (17 column tiles) of LDS-resident activations, at most one
    // MFMA tile of patches, and never fewer workgroups than CUs when the batch allows it
    int gmax = 272 / T;
    if (gmax < 1) gmax = 1;
    if (gmax > kMaxG) gmax = kMaxG;
    // ceil(N / 256): as few workgroups as there are CUs when the batch allows it -- 510 patches as 255 workgroups of two run one
    // round of the chip; as 510 workgroups of one (floor) they ran two, each with a quarter-empty fifth tile
    int G = (N + 255) / 256;
    if (G < 1) G = 1;
    if (G > gmax) G = gmax;
    if (const char *ev = getenv("SMH_TCN_G")) {  // tuning only (tools/tune_model.py)
        const int g = atoi(ev);
        if (g >= 1 && g <= gmax) G = g;
    }
    a.G = G;
    int GRP = ((G * T + 15) / 16) * 16;
    const int head_scratch = 8 * 4 * 128 + kMaxG * kPS;  // Dense partial sums (8 parts x 4 patches x <= 128) + pre[kMaxG][kPS]
    if (GRP * SX < head_scratch) GRP = (head_scratch + SX - 1) / SX;  // the head scratch lives in one buffer
    if (GRP * SX < m->FQ * 2 * 64) GRP = (m->FQ * 2 * 64 + SX - 1) / SX;  // layer-0 A operands are staged there
    GRP = ((GRP + 15) / 16) * 16;
    a.GRP = GRP;
    const size_t lds_x = sizeof(float) * 2 * (size_t)(GRP + 1) * SX, lds_w = sizeof(float) * 2 * (size_t)kBlockFloats;  // + the zero rows
    const size_t lds_xch = sizeof(float) * (2 * 64 * 4 + 4);  // the exchange area of split_last (half_tile_compute)
    a.wlds = lds_x + lds_w + lds_xch <= 156 * 1024 ? 1 : 0;  // activations (x2) + two weight slots, when they fit
    a.split_last = 0;
    *plds = lds_x + (a.wlds ? lds_w + lds_xch : 0);
}

int launch_forward(const smh_model *m, const float *d_x, int N, float *d_out, float *d_trunk, const TrainIO *tio,
                   hipStream_t st, int from_x0, int x0_shift, int x0_T) {
    TcnArgs a;
    size_t lds;
    fill_args(m, N, &a, &lds);
    a.from_x0 = from_x0, a.x0_shift = x0_shift, a.x0_T = x0_T;
    // timing probes (outputs invalid; training would read stale activations): only under SMH_ENABLE_PROBES=1, announced on stderr
    if (const char *ev = smh::probe_env("SMH_TCN_BLOCKS")) a.n_blocks = atoi(ev);  // tools/tune_model.py
    a.skip_heads = smh::probe_env("SMH_TCN_NOHEADS") ? 1 : 0;
    if (const char *ev = smh::probe_env("SMH_TCN_TUNE")) a.tune = atoi(ev);
    if (a.tune & 256) a.spin_limit = 1 << 10;  // test_skew_give_up_is_reported: a withheld flag must end in the error word quickly
    a.trace = g_trace ? g_trace + (g_trace_launches++ & 1) * kTraceWords : nullptr;  // consecutive launches alternate halves
    SMH_REQUIRE(lds <= 156 * 1024, "patch_size %d too long for the LDS-resident TCN", a.T);
    // waves per workgroup: the block time is (column tiles of the busiest wave) x (time per tile) plus a fixed part, so
    // take the fewest waves in 8..12 that minimise ceil(tiles / waves): 17 tiles (4 patches of 68 frames) -> 9 waves,
    // 2 tiles each and one with 1, instead of 8 waves of which one does 3
    const int units = (std::min(a.G, N) * a.T + 15) / 16;
    // (only without the LDS weight slots: the 9..12-wave build is held to 170 VGPRs and spills ~60 of them, which costs more
    // than the shorter tile list saves -- training forward, 9 tiles: 134 us with 9 waves, 121 us with 8 and two register sets)
    int nwaves = 8;
    for (int w = 9; w <= 12 && !a.wlds; ++w)
        if ((units + w - 1) / w < (units + nwaves - 1) / nwaves) nwaves = w;
    if (const char *ev = getenv("SMH_TCN_WAVES")) nwaves = std::max(4, std::min(12, atoi(ev)));  // tuning only
    // 8 waves (two per SIMD, 256 VGPRs each): two weight register sets, the next block's weights are read behind this
    // block's products.  More waves (170 VGPRs): one set, read at the top of the block.
    bool prefetch = nwaves <= 8 && a.wlds;
    if (const char *ev = getenv("SMH_TCN_PREFETCH")) prefetch = prefetch && atoi(ev) != 0;  // tuning only
    // the skewed task schedule (8 waves, flags instead of barriers) whenever its tables fit -- inference and training forward
    // It pays where the barrier schedule leaves wave slots empty AND there are enough tiles for the windows to overlap
    // (tools/time_model_sizes.py, W = 68: 17 tiles 149 against 160 us, 13 tiles 130 / 134; 9 tiles 107 / 105.5; 5 tiles -- up to 256
    // patches, one per workgroup -- 97 / 77: fewer tiles than waves, every task waits on the block before; W = 249, 16 tiles = two
    // full rounds of the 8 waves: 162 / 154).  SMH_TCN_SKEW=0 / 2: never / whenever it can run (tests, tuning).
    // (the schedule decodes task n into (block, tile) by multiplication, exact for n < 2048: smh_model_create accepts up to
    // nb_stacks x 16 dilations, the reference tunes nb_stacks up to 10 -- beyond the bound the barrier schedule runs)
    const bool skew_ok = a.wlds && units <= 32 && units >= 1 && a.n_blocks * units < 2048 && a.T >= 16;  // (T >= 16: issue_ops)
    // (13 tiles: since the lone last-round tile is shared by two waves the barrier schedule is ahead there -- W = 68 x 768 patches
    // 130.1 against 131.5 us, W = 99 x 510 131.5 / 134.7; at 17 tiles the skew schedule stays ahead, 149.8 / 157.5)
    bool skew = skew_ok && units >= 12 && 8 * ((units + 7) / 8) - units >= 3 && units != 13;
    if (const char *ev = getenv("SMH_TCN_SKEW")) skew = atoi(ev) == 2 ? skew_ok : (skew && atoi(ev) != 0);
    if (skew && !getenv("SMH_TCN_WAVES")) nwaves = 8;
    if (skew) nwaves = std::min(nwaves, 8);
    // The 16-wave form of the skew schedule (weights in an LDS ring, four waves per SIMD): inference, when its ring fits beside
    // the activations.  LAB BUILDS ONLY (-DSMH_LAB, then SMH_TCN_SKEW16=1; the production library does not instantiate it): measured on the bench shape it runs 135.7-136.2 us against 133.6-134.2 us for the
    // 8-wave form (tools/gpu/r3_net.sh) -- twice the waves per SIMD buy nothing, i.e. the loop is not short of waves to hide
    // latency behind: exact-f32 MFMA and the VALU work of the epilogues do not overlap.  Kept as the measured
    // experiment and as a third implementation the schedule-agreement test holds bit-identical to the other two.
    const size_t lds16 = sizeof(float) * (2 * (size_t)(a.GRP + 1) * SX + (size_t)kWeightRing * kBlockFloats + 128);  // + flags and scratch words
    bool skew16 = false;
    if (const char *ev = smh::lab_env("SMH_TCN_SKEW16")) skew16 = atoi(ev) != 0 && skew && !tio && !a.trace && lds16 <= 156 * 1024;
    if (skew16) nwaves = 16, lds = lds16;
    // barrier schedule with two register sets and ONE tile in its last round (5, 9, ... tiles on 8 waves): that tile as two halves on
    // two waves of different SIMDs (half_tile_compute).  SMH_TCN_SPLIT=0 switches it off (tests: the two forms agree bit for bit).
    a.split_last = (!skew && prefetch && nwaves == 8 && units >= 2 && (units % 8 == 1 || units % 8 == 5)) ? 1 : 0;
    if (const char *ev = getenv("SMH_TCN_SPLIT")) a.split_last = a.split_last && atoi(ev) != 0;
    const dim3 grid((N + a.G - 1) / a.G), block(64 * nwaves);
    TrainIO io{nullptr, nullptr, nullptr, nullptr};
    if (tio) io = *tio;
#define SMH_LAUNCH_FWD(TR, MD, TC)                                                                                      \
    do {                                                                                                                \
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)b3mtl_forward_kernel<TR, MD, TC>,                              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
        hipLaunchKernelGGL((b3mtl_forward_kernel<TR, MD, TC>), grid, block, lds, st, a, d_x, m->d_W0, m->d_Wb,         \
                           m->d_WhA, m->d_hp, d_trunk, d_out, io);                                                     \
    } while (0)
    if (tio) {
        if (skew) SMH_LAUNCH_FWD(true, kSkew, false);
        else if (prefetch) SMH_LAUNCH_FWD(true, kPrefetch, false);
        else SMH_LAUNCH_FWD(true, kOneSet, false);
    } else if (a.trace) {  // tools/trace_model.py: the stamped instantiations
        if (skew) SMH_LAUNCH_FWD(false, kSkew, true);
        else if (prefetch) SMH_LAUNCH_FWD(false, kPrefetch, true);
        else SMH_LAUNCH_FWD(false, kOneSet, true);
#ifdef SMH_LAB
    } else if (skew16) {
        SMH_LAUNCH_FWD(false, kSkew16, false);
#endif
    } else if (skew) {
        SMH_LAUNCH_FWD(false, kSkew, false);
    } else {
        if (prefetch) SMH_LAUNCH_FWD(false, kPrefetch, false);
        else SMH_LAUNCH_FWD(false, kOneSet, false);
    }
#undef SMH_LAUNCH_FWD
    return smh::launch_status("b3mtl_forward_kernel");
}

}  // namespace smh_tcn

extern "C" int smh_model_create(const smh_model_cfg *cfg, smh_model **out) {
    SMH_REQUIRE(cfg && out, "smh_model_create: null argument");
    SMH_REQUIRE(cfg->nb_filters == C, "B3_MTL kernel is tiled for nb_filters=32 (got %d)", cfg->nb_filters);
    SMH_REQUIRE(cfg->kernel_size == 3, "B3_MTL kernel supports kernel_size=3 (got %d)", cfg->kernel_size);
    SMH_REQUIRE(cfg->n_classes == 3 || cfg->n_classes == 5, "n_classes must be 3 or 5 (got %d)", cfg->n_classes);
    SMH_REQUIRE(cfg->n_feat >= 1 && cfg->patch_size >= 1 && cfg->patch_size <= 512, "bad n_feat/patch_size");
    SMH_REQUIRE(cfg->nb_stacks >= 1 && cfg->n_dilations >= 1 && cfg->n_dilations <= 16, "bad stacks/dilations");
    SMH_REQUIRE(smh_device_count() > 0, "no HIP device visible: libsmh has no CPU path");
    smh_model *m = new smh_model();
    m->cfg = *cfg;
    m->n_blocks = cfg->nb_stacks * cfg->n_dilations;
    if (cfg->n_classes == 5) {  // 5_class_classification.py:150-215: S, M, N, R(3)
        m->n_heads = 4;
        const int od[4] = {1, 1, 1, 3}, sg[4] = {1, 1, 1, 0};
        for (int i = 0; i < 4; ++i) m->head_odim[i] = od[i], m->head_sigmoid[i] = sg[i];
    } else {  // proposed_architectures.py:25-80: S, M, R(2)
        m->n_heads = 3;
        const int od[3] = {1, 1, 2}, sg[3] = {1, 1, 0};
        for (int i = 0; i < 3; ++i) m->head_odim[i] = od[i], m->head_sigmoid[i] = sg[i];
    }
    m->D = cfg->patch_size * C;
    m->NH = cfg->n_classes + kHidden * m->n_heads;
    m->n_mt = (m->NH + 15) / 16;
    m->out_dim = cfg->n_classes;
    for (int i = 0; i < m->n_heads; ++i) m->out_dim += m->head_odim[i];
    m->FQ = (cfg->n_feat + 3) / 4;
    SMH_REQUIRE(cfg->block_variant == 0 || cfg->block_variant == 1, "block_variant must be 0 (keras-tcn 2.3.x) or 1 (>= 2.8)");
    size_t n = (size_t)cfg->n_feat * C + C;
    n += (size_t)m->n_blocks * (3 * C * C + C + C * C + C);
    if (cfg->block_variant == 1)
        n = (size_t)3 * cfg->n_feat * C + C + 3 * C * C + C + (size_t)cfg->n_feat * C + C + (size_t)(m->n_blocks - 1) * 2 * (3 * C * C + C);
    n += (size_t)m->D * cfg->n_classes + cfg->n_classes;
    for (int i = 0; i < m->n_heads; ++i)
        n += (size_t)m->D * kHidden + kHidden + 4 * kHidden + (size_t)kHidden * m->head_odim[i] + m->head_odim[i];
    m->n_params = n;
    SMH_REQUIRE(n < (1u << 24), "model too large for the float-encoded gather map");
    m->nW0 = (size_t)m->FQ * 2 * 64 + 32;
    m->nWb = (size_t)m->n_blocks * kBlockFloats;
    if (cfg->block_variant == 1) m->nW0 = m->nWb = 0;
    m->nWhA = (size_t)m->D * 64 * ((m->NH + 63) / 64) + (size_t)m->n_mt * 16;
    m->nhp = 0;
    for (int i = 0; i < m->n_heads; ++i) m->nhp += 4 * kHidden + (size_t)kHidden * m->head_odim[i] + m->head_odim[i];
    // gather map = the host packing applied to 1, 2, 3, ... (0 marks padding)
    std::vector<float> iota(n), W0, Wb, WhA, hp;
    for (size_t i = 0; i < n; ++i) iota[i] = (float)(i + 1);
    pack_host(m, iota.data(), W0, Wb, WhA, hp);
    std::vector<int> map;
    map.reserve(m->nW0 + m->nWb + m->nWhA + m->nhp);
    for (auto *v : {&W0, &Wb, &WhA, &hp})
        for (float f : *v) map.push_back((int)f);
    const size_t npk = map.size();
    hipError_t e = hipMalloc((void **)&m->d_flat, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_W0, npk * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_map, npk * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(m->d_map, map.data(), npk * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(m->d_flat, 0, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_status, sizeof(int));
    if (e == hipSuccess) e = hipMemset(m->d_status, 0, sizeof(int));
    if (e != hipSuccess) {
        smh_model_destroy(m);
        return smh::set_error(SMH_E_HIP, "smh_model_create: device allocation failed: %s", hipGetErrorString(e));
    }
    m->d_Wb = m->d_W0 + m->nW0;
    m->d_WhA = m->d_Wb + m->nWb;
    m->d_hp = m->d_WhA + m->nWhA;
    *out = m;
    return SMH_OK;
}

extern "C" void smh_model_destroy(smh_model *m) {
    if (!m) return;
    (void)hipFree(m->d_flat);
    (void)hipFree(m->d_W0);  // base of the four operand buffers
    (void)hipFree(m->d_map);
    (void)hipFree(m->d_bf16);
    (void)hipFree(m->d_status);
    delete m;
}

extern "C" int smh_model_status(smh_model *m, void *stream) {
    SMH_REQUIRE(m, "smh_model_status: null argument");
    hipStream_t st = (hipStream_t)stream;
    int word = 0;
    SMH_CHECK_HIP(hipMemcpyAsync(&word, m->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
    SMH_CHECK_HIP(hipStreamSynchronize(st));
    if (word == 0) return SMH_OK;
    SMH_CHECK_HIP(hipMemsetAsync(m->d_status, 0, sizeof(int), st));
    return smh::set_error(SMH_E_DEVICE, "B3_MTL forward: device error word 0x%x (bit 0: a wave gave up waiting for a tile flag of the "
                          "skewed block schedule or for the partner half of a split tile; the affected outputs are not results)", word);
}

extern "C" size_t smh_model_num_params(const smh_model *m) { return m ? m->n_params : 0; }
extern "C" int smh_model_out_dim(const smh_model *m) { return m ? m->out_dim : SMH_E_INVALID; }

extern "C" int smh_model_set_weights(smh_model *m, const float *h, size_t n, void *stream) {
    SMH_REQUIRE(m && h, "smh_model_set_weights: null argument");
    SMH_REQUIRE(n == m->n_params, "smh_model_set_weights: got %zu floats, model has %zu", n, m->n_params);
    hipStream_t st = (hipStream_t)stream;
    SMH_CHECK_HIP(hipMemcpyAsync(m->d_flat, h, n * sizeof(float), hipMemcpyHostToDevice, st));
    int rc = smh_tcn::repack(m, st);
    if (rc) return rc;
    SMH_CHECK_HIP(hipStreamSynchronize(st));  // the caller's host buffer may die after return
    return SMH_OK;
}

extern "C" int smh_model_get_weights(const smh_model *m, float *h, size_t n, void *stream) {
    SMH_REQUIRE(m && h, "smh_model_get_weights: null argument");
    SMH_REQUIRE(n == m->n_params, "smh_model_get_weights: got room for %zu floats, model has %zu", n, m->n_params);
    hipStream_t st = (hipStream_t)stream;
    SMH_CHECK_HIP(hipMemcpyAsync(h, m->d_flat, n * sizeof(float), hipMemcpyDeviceToHost, st));
    SMH_CHECK_HIP(hipStreamSynchronize(st));
    return SMH_OK;
}

extern "C" const float *smh_model_w0_ptr(const smh_model *m) {
    return (m && m->cfg.block_variant == 0) ? m->d_flat + smh_tcn::offsets(m).w0_k : nullptr;
}

// ---- dense file-level inference (DAFx12_Speech_Music_Detection_B3_MTL_v2.py:634-665): every hop-`shift` patch of a
// standardised featuregram chunk.  The network's first layer is a 1x1 convolution -- pointwise in time -- and the chunk is standardised
// as a whole, so a frame has the same layer-0 output in every patch that contains it: l0_frames_kernel computes it ONCE per frame (as
// the two per-half partial sums the forward's from_x0 entry adds up) and the forward kernel reads each patch as a window of that
// (2, Tc, 32) array.  Per 10 000-frame chunk at hop 1 that replaces 648 MB of materialised (9 932, 68, 240) patches, read back by
// layer 0, by 2.6 MB.
namespace {
// one wave per (16-frame tile, half): D[c][t] = sum_r W0[half * rows + r][c] * fv[half * rows + r][t], both 16-channel M-tiles;
// k order and the two accumulator chains per M-tile as in the feature kernel's layer-0 phase (smh_feat.hip)
__global__ void __launch_bounds__(256)
l0_frames_kernel(const float *__restrict__ fv, const float *__restrict__ w0, float *__restrict__ x0, int rows, int Tc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const int tile = blockIdx.x * 4 + wave, half = blockIdx.y;
    if (tile * 16 >= Tc) return;
    const int t = min(tile * 16 + j, Tc - 1);
    const float *xr = fv + ((size_t)half * rows + q) * Tc + t;
    const float *wr = w0 + ((size_t)half * rows + q) * 32 + j;
    f32x4 c0a = {0.f, 0.f, 0.f, 0.f}, c0b = c0a, c1a = c0a, c1b = c0a;
    const int nst = rows >> 2;
    for (int s0 = 0; s0 < nst; s0 += 2) {
        const int s1 = min(s0 + 1, nst - 1);
        const float xa = xr[(size_t)4 * s0 * Tc], xb = s0 + 1 < nst ? xr[(size_t)4 * s1 * Tc] : 0.f;
        const float wa0 = wr[4 * s0 * 32], wa1 = wr[4 * s0 * 32 + 16], wb0 = wr[4 * s1 * 32], wb1 = wr[4 * s1 * 32 + 16];
        c0a = mfma4(wa0, xa, c0a), c1a = mfma4(wa1, xa, c1a);
        c0b = mfma4(wb0, xb, c0b), c1b = mfma4(wb1, xb, c1b);
    }
    c0a += c0b, c1a += c1b;
    if (tile * 16 + j < Tc) {
        float *o = x0 + (((size_t)half * Tc + tile * 16 + j) * 32 + 4 * q);
        *reinterpret_cast<f32x4 *>(o) = c0a;
        *reinterpret_cast<f32x4 *>(o + 16) = c1a;
    }
}
}  // namespace

extern "C" size_t smh_model_dense_workspace_bytes(const smh_model *m, int Tc) {
    return (m && Tc > 0) ? sizeof(float) * 2 * (size_t)Tc * 32 : 0;
}

extern "C" int smh_model_forward_dense_f32(const smh_model *m, const float *d_fv, int Tc, int shift, void *d_work, size_t work_bytes,
                                           float *d_out, void *stream) {
    SMH_REQUIRE(m && d_fv && d_work && d_out, "smh_model_forward_dense_f32: null argument");
    SMH_REQUIRE(m->cfg.block_variant == 0, "smh_model_forward_dense_f32: exists for block_variant 0 only");
    SMH_REQUIRE((reinterpret_cast<uintptr_t>(d_work) % 16) == 0 && (reinterpret_cast<uintptr_t>(d_fv) % 16) == 0 &&
                    (reinterpret_cast<uintptr_t>(d_out) % 4) == 0,
                "smh_model_forward_dense_f32: d_fv and d_work are read / written with 16-byte accesses and must start on 16-byte boundaries");
    const int W = m->cfg.patch_size, F = m->cfg.n_feat;
    SMH_REQUIRE(F % 8 == 0, "smh_model_forward_dense_f32: n_feat=%d must be a multiple of 8 (two halves of whole k steps)", F);
    SMH_REQUIRE(shift >= 1 && Tc >= W, "smh_model_forward_dense_f32: needs shift >= 1 and at least patch_size=%d frames (Tc=%d, shift=%d); "
                "shorter chunks are tiled by get_feature_patches and take smh_model_forward_f32", W, Tc, shift);
    SMH_REQUIRE(work_bytes >= smh_model_dense_workspace_bytes(m, Tc), "smh_model_forward_dense_f32: workspace of %zu bytes, need %zu",
                work_bytes, smh_model_dense_workspace_bytes(m, Tc));
    const int nP = smh_num_patches(Tc, W, shift);
    if (nP <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float *x0 = static_cast<float *>(d_work);
    hipLaunchKernelGGL(l0_frames_kernel, dim3((unsigned)((Tc + 63) / 64), 2), dim3(256), 0, st, d_fv, smh_model_w0_ptr(m), x0, F / 2, Tc);
    int rc = smh::launch_status("l0_frames_kernel");
    if (rc) return rc;
    rc = smh_tcn::launch_forward(m, x0, nP, d_out, nullptr, nullptr, st, 1, shift, Tc);
    return rc ? rc : nP;
}

// ---- model.evaluate's per-batch arithmetic on the device (Keras inference-mode losses of Proposed_Work_Results.py:160-165, 683-687) ----
// One 256-thread workgroup per batch: for every output the batch-mean loss -- binary cross-entropy on the sigmoid heads and categorical
// cross-entropy on '3C' with Keras' 1e-7 clipping, mean squared error on 'R' -- and the '3C' accuracy, in float64 like the host
// arithmetic it replaces (training.py: _losses_inference), ADDED (times `weight`) to d_sums: the caller reads the sums
// back once per evaluate call instead of synchronising behind every batch.
namespace {
struct EvalArgs {
    int N, n_heads, n_classes, out_dim;
    int head_odim[kMaxHeads], head_sigmoid[kMaxHeads];
    double weight, l2, lw[kMaxHeads + 1];
};
__global__ void __launch_bounds__(256) eval_losses_kernel(EvalArgs a, const float *__restrict__ out, const float *__restrict__ tgt,
                                                          double *__restrict__ sums) {
    __shared__ double red[256];
    __shared__ double mean[kMaxHeads + 2];
    const int tid = threadIdx.x;
    const double eps = 1e-7;
    int c3 = 0;
    for (int h = 0; h < a.n_heads; ++h) c3 += a.head_odim[h];
    int col = 0;
    for (int slot = 0; slot < a.n_heads + 2; ++slot) {
        double acc = 0.0;
        const int od = slot < a.n_heads ? a.head_odim[slot] : a.n_classes;
        const int c0 = slot < a.n_heads ? col : c3;
        for (int r = tid; r < a.N; r += 256) {
            const float *o = out + (size_t)r * a.out_dim + c0, *t = tgt + (size_t)r * a.out_dim + c0;
            if (slot < a.n_heads) {
                for (int c = 0; c < od; ++c) {
                    const double ov = (double)o[c], tv = (double)t[c];
                    if (a.head_sigmoid[slot]) {
                        const double oc = fmin(fmax(ov, eps), 1.0 - eps);
                        acc -= tv * log(oc + eps) + (1.0 - tv) * log(1.0 - oc + eps);
                    } else {
                        acc += (ov - tv) * (ov - tv);
                    }
                }
            } else if (slot == a.n_heads) {
                double sum = 0.0;
                for (int c = 0; c < od; ++c) sum += (double)o[c];
                for (int c = 0; c < od; ++c) acc -= (double)t[c] * log(fmin(fmax((double)o[c] / sum, eps), 1.0 - eps));
            } else {
                int bo = 0, bt = 0;  // first maximum, like numpy's argmax
                for (int c = 1; c < od; ++c) {
                    if (o[c] > o[bo]) bo = c;
                    if (t[c] > t[bt]) bt = c;
                }
                acc += bo == bt ? 1.0 : 0.0;
            }
        }
        red[tid] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) mean[slot] = red[0] / (slot < a.n_heads ? (double)a.N * od : (double)a.N);
        __syncthreads();
        if (slot < a.n_heads) col += od;
    }
    if (tid == 0) {
        // the batch's values first, then their weight -- the order of the host loop this replaces (training.py: _losses_inference,
        // evaluate), so that a one-rank data-parallel pass (row-weighted) and a plain one agree to the bit as they did there
        double total = 0.0;
        for (int i = 0; i <= a.n_heads; ++i) total += a.lw[i] * mean[i];
        total += a.l2;
        sums[0] += a.weight * total;
        for (int i = 0; i < a.n_heads + 2; ++i) sums[1 + i] += a.weight * mean[i];
    }
}
}  // namespace

extern "C" int smh_model_eval_losses_f32(const smh_model *m, const float *d_out, const float *d_targets, int N, double weight,
                                         const double *h_loss_weights, double l2_penalty, double *d_sums, void *stream) {
    SMH_REQUIRE(m && d_out && d_targets && d_sums && h_loss_weights, "smh_model_eval_losses_f32: null argument");
    SMH_REQUIRE(N >= 1, "smh_model_eval_losses_f32: N=%d", N);
    EvalArgs a;
    a.N = N, a.n_heads = m->n_heads, a.n_classes = m->cfg.n_classes, a.out_dim = m->out_dim, a.weight = weight, a.l2 = l2_penalty;
    for (int i = 0; i < kMaxHeads; ++i) a.head_odim[i] = m->head_odim[i], a.head_sigmoid[i] = m->head_sigmoid[i];
    for (int i = 0; i <= kMaxHeads; ++i) a.lw[i] = i <= m->n_heads ? h_loss_weights[i] : 0.0;
    hipLaunchKernelGGL(eval_losses_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, d_out, d_targets, d_sums);
    return smh::launch_status("eval_losses_kernel");
}

extern "C" int smh_model_forward_x0_f32(const smh_model *m, const float *d_x0p, int N, float *d_out, float *d_trunk,
                                        void *stream) {
    SMH_REQUIRE(m && d_x0p && d_out, "smh_model_forward_x0_f32: null argument");
    SMH_REQUIRE(m->cfg.block_variant == 0, "smh_model_forward_x0_f32: the layer-0 fusion exists for block_variant 0 only");
    SMH_REQUIRE(N >= 0, "smh_model_forward_x0_f32: N=%d", N);
    if (N == 0) return SMH_OK;
    return smh_tcn::launch_forward(m, d_x0p, N, d_out, d_trunk, nullptr, (hipStream_t)stream, 1);
}

extern "C" int smh_model_forward_f32(const smh_model *m, const float *d_x, int N, float *d_out, float *d_trunk,
                                     void *stream) {
    SMH_REQUIRE(m && d_x && d_out, "smh_model_forward_f32: null argument");
    SMH_REQUIRE(N >= 0, "smh_model_forward_f32: N=%d", N);
    if (N == 0) return SMH_OK;
    if (m->cfg.block_variant == 1) return smh_tcn::launch_forward_v2(m, d_x, N, d_out, d_trunk, (hipStream_t)stream);
    return smh_tcn::launch_forward(m, d_x, N, d_out, d_trunk, nullptr, (hipStream_t)stream);
}
