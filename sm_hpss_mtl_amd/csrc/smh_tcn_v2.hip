// B3_MTL forward with the LATER keras-tcn residual block (2.8 / 3.x; smh_model_cfg.block_variant = 1):
//   per (stack, dilation d):  y = relu(Conv1D(32, 3, dilation d, 'same')(x));  y = relu(Conv1D(32, 3, dilation d, 'same')(y));
//                             x = relu(shortcut(x) + y),  shortcut = identity, or Conv1D(32, 1) when the channel counts differ
//   (first block: n_feat -> 32); no initial 1x1 convolution, no activation behind the last block.
// The reference does not pin keras-tcn (lib/proposed_architectures.py:124-125 `from tcn import TCN`); its positional call
// binds under the 2.3.x API (smh_tcn.hip, the default and the measured path).  This variant exists so that a model trained
// with the other block can be served; restated in oracle/b3_mtl.py (tcn_forward_v2), inference only.
//
// gfx950 mapping: the same transposed exact-f32 MFMA products as smh_tcn.hip (D[channel][time], v_mfma_f32_16x16x4_f32),
// activations of G patches resident in LDS.  A block is two phases separated by a barrier: conv0 reads x (neighbour rows
// at +-d) and writes y; conv1 reads y (neighbour rows) and x's OWN rows (shortcut) and overwrites those rows of x.
// Weights are read from the canonical tensor (Keras layout) with the A-operand gather built into the addressing: 48
// operand registers per convolution, loaded once per block and wave.  The first block streams its 240-channel input rows
// from HBM (three taps + the matching convolution).  Dense layers and heads: smh_tcn_heads.h.
#include <algorithm>
#include <cstdlib>

#include "smh_model.h"
#include "smh_tcn_heads.h"

using namespace smh_tcn;

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct ConvW {
    float w[24][2];  // [tap*8 + s8][M-tile]: A operand of step s8 of a tap = W[tap][c = 8 q + s8][16 mt + i]
    f32x4 blo, bhi;
};

// canonical kernel (3, 32, 32) [tap][cin][cout] + bias (32) -> this lane's operand registers
__device__ __forceinline__ void load_conv(ConvW &cw, const float *__restrict__ k, const float *__restrict__ b, int q, int i) {
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const float *p = k + ((size_t)tap * C + 8 * q + s8) * C + i;
            cw.w[tap * 8 + s8][0] = p[0];
            cw.w[tap * 8 + s8][1] = p[16];
        }
    cw.blo = *reinterpret_cast<const f32x4 *>(b + 4 * q);
    cw.bhi = *reinterpret_cast<const f32x4 *>(b + 16 + 4 * q);
}

// one dilated convolution of a 16-row tile from an LDS buffer: acc[channel 4q + r (+16)][row j]
__device__ __forceinline__ void conv_tile(const ConvW &cw, const float *src, int Rc, int t, int d, int T, int ZR, int q,
                                          f32x4 &acc0, f32x4 &acc1) {
    acc0 = cw.blo, acc1 = cw.bhi;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int off = (tap - 1) * d;
        const bool ok = (tap == 1) || ((t + off >= 0) && (t + off < T));
        if (tap != 1 && !__any(ok)) continue;
        const float *row = src + (size_t)(ok ? Rc + off : ZR) * SX + 8 * q;
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(row), b1 = *reinterpret_cast<const f32x4 *>(row + 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = mfma4(cw.w[tap * 8 + s][0], b0[s], acc0);
            acc1 = mfma4(cw.w[tap * 8 + s][1], b0[s], acc1);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = mfma4(cw.w[tap * 8 + 4 + s][0], b1[s], acc0);
            acc1 = mfma4(cw.w[tap * 8 + 4 + s][1], b1[s], acc1);
        }
    }
}

__global__ void __launch_bounds__(512)
b3mtl_forward_v2_kernel(TcnArgs a, const float *__restrict__ X, const float *__restrict__ flat, const float *__restrict__ WhA,
                        const float *__restrict__ hp, float *__restrict__ trunk, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int n0 = blockIdx.x * a.G;
    const int g_here = min(a.G, a.N - n0);
    const int T = a.T, F = a.F;
    const int GR = g_here * T;
    const int units = (GR + 15) >> 4;
    const int ZR = a.GRP;
    float *xa = lds, *ya = lds + (size_t)(a.GRP + 1) * SX;
    if (threadIdx.x < SX) xa[(size_t)ZR * SX + threadIdx.x] = 0.f, ya[(size_t)ZR * SX + threadIdx.x] = 0.f;

    // canonical offsets: block 0 = [conv0 (3,F,32), b, conv1 (3,32,32), b, matching (1,F,32), b]; later blocks = [conv0, b, conv1, b]
    const size_t blk0_floats = (size_t)3 * F * C + C + 3 * C * C + C + (size_t)F * C + C;
    const size_t blk_floats = 2 * (3 * C * C + C);

    // ---- block 0, phase A: y = relu(conv0(x)) straight from the HBM patches; K order per tap: channel q*FQ + s ----
    for (int u = wave; u < units; u += nw) {
        const int R = 16 * u + j, Rc = min(R, GR - 1);
        const int g = Rc / T, t = Rc - g * T;
        f32x4 acc0 = *reinterpret_cast<const f32x4 *>(flat + (size_t)3 * F * C + 4 * q);
        f32x4 acc1 = *reinterpret_cast<const f32x4 *>(flat + (size_t)3 * F * C + 16 + 4 * q);
        for (int tap = 0; tap < 3; ++tap) {
            const int off = tap - 1;  // dilation 1
            const bool ok = (t + off >= 0) && (t + off < T);
            const float *xr = X + ((size_t)(n0 + g) * T + (ok ? t + off : t)) * F + (size_t)q * a.FQ;
            const float *wk = flat + ((size_t)tap * F + (size_t)q * a.FQ) * C + j;
            for (int s = 0; s < a.FQ; ++s) {
                const bool live = ok && (q * a.FQ + s < F);
                const float xv = live ? xr[s] : 0.f;
                const float w0v = (q * a.FQ + s < F) ? wk[(size_t)s * C] : 0.f, w1v = (q * a.FQ + s < F) ? wk[(size_t)s * C + 16] : 0.f;
                acc0 = mfma4(w0v, xv, acc0);
                acc1 = mfma4(w1v, xv, acc1);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0[r] = fmaxf(acc0[r], 0.f), acc1[r] = fmaxf(acc1[r], 0.f);
        float *dst = ya + (size_t)R * SX + 4 * q;
        *reinterpret_cast<f32x4 *>(dst) = acc0;
        *reinterpret_cast<f32x4 *>(dst + 16) = acc1;
    }
    __syncthreads();
    // ---- block 0, phase B: x = relu(matching(x) + relu(conv1(y))) ----
    {
        ConvW cw;
        const float *k1 = flat + (size_t)3 * F * C + C;
        load_conv(cw, k1, k1 + 3 * C * C, q, j);
        const float *km = k1 + 3 * C * C + C;  // matching kernel (1, F, 32), bias
        for (int u = wave; u < units; u += nw) {
            const int R = 16 * u + j, Rc = min(R, GR - 1);
            const int g = Rc / T, t = Rc - g * T;
            f32x4 y0, y1;
            conv_tile(cw, ya, Rc, t, 1, T, ZR, q, y0, y1);
            f32x4 s0 = *reinterpret_cast<const f32x4 *>(km + (size_t)F * C + 4 * q);
            f32x4 s1 = *reinterpret_cast<const f32x4 *>(km + (size_t)F * C + 16 + 4 * q);
            const float *xr = X + ((size_t)(n0 + g) * T + t) * F + (size_t)q * a.FQ;
            const float *wk = km + (size_t)q * a.FQ * C + j;
            for (int s = 0; s < a.FQ; ++s) {
                const bool live = q * a.FQ + s < F;
                const float xv = live ? xr[s] : 0.f;
                s0 = mfma4(live ? wk[(size_t)s * C] : 0.f, xv, s0);
                s1 = mfma4(live ? wk[(size_t)s * C + 16] : 0.f, xv, s1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = fmaxf(s0[r] + fmaxf(y0[r], 0.f), 0.f);
                s1[r] = fmaxf(s1[r] + fmaxf(y1[r], 0.f), 0.f);
            }
            float *dst = xa + (size_t)R * SX + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = s0;
            *reinterpret_cast<f32x4 *>(dst + 16) = s1;
        }
    }
    // ---- blocks 1 .. n_blocks-1 ----
    for (int blk = 1; blk < a.n_blocks; ++blk) {
        const int d = 1 << (blk % a.n_dil);
        const float *k0 = flat + blk0_floats + (size_t)(blk - 1) * blk_floats;
        const float *k1 = k0 + 3 * C * C + C;
        ConvW cw;
        load_conv(cw, k0, k0 + 3 * C * C, q, j);
        __syncthreads();  // x of the previous block complete
        for (int u = wave; u < units; u += nw) {
            const int R = 16 * u + j, Rc = min(R, GR - 1);
            const int t = Rc % T;
            f32x4 y0, y1;
            conv_tile(cw, xa, Rc, t, d, T, ZR, q, y0, y1);
#pragma unroll
            for (int r = 0; r < 4; ++r) y0[r] = fmaxf(y0[r], 0.f), y1[r] = fmaxf(y1[r], 0.f);
            float *dst = ya + (size_t)R * SX + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = y0;
            *reinterpret_cast<f32x4 *>(dst + 16) = y1;
        }
        load_conv(cw, k1, k1 + 3 * C * C, q, j);
        __syncthreads();  // y complete
        for (int u = wave; u < units; u += nw) {
            const int R = 16 * u + j, Rc = min(R, GR - 1);
            const int t = Rc % T;
            f32x4 y0, y1;
            conv_tile(cw, ya, Rc, t, d, T, ZR, q, y0, y1);
            float *xr = xa + (size_t)Rc * SX + 4 * q;  // identity shortcut: this lane's own rows
            f32x4 s0 = *reinterpret_cast<const f32x4 *>(xr), s1 = *reinterpret_cast<const f32x4 *>(xr + 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = fmaxf(s0[r] + fmaxf(y0[r], 0.f), 0.f);
                s1[r] = fmaxf(s1[r] + fmaxf(y1[r], 0.f), 0.f);
            }
            float *dst = xa + (size_t)R * SX + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = s0;
            *reinterpret_cast<f32x4 *>(dst + 16) = s1;
        }
    }
    __syncthreads();
    if (trunk)  // the TCN output (already through its last relu) as (N, T, 32) == Keras Flatten order
        for (int i = threadIdx.x; i < GR * (C / 4); i += blockDim.x) {
            const int R = i >> 3, c4 = (i & 7) * 4;
            *reinterpret_cast<f32x4 *>(trunk + ((size_t)n0 * T + R) * C + c4) = *reinterpret_cast<const f32x4 *>(xa + (size_t)R * SX + c4);
        }
    TrainIO none{nullptr, nullptr, nullptr, nullptr};
    dense_and_heads<false>(a, xa, ya, WhA, hp, out, none, n0, g_here);
}

}  // namespace

namespace smh_tcn {

int launch_forward_v2(const smh_model *m, const float *d_x, int N, float *d_out, float *d_trunk, hipStream_t st) {
    TcnArgs a;
    size_t lds;
    fill_args(m, N, &a, &lds);
    lds = sizeof(float) * 2 * (size_t)(a.GRP + 1) * SX;  // x and y, each with its zero row; no weight slots
    SMH_REQUIRE(lds <= 156 * 1024, "patch_size %d too long for the LDS-resident TCN", a.T);
    const dim3 grid((N + a.G - 1) / a.G), block(512);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)b3mtl_forward_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(b3mtl_forward_v2_kernel, grid, block, lds, st, a, d_x, (const float *)m->d_flat, (const float *)m->d_WhA,
                       (const float *)m->d_hp, d_trunk, d_out);
    return smh::launch_status("b3mtl_forward_v2_kernel");
}

}  // namespace smh_tcn
