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
// 24 blocks (double buffered, one barrier per block); weights stream from L2 in MFMA A-operand order.
#include <cmath>
#include <cstring>
#include <vector>

#include "smh_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int C = 32;          // nb_filters (fixed by the MFMA tiling)
constexpr int SX = 36;         // LDS row stride of x in floats (16-byte aligned rows)
constexpr int kMaxHeads = 4;
constexpr int kHidden = 16;    // Dense(16) of every MTL head
constexpr int kHeadPatches = 4;  // patches per workgroup in the heads kernel
constexpr float kNormEps = 1e-5f;
constexpr float kBnEps = 1e-3f;

struct TcnArgs {
    int N, T, F, FQ, G, n_blocks, n_dil, vec_ok;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Packed per-block weights: [conv A: 24 steps x 2 M-tiles x 64 lanes][1x1 A: 8 x 2 x 64][b1 32][b2 32]
constexpr int kBlockFloats = 24 * 2 * 64 + 8 * 2 * 64 + 32 + 32;

__global__ void __launch_bounds__(256, 2)
tcn_trunk_kernel(TcnArgs a, const float *__restrict__ X, const float *__restrict__ W0, const float *__restrict__ Wb,
                 float *__restrict__ trunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int n0 = blockIdx.x * a.G;
    const int g_here = min(a.G, a.N - n0);
    const int T = a.T;
    const int GR = g_here * T;
    const int units = (GR + 15) >> 4;
    const int GRP = ((a.G * T + 15) >> 4) << 4;
    float *xa = lds, *xb = lds + (size_t)GRP * SX;

    // ---- initial Conv1D(32, 1): K order f = q*FQ + s so that every lane streams a contiguous run ----
    {
        const float *bias0 = W0 + (size_t)a.FQ * 2 * 64;
        for (int u = wave; u < units; u += nw) {
            const int R = 16 * u + j;
            const int Rc = min(R, GR - 1);
            const float *xrow = X + ((size_t)n0 * T + Rc) * a.F + (size_t)q * a.FQ;
            f32x4 acc0 = *reinterpret_cast<const f32x4 *>(bias0 + 4 * q);
            f32x4 acc1 = *reinterpret_cast<const f32x4 *>(bias0 + 16 + 4 * q);
            const float *wa = W0 + lane;
            if (a.vec_ok) {
                for (int s = 0; s < a.FQ; s += 4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(xrow + s);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = wa[(size_t)((s + e) * 2 + 0) * 64];
                        const float a1 = wa[(size_t)((s + e) * 2 + 1) * 64];
                        acc0 = mfma4(a0, xv[e], acc0);
                        acc1 = mfma4(a1, xv[e], acc1);
                    }
                }
            } else {
                for (int s = 0; s < a.FQ; ++s) {
                    const float xv = (q * a.FQ + s < a.F) ? xrow[s] : 0.f;
                    acc0 = mfma4(wa[(size_t)(s * 2 + 0) * 64], xv, acc0);
                    acc1 = mfma4(wa[(size_t)(s * 2 + 1) * 64], xv, acc1);
                }
            }
            float *dst = xa + (size_t)R * SX + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = acc0;
            *reinterpret_cast<f32x4 *>(dst + 16) = acc1;
        }
    }

    // ---- residual blocks ----
    float *xin = xa, *xout = xb;
    for (int blk = 0; blk < a.n_blocks; ++blk) {
        const int d = 1 << (blk % a.n_dil);
        const float *wblk = Wb + (size_t)blk * kBlockFloats;
        // A operands of this block in registers (reused by all of this wave's units)
        float wc[24][2], wp[8][2];
        const bool side_taps = d < T;  // |offset| >= T: the side taps only ever see zero padding
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            wc[s][0] = wblk[(s * 2 + 0) * 64 + lane];
            wc[s][1] = wblk[(s * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            wp[s][0] = wblk[24 * 2 * 64 + (s * 2 + 0) * 64 + lane];
            wp[s][1] = wblk[24 * 2 * 64 + (s * 2 + 1) * 64 + lane];
        }
        const float *b1 = wblk + 24 * 2 * 64 + 8 * 2 * 64;
        const float *b2 = b1 + 32;
        const f32x4 b1lo = *reinterpret_cast<const f32x4 *>(b1 + 4 * q);
        const f32x4 b1hi = *reinterpret_cast<const f32x4 *>(b1 + 16 + 4 * q);
        const f32x4 b2lo = *reinterpret_cast<const f32x4 *>(b2 + 4 * q);
        const f32x4 b2hi = *reinterpret_cast<const f32x4 *>(b2 + 16 + 4 * q);
        __syncthreads();  // xin complete (written by the previous stage)

        for (int u = wave; u < units; u += nw) {
            const int R = 16 * u + j;
            const int Rc = min(R, GR - 1);
            const int t = Rc % T;
            f32x4 acc0 = b1lo, acc1 = b1hi;
            // dilated conv: k index = tap*32 + c, step s covers c = (4s % 32) + q of tap s/8
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                if (tap != 1 && !side_taps) continue;
                const int off = (tap - 1) * d;
                const bool ok = (t + off >= 0) && (t + off < T);
                const float *src = xin + (size_t)(ok ? Rc + off : Rc) * SX + q;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    float bv = src[4 * s8];
                    bv = ok ? bv : 0.f;
                    acc0 = mfma4(wc[tap * 8 + s8][0], bv, acc0);
                    acc1 = mfma4(wc[tap * 8 + s8][1], bv, acc1);
                }
            }
            // relu + channel-max normalisation ('norm_relu')
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] = fmaxf(acc0[r], 0.f);
                acc1[r] = fmaxf(acc1[r], 0.f);
                mx = fmaxf(mx, fmaxf(acc0[r], acc1[r]));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float inv = 1.0f / (mx + kNormEps);
            // 1x1 conv on the normalised activations + bias + residual, all from registers
            const float *res = xin + (size_t)Rc * SX + 4 * q;
            f32x4 o0 = *reinterpret_cast<const f32x4 *>(res) + b2lo;
            f32x4 o1 = *reinterpret_cast<const f32x4 *>(res + 16) + b2hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y0 = acc0[r] * inv;  // channel 4q + r
                o0 = mfma4(wp[r][0], y0, o0);
                o1 = mfma4(wp[r][1], y0, o1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y1 = acc1[r] * inv;  // channel 16 + 4q + r
                o0 = mfma4(wp[4 + r][0], y1, o0);
                o1 = mfma4(wp[4 + r][1], y1, o1);
            }
            float *dst = xout + (size_t)R * SX + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = o0;
            *reinterpret_cast<f32x4 *>(dst + 16) = o1;
        }
        float *tmp = xin;
        xin = xout;
        xout = tmp;
    }
    __syncthreads();
    // final relu; trunk (N, T, 32) row-major == Keras Flatten order
    float *out = trunk + (size_t)n0 * T * C;
    for (int i = threadIdx.x; i < GR * (C / 4); i += blockDim.x) {
        const int R = i >> 3, c4 = (i & 7) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(xin + (size_t)R * SX + c4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *reinterpret_cast<f32x4 *>(out + (size_t)R * C + c4) = v;
    }
}

struct HeadArgs {
    int N, D, NH, n_classes, n_heads, out_dim;
    int head_odim[kMaxHeads];
    int head_sigmoid[kMaxHeads];
};

// Heads: one workgroup per kHeadPatches patches.  Stage 1: all Dense layers that read the flattened
// trunk as one (D x NH) product (NH = n_classes + 16*n_heads); stage 2: BN/relu/out-Dense/activations.
// Packed head params after Wh (D*NH) and bh (NH): per head [gamma16 beta16 mean16 var16 Wout(16*odim) bout(odim)].
__global__ void __launch_bounds__(256)
heads_kernel(HeadArgs a, const float *__restrict__ trunk, const float *__restrict__ Wh, const float *__restrict__ hp,
             float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *flat = sm;                                   // kHeadPatches * D
    float *part = sm + (size_t)kHeadPatches * a.D;      // parts * NH * kHeadPatches
    float *pre = part + (size_t)256 * kHeadPatches;     // NH * kHeadPatches
    const int n0 = blockIdx.x * kHeadPatches;
    const int np = min(kHeadPatches, a.N - n0);
    for (int i = threadIdx.x; i < np * a.D; i += blockDim.x) flat[i] = trunk[(size_t)n0 * a.D + i];
    for (int i = np * a.D + threadIdx.x; i < kHeadPatches * a.D; i += blockDim.x) flat[i] = 0.f;
    __syncthreads();
    const int parts = 256 / a.NH;
    const int o = threadIdx.x % a.NH, pt = threadIdx.x / a.NH;
    if (pt < parts) {
        const int chunk = (a.D + parts - 1) / parts;
        const int i0 = pt * chunk, i1 = min(a.D, i0 + chunk);
        float acc[kHeadPatches];
#pragma unroll
        for (int p = 0; p < kHeadPatches; ++p) acc[p] = 0.f;
        for (int i = i0; i < i1; ++i) {
            const float w = Wh[(size_t)i * a.NH + o];
#pragma unroll
            for (int p = 0; p < kHeadPatches; ++p) acc[p] = fmaf(flat[p * a.D + i], w, acc[p]);
        }
#pragma unroll
        for (int p = 0; p < kHeadPatches; ++p) part[(pt * a.NH + o) * kHeadPatches + p] = acc[p];
    }
    __syncthreads();
    const float *bh = Wh + (size_t)a.D * a.NH;
    for (int i = threadIdx.x; i < a.NH * kHeadPatches; i += blockDim.x) {
        const int oo = i / kHeadPatches, p = i - oo * kHeadPatches;
        float s = bh[oo];
        for (int k = 0; k < parts; ++k) s += part[(k * a.NH + oo) * kHeadPatches + p];
        pre[oo * kHeadPatches + p] = s;
    }
    __syncthreads();
    // stage 2: one thread per (patch, head) and one per patch for the softmax
    const int tid = threadIdx.x;
    if (tid < np * a.n_heads) {
        const int p = tid / a.n_heads, h = tid - p * a.n_heads;
        const float *ph = hp;
        int col = 0;
        for (int k = 0; k < h; ++k) {
            ph += 4 * kHidden + kHidden * a.head_odim[k] + a.head_odim[k];
            col += a.head_odim[k];
        }
        const float *gamma = ph, *beta = ph + 16, *mean = ph + 32, *var = ph + 48, *wo = ph + 64;
        const int od = a.head_odim[h];
        const float *bo = wo + kHidden * od;
        float hid[kHidden];
#pragma unroll
        for (int i = 0; i < kHidden; ++i) {
            float v = pre[(a.n_classes + h * kHidden + i) * kHeadPatches + p];
            v = (v - mean[i]) / sqrtf(var[i] + kBnEps);
            v = v * gamma[i] + beta[i];
            hid[i] = fmaxf(v, 0.f);
        }
        for (int c = 0; c < od; ++c) {
            float s = bo[c];
#pragma unroll
            for (int i = 0; i < kHidden; ++i) s = fmaf(hid[i], wo[i * od + c], s);
            if (a.head_sigmoid[h]) s = 1.0f / (1.0f + expf(-s));
            out[(size_t)(n0 + p) * a.out_dim + col + c] = s;
        }
    } else if (tid >= 128 && tid < 128 + np) {
        const int p = tid - 128;
        float mxl = -INFINITY;
        for (int c = 0; c < a.n_classes; ++c) mxl = fmaxf(mxl, pre[c * kHeadPatches + p]);
        float den = 0.f;
        for (int c = 0; c < a.n_classes; ++c) den += expf(pre[c * kHeadPatches + p] - mxl);
        const int col = a.out_dim - a.n_classes;
        for (int c = 0; c < a.n_classes; ++c)
            out[(size_t)(n0 + p) * a.out_dim + col + c] = expf(pre[c * kHeadPatches + p] - mxl) / den;
    }
}

}  // namespace

struct smh_model {
    smh_model_cfg cfg;
    int n_blocks, n_heads, NH, D, out_dim, FQ;
    int head_odim[kMaxHeads], head_sigmoid[kMaxHeads];
    size_t n_params;
    float *d_W0 = nullptr;     // layer-0 A operands + bias0
    float *d_Wb = nullptr;     // per-block packed weights
    float *d_Wh = nullptr;     // (D x NH) + bh
    float *d_hp = nullptr;     // per-head BN / out params
    float *d_trunk = nullptr;  // scratch when the caller passes no tap
    size_t trunk_cap = 0;
    size_t nW0, nWb, nWh, nhp;
};

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
    m->out_dim = cfg->n_classes;
    for (int i = 0; i < m->n_heads; ++i) m->out_dim += m->head_odim[i];
    m->FQ = (cfg->n_feat + 3) / 4;
    if ((cfg->n_feat % 4 == 0) && (m->FQ % 4 != 0)) { /* contiguous runs still fine, float4 path needs FQ%4==0 */ }
    size_t n = (size_t)cfg->n_feat * C + C;
    n += (size_t)m->n_blocks * (3 * C * C + C + C * C + C);
    n += (size_t)m->D * cfg->n_classes + cfg->n_classes;
    for (int i = 0; i < m->n_heads; ++i)
        n += (size_t)m->D * kHidden + kHidden + 4 * kHidden + (size_t)kHidden * m->head_odim[i] + m->head_odim[i];
    m->n_params = n;
    m->nW0 = (size_t)m->FQ * 2 * 64 + 32;
    m->nWb = (size_t)m->n_blocks * kBlockFloats;
    m->nWh = (size_t)m->D * m->NH + m->NH;
    m->nhp = 0;
    for (int i = 0; i < m->n_heads; ++i) m->nhp += 4 * kHidden + (size_t)kHidden * m->head_odim[i] + m->head_odim[i];
    hipError_t e = hipMalloc((void **)&m->d_W0, m->nW0 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_Wb, m->nWb * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_Wh, m->nWh * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_hp, m->nhp * sizeof(float));
    if (e != hipSuccess) {
        smh_model_destroy(m);
        return smh::set_error(SMH_E_HIP, "smh_model_create: hipMalloc failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return SMH_OK;
}

extern "C" void smh_model_destroy(smh_model *m) {
    if (!m) return;
    (void)hipFree(m->d_W0);
    (void)hipFree(m->d_Wb);
    (void)hipFree(m->d_Wh);
    (void)hipFree(m->d_hp);
    (void)hipFree(m->d_trunk);
    delete m;
}

extern "C" size_t smh_model_num_params(const smh_model *m) { return m ? m->n_params : 0; }
extern "C" int smh_model_out_dim(const smh_model *m) { return m ? m->out_dim : SMH_E_INVALID; }

// Canonical flat order (Keras array layouts):
//   initial_conv kernel (1,F,32), bias(32); per block [conv kernel (3,32,32), bias, conv1x1 kernel (1,32,32), bias];
//   3C kernel (D,ncls), bias; per head [dense kernel (D,16), bias, gamma, beta, moving_mean, moving_var,
//   out kernel (16,odim), out bias].
extern "C" int smh_model_set_weights(smh_model *m, const float *h, size_t n, void *stream) {
    SMH_REQUIRE(m && h, "smh_model_set_weights: null argument");
    SMH_REQUIRE(n == m->n_params, "smh_model_set_weights: got %zu floats, model has %zu", n, m->n_params);
    const int F = m->cfg.n_feat, FQ = m->FQ, D = m->D, NH = m->NH, ncls = m->cfg.n_classes;
    std::vector<float> W0(m->nW0, 0.f), Wb(m->nWb, 0.f), Wh(m->nWh, 0.f), hp(m->nhp, 0.f);
    const float *p = h;
    // layer 0: A[s][m'][lane] = W0[f = q*FQ + s][16m' + i]
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
                    const int tap = s / 8, c = (4 * s) % 32 + q;
                    wb[(s * 2 + mt) * 64 + lane] = k1[((size_t)tap * C + c) * C + 16 * mt + i];
                }
        for (int e = 0; e < 8; ++e)
            for (int mt = 0; mt < 2; ++mt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, i = lane & 15;
                    const int cin = 16 * (e / 4) + 4 * q + (e % 4);
                    wb[24 * 2 * 64 + (e * 2 + mt) * 64 + lane] = k2[(size_t)cin * C + 16 * mt + i];
                }
        std::memcpy(wb + 24 * 2 * 64 + 8 * 2 * 64, b1, C * sizeof(float));
        std::memcpy(wb + 24 * 2 * 64 + 8 * 2 * 64 + 32, b2, C * sizeof(float));
    }
    // heads: column order [3C | head0 dense16 | head1 dense16 | ...]
    const float *k3c = p;
    p += (size_t)D * ncls;
    const float *b3c = p;
    p += ncls;
    for (int i = 0; i < D; ++i)
        for (int c = 0; c < ncls; ++c) Wh[(size_t)i * NH + c] = k3c[(size_t)i * ncls + c];
    for (int c = 0; c < ncls; ++c) Wh[(size_t)D * NH + c] = b3c[c];
    float *php = hp.data();
    for (int hd = 0; hd < m->n_heads; ++hd) {
        const float *kd = p;
        p += (size_t)D * kHidden;
        const float *bd = p;
        p += kHidden;
        for (int i = 0; i < D; ++i)
            for (int c = 0; c < kHidden; ++c) Wh[(size_t)i * NH + ncls + hd * kHidden + c] = kd[(size_t)i * kHidden + c];
        for (int c = 0; c < kHidden; ++c) Wh[(size_t)D * NH + ncls + hd * kHidden + c] = bd[c];
        const int od = m->head_odim[hd];
        const size_t cnt = 4 * kHidden + (size_t)kHidden * od + od;
        std::memcpy(php, p, cnt * sizeof(float));  // gamma beta mean var Wout bout are contiguous in canonical order
        php += cnt;
        p += cnt;
    }
    hipStream_t st = (hipStream_t)stream;
    SMH_CHECK_HIP(hipMemcpyAsync(m->d_W0, W0.data(), W0.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SMH_CHECK_HIP(hipMemcpyAsync(m->d_Wb, Wb.data(), Wb.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SMH_CHECK_HIP(hipMemcpyAsync(m->d_Wh, Wh.data(), Wh.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SMH_CHECK_HIP(hipMemcpyAsync(m->d_hp, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SMH_CHECK_HIP(hipStreamSynchronize(st));  // the staging vectors die here
    return SMH_OK;
}

extern "C" int smh_model_forward_f32(const smh_model *mc, const float *d_x, int N, float *d_out, float *d_trunk,
                                     void *stream) {
    smh_model *m = const_cast<smh_model *>(mc);
    SMH_REQUIRE(m && d_x && d_out, "smh_model_forward_f32: null argument");
    SMH_REQUIRE(N >= 0, "smh_model_forward_f32: N=%d", N);
    if (N == 0) return SMH_OK;
    hipStream_t st = (hipStream_t)stream;
    const int T = m->cfg.patch_size;
    float *trunk = d_trunk;
    if (!trunk) {
        const size_t need = (size_t)N * T * C;
        if (need > m->trunk_cap) {
            // grows outside any stream capture: callers that capture graphs pass d_trunk or warm up first
            (void)hipFree(m->d_trunk);
            m->d_trunk = nullptr, m->trunk_cap = 0;
            SMH_CHECK_HIP(hipMalloc((void **)&m->d_trunk, need * sizeof(float)));
            m->trunk_cap = need;
        }
        trunk = m->d_trunk;
    }
    TcnArgs a;
    a.N = N, a.T = T, a.F = m->cfg.n_feat, a.FQ = m->FQ, a.n_blocks = m->n_blocks, a.n_dil = m->cfg.n_dilations;
    a.vec_ok = (a.F % 4 == 0) && (a.FQ % 4 == 0) && (a.FQ * 4 == a.F);
    // patches per workgroup: up to 272 rows (17 column tiles) of LDS-resident activations, but never
    // fewer workgroups than CUs when the batch allows it
    int gmax = 272 / T;
    if (gmax < 1) gmax = 1;
    int G = N / 256;
    if (G < 1) G = 1;
    if (G > gmax) G = gmax;
    a.G = G;
    const int GRP = ((G * T + 15) / 16) * 16;
    const size_t lds = sizeof(float) * 2 * (size_t)GRP * SX;
    SMH_REQUIRE(lds <= 156 * 1024, "patch_size %d too long for the LDS-resident TCN", T);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)tcn_trunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(tcn_trunk_kernel, dim3((N + G - 1) / G), dim3(256), lds, st, a, d_x, m->d_W0, m->d_Wb, trunk);
    int rc = smh::launch_status("tcn_trunk_kernel");
    if (rc) return rc;

    HeadArgs ha;
    ha.N = N, ha.D = m->D, ha.NH = m->NH, ha.n_classes = m->cfg.n_classes, ha.n_heads = m->n_heads, ha.out_dim = m->out_dim;
    for (int i = 0; i < kMaxHeads; ++i) ha.head_odim[i] = m->head_odim[i], ha.head_sigmoid[i] = m->head_sigmoid[i];
    const size_t hl = sizeof(float) * ((size_t)kHeadPatches * m->D + 256 * kHeadPatches + (size_t)m->NH * kHeadPatches);
    SMH_REQUIRE(hl <= 156 * 1024, "patch_size %d too long for the heads kernel", T);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
    hipLaunchKernelGGL(heads_kernel, dim3((N + kHeadPatches - 1) / kHeadPatches), dim3(256), hl, st, ha, trunk, m->d_Wh,
                       m->d_hp, d_out);
    return smh::launch_status("heads_kernel");
}
