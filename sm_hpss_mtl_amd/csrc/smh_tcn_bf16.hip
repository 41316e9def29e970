// B3_MTL inference forward with bf16 matrix-core operands (BASELINE config 5: "mixed bf16 CNN + fp32 HPSS").
// Same graph, same transposed-product mapping and the same LDS-resident f32 activation stream as smh_tcn.hip; only
// the MFMA operands are rounded to bf16 (v_mfma_f32_16x16x32_bf16, f32 accumulation): every weight once when the
// operand buffer is (re)built, every activation right before it is multiplied.  Bias, residual sum, relu, the
// channel-max normalisation, BatchNorm and the output activations stay f32.  Not bit-compatible with the f32 path
// (stated tolerance in tests/test_bf16_gpu.py); `smh_model_forward_f32` remains the parity path.
//
// k orders (one MFMA step covers 32 k values, lane (q = l>>4) supplies k = 8q .. 8q+7):
//   layer 0   k = feature, 8 steps for 240 features (zero padded to 256)
//   conv      one step per tap, k' = input channel in the order of the accumulator registers: lane group q holds channels
//             4q+r (r < 4), then 16+4q+r -- a lane's eight outputs of one block are, as they stand, the eight k' values the
//             same lane group supplies to the next block's products (one 16-byte LDS row piece written, one read)
//   1x1 conv  the same k' order: the normalised activations go from registers straight into the B operand
//   heads     one step per frame t, k' = channel;  D[output][patch]
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "smh_model.h"

using namespace smh_tcn;

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 to_bf16x8(f32x4 lo, f32x4 hi) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (__bf16)lo[i], r[4 + i] = (__bf16)hi[i];
    return r;
}
__device__ __forceinline__ bf16x8 zero_bf16x8() {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)0.0f;
    return r;
}

// An MFMA operand as bf16 "hi" (the rounded value) and, in the SPLIT variant, "lo" = bf16(x - hi): x ~ hi + lo to 16
// mantissa bits.  product<SPLIT> = hi*hi (+ hi*lo + lo*hi): three bf16 products reproduce the f32 product to ~2^-16
// relative -- the variant that meets the 2e-2 tolerance of SURVEY 8(d') through 24 normalised blocks; the lo*lo term
// (2^-18) is dropped.
struct Op {
    bf16x8 hi, lo;
};
template <bool SPLIT>
__device__ __forceinline__ Op split_op(f32x4 a, f32x4 b) {
    Op r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.hi[i] = (__bf16)a[i], r.hi[4 + i] = (__bf16)b[i];
        if (SPLIT) r.lo[i] = (__bf16)(a[i] - (float)r.hi[i]), r.lo[4 + i] = (__bf16)(b[i] - (float)r.hi[4 + i]);
    }
    if (!SPLIT) r.lo = r.hi;  // unused
    return r;
}
template <bool SPLIT>
__device__ __forceinline__ Op zero_op() {
    Op r;
    r.hi = zero_bf16x8(), r.lo = r.hi;
    return r;
}
// packed weights: hi at [idx], lo at [lo_off + idx]
template <bool SPLIT, typename P>
__device__ __forceinline__ Op load_op(P pk, size_t idx, size_t lo_off) {
    Op r;
    r.hi = pk[idx];
    r.lo = SPLIT ? pk[lo_off + idx] : r.hi;
    return r;
}
template <bool SPLIT>
__device__ __forceinline__ f32x4 product(const Op &w, const Op &x, f32x4 c) {
    if (SPLIT) {
        c = mfma_bf16(w.lo, x.hi, c);  // small terms first
        c = mfma_bf16(w.hi, x.lo, c);
    }
    return mfma_bf16(w.hi, x.hi, c);
}

struct PackInfo {
    size_t l0, blk0, blk_stride, heads, total;  // offsets in bf16x8 units (16 bytes)
    int steps0;
};
PackInfo pack_info(const smh_model *m) {
    PackInfo p;
    p.steps0 = (m->cfg.n_feat + 31) / 32;
    p.l0 = 0;
    p.blk0 = (size_t)p.steps0 * 2 * 64;
    p.blk_stride = 8 * 64 + 64;  // 3 taps x 2 M-tiles + 2 M-tiles of the 1x1, then one more row of 64 units whose first 16 are
                                 // [b1 32 floats | b2 32 floats] as raw f32 (a multiple of 64 units: `lane = idx & 63` below)
    p.heads = p.blk0 + (size_t)m->n_blocks * p.blk_stride;
    p.total = p.heads + (size_t)m->n_mt * m->cfg.patch_size * 64;
    return p;
}

// canonical f32 weights -> bf16 operands, one thread per bf16x8
__global__ void pack_bf16_kernel(const float *__restrict__ flat, Offsets off, PackInfo pi, int F, int T, int n_blocks,
                                 int n_classes, int n_heads, int NH, bf16x8 *__restrict__ dst) {
    // every branch below fills vf (the eight f32 weights of this operand); hi goes to dst[idx], lo to dst[total + idx]
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pi.total) return;
    const int lane = (int)(idx & 63), q = lane >> 4, j = lane & 15;
    float vf[8];
    if (idx < pi.blk0) {  // layer 0: [s][mt][lane]
        const int smt = (int)(idx >> 6), s = smt >> 1, mt = smt & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = 32 * s + 8 * q + i;
            vf[i] = f < F ? flat[off.w0_k + (size_t)f * C + 16 * mt + j] : 0.f;
        }
    } else if (idx < pi.heads) {  // blocks: [blk][tap*2 + mt | 6 + mt][lane], then 16 units of f32 biases
        const size_t r = idx - pi.blk0;
        const int blk = (int)(r / pi.blk_stride), e = (int)((r % pi.blk_stride) >> 6);
        const size_t wo = off.blk0 + (size_t)blk * off.blk_stride;
        if (e >= 8) {  // unit k of the bias tail: floats 4k .. 4k+3 of [b1 | b2], stored as they are (not bf16)
            const int k = (int)(r % pi.blk_stride) - 8 * 64;
            const float *b1 = flat + wo + 3 * C * C, *b2 = b1 + C + C * C;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (k < 16) v[i] = 4 * k + i < C ? b1[4 * k + i] : b2[4 * k + i - C];
            *reinterpret_cast<f32x4 *>(dst + idx) = v;
            *reinterpret_cast<f32x4 *>(dst + pi.total + idx) = v;
            return;
        }
        if (e < 6) {
            const int tap = e >> 1, mt = e & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cin = i < 4 ? 4 * q + i : 16 + 4 * q + (i - 4);  // k' = the accumulator's channel order (see above)
                vf[i] = flat[wo + ((size_t)tap * C + cin) * C + 16 * mt + j];
            }
        } else {
            const int mt = e - 6;
            const size_t k2 = wo + 3 * C * C + C;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cin = i < 4 ? 4 * q + i : 16 + 4 * q + (i - 4);
                vf[i] = flat[k2 + (size_t)cin * C + 16 * mt + j];
            }
        }
    } else {  // Dense-on-trunk: [mt][t][lane], output o = 16 mt + j, k = t*32 + 8q + i
        const size_t r = idx - pi.heads;
        const int mt = (int)(r / ((size_t)T * 64)), t = (int)((r >> 6) % T);
        const int o = 16 * mt + j;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t k = (size_t)t * C + (i < 4 ? 4 * q + i : 16 + 4 * q + (i - 4));  // k' order
            float w = 0.f;
            if (o < n_classes) w = flat[off.c3_k + k * n_classes + o];
            else if (o < NH) {
                const int h = (o - n_classes) / kHidden, jj = (o - n_classes) % kHidden;
                w = flat[off.head[h] + k * kHidden + jj];
            }
            vf[i] = w;
        }
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (__bf16)vf[i], lo[i] = (__bf16)(vf[i] - (float)hi[i]);
    dst[idx] = hi;
    dst[pi.total + idx] = lo;
}

struct BlockWb {
    Op wc[3][2], wp[2];
    f32x4 b1lo, b1hi, b2lo, b2hi;
};
template <bool SPLIT>
__device__ __forceinline__ void load_block(BlockWb &w, const bf16x8 *__restrict__ pk, size_t lo_off, const float *__restrict__ flat,
                                           size_t wo, int lane, int q) {
#pragma unroll
    for (int e = 0; e < 6; ++e) w.wc[e >> 1][e & 1] = load_op<SPLIT>(pk, (size_t)e * 64 + lane, lo_off);
    w.wp[0] = load_op<SPLIT>(pk, (size_t)6 * 64 + lane, lo_off);
    w.wp[1] = load_op<SPLIT>(pk, (size_t)7 * 64 + lane, lo_off);
    const float *b1 = flat + wo + 3 * C * C, *b2 = b1 + C + C * C;
    w.b1lo = *reinterpret_cast<const f32x4 *>(b1 + 4 * q);
    w.b1hi = *reinterpret_cast<const f32x4 *>(b1 + 16 + 4 * q);
    w.b2lo = *reinterpret_cast<const f32x4 *>(b2 + 4 * q);
    w.b2hi = *reinterpret_cast<const f32x4 *>(b2 + 16 + 4 * q);
}

template <bool SPLIT>
__device__ __forceinline__ void run_block(const BlockWb &w, int d, int T, int GR, int units, int wave, int nw, int q,
                                          int j, const float *__restrict__ xin, float *__restrict__ xout) {
    for (int u = wave; u < units; u += nw) {
        const int R = 16 * u + j;
        const int Rc = min(R, GR - 1);
        const int t = Rc % T;
        f32x4 acc0 = w.b1lo, acc1 = w.b1hi;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int off = (tap - 1) * d;
            const bool ok = (t + off >= 0) && (t + off < T);
            if (tap != 1 && !__any(ok)) continue;
            const float *src = xin + (size_t)(ok ? Rc + off : Rc) * SX + 4 * q;  // k' order: channels 4q.., then 16+4q..
            Op b = split_op<SPLIT>(*reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 16));
            if (!ok) b = zero_op<SPLIT>();
            acc0 = product<SPLIT>(w.wc[tap][0], b, acc0);
            acc1 = product<SPLIT>(w.wc[tap][1], b, acc1);
        }
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
        acc0 *= inv;
        acc1 *= inv;
        const Op yb = split_op<SPLIT>(acc0, acc1);  // k' = 4q+r, then 16+4q+r: the order the 1x1 operand was packed in
        const float *res = xin + (size_t)Rc * SX + 4 * q;
        f32x4 o0 = *reinterpret_cast<const f32x4 *>(res) + w.b2lo;
        f32x4 o1 = *reinterpret_cast<const f32x4 *>(res + 16) + w.b2hi;
        o0 = product<SPLIT>(w.wp[0], yb, o0);
        o1 = product<SPLIT>(w.wp[1], yb, o1);
        float *dst = xout + (size_t)R * SX + 4 * q;
        *reinterpret_cast<f32x4 *>(dst) = o0;
        *reinterpret_cast<f32x4 *>(dst + 16) = o1;
    }
}

template <bool SPLIT>
__global__ void __launch_bounds__(512)
b3mtl_forward_bf16_kernel(TcnArgs a, PackInfo pi, Offsets off, const float *__restrict__ X, const float *__restrict__ flat,
                          const bf16x8 *__restrict__ pk, const float *__restrict__ hp, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int n0 = blockIdx.x * a.G;
    const int g_here = min(a.G, a.N - n0);
    const int T = a.T;
    const int GR = g_here * T;
    const int units = (GR + 15) >> 4;
    float *xa = lds, *xb = lds + (size_t)a.GRP * SX;

    // ---- initial Conv1D(32, 1) ------------------------------------------------------------------------------------
    if (a.from_x0) {
        // layer 0 was computed by the feature kernel in exact f32 (smh_features_l0_f32): X holds its two per-half partials
        // (N, 2, T, 32); sum them, add the bias (the f32 kernel's prologue, smh_tcn.hip)
        const float *bias0 = flat + off.w0_b;
        constexpr int kX0R = 4;
        const int n4 = GR * (C / 4);
        for (int i0 = threadIdx.x; i0 < n4; i0 += kX0R * blockDim.x) {
            f32x4 pa[kX0R], pb[kX0R];
#pragma unroll
            for (int r = 0; r < kX0R; ++r) {
                const int i = min(i0 + r * (int)blockDim.x, n4 - 1);
                const int R = i >> 3, c4 = (i & 7) * 4;
                const int g = R / T, t = R - g * T;
                const float *p0 = X + ((((size_t)(n0 + g) * 2) * T + t) * C + c4);
                pa[r] = *reinterpret_cast<const f32x4 *>(p0);
                pb[r] = *reinterpret_cast<const f32x4 *>(p0 + (size_t)T * C);
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
    } else {
        bf16x8 *w0s = reinterpret_cast<bf16x8 *>(xb);  // layer-0 A operands staged in the not-yet-used buffer
        const int nW0 = pi.steps0 * 2 * 64;
        for (int i = threadIdx.x; i < nW0; i += blockDim.x) {
            w0s[i] = pk[pi.l0 + i];
            if (SPLIT) w0s[nW0 + i] = pk[pi.total + pi.l0 + i];
        }
        __syncthreads();
        const float *b0 = flat + off.w0_b;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(b0 + 4 * q), bh = *reinterpret_cast<const f32x4 *>(b0 + 16 + 4 * q);
        for (int u = wave; u < units; u += nw) {
            const int R = 16 * u + j;
            const int Rc = min(R, GR - 1);
            const float *xr = X + ((size_t)n0 * T + Rc) * a.F + 8 * q;
            f32x4 xv[16];
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // all loads of the tile first (8 steps x 2 float4)
                const int f = 32 * s + 8 * q;
                const bool in = s < pi.steps0 && f + 7 < a.F && (a.F & 3) == 0;
                xv[2 * s] = in ? *reinterpret_cast<const f32x4 *>(xr + 32 * s) : f32x4{0.f, 0.f, 0.f, 0.f};
                xv[2 * s + 1] = in ? *reinterpret_cast<const f32x4 *>(xr + 32 * s + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (!in && s < pi.steps0) {  // ragged tail / unaligned feature count: scalar, bounds-checked
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = (f + i < a.F) ? xr[32 * s + i] : 0.f;
                        if (i < 4) xv[2 * s][i] = v;
                        else xv[2 * s + 1][i - 4] = v;
                    }
                }
            }
            f32x4 c0 = bl, c1 = bh;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s >= pi.steps0) break;
                const Op b = split_op<SPLIT>(xv[2 * s], xv[2 * s + 1]);
                c0 = product<SPLIT>(load_op<SPLIT>(w0s, (size_t)(s * 2 + 0) * 64 + lane, (size_t)nW0), b, c0);
                c1 = product<SPLIT>(load_op<SPLIT>(w0s, (size_t)(s * 2 + 1) * 64 + lane, (size_t)nW0), b, c1);
            }
            float *dst = xa + (size_t)R * SX + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = c0;
            *reinterpret_cast<f32x4 *>(dst + 16) = c1;
        }
    }

    float *xin = xa, *xout = xb;
    if constexpr (SPLIT) {  // one register set (hi + lo: 64 operand registers), loaded at the top of the block
        BlockWb w;
        for (int blk = 0; blk < a.n_blocks; ++blk) {
            load_block<true>(w, pk + pi.blk0 + (size_t)blk * pi.blk_stride, pi.total, flat, off.blk0 + (size_t)blk * off.blk_stride, lane, q);
            __syncthreads();
            run_block<true>(w, 1 << (blk % a.n_dil), T, GR, units, wave, nw, q, j, xin, xout);
            float *tmp = xin;
            xin = xout;
            xout = tmp;
        }
    } else {
        BlockWb wA, wB;
        load_block<false>(wA, pk + pi.blk0, pi.total, flat, off.blk0, lane, q);
        for (int blk = 0; blk < a.n_blocks; blk += 2) {
            if (blk + 1 < a.n_blocks)
                load_block<false>(wB, pk + pi.blk0 + (size_t)(blk + 1) * pi.blk_stride, pi.total, flat, off.blk0 + (size_t)(blk + 1) * off.blk_stride, lane, q);
            __syncthreads();
            run_block<false>(wA, 1 << (blk % a.n_dil), T, GR, units, wave, nw, q, j, xin, xout);
            if (blk + 1 < a.n_blocks) {
                if (blk + 2 < a.n_blocks)
                    load_block<false>(wA, pk + pi.blk0 + (size_t)(blk + 2) * pi.blk_stride, pi.total, flat, off.blk0 + (size_t)(blk + 2) * off.blk_stride, lane, q);
                __syncthreads();
                run_block<false>(wB, 1 << ((blk + 1) % a.n_dil), T, GR, units, wave, nw, q, j, xout, xin);
            } else {
                float *tmp = xin;
                xin = xout;
                xout = tmp;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GR * (C / 4); i += blockDim.x) {  // final relu in place
        const int R = i >> 3, c4 = (i & 7) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(xin + (size_t)R * SX + c4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *reinterpret_cast<f32x4 *>(xin + (size_t)R * SX + c4) = v;
    }
    __syncthreads();

    // ---- Dense layers on the flattened trunk: D[o][g] = sum_t sum_c WhT[o][t*32+c] x[g][t][c] ------------------------
    float *pre = xout;  // scratch [nks][kMaxG][kPS]
    const int nks = max(1, nw / a.n_mt);
    {
        const int mt = wave % a.n_mt, ks = wave / a.n_mt;
        if (ks < nks) {
            const int t_lo = (int)((long)T * ks / nks), t_hi = (int)((long)T * (ks + 1) / nks);
            const bf16x8 *wa = pk + pi.heads + (size_t)mt * T * 64 + lane;
            f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
            const bool live = j < g_here;
            const float *xg = xin + (size_t)(live ? j : 0) * T * SX + 4 * q;
            for (int t = t_lo; t < t_hi; ++t) {
                const float *xr = xg + (size_t)t * SX;
                Op b = split_op<SPLIT>(*reinterpret_cast<const f32x4 *>(xr), *reinterpret_cast<const f32x4 *>(xr + 16));
                if (!live) b = zero_op<SPLIT>();
                const Op wv = load_op<SPLIT>(wa, (size_t)t * 64, pi.total);
                if (t & 1) accB = product<SPLIT>(wv, b, accB);
                else accA = product<SPLIT>(wv, b, accA);
            }
            accA += accB;
#pragma unroll
            for (int r = 0; r < 4; ++r) pre[(ks * kMaxG + j) * kPS + 16 * mt + 4 * q + r] = accA[r];
        }
    }
    __syncthreads();
    if (nks > 1) {
        for (int i = threadIdx.x; i < kMaxG * kPS; i += blockDim.x) {
            float v = pre[i];
            for (int k2 = 1; k2 < nks; ++k2) v += pre[k2 * kMaxG * kPS + i];
            pre[i] = v;
        }
        __syncthreads();
    }
    // ---- BN / relu / output Dense / activations (f32), as in the f32 kernel -------------------------------------------
    auto bias_of = [&](int o) {
        if (o < a.n_classes) return flat[off.c3_b + o];
        const int h = (o - a.n_classes) / kHidden, jj = (o - a.n_classes) % kHidden;
        return flat[off.head[h] + (size_t)a.D * kHidden + jj];
    };
    const int tid = threadIdx.x;
    if (tid < g_here * a.n_heads) {
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
            const int o = a.n_classes + h * kHidden + i;
            float v = pre[p * kPS + o] + bias_of(o);
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
    } else if (tid >= 128 && tid < 128 + g_here) {
        const int p = tid - 128;
        float mxl = -INFINITY;
        for (int c = 0; c < a.n_classes; ++c) mxl = fmaxf(mxl, pre[p * kPS + c] + bias_of(c));
        float den = 0.f;
        for (int c = 0; c < a.n_classes; ++c) den += expf(pre[p * kPS + c] + bias_of(c) - mxl);
        const int col = a.out_dim - a.n_classes;
        for (int c = 0; c < a.n_classes; ++c)
            out[(size_t)(n0 + p) * a.out_dim + col + c] = expf(pre[p * kPS + c] + bias_of(c) - mxl) / den;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: the split-operand network with the activations kept in LDS ALREADY SPLIT (`b3mtl_forward_bf16s_kernel`, what
// smh_model_forward_bf16 runs).  The round-3 kernel above held the activation stream as f32 rows and split every value into hi + lo
// bf16 each time a tap used it (three times per block, ~30 vector instructions per use) -- with the matrix time down 5x it was that
// vector work, the block's weights fetched in front of every barrier and the lone third-round tile that made it slower than the
// exact-f32 kernel (133 against 127 us).  Here a block's output is split ONCE, by the lane that computed it, and stored as two bf16
// images [row][32 channels in k' order] (hi, lo; 80-byte rows: the 16-byte row pieces of 16 consecutive rows fall on 8 different
// bank quads -- the two cycles a 256-byte access takes anyway); a tap's B operand is then two 16-byte LDS reads and no arithmetic.
// The residual stream stays exact f32 in REGISTERS: a wave keeps its column tiles for the whole kernel (8 values per lane and tile).
// A block's 16 KB of operands (pre-split at pack time) reach the CU ONCE, by LDS-DMA into one of two slots while the block before
// runs; every wave then fills its 64 operand registers from the slot -- fetched by each of the 8 waves from L2 they were 128 KB
// per block and workgroup, 1.8 us at the 70 GB/s a CU reads L2 at: as long as the block itself now takes.
// Dead taps (dilation >= T) are skipped, out-of-range taps read an all-zero row.  Bias, residual, relu, channel-max normalisation,
// BatchNorm and the output activations are f32, accumulation is f32: results within 1e-4 of the f32 path (tests/test_bf16_gpu.py).
// LDS: 4 x 80 (GRP + 1) bytes of images = 85 KB for 272 rows, + 2 x 16 KB of operand slots + the Dense scratch = 122 KB.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kRS = 40;  // bf16 elements per row of a split image (32 + 8 of padding = 80 bytes)

struct SplitW {  // one block's A operands: [tap][M-tile] of the dilated conv, [M-tile] of the 1x1 conv, hi and lo; the biases
    bf16x8 ch[3][2], cl[3][2], ph[2], pl[2];
    f32x4 b1lo, b1hi, b2lo, b2hi;
};
constexpr int kSlotOps = 8 * 64 + 64;  // 16-byte units per half (hi / lo) of a weight slot: the operands [tap * 2 + mt | 6 + mt][lane], then
                                       // one more 1 KiB piece whose first 256 bytes are the block's biases [b1 | b2] (hi half only)
// pk: the block's slot in LDS ([hi 9 KB][lo 9 KB]); one ds_read_b128 per operand, lane-contiguous; the biases come with them -- as
// global loads behind the block's barrier they were an L2 round trip in front of every block's first product
__device__ __forceinline__ void load_split_w(SplitW &w, const bf16x8 *pk, int lane, int q) {
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        w.ch[e >> 1][e & 1] = pk[e * 64 + lane];
        w.cl[e >> 1][e & 1] = pk[kSlotOps + e * 64 + lane];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        w.ph[e] = pk[(6 + e) * 64 + lane];
        w.pl[e] = pk[kSlotOps + (6 + e) * 64 + lane];
    }
    const float *b1 = reinterpret_cast<const float *>(pk + 8 * 64), *b2 = b1 + C;
    w.b1lo = *reinterpret_cast<const f32x4 *>(b1 + 4 * q);
    w.b1hi = *reinterpret_cast<const f32x4 *>(b1 + 16 + 4 * q);
    w.b2lo = *reinterpret_cast<const f32x4 *>(b2 + 4 * q);
    w.b2hi = *reinterpret_cast<const f32x4 *>(b2 + 16 + 4 * q);
}
// eight f32 values (a lane's 4 + 4 channels in k' order) -> hi = bf16(x), lo = bf16(x - hi)
// (pairs: one v_cvt_pk_bf16_f32 rounds two values; the pair's float images are a shift and a mask of the packed word -- written
// element by element hipcc converted every hi value on its own, 12 conversions per call instead of 8)
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo) {
    const bf16x2s h = __builtin_convertvector(f32x2s{x0, x1}, bf16x2s);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2s hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    const bf16x2s l = __builtin_convertvector(f32x2s{x0, x1} - hf, bf16x2s);
    lo = __builtin_bit_cast(unsigned, l);
}
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8(f32x4 a, f32x4 b, bf16x8 &hi, bf16x8 &lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2(a[0], a[1], h0, l0);
    split2(a[2], a[3], h1, l1);
    split2(b[0], b[1], h2, l2);
    split2(b[2], b[3], h3, l3);
    hi = __builtin_bit_cast(bf16x8, u32x4s{h0, h1, h2, h3});
    lo = __builtin_bit_cast(bf16x8, u32x4s{l0, l1, l2, l3});
}
__device__ __forceinline__ f32x4 product3(bf16x8 wh, bf16x8 wl, bf16x8 xh, bf16x8 xl, f32x4 c) {
    c = mfma_bf16(wl, xh, c);  // small terms first
    c = mfma_bf16(wh, xl, c);
    return mfma_bf16(wh, xh, c);
}

// one block for this wave's column tiles: split images (xh, xl) of the block's input -> (yh, yl), residual image xres in place
// max over the four lanes that hold the same time step (l, l^16, l^32, l^48) on gfx950's VALU lane swaps (smh_tcn.hip: quad_max)
__device__ __forceinline__ float quad_max_s(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    const unsigned ua = __float_as_uint(a);
    const auto r16 = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
    return fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
}

// NT = 1 or 2 column tiles of one block as ONE straight-line region: with two tiles the compiler interleaves their independent
// chains (LDS reads -> 18 dependent products -> channel maximum -> 6 products -> stores), which a wave alone cannot overlap -- a
// tile's critical path is about twice its issue time.  Split images (xh, xl) of the block's input -> (yh, yl), residual image in place.
// what the TRAINING forward saves / applies per block (smh_model.h: TrainIO), for this workgroup's first patch
struct TrainBlk {
    float *upre;        // this block's slice of the saved dilated-conv outputs (incl. bias, before the relu), patch stride ustride
    const float *drop;  // this block's SpatialDropout1D mask of the first patch (32 channels), patch stride dstride; or nullptr
    int ustride, dstride, GR;
};
// ro / wo: this lane's element offsets into an image for row Rc (operand reads, taps add off * kRS) and row R (its own output);
// zo: the same lane position in the all-zero row -- computed once per kernel, a tap is then one add, two compares and a select
// ws: the block's operand slot in LDS ([hi 9 KB][lo 9 KB], load_split_w has the layout).  The A operands are read WHERE THEY ARE USED,
// tap by tap behind that tap's B operands: LDS returns in order, so the first product waits for six reads, not for the 36 of a
// whole block's operands up front (nine waves' worth of them kept every wave's first product 1.3 k cycles behind the barrier).
template <int NT, bool TRAIN>
__device__ __forceinline__ void tiles_split(const bf16x8 *ws, int lane, int d, int T, unsigned zo, const int *R, const unsigned *ro,
                                            const unsigned *wo, const int *t, const int *g, int q, const __bf16 *__restrict__ xh,
                                            const __bf16 *__restrict__ xl, __bf16 *__restrict__ yh, __bf16 *__restrict__ yl, f32x4 *res0,
                                            f32x4 *res1, const TrainBlk &tb) {
    const float *bias = reinterpret_cast<const float *>(ws + 8 * 64);  // [b1 32 | b2 32]
    f32x4 acc0[NT], acc1[NT];
    {
        const f32x4 b1lo = *reinterpret_cast<const f32x4 *>(bias + 4 * q), b1hi = *reinterpret_cast<const f32x4 *>(bias + 16 + 4 * q);
#pragma unroll
        for (int k = 0; k < NT; ++k) acc0[k] = b1lo, acc1[k] = b1hi;
    }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int off = (tap - 1) * d;
        bool ok[NT], any = false;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            ok[k] = (t[k] + off >= 0) && (t[k] + off < T);
            any = any || __any(ok[k]);
        }
        if (tap != 1 && !any) continue;  // (wave-uniform: the tap only sees zero padding)
        bf16x8 bh[NT], bl[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const unsigned row = ok[k] ? ro[k] + (unsigned)(off * kRS) : zo;
            bh[k] = *reinterpret_cast<const bf16x8 *>(xh + row);
            bl[k] = *reinterpret_cast<const bf16x8 *>(xl + row);
        }
        const bf16x8 c0h = ws[(2 * tap) * 64 + lane], c0l = ws[kSlotOps + (2 * tap) * 64 + lane];
        const bf16x8 c1h = ws[(2 * tap + 1) * 64 + lane], c1l = ws[kSlotOps + (2 * tap + 1) * 64 + lane];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            acc0[k] = product3(c0h, c0l, bh[k], bl[k], acc0[k]);
            acc1[k] = product3(c1h, c1l, bh[k], bl[k], acc1[k]);
        }
    }
    const bf16x8 p0h = ws[6 * 64 + lane], p0l = ws[kSlotOps + 6 * 64 + lane], p1h = ws[7 * 64 + lane], p1l = ws[kSlotOps + 7 * 64 + lane];
    const f32x4 b2lo = *reinterpret_cast<const f32x4 *>(bias + C + 4 * q), b2hi = *reinterpret_cast<const f32x4 *>(bias + C + 16 + 4 * q);
    bf16x8 nh[NT], nl[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if constexpr (TRAIN) {  // the backward's relu / channel-max gates: the conv output as this forward computed it
            if (tb.upre && R[k] < tb.GR) {
                const unsigned uo = (unsigned)(g[k] * tb.ustride + t[k] * C + 4 * q);
                *reinterpret_cast<f32x4 *>(tb.upre + uo) = acc0[k];
                *reinterpret_cast<f32x4 *>(tb.upre + uo + 16) = acc1[k];
            }
        }
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0[k][r] = fmaxf(acc0[k][r], 0.f);
            acc1[k][r] = fmaxf(acc1[k][r], 0.f);
            mx = fmaxf(mx, fmaxf(acc0[k][r], acc1[k][r]));
        }
        mx = quad_max_s(mx);
        const float inv = __builtin_amdgcn_rcpf(mx + kNormEps);
        acc0[k] *= inv;
        acc1[k] *= inv;
        if constexpr (TRAIN) {
            if (tb.drop) {  // SpatialDropout1D: one mask value per (patch, channel)
                const unsigned dofs = (unsigned)(g[k] * tb.dstride + 4 * q);
                acc0[k] *= *reinterpret_cast<const f32x4 *>(tb.drop + dofs);
                acc1[k] *= *reinterpret_cast<const f32x4 *>(tb.drop + dofs + 16);
            }
        }
        split8(acc0[k], acc1[k], nh[k], nl[k]);  // k' order = the order the 1x1 operand was packed in
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        f32x4 o0 = res0[k] + b2lo, o1 = res1[k] + b2hi;  // the residual stream: exact f32, in this lane's registers
        o0 = product3(p0h, p0l, nh[k], nl[k], o0);
        o1 = product3(p1h, p1l, nh[k], nl[k], o1);
        res0[k] = o0, res1[k] = o1;
        bf16x8 oh, ol;
        split8(o0, o1, oh, ol);
        *reinterpret_cast<bf16x8 *>(yh + wo[k]) = oh;
        *reinterpret_cast<bf16x8 *>(yl + wo[k]) = ol;
    }
}

// Column tiles per wave, MAXT = 2 or 4.  A block costs the workgroup its busiest WAVE's tiles one pair after the other, so the
// launch takes as many waves as halve the tile count: 272 rows = 17 tiles run on 9 waves (eight pairs and a single: one pair's time
// per block) rather than on 8 (wave 0: a pair, then a single).  Up to 12 waves (three per SIMD: 168 VGPRs); beyond 24 tiles, 8 waves
// with up to four tiles each.
template <int MAXT>
struct TilesS {
    int n, R[MAXT], t[MAXT], g[MAXT];
    unsigned ro[MAXT], wo[MAXT];  // element offsets of this lane's row pieces in an image: row Rc (reads), row R (its own output)
    f32x4 r0[MAXT], r1[MAXT];  // the residual stream of this lane's rows: channels 4q + r and 16 + 4q + r
};
template <int kMaxTilesS, bool TRAIN>
__device__ __forceinline__ void run_block_split(const bf16x8 *w, int lane, int d, int T, int ZR, TilesS<kMaxTilesS> &ti, int q,
                                                const __bf16 *__restrict__ xh, const __bf16 *__restrict__ xl, __bf16 *__restrict__ yh,
                                                __bf16 *__restrict__ yl, const TrainBlk &tb) {
    const unsigned zo = (unsigned)ZR * kRS + 8 * q;
#pragma unroll
    for (int i = 0; i < kMaxTilesS; i += 2) {
        if (i + 1 < ti.n)
            tiles_split<2, TRAIN>(w, lane, d, T, zo, ti.R + i, ti.ro + i, ti.wo + i, ti.t + i, ti.g + i, q, xh, xl, yh, yl, ti.r0 + i, ti.r1 + i, tb);
        else if (i < ti.n)
            tiles_split<1, TRAIN>(w, lane, d, T, zo, ti.R + i, ti.ro + i, ti.wo + i, ti.t + i, ti.g + i, q, xh, xl, yh, yl, ti.r0 + i, ti.r1 + i, tb);
    }
}

// TRAIN: the training-mode forward (smh_train_step_f32 with the trainer's dtype set to bf16): saves every block's input and its
// dilated-conv output for the backward pass -- straight from the lanes' registers --, applies the SpatialDropout1D masks, and
// hands the Dense-on-trunk outputs to heads_train_kernel instead of running the inference heads (smh_model.h: TrainIO).
template <int kMaxTilesS, bool TRAIN>
__global__ void __launch_bounds__(kMaxTilesS == 2 ? 768 : 512)
b3mtl_forward_bf16s_kernel(TcnArgs a, PackInfo pi, Offsets off, const float *__restrict__ X, const float *__restrict__ flat,
                           const bf16x8 *__restrict__ pk, const float *__restrict__ hp, float *__restrict__ out, TrainIO tio) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int n0 = blockIdx.x * a.G;
    const int g_here = min(a.G, a.N - n0);
    const int T = a.T;
    const int GR = g_here * T;
    const int units = (GR + 15) >> 4;
    const int ZR = a.GRP;  // the all-zero row of every split image
    const size_t img = (size_t)(a.GRP + 1) * kRS;  // bf16 elements per image
    __bf16 *im = reinterpret_cast<__bf16 *>(lds);
    __bf16 *h0 = im, *l0 = im + img, *h1 = im + 2 * img, *l1 = im + 3 * img;
    bf16x8 *wslot = reinterpret_cast<bf16x8 *>(im + 4 * img);  // two 16 KB operand slots
    float *scratch = reinterpret_cast<float *>(wslot + 2 * 2 * kSlotOps);  // Dense-on-trunk partial sums
    // block b's operands travel L2 -> LDS slot b % 2 by LDS-DMA (17 wave-instructions of 1 KiB: the 8 KB of hi operands + the piece
    // that starts with the biases, then the 8 KB of lo operands), issued at the top of block b - 1 -- behind the barrier every wave passes only after it has filled its registers
    // from that slot for block b - 2 -- and drained (vmcnt) in front of block b's barrier
    auto stage = [&](int blk) {
        const char *hi = reinterpret_cast<const char *>(pk + pi.blk0 + (size_t)blk * pi.blk_stride);
        const char *lo = reinterpret_cast<const char *>(pk + pi.total + pi.blk0 + (size_t)blk * pi.blk_stride);
        char *dst = reinterpret_cast<char *>(wslot + (size_t)(blk & 1) * 2 * kSlotOps);
        for (int i = wave; i < 17; i += nw) {  // pieces 0..8: the hi half (8 operand rows + the bias piece), 9..16: the lo operands
            const char *src = (i < 9 ? hi + (size_t)(i * 64 + lane) * 16 : lo + (size_t)((i - 9) * 64 + lane) * 16);
            char *d = dst + (i < 9 ? i : i + 0) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)d, 16, 0, 0);
        }
    };
    if (a.n_blocks > 0) stage(0);
    // this wave's column tiles for the whole kernel: tile u = wave + i * nw, lane (q, j) = row 16 u + j, channels 4q + r / 16 + 4q + r
    TilesS<kMaxTilesS> ti;
    ti.n = 0;
#pragma unroll
    for (int i = 0; i < kMaxTilesS; ++i) {
        const int u = wave + i * nw;
        const int R = 16 * u + j, Rc = min(R, GR - 1);
        ti.R[i] = R, ti.g[i] = Rc / T, ti.t[i] = Rc - ti.g[i] * T;
        ti.ro[i] = (unsigned)Rc * kRS + 8 * q, ti.wo[i] = (unsigned)R * kRS + 8 * q;
        if (u < units) ti.n = i + 1;
    }
    const int nslot = a.n_blocks + 1;
    auto save_acts = [&](int slot) {  // (training) the residual registers = block `slot`'s input -> acts[n][slot][t][c]
        if constexpr (TRAIN) {
#pragma unroll
            for (int i = 0; i < kMaxTilesS; ++i) {
                if (i < ti.n && ti.R[i] < GR) {
                    float *dst = tio.acts + (((size_t)(n0 + ti.g[i]) * nslot + slot) * T + ti.t[i]) * C + 4 * q;
                    *reinterpret_cast<f32x4 *>(dst) = ti.r0[i];
                    *reinterpret_cast<f32x4 *>(dst + 16) = ti.r1[i];
                }
            }
        }
    };

    // ---- initial Conv1D(32, 1): into the residual registers and the split images of buffer 0 --------------------------------------
    if (a.from_x0) {
        // layer 0 was computed by the feature kernel in exact f32 (smh_features_l0_f32): X holds its two per-half partials
        // (N, 2, T, 32); sum them, add the bias (the f32 kernel's prologue, smh_tcn.hip)
        const float *bias0 = flat + off.w0_b;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(bias0 + 4 * q), bh = *reinterpret_cast<const f32x4 *>(bias0 + 16 + 4 * q);
        f32x4 pa[kMaxTilesS][4];
#pragma unroll
        for (int i = 0; i < kMaxTilesS; ++i) {  // all loads first
            const float *p0 = X + ((((size_t)(n0 + ti.g[i]) * 2) * T + ti.t[i]) * C + 4 * q);
            const bool on = i < ti.n;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                pa[i][e] = on ? *reinterpret_cast<const f32x4 *>(p0 + (size_t)(e >> 1) * T * C + 16 * (e & 1)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < kMaxTilesS; ++i) {
            ti.r0[i] = pa[i][0] + pa[i][2] + bl;
            ti.r1[i] = pa[i][1] + pa[i][3] + bh;
        }
    } else {
        bf16x8 *w0s = reinterpret_cast<bf16x8 *>(h1);  // layer-0 A operands staged in the not-yet-used images of buffer 1 (43 KB)
        const int nW0 = pi.steps0 * 2 * 64;
        for (int i = threadIdx.x; i < nW0; i += blockDim.x) {
            w0s[i] = pk[pi.l0 + i];
            w0s[nW0 + i] = pk[pi.total + pi.l0 + i];
        }
        __syncthreads();
        const float *b0 = flat + off.w0_b;
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(b0 + 4 * q), bh = *reinterpret_cast<const f32x4 *>(b0 + 16 + 4 * q);
#pragma unroll
        for (int it = 0; it < kMaxTilesS; ++it) {
            ti.r0[it] = bl, ti.r1[it] = bh;
            if (it >= ti.n) continue;
            const float *xr = X + ((size_t)n0 * T + ti.g[it] * T + ti.t[it]) * a.F + 8 * q;
            f32x4 xv[16];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {  // all loads of the tile first (8 steps x 2 float4)
                const int f = 32 * s2 + 8 * q;
                const bool in = s2 < pi.steps0 && f + 7 < a.F && (a.F & 3) == 0;
                xv[2 * s2] = in ? *reinterpret_cast<const f32x4 *>(xr + 32 * s2) : f32x4{0.f, 0.f, 0.f, 0.f};
                xv[2 * s2 + 1] = in ? *reinterpret_cast<const f32x4 *>(xr + 32 * s2 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (!in && s2 < pi.steps0) {  // ragged tail / unaligned feature count: scalar, bounds-checked
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = (f + i < a.F) ? xr[32 * s2 + i] : 0.f;
                        if (i < 4) xv[2 * s2][i] = v;
                        else xv[2 * s2 + 1][i - 4] = v;
                    }
                }
            }
            f32x4 c0 = bl, c1 = bh;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                if (s2 >= pi.steps0) break;
                bf16x8 xh8, xl8;
                split8(xv[2 * s2], xv[2 * s2 + 1], xh8, xl8);
                c0 = product3(w0s[(s2 * 2 + 0) * 64 + lane], w0s[nW0 + (s2 * 2 + 0) * 64 + lane], xh8, xl8, c0);
                c1 = product3(w0s[(s2 * 2 + 1) * 64 + lane], w0s[nW0 + (s2 * 2 + 1) * 64 + lane], xh8, xl8, c1);
            }
            ti.r0[it] = c0, ti.r1[it] = c1;
        }
        __syncthreads();  // every wave is done with the staged layer-0 operands: buffer 1 is free for block 0's output
    }
#pragma unroll
    for (int i = 0; i < kMaxTilesS; ++i) {  // the accumulator layout IS the k' order: positions 8q .. 8q+7 of the row
        if (i >= ti.n) break;
        bf16x8 oh, ol;
        split8(ti.r0[i], ti.r1[i], oh, ol);
        *reinterpret_cast<bf16x8 *>(h0 + ti.wo[i]) = oh;
        *reinterpret_cast<bf16x8 *>(l0 + ti.wo[i]) = ol;
    }

    // the all-zero row of every image (what out-of-range taps read): written HERE, behind the layer-0 phase, whose operand staging
    // covers buffer 1's images including its zero row; the first block's barrier orders it
    for (int i = threadIdx.x; i < kRS; i += blockDim.x) {
        const __bf16 z = (__bf16)0.0f;
        h0[(size_t)ZR * kRS + i] = z, l0[(size_t)ZR * kRS + i] = z, h1[(size_t)ZR * kRS + i] = z, l1[(size_t)ZR * kRS + i] = z;
    }
    // ---- the residual blocks ----------------------------------------------------------------------------------------------------
    __bf16 *xh = h0, *xl = l0, *yh = h1, *yl = l1;
    {
        for (int blk = 0; blk < a.n_blocks; ++blk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of block blk's operands have landed ...
            __syncthreads();                                  // ... everybody's have, and block blk - 1 is complete in the images
            if (blk + 1 < a.n_blocks && !(a.tune & 2)) stage(blk + 1);
            const bf16x8 *w = wslot + (size_t)(blk & 1) * 2 * kSlotOps;
            TrainBlk tb{nullptr, nullptr, 0, 0, GR};
            if constexpr (TRAIN) {
                if (!tio.acts_last_only) save_acts(blk);
                tb.ustride = a.n_blocks * T * C, tb.dstride = a.n_blocks * C;
                tb.upre = tio.upre ? tio.upre + (size_t)n0 * tb.ustride + (size_t)blk * T * C : nullptr;
                tb.drop = tio.drop_tcn ? tio.drop_tcn + (size_t)n0 * tb.dstride + (size_t)blk * C : nullptr;
            }
            if (!(a.tune & 1)) run_block_split<kMaxTilesS, TRAIN>(w, lane, 1 << (blk % a.n_dil), T, ZR, ti, q, xh, xl, yh, yl, tb);
            __bf16 *th = xh, *tl = xl;
            xh = yh, xl = yl, yh = th, yl = tl;
        }
    }
    save_acts(a.n_blocks);  // (training) the pre-relu TCN output
    // final relu of the TCN output (f32, on the residual registers) -> split images of the Dense layers' input: over this lane's OWN
    // rows of the last block's output (nobody reads them before the barrier below; the other buffer may still be being read)
#pragma unroll
    for (int i = 0; i < kMaxTilesS; ++i) {
        if (i >= ti.n) break;
        f32x4 v0 = ti.r0[i], v1 = ti.r1[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) v0[r] = fmaxf(v0[r], 0.f), v1[r] = fmaxf(v1[r], 0.f);
        bf16x8 oh, ol;
        split8(v0, v1, oh, ol);
        *reinterpret_cast<bf16x8 *>(xh + ti.wo[i]) = oh;
        *reinterpret_cast<bf16x8 *>(xl + ti.wo[i]) = ol;
    }
    __syncthreads();

    // ---- Dense layers on the flattened trunk: D[o][g] = sum_t sum_c WhT[o][t*32+c] x[g][t][c] ------------------------
    // Every wave takes the frames t = wave, wave + nw, ... with ALL M-tiles of outputs (the 3C logits and the heads' Dense(16)s: up to
    // five tiles of 16): the frame's B operand is read once, the 2 KB of operands per (frame, M-tile) stream from L2 -- 680 KB per
    // workgroup, the L2 -> CU rate bounds this phase -- and the waves' partial sums are added in wave order (fixed: same bits
    // whatever the batch).  (With one M-tile per wave, five of eight waves worked and each walked all 68 frames.)
    float *pre = scratch;  // [nw][G][kPS] partial sums, then [G][kPS]
    {
        constexpr int kMaxMt = 5;
        f32x4 acc[kMaxMt];
#pragma unroll
        for (int mt = 0; mt < kMaxMt; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool live = j < g_here;
        const unsigned rowg = live ? (unsigned)j * T : (unsigned)ZR;
        const bf16x8 *wa = pk + pi.heads + lane;
        for (int t = wave; t < T; t += nw) {
            const unsigned row = (live ? rowg + t : rowg) * kRS + 8 * q;
            const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(xh + row);
            const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(xl + row);
            bf16x8 wh[kMaxMt], wl[kMaxMt];
#pragma unroll
            for (int mt = 0; mt < kMaxMt; ++mt) {  // all loads of the frame first
                const size_t o = ((size_t)min(mt, a.n_mt - 1) * T + t) * 64;
                wh[mt] = wa[o], wl[mt] = wa[pi.total + o];
            }
#pragma unroll
            for (int mt = 0; mt < kMaxMt; ++mt)
                if (mt < a.n_mt) acc[mt] = product3(wh[mt], wl[mt], bh, bl, acc[mt]);
        }
        if (live) {
#pragma unroll
            for (int mt = 0; mt < kMaxMt; ++mt)
                if (mt < a.n_mt) *reinterpret_cast<f32x4 *>(pre + ((size_t)wave * a.G + j) * kPS + 16 * mt + 4 * q) = acc[mt];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g_here * kPS; i += blockDim.x) {
        float v = pre[i];
        for (int k2 = 1; k2 < nw; ++k2) v += pre[(size_t)k2 * a.G * kPS + i];
        pre[i] = v;
    }
    __syncthreads();
    // ---- BN / relu / output Dense / activations (f32), as in the f32 kernel -------------------------------------------
    auto bias_of = [&](int o) {
        if (o < a.n_classes) return flat[off.c3_b + o];
        const int h = (o - a.n_classes) / kHidden, jj = (o - a.n_classes) % kHidden;
        return flat[off.head[h] + (size_t)a.D * kHidden + jj];
    };
    const int tid = threadIdx.x;
    if constexpr (TRAIN) {  // training: the batch-statistics heads run in smh_train.hip on `pre` (incl. bias; padding columns zero)
        for (int i = tid; i < g_here * kPS; i += blockDim.x) {
            const int p = i / kPS, o = i - p * kPS;
            tio.pre[(size_t)(n0 + p) * kPS + o] = o < a.NH ? pre[p * kPS + o] + bias_of(o) : 0.f;
        }
        return;
    }
    if (tid < g_here * a.n_heads) {
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
            const int o = a.n_classes + h * kHidden + i;
            float v = pre[p * kPS + o] + bias_of(o);
            v = (v - mean[i]) / sqrtf(var[i] + kBnEps);
            v = v * gamma[i] + beta[i];
            hid[i] = fmaxf(v, 0.f);
        }
        for (int c = 0; c < od; ++c) {
            float s2 = bo[c];
#pragma unroll
            for (int i = 0; i < kHidden; ++i) s2 = fmaf(hid[i], wo[i * od + c], s2);
            if (a.head_sigmoid[h]) s2 = 1.0f / (1.0f + expf(-s2));
            out[(size_t)(n0 + p) * a.out_dim + col + c] = s2;
        }
    } else if (tid >= 128 && tid < 128 + g_here) {
        const int p = tid - 128;
        float mxl = -INFINITY;
        for (int c = 0; c < a.n_classes; ++c) mxl = fmaxf(mxl, pre[p * kPS + c] + bias_of(c));
        float den = 0.f;
        for (int c = 0; c < a.n_classes; ++c) den += expf(pre[p * kPS + c] + bias_of(c) - mxl);
        const int col = a.out_dim - a.n_classes;
        for (int c = 0; c < a.n_classes; ++c)
            out[(size_t)(n0 + p) * a.out_dim + col + c] = expf(pre[p * kPS + c] + bias_of(c) - mxl) / den;
    }
}

}  // namespace

static int forward_bf16(smh_model *m, const float *d_x, int N, float *d_out, int split, int from_x0, void *stream,
                        const TrainIO *tio = nullptr) {
    SMH_REQUIRE(m && d_x && (d_out || tio), "smh_model_forward_bf16: null argument");
    SMH_REQUIRE(!tio || split, "smh_model_forward_bf16: the training forward exists for split operands only");
    SMH_REQUIRE(N >= 0, "smh_model_forward_bf16: N=%d", N);
    SMH_REQUIRE(m->cfg.block_variant == 0, "smh_model_forward_bf16: built for block_variant 0 only");
    SMH_REQUIRE(m->cfg.n_feat <= 256, "smh_model_forward_bf16: n_feat=%d exceeds the 256 features of the bf16 layer-0 tiling",
                m->cfg.n_feat);
    if (N == 0) return SMH_OK;
    hipStream_t st = (hipStream_t)stream;
    const PackInfo pi = pack_info(m);
    const Offsets off = offsets(m);
    if (!m->d_bf16) SMH_CHECK_HIP(hipMalloc(&m->d_bf16, 2 * pi.total * 16));  // hi operands, then lo operands
    if (m->bf16_version != m->version) {  // weights changed since the operands were built
        hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((pi.total + 255) / 256)), dim3(256), 0, st, m->d_flat, off, pi,
                           m->cfg.n_feat, m->cfg.patch_size, m->n_blocks, m->cfg.n_classes, m->n_heads, m->NH,
                           (bf16x8 *)m->d_bf16);
        int rc = smh::launch_status("pack_bf16_kernel");
        if (rc) return rc;
        m->bf16_version = m->version;
    }
    TcnArgs a;
    size_t lds;
    fill_args(m, N, &a, &lds);
    a.from_x0 = from_x0;
    if (const char *ev = smh::probe_env("SMH_TCN_BLOCKS")) a.n_blocks = atoi(ev);  // timing probe (outputs invalid): fewer residual blocks
    if (const char *ev = smh::probe_env("SMH_TCN_TUNE")) a.tune = atoi(ev);  // timing probes: 1 = no tile work, 2 = no operand staging
    SMH_REQUIRE(lds <= 156 * 1024, "patch_size %d too long for the LDS-resident TCN", a.T);
    if (split) {
        const int units_s = (std::min(a.G, N) * a.T + 15) / 16;
        const bool two = units_s <= 24;  // at most two column tiles per wave on up to 12 waves
        int nwaves_s = two ? std::min(12, std::max(8, (units_s + 1) / 2)) : 8;
        if (const char *ev = getenv("SMH_BF16_NW")) nwaves_s = two ? std::min(12, std::max((units_s + 1) / 2, atoi(ev))) : 8;  // tuning
        // residual image + four split images (hi / lo of two buffers, a zero row each)
        SMH_REQUIRE(a.n_mt <= 5, "smh_model_forward_bf16: more than five M-tiles of Dense-on-trunk outputs");
        const size_t lds_s = 4 * (size_t)(a.GRP + 1) * kRS * sizeof(__bf16) + 2 * 2 * kSlotOps * 16 + sizeof(float) * (size_t)nwaves_s * a.G * kPS;
        SMH_REQUIRE((units_s + nwaves_s - 1) / nwaves_s <= (two ? 2 : 4), "smh_model_forward_bf16: too many column tiles per wave");
        SMH_REQUIRE(lds_s <= 156 * 1024, "patch_size %d too long for the LDS-resident split-bf16 TCN", a.T);
        SMH_REQUIRE((size_t)pi.steps0 * 2 * 64 * 2 * 16 <= 2 * (size_t)(a.GRP + 1) * kRS * sizeof(__bf16) || from_x0,
                    "smh_model_forward_bf16: n_feat=%d too wide for the layer-0 operand staging", m->cfg.n_feat);
        const TrainIO io = tio ? *tio : TrainIO{nullptr, nullptr, nullptr, nullptr, 0};
#define SMH_LAUNCH_BF16S(MT, TR)                                                                                                       \
    do {                                                                                                                              \
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)b3mtl_forward_bf16s_kernel<MT, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)lds_s));                                                                               \
        hipLaunchKernelGGL((b3mtl_forward_bf16s_kernel<MT, TR>), dim3((N + a.G - 1) / a.G), dim3(64 * nwaves_s), lds_s, st, a, pi, off,  \
                           d_x, m->d_flat, (const bf16x8 *)m->d_bf16, m->d_hp, d_out, io);                                            \
    } while (0)
        if (two) {
            if (tio) SMH_LAUNCH_BF16S(2, true);
            else SMH_LAUNCH_BF16S(2, false);
        } else {
            if (tio) SMH_LAUNCH_BF16S(4, true);
            else SMH_LAUNCH_BF16S(4, false);
        }
#undef SMH_LAUNCH_BF16S
        return smh::launch_status("b3mtl_forward_bf16s_kernel");
    } else {
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)b3mtl_forward_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(b3mtl_forward_bf16_kernel<false>, dim3((N + a.G - 1) / a.G), dim3(512), lds, st, a, pi, off, d_x, m->d_flat,
                           (const bf16x8 *)m->d_bf16, m->d_hp, d_out);
    }
    return smh::launch_status("b3mtl_forward_bf16_kernel");
}

namespace smh_tcn {
// the training-mode forward on split bf16 operands (smh_train_step_f32 when the trainer's dtype is bf16): same TrainIO as launch_forward
int launch_forward_bf16_train(smh_model *m, const float *d_x, int N, const TrainIO *tio, hipStream_t st) {
    return forward_bf16(m, d_x, N, nullptr, 1, 0, (void *)st, tio);
}
}  // namespace smh_tcn

extern "C" int smh_model_forward_bf16_ex(smh_model *m, const float *d_x, int N, float *d_out, int split, void *stream) {
    return forward_bf16(m, d_x, N, d_out, split, 0, stream);
}

extern "C" int smh_model_forward_bf16(smh_model *m, const float *d_x, int N, float *d_out, void *stream) {
    return forward_bf16(m, d_x, N, d_out, 1, 0, stream);
}

extern "C" int smh_model_forward_x0_bf16(smh_model *m, const float *d_x0p, int N, float *d_out, int split, void *stream) {
    return forward_bf16(m, d_x0p, N, d_out, split, 1, stream);
}
