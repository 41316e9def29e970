// B3_MTL inference forward with bf16 matrix-core operands (BASELINE config 5: "mixed bf16 CNN + fp32 HPSS").
// Same graph, same transposed-product mapping and the same LDS-resident f32 activation stream as smh_tcn.hip; only
// the MFMA operands are rounded to bf16 (v_mfma_f32_16x16x32_bf16, f32 accumulation): every weight once when the
// operand buffer is (re)built, every activation right before it is multiplied.  Bias, residual sum, relu, the
// channel-max normalisation, BatchNorm and the output activations stay f32.  Not bit-compatible with the f32 path
// (stated tolerance in tests/test_bf16_gpu.py); `smh_model_forward_f32` remains the parity path.
//
// k orders (one MFMA step covers 32 k values, lane (q = l>>4) supplies k = 8q .. 8q+7):
//   layer 0   k = feature, 8 steps for 240 features (zero padded to 256)
//   conv      one step per tap, k = input channel
//   1x1 conv  k' = the channel order of the accumulator registers (4q+r, then 16+4q+r), so the normalised
//             activations go from registers straight into the B operand
//   heads     one step per frame t, k = channel;  D[output][patch]
#include <cstdint>

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
    p.blk_stride = 8 * 64;  // 3 taps x 2 M-tiles + 2 M-tiles of the 1x1
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
    } else if (idx < pi.heads) {  // blocks: [blk][tap*2 + mt | 6 + mt][lane]
        const size_t r = idx - pi.blk0;
        const int blk = (int)(r / pi.blk_stride), e = (int)((r % pi.blk_stride) >> 6);
        const size_t wo = off.blk0 + (size_t)blk * off.blk_stride;
        if (e < 6) {
            const int tap = e >> 1, mt = e & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) vf[i] = flat[wo + ((size_t)tap * C + 8 * q + i) * C + 16 * mt + j];
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
            const size_t k = (size_t)t * C + 8 * q + i;
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
            const float *src = xin + (size_t)(ok ? Rc + off : Rc) * SX + 8 * q;
            Op b = split_op<SPLIT>(*reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
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
            const float *xg = xin + (size_t)(live ? j : 0) * T * SX + 8 * q;
            for (int t = t_lo; t < t_hi; ++t) {
                const float *xr = xg + (size_t)t * SX;
                Op b = split_op<SPLIT>(*reinterpret_cast<const f32x4 *>(xr), *reinterpret_cast<const f32x4 *>(xr + 4));
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

}  // namespace

static int forward_bf16(smh_model *m, const float *d_x, int N, float *d_out, int split, int from_x0, void *stream) {
    SMH_REQUIRE(m && d_x && d_out, "smh_model_forward_bf16: null argument");
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
    SMH_REQUIRE(lds <= 156 * 1024, "patch_size %d too long for the LDS-resident TCN", a.T);
    if (split) {
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)b3mtl_forward_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(b3mtl_forward_bf16_kernel<true>, dim3((N + a.G - 1) / a.G), dim3(512), lds, st, a, pi, off, d_x, m->d_flat,
                           (const bf16x8 *)m->d_bf16, m->d_hp, d_out);
    } else {
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)b3mtl_forward_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(b3mtl_forward_bf16_kernel<false>, dim3((N + a.G - 1) / a.G), dim3(512), lds, st, a, pi, off, d_x, m->d_flat,
                           (const bf16x8 *)m->d_bf16, m->d_hp, d_out);
    }
    return smh::launch_status("b3mtl_forward_bf16_kernel");
}

extern "C" int smh_model_forward_bf16_ex(smh_model *m, const float *d_x, int N, float *d_out, int split, void *stream) {
    return forward_bf16(m, d_x, N, d_out, split, 0, stream);
}

extern "C" int smh_model_forward_bf16(smh_model *m, const float *d_x, int N, float *d_out, void *stream) {
    return forward_bf16(m, d_x, N, d_out, 1, 0, stream);
}

extern "C" int smh_model_forward_x0_bf16(smh_model *m, const float *d_x0p, int N, float *d_out, int split, void *stream) {
    return forward_bf16(m, d_x0p, N, d_out, split, 1, stream);
}
