// Shared tail of the B3_MTL forward kernels (smh_tcn.hip: the keras-tcn 2.3.x block; smh_tcn_v2.hip: the two-convolution
// block): the Dense layers on the flattened trunk and the MTL heads (lib/proposed_architectures.py:148-154, 25-80).
// Called by every thread of the workgroup with xin = the relu'd trunk (rows g*T + t, stride SX) in LDS and xout = a free
// LDS buffer of at least 8*4*128 + kMaxG*kPS floats.
#pragma once
#include <type_traits>

#include "smh_model.h"

namespace smh_tcn {

template <bool TRAIN>
__device__ __forceinline__ void dense_and_heads(const TcnArgs &a, const float *xin, float *xout, const float *__restrict__ WhA,
                                                const float *__restrict__ hp, float *__restrict__ out, const TrainIO &tio,
                                                int n0, int g_here) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int T = a.T;
    // ---- Dense layers on the flattened trunk: pre[g][o] = sum_k flat[g][k] * Wh[k][o], k = t*32 + c, o < NH (51 or 69) ----
    // On the VALU, not on the matrix cores: with 4 patches per workgroup a 16-column MFMA tile is 75 % padding, and exact-f32
    // MFMA has the same peak as v_fma_f32 -- the padded product cost 4x the arithmetic (and ran at 17 us per launch,
    // waiting for its weights).  Here a lane owns one output (two when NH > 64), a wave owns a range of frames; per frame
    // it streams 32 rows of Wh (256 contiguous bytes per row and wave) and multiplies them with the four patches'
    // activations, which are wave-uniform LDS reads.  The waves' partial sums meet in LDS.
    const int OPL = (a.NH + 63) >> 6;           // outputs per lane
    const int ld = OPL * 64;                    // row length of the packed weights Wh[k][ld]
    // The frames are cut into kDenseParts fixed ranges (a function of T alone) and the ranges are summed in a fixed order, so
    // a patch gets the same bits whatever the batch size, the patches per workgroup or the number of waves.
    constexpr int kDenseParts = 8;
    float *part = xout;                         // [kDenseParts][4][ld] partial sums, then pre[kMaxG][kPS] behind them
    float *pre = xout + (size_t)kDenseParts * 4 * ld;
    const int rpp = (T + kDenseParts - 1) / kDenseParts;
    auto dense_on_trunk = [&](auto opl_c) {
        constexpr int kOPL = decltype(opl_c)::value;
        for (int g0 = 0; g0 < g_here; g0 += 4) {
            const float *xg[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[g] = xin + (size_t)min(g0 + g, g_here - 1) * T * SX;
            for (int pt = wave; pt < kDenseParts; pt += nw) {
                const int t_lo = min(T, pt * rpp), t_hi = min(T, t_lo + rpp);
                float acc[4][kOPL];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int p = 0; p < kOPL; ++p) acc[g][p] = 0.f;
                for (int t = t_lo; t < t_hi; ++t) {
                    // weights: [k / 4][ld][4] (pack_host): the four channels 4 h .. 4 h + 3 of frame t for this lane's output
                    const f32x4 *wrow = reinterpret_cast<const f32x4 *>(WhA) + (size_t)t * (C / 4) * ld + lane;
#pragma unroll 2
                    for (int c8 = 0; c8 < 4; ++c8) {
                        f32x4 wa[kOPL], wb[kOPL];
#pragma unroll
                        for (int p = 0; p < kOPL; ++p) {
                            wa[p] = wrow[(size_t)(2 * c8) * ld + 64 * p];
                            wb[p] = wrow[(size_t)(2 * c8 + 1) * ld + 64 * p];
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 xa = *reinterpret_cast<const f32x4 *>(xg[g] + (size_t)t * SX + 8 * c8);
                            const f32x4 xb = *reinterpret_cast<const f32x4 *>(xg[g] + (size_t)t * SX + 8 * c8 + 4);
#pragma unroll
                            for (int c = 0; c < 4; ++c)
#pragma unroll
                                for (int p = 0; p < kOPL; ++p) {  // (the summation order of the [k][ld] layout: bit-identical results)
                                    acc[g][p] = fmaf(xa[c], wa[p][c], acc[g][p]);
                                    acc[g][p] = fmaf(xb[c], wb[p][c], acc[g][p]);
                                }
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int p = 0; p < kOPL; ++p) part[((size_t)pt * 4 + g) * ld + 64 * p + lane] = acc[g][p];
            }
            __syncthreads();
            for (int i = threadIdx.x; i < 4 * ld; i += blockDim.x) {  // ordered sum over the parts: deterministic
                const int g = i / ld, o = i - g * ld;
                if (g0 + g < g_here && o < kPS) {
                    float v = 0.f;
#pragma unroll
                    for (int pt = 0; pt < kDenseParts; ++pt) v += part[((size_t)pt * 4 + g) * ld + o];
                    pre[(g0 + g) * kPS + o] = v;
                }
            }
            __syncthreads();
        }
    };
    if (!a.skip_heads) {
        if (OPL == 1) dense_on_trunk(std::integral_constant<int, 1>{});
        else dense_on_trunk(std::integral_constant<int, 2>{});
    }
    // ---- BN / relu / output Dense / activations: one thread per (patch, head), one per patch for 3C ----
    const float *bh = WhA + (size_t)a.D * ld;  // NH biases follow the packed weights
    if constexpr (TRAIN) {  // training: the batch-statistics heads run in smh_train.hip on `pre`
        for (int i = threadIdx.x; i < g_here * kPS; i += blockDim.x) {
            const int p = i / kPS, o = i - p * kPS;
            tio.pre[(size_t)(n0 + p) * kPS + o] = o < a.NH ? pre[p * kPS + o] + bh[o] : 0.f;
        }
        return;
    }
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
            float v = pre[p * kPS + o] + bh[o];
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
        for (int c = 0; c < a.n_classes; ++c) mxl = fmaxf(mxl, pre[p * kPS + c] + bh[c]);
        float den = 0.f;
        for (int c = 0; c < a.n_classes; ++c) den += expf(pre[p * kPS + c] + bh[c] - mxl);
        const int col = a.out_dim - a.n_classes;
        for (int c = 0; c < a.n_classes; ++c)
            out[(size_t)(n0 + p) * a.out_dim + col + c] = expf(pre[p * kPS + c] + bh[c] - mxl) / den;
    }
}

}  // namespace smh_tcn
