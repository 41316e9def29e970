// a14: one training step of B3_MTL -- what `model.fit` runs per batch for the model compiled at
// lib/proposed_architectures.py:156-165 (losses S,M(,N): binary_crossentropy, R: mean_squared_error,
// 3C: categorical_crossentropy, l2(0.01) on the Dense(16) kernels, SGD(momentum 0.9, clipnorm=1)).
// Arithmetic restated in oracle/b3_mtl_train.py (pinned against torch autograd on the CPU).
//
//   smh_train_step_f32      = forward (MFMA kernel of smh_tcn.hip, TRAIN variant: saves every block input,
//                             applies the SpatialDropout1D masks) -> heads_train_kernel (batch-statistics
//                             BN, Dropout, losses, d loss / d pre) -> tcn_backward_kernel.
//   smh_trainer_apply_sgd_f32 = per-tensor clipnorm, momentum, BN moving averages, device-side repack.
// Round-1 scope: CORRECT and device-resident.  The backward kernel recomputes each block from its saved
// input with plain VALU loops in LDS and accumulates weight gradients with float atomics; moving it onto
// the MFMA tiling of the forward kernel is the obvious next step.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <type_traits>

#include "smh_model.h"
#include "smh_train_bwd.h"

using namespace smh_tcn;

namespace {

constexpr float kKerasEps = 1e-7f;
constexpr float kBnMomentum = 0.99f;
constexpr float kL2 = 0.01f;
constexpr int kBG = 1;        // patches per workgroup in the backward kernel
constexpr int kBS = 33;       // LDS row stride (floats) of the backward buffers
constexpr int kBThreads = 1024;

using smh_tcn::HeadsArgs;

// Sum over the 64 lanes, result in every lane, on the VALU's lane-permute paths: four DPP steps inside each row of 16 lanes
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then gfx950's v_permlane32_swap / v_permlane16_swap
// across the rows.  (__shfl_xor is a ds_bpermute per step -- an LDS round trip, six of them in a dependent chain: the
// heads kernel makes ~350 of these sums per step and spent most of its time in them.)
__device__ __forceinline__ float wave_sum_f(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror: every lane holds the sum of its row of 16
    const unsigned u = __float_as_uint(v);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    const unsigned u2 = __float_as_uint(v);
    const auto r16 = __builtin_amdgcn_permlane16_swap(u2, u2, false, false);
    return __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
}
// sum over the wave, added to THIS wave's own accumulator row (all lanes must call it; idle lanes pass 0): a plain
// read-add-write by lane 0 -- no other wave touches the row, so no float atomic (ds_add_f32 is slow and orders nothing);
// the rows are added up in wave order afterwards, which also makes the sums reproducible
// first = this wave's first contribution to slot q: a plain store (the read-add-write chain is an LDS round trip per sum)
__device__ __forceinline__ void wave_add(float *row, int q, float v, bool first = false) {
    v = wave_sum_f(v);
    if ((threadIdx.x & 63) == 0) row[q] = first ? v : row[q] + v;
}
// accumulator slots of heads_train_kernel (per wave, then totals)
enum { Q_SUM1 = 0, Q_SUM2 = 64, Q_DGAMMA = 128, Q_DBETA = 192, Q_DBIAS = 256, Q_DWO = 320, Q_DBO = Q_DWO + smh_tcn::kMaxHeads * smh_tcn::kHidden * 3,
       Q_LOSS = Q_DBO + smh_tcn::kMaxHeads * 3, Q_DB3 = Q_LOSS + smh_tcn::kMaxHeads + 2, Q_ACC = Q_DB3 + 8, kQ = Q_ACC + smh_tcn::kMaxHeads };

// One workgroup per output head (blockIdx.x < n_heads: the Dense(16) -> BN -> relu -> Dropout -> Dense head of that index;
// blockIdx.x == n_heads: the '3C' softmax): the heads share nothing but the total loss -- BatchNorm's batch statistics are per
// hidden unit -- so each stages, reduces and writes only its own 16 (or n_classes) columns, and the last workgroup to arrive
// (ticket) adds the weighted losses in a fixed order.  As ONE workgroup running the heads one after the other this kernel was
// 79 us of a 510-patch step on an otherwise idle chip.
// Inside a workgroup: the batch-statistics part of the network is tiny (N x 51 values).  Every reduction over the
// batch is spread over the lanes (16 lanes per hidden unit for the statistics; one head at a time with wave-level
// sums for the gradients of the small tensors) -- a thread looping over the whole batch was 60 % of this kernel.
// STAGED (the batch fits: N <= kHeadsStageMax): `pre` and the targets are copied to LDS once, coalesced and all in flight,
// and every later access -- most of them with lane = sample, i.e. 320-byte strides in global memory, 64 cache lines per
// wave load -- is an LDS read at an odd stride; d loss / d pre is built in place in that tile and leaves in one coalesced
// copy; dxh is kept transposed ([unit][sample]: coalesced both ways).  One workgroup on one CU is latency-bound: the
// global-memory form of this kernel took 181 us for 510 patches.
// THREADS: 512 when the batch fits (one sample per lane in the per-head pass, 256 VGPRs: at 1024 threads the pass lives on
// 128 VGPRs and spills 46 of them into its inner loops), else 1024.
template <bool STAGED, int THREADS>
__global__ void __launch_bounds__(THREADS)
heads_train_kernel(HeadsArgs a, const float *__restrict__ pre, const float *__restrict__ y, const float *__restrict__ hp,
                   const float *__restrict__ drop, float *__restrict__ dpre, float *__restrict__ dxh,
                   float *__restrict__ grad, float *__restrict__ bnstat, float *__restrict__ losses, unsigned *__restrict__ ticket) {
    extern __shared__ __attribute__((aligned(16))) float stage[];
    __shared__ float s_mean[64], s_inv[64];
    __shared__ float red[16][kQ], tot[kQ];  // per-wave accumulator rows (blockDim.x <= 1024), their totals
    // non-STAGED path only: the Dense(16) bias gradients, summed on a 2^-36 grid with integer atomics (order-independent: a float
    // ds_add here made that tensor differ from run to run for batches too large for the LDS tile)
    __shared__ unsigned long long dbq[64];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (!STAGED && tid < 64) dbq[tid] = 0ull;
    float *row = red[tid >> 6];
    unsigned long long tk[8];
    int ntk = 0;
    auto stamp = [&]() { if (a.stamps && tid == 0 && blockIdx.x == 0 && ntk < 8) tk[ntk++] = __builtin_amdgcn_s_memrealtime(); };
    stamp();
    auto totals = [&]() {  // every wave's row -> tot, in wave order; called by all threads between barriers
        __syncthreads();
        for (int q = tid; q < kQ; q += nt) {
            float v = 0.f;
            for (int w = 0; w < (nt >> 6); ++w) v += red[w][q];
            tot[q] = v;
        }
        __syncthreads();
    };
    const int N = a.N, ncls = a.n_classes, nh = a.n_heads, NJ = nh * kHidden, NHc = ncls + NJ;
    const int role = blockIdx.x;  // < nh: head `role`; == nh: the '3C' softmax
    const bool is_cls = role == nh;
    const int j_lo = is_cls ? 0 : role * kHidden, j_hi = is_cls ? 0 : j_lo + kHidden;  // this workgroup's hidden units
    const int c_lo = is_cls ? 0 : ncls + j_lo, c_n = is_cls ? ncls : kHidden;           // ... and its columns of pre / dpre
    const int PST = NHc | 1, YST = a.out_dim | 1;  // odd row strides: lane = sample reads are conflict-free
    float *tile = stage, *ty = stage + (size_t)(STAGED ? N : 0) * PST;
    auto P = [&](int n, int c) -> float { return STAGED ? tile[n * PST + c] : pre[(size_t)n * kPS + c]; };
    auto Y = [&](int n, int c) -> float { return STAGED ? ty[n * YST + c] : y[(size_t)n * a.out_dim + c]; };
    if constexpr (STAGED) {
        for (int i = tid; i < N * c_n; i += nt) {
            const int n = i / c_n, c = c_lo + (i - n * c_n);
            tile[n * PST + c] = pre[(size_t)n * kPS + c];
        }
        for (int i = tid; i < N * a.out_dim; i += nt) {
            const int n = i / a.out_dim, c = i - n * a.out_dim;
            ty[n * YST + c] = y[i];
        }
    }
    for (int i = tid; i < 16 * kQ; i += nt) (&red[0][0])[i] = 0.f;
    if constexpr (STAGED) __syncthreads();
    stamp();  // staged
    // A: batch statistics of every hidden unit (population variance, two passes): 16 lanes per unit
    for (int j0 = j_lo; j0 < j_hi; j0 += nt >> 4) {
        const int j = j0 + (tid >> 4), sub = tid & 15;
        const bool on = j < j_hi;
        float s = 0.f;
        if (on)
            for (int n = sub; n < N; n += 16) s += P(n, ncls + j);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)N;
        float q = 0.f;
        if (on)
            for (int n = sub; n < N; n += 16) {
                const float d = P(n, ncls + j) - mean;
                q += d * d;
            }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
        if (on && sub == 0) {
            const float var = q / (float)N;
            s_mean[j] = mean;
            s_inv[j] = 1.0f / sqrtf(var + kBnEps);
            const int h = j / kHidden, i = j % kHidden;
            bnstat[h * 32 + i] = mean;
            bnstat[h * 32 + 16 + i] = var;
        }
    }
    __syncthreads();
    stamp();  // A
    // B: one head at a time, lanes over the samples: forward through BN / relu / dropout / output Dense, loss,
    // gradients; sums over the batch are wave-reduced before they touch LDS
    for (int h = role; h < (is_cls ? role : role + 1); ++h) {
        const float *ph = hp + a.hp_off[h];
        int col = 0;
        for (int k = 0; k < h; ++k) col += a.head_odim[k];
        const float *gamma = ph, *beta = ph + 16, *wo = ph + 64;
        const int od = a.head_odim[h];
        const float *bo = wo + kHidden * od;
        for (int n0 = 0; n0 < N; n0 += nt) {
            const int n = n0 + tid;
            const bool on = n < N;
            const int nc = on ? n : N - 1;
            float xh[kHidden], bn[kHidden], ad[kHidden], dm[kHidden];
            float zo[3] = {0.f, 0.f, 0.f};
            for (int c = 0; c < od; ++c) zo[c] = bo[c];
            if (drop) {  // this sample's 16 masks: four float4 (64 contiguous bytes)
                const float4 *dp = reinterpret_cast<const float4 *>(drop + ((size_t)nc * nh + h) * kHidden);
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const float4 v = dp[i4];
                    dm[4 * i4] = v.x, dm[4 * i4 + 1] = v.y, dm[4 * i4 + 2] = v.z, dm[4 * i4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < kHidden; ++i) dm[i] = 1.0f;
            }
#pragma unroll
            for (int i = 0; i < kHidden; ++i) {
                const int j = h * kHidden + i;
                xh[i] = (P(nc, ncls + j) - s_mean[j]) * s_inv[j];
                bn[i] = xh[i] * gamma[i] + beta[i];
                ad[i] = fmaxf(bn[i], 0.f) * dm[i];
                for (int c = 0; c < od; ++c) zo[c] = fmaf(ad[i], wo[i * od + c], zo[c]);
            }
            float dzo[3] = {0.f, 0.f, 0.f}, lsum = 0.f, hit = 0.f;
            for (int c = 0; c < od; ++c) {
                const float t = Y(nc, col + c);
                if (a.head_sigmoid[h]) {
                    const float o = 1.0f / (1.0f + expf(-zo[c]));
                    hit += ((o > 0.5f) == (t > 0.5f)) ? 1.0f : 0.f;  // Keras binary_accuracy, threshold 0.5
                    const float oc = fminf(fmaxf(o, kKerasEps), 1.0f - kKerasEps);
                    lsum += -(t * logf(oc + kKerasEps) + (1.0f - t) * logf(1.0f - oc + kKerasEps));
                    const bool inside = (o > kKerasEps) && (o < 1.0f - kKerasEps);
                    const float doc = -(t / (oc + kKerasEps) - (1.0f - t) / (1.0f - oc + kKerasEps)) / (float)(N * od);
                    dzo[c] = inside ? doc * o * (1.0f - o) : 0.f;
                } else {
                    const float d = zo[c] - t;
                    lsum += d * d;
                    dzo[c] = 2.0f * d / (float)(N * od);
                }
                dzo[c] = on ? dzo[c] * a.lw[h] : 0.f;
                wave_add(row, Q_DBO + h * 3 + c, dzo[c], n0 == 0);
            }
            wave_add(row, Q_LOSS + h, on ? lsum / (float)(N * od) : 0.f, n0 == 0);
            wave_add(row, Q_ACC + h, on ? hit / (float)(N * od) : 0.f, n0 == 0);
#pragma unroll
            for (int i = 0; i < kHidden; ++i) {
                const int j = h * kHidden + i;
                float da = 0.f;
                for (int c = 0; c < od; ++c) {
                    da = fmaf(dzo[c], wo[i * od + c], da);
                    wave_add(row, Q_DWO + (h * kHidden + i) * 3 + c, ad[i] * dzo[c], n0 == 0);
                }
                const float dbn = (on && bn[i] > 0.f) ? da * dm[i] : 0.f;
                const float dxhat = dbn * gamma[i];
                if (on) {
                    if constexpr (STAGED) dxh[(size_t)j * N + n] = dxhat;  // [unit][sample]
                    else dxh[(size_t)n * kPS + j] = dxhat;
                }
                wave_add(row, Q_DGAMMA + j, dbn * xh[i], n0 == 0);
                wave_add(row, Q_DBETA + j, dbn, n0 == 0);
            }
        }
    }
    totals();  // dbeta / dgamma of every unit; the BatchNorm backward's sums follow from them: dxhat = gamma dbn, so
    // sum_n dxhat = gamma dbeta and sum_n dxhat xhat = gamma dgamma (32 fewer wave sums per head)
    for (int j = j_lo + tid; j < j_hi; j += nt) {
        const float gm = hp[a.hp_off[j / kHidden] + (j % kHidden)];
        tot[Q_SUM1 + j] = gm * tot[Q_DBETA + j];
        tot[Q_SUM2 + j] = gm * tot[Q_DGAMMA + j];
    }
    __syncthreads();
    stamp();  // B
    if constexpr (STAGED) {
        // C: BN backward to the Dense(16) pre-activations, lanes over the samples of one unit; d replaces pre in the tile
        const int NR = ((N + 63) >> 6) << 6;  // whole waves per unit: the wave sum below needs every lane
        constexpr int kCI = 8;  // dxh values requested ahead per thread (one L2 trip per eight iterations instead of one each)
        const int NJr = j_hi - j_lo;
        for (int it0 = tid; it0 < NJr * NR; it0 += kCI * nt) {
            float dv[kCI];
#pragma unroll
            for (int e = 0; e < kCI; ++e) {
                const int it = it0 + e * nt, jr = it / NR, n = it - jr * NR, j = j_lo + jr;
                dv[e] = (it < NJr * NR && n < N) ? dxh[(size_t)j * N + n] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < kCI; ++e) {
                const int it = it0 + e * nt;
                if (it >= NJr * NR) break;  // wave-uniform (NR and nt are multiples of 64)
                const int jr = it / NR, n = it - jr * NR, j = j_lo + jr;
                float d = 0.f;
                if (n < N) {
                    const float xhat = (tile[n * PST + ncls + j] - s_mean[j]) * s_inv[j];
                    d = s_inv[j] / (float)N * ((float)N * dv[e] - tot[Q_SUM1 + j] - xhat * tot[Q_SUM2 + j]);
                    tile[n * PST + ncls + j] = d;
                }
                wave_add(row, Q_DBIAS + j, d, true);  // a wave meets a unit once: NR <= nt
            }
        }
    } else {
        // C: BN backward to the Dense(16) pre-activations
        const int NJr = j_hi - j_lo;
        for (int it = tid; it < N * NJr; it += nt) {
            const int n = it / NJr, j = j_lo + (it - n * NJr);
            const float xhat = (pre[(size_t)n * kPS + ncls + j] - s_mean[j]) * s_inv[j];
            const float d = s_inv[j] / (float)N * ((float)N * dxh[(size_t)n * kPS + j] - tot[Q_SUM1 + j] - xhat * tot[Q_SUM2 + j]);
            dpre[(size_t)n * kPS + ncls + j] = d;
            atomicAdd(&dbq[j - j_lo], (unsigned long long)__float2ll_rn(d * 68719476736.0f));  // lanes of one wave hold different units here
        }
    }
    stamp();  // C
    // D: softmax + categorical cross-entropy (lanes over the samples, wave-reduced sums)
    for (int n0 = 0; n0 < (is_cls ? N : 0); n0 += nt) {
        const int n = n0 + tid;
        const bool on = n < N;
        const int nc = on ? n : N - 1;
        float mx = -INFINITY, p[8], t[8];
        for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, P(nc, c));
        float den = 0.f;
        for (int c = 0; c < ncls; ++c) den += (p[c] = expf(P(nc, c) - mx));
        int am = 0, at = 0;
        float l = 0.f;
        for (int c = 0; c < ncls; ++c) {
            p[c] /= den;
            t[c] = Y(nc, (a.out_dim - ncls) + c);
            l -= t[c] * logf(fminf(fmaxf(p[c], kKerasEps), 1.0f - kKerasEps));
            if (p[c] > p[am]) am = c;
            if (t[c] > t[at]) at = c;
        }
        for (int c = 0; c < ncls; ++c) {
            const float d = on ? (p[c] - t[c]) / (float)N * a.lw[nh] : 0.f;
            if (on) {
                if constexpr (STAGED) tile[n * PST + c] = d;  // a lane's own row: its reads of P(n, .) are done
                else dpre[(size_t)n * kPS + c] = d;
            }
            wave_add(row, Q_DB3 + c, d);
        }
        if constexpr (!STAGED)
            if (on)
                for (int c = ncls + NJ; c < kPS; ++c) dpre[(size_t)n * kPS + c] = 0.f;
        wave_add(row, Q_LOSS + nh, on ? l / (float)N : 0.f);
        wave_add(row, Q_LOSS + nh + 1, (on && am == at) ? 1.0f / (float)N : 0.f);
    }
    if constexpr (STAGED) {
        __syncthreads();
        for (int i = tid; i < N * c_n; i += nt) {  // d loss / d pre: this workgroup's columns (64-byte runs per sample for a head)
            const int n = i / c_n, c = c_lo + (i - n * c_n);
            dpre[(size_t)n * kPS + c] = tile[n * PST + c];
        }
        if (is_cls)  // the padding columns behind the last head
            for (int i = tid; i < N * (kPS - NHc); i += nt) {
                const int n = i / (kPS - NHc), c = NHc + (i - n * (kPS - NHc));
                dpre[(size_t)n * kPS + c] = 0.f;
            }
    }
    stamp();  // D (+ copy-out)
    if constexpr (!STAGED) {
        __syncthreads();
        if (tid < j_hi - j_lo) red[0][Q_DBIAS + j_lo + tid] = (float)((double)(long long)dbq[tid] * (1.0 / 68719476736.0));
    }
    totals();
    // E: gradients of the small tensors (this workgroup is their only writer) and the losses
    for (int j = j_lo + tid; j < j_hi; j += nt) {
        const int h = j / kHidden, i = j % kHidden;
        float *gh = grad + a.goff_head[h] + (size_t)a.D * kHidden;  // after the dense kernel
        gh[i] = tot[Q_DBIAS + j];
        gh[16 + i] = tot[Q_DGAMMA + j];
        gh[32 + i] = tot[Q_DBETA + j];
        const int od = a.head_odim[h];
        for (int c = 0; c < od; ++c) gh[16 + 64 + i * od + c] = tot[Q_DWO + j * 3 + c];
        if (i < od) gh[16 + 64 + kHidden * od + i] = tot[Q_DBO + h * 3 + i];
    }
    if (is_cls && tid < ncls) grad[a.goff_c3b + tid] = tot[Q_DB3 + tid];
    if (tid == 0) {
        if (is_cls) {
            losses[nh] = tot[Q_LOSS + nh];
            losses[nh + 2] = tot[Q_LOSS + nh + 1];  // 3C accuracy
        } else {
            losses[role] = tot[Q_LOSS + role];
            if (a.ext_losses) losses[2 * nh + 4 + role] = tot[Q_ACC + role];  // B3_MTL trainer: binary accuracy of head h (training-mode outputs)
        }
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {  // the last head to finish: the weighted total, in head order
            __threadfence();
            float total = 0.f;
            for (int h = 0; h <= nh; ++h) total += a.lw[h] * __builtin_nontemporal_load(losses + h);
            losses[nh + 1] = total;  // without the l2 term (losses[nh + 3], l2_penalty_kernel)
            *ticket = 0;             // ready for the next step (stream order)
        }
    }
    stamp();
    if (a.stamps && tid == 0 && blockIdx.x == 0)
        printf("heads_train_kernel (head 0) N=%d (x10 ns): stage %llu  A %llu  B %llu  C %llu  D+copy %llu  E %llu\n", N, tk[1] - tk[0], tk[2] - tk[1],
               tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5]);
}

__global__ void det_finalize_kernel(unsigned long long *__restrict__ gq, float *__restrict__ grad, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const long long q = (long long)gq[i];
        if (q != 0) {
            grad[i] += (float)((double)q * kDetInvScale);  // grad[i] is 0 here (cleared at the top of the step) for every atomically summed tensor
            gq[i] = 0ull;
        }
    }
}

// One workgroup = kBG patches.  All buffers are (rows = g*T + t, channel) with stride kBS.
__global__ void __launch_bounds__(kBThreads)
tcn_backward_kernel(BwdArgs a, const float *__restrict__ X, const float *__restrict__ flatw, const float *__restrict__ acts,
                    const float *__restrict__ drop, const float *__restrict__ dpre, float *__restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T = a.T, nslot = a.n_blocks + 1;
    const int n0 = blockIdx.x * kBG;
    const int g_here = min(kBG, a.N - n0);
    const int rows = g_here * T, RP = kBG * T;
    float *G = sm, *Xs = G + (size_t)RP * kBS, *U = Xs + (size_t)RP * kBS, *Y = U + (size_t)RP * kBS;
    float *W1 = Y + (size_t)RP * kBS;       // [3][32][32]
    float *W2 = W1 + 3 * C * C;             // [32][32]
    // transposed copies for the two phases whose lanes run over the INPUT channel: W1T[tap][co][c], W2T[co][c]
    // (reading W[c*32 + co] with 32 lanes over c is a 32-way LDS bank conflict: it was 60 % of the kernel's time)
    // (patches of more than ~200 frames leave no room for them: use_wt = 0 reads W1 / W2 transposed, conflicts and all)
    float *W1T = W2 + C * C;                // [3][32][32]
    float *W2T = W1T + 3 * C * C;           // [32][32]
    float *B1 = a.use_wt ? W2T + C * C : W2 + C * C;  // [32]
    float *rowm = B1 + C;                   // per row: m
    float *rowmx = rowm + RP;               // per row: max
    float *dps = rowmx + RP;                // [kBG][kPS] d loss / d pre
    int *rowt = reinterpret_cast<int *>(dps + kBG * kPS);  // per row: frame index inside its patch
    const int tid = threadIdx.x, nt = blockDim.x;

    for (int i = tid; i < g_here * kPS; i += nt) dps[i] = dpre[(size_t)n0 * kPS + i];
    for (int i = tid; i < RP; i += nt) rowt[i] = i % T;
    __syncthreads();
    // ---- Dense-on-trunk backward: G = relu'(x) * (dpre @ Wh^T);  dWh += flat^T dpre --------------------
    for (int i = tid; i < rows * C; i += nt) {
        const int R = i / C, c = i - R * C;
        const int g = R / T, t = R - g * T;
        const size_t k = (size_t)t * C + c;
        const float xpre = acts[(((size_t)(n0 + g) * nslot + a.n_blocks) * T + t) * C + c];
        const float fl = fmaxf(xpre, 0.f);
        float acc = 0.f;
        const float *dp = dps + g * kPS;
        for (int o = 0; o < a.n_classes; ++o) {
            acc = fmaf(dp[o], flatw[a.off.c3_k + k * a.n_classes + o], acc);
            if (fl != 0.f) gadd(grad, a.gq, a.off.c3_k + k * a.n_classes + o, fl * dp[o]);
        }
        for (int h = 0; h < a.n_heads; ++h) {
            const size_t base = a.off.head[h] + k * kHidden;
            for (int j = 0; j < kHidden; ++j) {
                const float d = dp[a.n_classes + h * kHidden + j];
                acc = fmaf(d, flatw[base + j], acc);
                if (fl != 0.f) gadd(grad, a.gq, base + j, fl * d);
            }
        }
        G[R * kBS + c] = xpre > 0.f ? acc : 0.f;
    }
    // ---- residual blocks, last to first ----------------------------------------------------------------
    for (int blk = a.n_blocks - 1; blk >= 0; --blk) {
        const int d = 1 << (blk % a.n_dil);
        const size_t wo = a.off.blk0 + (size_t)blk * a.off.blk_stride;
        const size_t o_k1 = wo, o_b1 = wo + 3 * C * C, o_k2 = o_b1 + C, o_b2 = o_k2 + C * C;
        __syncthreads();
        for (int i = tid; i < rows * C; i += nt) {
            const int R = i / C, c = i - R * C;
            const int g = R / T, t = R - g * T;
            Xs[R * kBS + c] = acts[(((size_t)(n0 + g) * nslot + blk) * T + t) * C + c];
        }
        for (int i = tid; i < 3 * C * C; i += nt) {
            const float v = flatw[o_k1 + i];
            W1[i] = v;
            const int tap = i / (C * C), c = (i / C) % C, co = i % C;
            if (a.use_wt) W1T[(tap * C + co) * C + c] = v;
        }
        for (int i = tid; i < C * C; i += nt) {
            const float v = flatw[o_k2 + i];
            W2[i] = v;
            if (a.use_wt) W2T[(i % C) * C + i / C] = v;
        }
        if (tid < C) B1[tid] = flatw[o_b1 + tid];
        __syncthreads();
        // recompute u = conv_d(x) + b1
        for (int i = tid; i < rows * C; i += nt) {
            const int R = i / C, co = i - R * C;
            const int t = rowt[R];
            float acc = B1[co];
            for (int tap = 0; tap < 3; ++tap) {
                const int off = (tap - 1) * d;
                if (t + off < 0 || t + off >= T) continue;
                const float *xr = Xs + (R + off) * kBS;
                const float *w = W1 + tap * C * C + co;
#pragma unroll 8
                for (int c = 0; c < C; ++c) acc = fmaf(xr[c], w[c * C], acc);
            }
            U[R * kBS + co] = acc;
        }
        __syncthreads();
        // row statistics: m = max_c relu(u) + eps ; y = relu(u)/m * mask
        for (int i = tid; i < rows * C; i += nt) {  // 32 consecutive threads = one row (C == 32)
            const int R = i / C, c = i - R * C;
            const float r = fmaxf(U[R * kBS + c], 0.f);
            float mx = r;
            for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float m = mx + kNormEps;
            const float mask = drop ? drop[((size_t)(n0 + R / T) * a.n_blocks + blk) * C + c] : 1.0f;
            Y[R * kBS + c] = r / m * mask;
            if (c == 0) rowm[R] = m, rowmx[R] = mx;
        }
        __syncthreads();
        // dW2[c][co] += sum_R y[R][c] g[R][co] ; db2[co] += sum_R g[R][co]
        for (int i = tid; i < C * C; i += nt) {
            const int c = i / C, co = i - c * C;
            float acc = 0.f, accb = 0.f;
            for (int R = 0; R < rows; ++R) {
                const float gv = G[R * kBS + co];
                acc = fmaf(Y[R * kBS + c], gv, acc);
                accb += gv;
            }
            gadd(grad, a.gq, o_k2 + i, acc);
            if (c == 0) gadd(grad, a.gq, o_b2 + co, accb);
        }
        __syncthreads();
        // dyn = g @ W2^T (masked) ; norm backward ; du = dr * (u > 0)  -> U
        for (int i = tid; i < rows * C; i += nt) {
            const int R = i / C, c = i - R * C;
            const float *gr = G + R * kBS;
            const float *w = a.use_wt ? W2T + c : W2 + c * C;  // element (co, c) of W2^T
            const int ws = a.use_wt ? C : 1;
            float dyn = 0.f;
#pragma unroll 8
            for (int co = 0; co < C; ++co) dyn = fmaf(gr[co], w[co * ws], dyn);
            const float mask = drop ? drop[((size_t)(n0 + R / T) * a.n_blocks + blk) * C + c] : 1.0f;
            dyn *= mask;
            const float u = U[R * kBS + c];
            const float r = fmaxf(u, 0.f);
            const float m = rowm[R], mx = rowmx[R];
            float s1 = dyn * r;
            float cnt = (r == mx) ? 1.f : 0.f;
            for (int o = 16; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o);
                cnt += __shfl_xor(cnt, o);
            }
            float dr = dyn / m;
            if (r == mx && r > 0.f) dr -= s1 / (m * m) / cnt;
            U[R * kBS + c] = u > 0.f ? dr : 0.f;
        }
        __syncthreads();
        // dW1[tap][c][co] += sum_R x[R+off][c] du[R][co] ; db1[co] += sum_R du[R][co]
        for (int i = tid; i < 3 * C * C; i += nt) {
            const int tap = i / (C * C), c = (i / C) % C, co = i % C;
            const int off = (tap - 1) * d;
            float acc = 0.f, accb = 0.f;
            for (int R = 0; R < rows; ++R) {
                const int t = rowt[R];
                const float duv = U[R * kBS + co];
                accb += duv;
                if (t + off >= 0 && t + off < T) acc = fmaf(Xs[(R + off) * kBS + c], duv, acc);
            }
            gadd(grad, a.gq, o_k1 + i, acc);
            if (tap == 0 && c == 0) gadd(grad, a.gq, o_b1 + co, accb);
        }
        __syncthreads();
        // g[R][c] += sum_tap sum_co du[R - off][co] W1[tap][c][co]
        for (int i = tid; i < rows * C; i += nt) {
            const int R = i / C, c = i - R * C;
            const int t = rowt[R];
            float acc = G[R * kBS + c];
            for (int tap = 0; tap < 3; ++tap) {
                const int off = (tap - 1) * d;
                if (t - off < 0 || t - off >= T) continue;
                const float *dur = U + (R - off) * kBS;
                const float *w = a.use_wt ? W1T + tap * C * C + c : W1 + (tap * C + c) * C;
                const int ws = a.use_wt ? C : 1;
#pragma unroll 8
                for (int co = 0; co < C; ++co) acc = fmaf(dur[co], w[co * ws], acc);
            }
            G[R * kBS + c] = acc;
        }
    }
    __syncthreads();
    // ---- initial Conv1D(32,1): dW0[f][c] += sum_R x[R][f] g[R][c] ; db0[c] += sum_R g[R][c] ----------------
    for (int i = tid; i < a.F * C; i += nt) {
        const int f = i / C, c = i - f * C;
        float acc = 0.f, accb = 0.f;
        for (int R = 0; R < rows; ++R) {
            const float gv = G[R * kBS + c];
            acc = fmaf(X[((size_t)n0 * T + R) * a.F + f], gv, acc);
            accb += gv;
        }
        gadd(grad, a.gq, a.off.w0_k + i, acc);
        if (f == 0) gadd(grad, a.gq, a.off.w0_b + c, accb);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA form of the backward pass (the default).  One workgroup = kMG patches (rows = kMG*T activations rows of 32
// channels in LDS, stride SX), 8 waves.  Per residual block, last to first, with the block input x re-read from the
// activations the TRAIN forward saved:
//   1. per 16-row tile (one wave each): u = conv_d(x) + b1 as the forward saved it (TrainIO::upre, register-prefetched in
//      accumulator layout; it used to be recomputed: 48 MFMA per tile), relu / channel-max norm -> y (to LDS), dyn = W2 . g
//      as D[c][time] (16 MFMA, g from LDS as B operand), the norm / relu backward in registers (same lane owns the same
//      channels of u and dyn) -> du (to LDS);
//   2. weight gradients as 16 x 16 output tiles with K = rows: dW2[c][co] = sum_t y[t][c] g[t][co],
//      dW1[tap][c][co] = sum_t x[t+off][c] du[t][co], the bias gradients as column sums on the VALU; 16 tiles, two per wave,
//      results added to the global gradient with hardware float atomics;
//   3. per tile: g += sum_tap W1[tap] . du[t - off] (48 MFMA), in place.
// A operands come from LDS copies of the block's canonical kernels (rows = the lane's channel, stride kWS).
// The Dense-on-trunk weight gradient (a rank-N update of a D x 51 matrix) is its own kernel without atomics.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMG = 1;
constexpr int kWS = 36;  // row stride of the weight copies in LDS (tcn_backward_mfma_kernel)
constexpr int kMThreads = 512;
constexpr int kZW = 3 * SX + 4;  // zero words behind the images (tile_jobs)
constexpr int kMfmaMaxT = 128;   // two float4 of saved activations per thread in the register prefetch
constexpr int kMfmaLongT = 256;  // four; kernels read from global memory (no LDS left)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// combine over the four lanes that hold the same time step (l, l ^ 16, l ^ 32, l ^ 48) with gfx950's VALU lane swaps
// instead of two ds_bpermute round trips
template <class F>
__device__ __forceinline__ float quad_reduce(float v, F f) {
    const unsigned u = __float_as_uint(v);
    const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = f(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    const unsigned ua = __float_as_uint(a);
    const auto r16 = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
    return f(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
}

// WLDS: the block's canonical kernels are parked in LDS.  Patches longer than 128 frames (the reference's W = 249) leave no room
// for them next to the four activation images: WLDS = false reads them from the weight vector itself.
template <bool WLDS, int MAXT>
// (WLDS: held to 128 VGPRs = four waves per SIMD = two workgroups per CU; at 129 the second workgroup is lost and the step costs 100 us more)
__global__ void __launch_bounds__(kMThreads, WLDS ? 4 : 1)
tcn_backward_mfma_kernel(BwdArgs a, const float *__restrict__ X, const float *__restrict__ flatw,
                         const float *__restrict__ acts, const float *__restrict__ drop, const float *__restrict__ dpre,
                         float *__restrict__ grad, int RPm, const float *__restrict__ upre, const float *__restrict__ gt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T = a.T, nslot = a.n_blocks + 1;
    const int n0 = blockIdx.x * kMG;
    const int g_here = min(kMG, a.N - n0);
    const int rows = g_here * T;
    const int units = RPm >> 4;
    float *Xs = sm, *G = Xs + (size_t)RPm * SX, *DU = G + (size_t)RPm * SX, *Y = DU + (size_t)RPm * SX;
    // A operands of the three products are rows of these copies (row = the lane's channel, eight consecutive k per lane
    // group: two float4 per tap); rows are kWS = 36 floats apart so that 16 lanes reading 16 rows spread over the banks
    float *W1 = Y + (size_t)RPm * SX;   // [3][32 cin][kWS: 32 cout]   (canonical)        -> phase 3
    float *W2 = W1 + 3 * C * kWS;       // [32 cin][kWS: 32 cout]      (canonical)        -> dyn
    float *dps = WLDS ? W2 + C * kWS : Y + (size_t)RPm * SX;  // [kMG][kPS]
    float *DM = dps + kMG * kPS;                         // [32] SpatialDropout1D mask of the current block (ones without dropout)
    float *ZW = DM + C;                                  // [kZW] zeros: what an operand row outside the patch reads (+ up to 3 rows of offset)
    const int tid = threadIdx.x, nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;  // (wave-uniform: job and tile indices in SGPRs)
    const int q = lane >> 4, j = lane & 15;

    const unsigned long long t_entry = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (int i = tid; i < 4 * RPm * SX; i += nt) sm[i] = 0.f;  // padded rows stay zero in every buffer
    for (int i = tid; i < g_here * kPS; i += nt) dps[i] = dpre[(size_t)n0 * kPS + i];
    if (tid < kZW) ZW[tid] = 0.f;
    __syncthreads();
    // ---- Dense-on-trunk backward: G = relu'(x) * (dpre @ Wh^T) ------------------------------------------------
    // gt: the product as dtrunk_kernel computed it for the whole batch (smh_train_bf16.hip) -- here every workgroup read the whole
    // 444 KB Dense kernel from L2 for its one patch; nullptr (SMH_DTRUNK=0): the loop below
    if (gt) {
        for (int i = tid; i < rows * (C / 4); i += nt) {
            const int R = i >> 3, c4 = (i & 7) * 4;
            *reinterpret_cast<f32x4 *>(G + (size_t)R * SX + c4) = *reinterpret_cast<const f32x4 *>(gt + ((size_t)n0 * T + R) * C + c4);
        }
    }
    for (int i = tid; i < (gt ? 0 : rows * C); i += nt) {
        const int R = i / C, c = i - R * C;
        const int g = R / T, t = R - g * T;
        const size_t k = (size_t)t * C + c;
        const float xpre = acts[(((size_t)(n0 + g) * nslot + a.n_blocks) * T + t) * C + c];
        float acc = 0.f;
        const float *dp = dps + g * kPS;
        for (int o = 0; o < a.n_classes; ++o) acc = fmaf(dp[o], flatw[a.off.c3_k + k * a.n_classes + o], acc);
        for (int h = 0; h < a.n_heads; ++h) {
            const float *wr = flatw + a.off.head[h] + k * kHidden;
#pragma unroll
            for (int jj = 0; jj < kHidden; ++jj) acc = fmaf(dp[a.n_classes + h * kHidden + jj], wr[jj], acc);
        }
        G[R * SX + c] = xpre > 0.f ? acc : 0.f;
    }

    unsigned long long tph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = t_entry;
    const bool stamping = a.stamps && blockIdx.x == 0 && tid == 0;
    auto lap = [&](int i) {
        if (stamping) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            tph[i] += now - tlast;
            tlast = now;
        }
    };
    // Weight-gradient tiles: D[row = 16 mt + ..][col = 16 nt + j] = sum_k A[k + a_shift][row] B[k][col] with k = activation
    // row.  with_bias: the same pass also sums the columns of B (A = ones on a second accumulator) -- the bias gradient of that
    // column tile; as jobs of their own the four column sums cost a third round of the eight waves.
    // A wave runs TWO tiles at once (16 jobs = one round of the eight waves): four k steps of both per iteration, their sixteen
    // LDS reads issued together, four independent accumulator chains -- one tile per wave left the wave waiting for its own
    // LDS reads and for its own previous product in turn (2.4 us per tile, two rounds: half of a block's time).
    struct TileJob {
        int A, B;  // offsets of the two operand images in `sm`
        int a_shift, mt, nt;
        unsigned gbase, gbias;
        bool with_bias;
    };
    auto tile_jobs = [&](const TileJob &j0, const TileJob &j1) {
        // One patch per workgroup: the A row of step kr is kr + a_shift, valid inside [0, T) -- ONE unsigned compare (rows
        // kr >= T meet B = 0: every image keeps its rows behind the patch at zero, and the A row they pair with is a real
        // row, i.e. finite).  An invalid row reads from the zero words ZW instead of skipping the read under an exec mask.
        // Which rows a product step takes is free (any order of k is the same sum): the sixteen rows 16 m .. 16 m + 15 are
        // taken as row 16 m + c + 4 q by lane group q in step c = 0..3, so that the four groups of a read sit 4 rows = 16 banks
        // apart (two lanes per bank, the minimum for 64 lanes), and every address is a per-lane base that advances by 16 rows
        // per m plus a compile-time offset.  Two steps per pipeline stage (c = 0, 1 and c = 2, 3), the reads of the next stage
        // in flight under the products of this one.  The bias gradients (column sums of B) are VALU adds on the operands the
        // lanes hold anyway -- as products with a ones operand they were a fifth of the phase's matrix-core time.
        static_assert(kMG == 1, "tile_jobs: one patch per workgroup (row index == frame index)");
        f32x4 acc[2][2];
        float bsum[2] = {0.f, 0.f};
        // j0 is always an unshifted tile (dW2 or the centre tap): its operand rows 0 .. RPm - 1 all exist (zero behind the patch)
        // and need no validity test; byte offsets throughout (a select on a word index costs a shift per read on top).
        int pa[2], pb[2], ua[2];  // A / B byte offset of row (16 m + 4 q) for this lane; A's row index itself
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const TileJob &jb = i ? j1 : j0;
            acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            ua[i] = 4 * q + jb.a_shift;
            pa[i] = 4 * (jb.A + ua[i] * SX + 16 * jb.mt + j), pb[i] = 4 * (jb.B + 4 * q * SX + 16 * jb.nt + j);
        }
        const int zero_off = 4 * (int)(ZW - sm);
        auto lds_at = [&](int byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sm) + byte_off); };
        auto fetch = [&](int h, int adv, float (&av)[2][2], float (&bv)[2][2]) {  // stage h of the block of 16 rows `adv` blocks ahead
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int c = 2 * h + e;
                    const bool ok = i == 0 || (unsigned)(ua[i] + 16 * adv + c) < (unsigned)T;
                    av[i][e] = lds_at((ok ? pa[i] + 64 * adv * SX : zero_off) + 4 * c * SX);
                    bv[i][e] = lds_at(pb[i] + 64 * adv * SX + 4 * c * SX);
                }
        };
        auto products = [&](float (&av)[2][2], float (&bv)[2][2]) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i][e] = mfma4(av[i][e], bv[i][e], acc[i][e]);
                    bsum[i] += bv[i][e];
                }
        };
        float a0[2][2], b0[2][2], a1[2][2], b1[2][2];
        fetch(0, 0, a0, b0);
        const int M = RPm >> 4;
        for (int m = 0; m < M; ++m) {
            fetch(1, 0, a1, b1);
            products(a0, b0);
            const int adv = m + 1 < M ? 1 : 0;  // (behind the last block: the same rows again, unused)
            fetch(0, adv, a0, b0);
            products(a1, b1);
#pragma unroll
            for (int i = 0; i < 2; ++i) pa[i] += 64 * SX, pb[i] += 64 * SX, ua[i] += 16;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const TileJob &jb = i ? j1 : j0;
            const f32x4 v = acc[i][0] + acc[i][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) gadd(grad, a.gq, jb.gbase + (unsigned)((16 * jb.mt + 4 * q + r) * C + 16 * jb.nt + j), v[r]);
            if (jb.with_bias) {  // (uniform) column sums: over this lane's rows, then over the four lane groups
                const float cs = quad_reduce(bsum[i], [](float x, float y) { return x + y; });
                if (q == 0) gadd(grad, a.gq, jb.gbias + 16 * jb.nt + j, cs);
            }
        }
    };

    // register prefetch of a block's inputs: saved activations (rows x 8 float4) and its two kernels + bias
    constexpr int kPfX = (kMG * MAXT * (C / 4) + kMThreads - 1) / kMThreads;  // T <= MAXT per patch (checked on the host)
    f32x4 pf_x[kPfX];
    float pf_w1[6], pf_w2[2];
    // the block's dilated-conv outputs before the relu, as the training forward saved them (TrainIO::upre), for this wave's
    // tiles in accumulator layout: the gates of the relu / channel-max backward.  (They used to be recomputed from the block
    // input: 48 products per tile, a transposed LDS copy of the kernel and half of phase 1's time.)
    constexpr int kTPW = (MAXT / 16 + 7) / 8;  // tiles per wave: 1 up to 128 frames, 2 up to 256
    f32x4 pf_u[kTPW][2];
    // the SpatialDropout1D mask of (patch, block): one patch per workgroup, so it is the same for every tile of the block --
    // fetched with the block's other inputs and parked in LDS (DM) instead of read from L2 once per tile right before its use
    // (an exposed L2 round trip in every tile of phase 1)
    float pf_dm = 1.f;
    auto prefetch = [&](int blk) {
        const size_t wo = a.off.blk0 + (size_t)blk * a.off.blk_stride;
        if (kMG == 1 && drop && tid < C) pf_dm = drop[((size_t)n0 * a.n_blocks + blk) * C + tid];
#pragma unroll
        for (int e = 0; e < kPfX; ++e) {
            const int i = tid + e * kMThreads;
            if (i < rows * (C / 4)) {
                const int R = i >> 3, c4 = (i & 7) * 4;
                const int g = R / T, t = R - g * T;
                pf_x[e] = *reinterpret_cast<const f32x4 *>(acts + (((size_t)(n0 + g) * nslot + blk) * T + t) * C + c4);
            }
        }
        if (WLDS) {
#pragma unroll
            for (int e = 0; e < 6; ++e) pf_w1[e] = flatw[wo + tid + e * kMThreads];
#pragma unroll
            for (int e = 0; e < 2; ++e) pf_w2[e] = flatw[wo + 3 * C * C + C + tid + e * kMThreads];
        }
    };
    // requested in front of phase 3 of the block before (not with the block's other inputs: eight more registers across the
    // weight-gradient phase would cost the second workgroup per CU)
    auto prefetch_u = [&](int blk) {
#pragma unroll
        for (int i = 0; i < kTPW; ++i) {
            const int R = 16 * (wave + i * 8) + j;  // (8 waves)
            pf_u[i][0] = pf_u[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (R < rows) {
                const float *up = upre + (((size_t)n0 * a.n_blocks + blk) * T + R) * C + 4 * q;  // one patch per workgroup: row == frame
                pf_u[i][0] = *reinterpret_cast<const f32x4 *>(up);
                pf_u[i][1] = *reinterpret_cast<const f32x4 *>(up + 16);
            }
        }
    };
    lap(5);  // zeroing, dpre, the Dense-on-trunk backward
    prefetch(a.n_blocks - 1);
    prefetch_u(a.n_blocks - 1);
    // ---- residual blocks, last to first --------------------------------------------------------------------------
    for (int blk = a.n_blocks - 1; blk >= 0; --blk) {
        const int d = 1 << (blk % a.n_dil);
        const size_t wo = a.off.blk0 + (size_t)blk * a.off.blk_stride;
        const size_t o_k1 = wo, o_b1 = wo + 3 * C * C, o_k2 = o_b1 + C, o_b2 = o_k2 + C * C;
        __syncthreads();  // previous block finished with Xs / W*
        // this block's inputs were requested during the previous block (registers pf_*); park them in LDS
#pragma unroll
        for (int e = 0; e < kPfX; ++e) {
            const int i = tid + e * kMThreads;
            if (i < rows * (C / 4)) *reinterpret_cast<f32x4 *>(Xs + (size_t)(i >> 3) * SX + (i & 7) * 4) = pf_x[e];
        }
        if (WLDS) {
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int i = tid + e * kMThreads;  // 3*C*C = 6 * 512
                const int tap = i / (C * C), c = (i / C) % C, co = i % C;
                W1[(tap * C + c) * kWS + co] = pf_w1[e];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = tid + e * kMThreads;  // C*C = 2 * 512
                W2[(i / C) * kWS + i % C] = pf_w2[e];
            }
        }
        if (tid < C) DM[tid] = pf_dm;
        constexpr int WS = WLDS ? kWS : C;  // the global copies keep the canonical stride
        const float *W1p = WLDS ? W1 : flatw + o_k1, *W2p = WLDS ? W2 : flatw + o_k2;
        __syncthreads();
        lap(0);  // barrier + inputs parked in LDS
        f32x4 cur_u[kTPW][2];
#pragma unroll
        for (int i = 0; i < kTPW; ++i) cur_u[i][0] = pf_u[i][0], cur_u[i][1] = pf_u[i][1];
        if (blk > 0) prefetch(blk - 1);  // overlaps with the three phases below
        // ---- phase 1: norm (from the saved conv outputs), dyn, norm backward -> Y, DU ------------------------------
#pragma unroll
        for (int ti = 0; ti < kTPW; ++ti) {
            const int u = wave + ti * 8;
            if (u >= units) break;
            const int R = 16 * u + j;
            const bool live = R < rows;
            const f32x4 acc0 = cur_u[ti][0], acc1 = cur_u[ti][1];
            float r0[4], r1[4], mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                r0[r] = fmaxf(acc0[r], 0.f), r1[r] = fmaxf(acc1[r], 0.f);
                mx = fmaxf(mx, fmaxf(r0[r], r1[r]));
            }
            mx = quad_reduce(mx, [](float x, float y) { return fmaxf(x, y); });
            const float m = mx + kNormEps;
            const float inv_m = __builtin_amdgcn_rcpf(m);  // the forward's own 1 / (max + eps)
            f32x4 dm0 = *reinterpret_cast<const f32x4 *>(DM + 4 * q), dm1 = *reinterpret_cast<const f32x4 *>(DM + 16 + 4 * q);
            static_assert(kMG == 1, "one patch per workgroup: one dropout mask per block (DM)");
            f32x4 y0, y1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y0[r] = live ? r0[r] * inv_m * dm0[r] : 0.f;
                y1[r] = live ? r1[r] * inv_m * dm1[r] : 0.f;
            }
            *reinterpret_cast<f32x4 *>(Y + (size_t)R * SX + 4 * q) = y0;
            *reinterpret_cast<f32x4 *>(Y + (size_t)R * SX + 16 + 4 * q) = y1;
            // dyn[c][time] = sum_co W2[c][co] g[time][co]
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
            {
                const float *gs = G + (size_t)R * SX + 8 * q;  // (rows behind the patch are zero in G)
                const f32x4 bA = *reinterpret_cast<const f32x4 *>(gs), bB = *reinterpret_cast<const f32x4 *>(gs + 4);
                const float *wa = W2p + (size_t)j * WS + 8 * q;  // W2[c = j (+16)][co = 8 q + s8]
                const f32x4 a0A = *reinterpret_cast<const f32x4 *>(wa), a0B = *reinterpret_cast<const f32x4 *>(wa + 4);
                const f32x4 a1A = *reinterpret_cast<const f32x4 *>(wa + 16 * WS), a1B = *reinterpret_cast<const f32x4 *>(wa + 16 * WS + 4);
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const float bv = s8 < 4 ? bA[s8 & 3] : bB[s8 & 3];
                    d0 = mfma4(s8 < 4 ? a0A[s8 & 3] : a0B[s8 & 3], bv, d0);
                    d1 = mfma4(s8 < 4 ? a1A[s8 & 3] : a1B[s8 & 3], bv, d1);
                }
            }
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
                du0[r] = (live && acc0[r] > 0.f) ? a0 : 0.f;
                du1[r] = (live && acc1[r] > 0.f) ? a1 : 0.f;
            }
            *reinterpret_cast<f32x4 *>(DU + (size_t)R * SX + 4 * q) = du0;
            *reinterpret_cast<f32x4 *>(DU + (size_t)R * SX + 16 + 4 * q) = du1;
        }
        lap(7);  // wave 0's own tile
        __syncthreads();
        lap(1);
        // ---- phase 2: weight gradients (20 tile jobs) ------------------------------------------------------------
        auto job_of = [&](int job) {  // 0..3: dW2[c][co] (db2[co] = sum_t g);  4..15: dW1[tap] (db1 = sum_t du)
            TileJob tj;
            if (job < 4) {
                const int mt = job >> 1;
                tj = TileJob{(int)(Y - sm), (int)(G - sm), 0, mt, job & 1, (unsigned)o_k2, (unsigned)o_b2, mt == 0};
            } else {
                const int tap = (job - 4) >> 2, mt = ((job - 4) >> 1) & 1;
                tj = TileJob{(int)(Xs - sm), (int)(DU - sm), (tap - 1) * d, mt, (job - 4) & 1, (unsigned)(o_k1 + (size_t)tap * C * C),
                             (unsigned)o_b1, tap == 1 && mt == 0};
            }
            return tj;
        };
        // one round of the eight waves, an unshifted tile (dW2, centre tap) first and a side-tap tile second in every pair
        for (int w = wave; w < 8; w += nw) tile_jobs(job_of(w < 4 ? w : w + 4), job_of(w < 4 ? w + 4 : w + 8));
        lap(6);  // wave 0's own two tiles
        __syncthreads();
        lap(2);
        if (blk > 0) prefetch_u(blk - 1);
        // ---- phase 3: g[time][c] += sum_tap sum_co W1[tap][c][co] du[time - off][co] ---------------------------------
        // A lone tile in the last round of the eight waves (5 tiles: one 68-frame patch; 9, 13) is a second whole tile on one SIMD
        // while three waves idle: its two accumulator chains (channels 0-15, 16-31) are independent, so two waves on different SIMDs
        // take one each -- same products, same order, same bits; the busiest SIMD runs 72 instead of 96 products per block.
        const bool split3 = a.split3 && nw == 8 && (units & 3) == 1 && units >= 2;
        const int hw0 = (units - 1) & 7, hw1 = (hw0 + 1) & 7;
        auto phase3_tile = [&](int u, auto part_c) {
            constexpr int PART = decltype(part_c)::value;  // 0: both channel halves, 1: channels 0-15, 2: channels 16-31
            const int R = 16 * u + j;
            if (16 * u >= rows) return;
            const bool live = R < rows;
            const int Rc = min(R, rows - 1);
            const int t = Rc % T;
            f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
            if (PART != 2) g0 = *reinterpret_cast<const f32x4 *>(G + (size_t)Rc * SX + 4 * q);
            if (PART != 1) g1 = *reinterpret_cast<const f32x4 *>(G + (size_t)Rc * SX + 16 + 4 * q);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int off = (tap - 1) * d;
                const bool ok = (t - off >= 0) && (t - off < T);
                if (tap != 1 && !__any(ok)) continue;
                const float *src = (ok ? DU + (size_t)(Rc - off) * SX : ZW) + 8 * q;
                const f32x4 bA = *reinterpret_cast<const f32x4 *>(src), bB = *reinterpret_cast<const f32x4 *>(src + 4);
                const float *wa = W1p + (size_t)(tap * C + j) * WS + 8 * q;  // W1[tap][c = j (+16)][co = 8 q + s8]
                f32x4 a0A = {0.f, 0.f, 0.f, 0.f}, a0B = a0A, a1A = a0A, a1B = a0A;
                if (PART != 2) a0A = *reinterpret_cast<const f32x4 *>(wa), a0B = *reinterpret_cast<const f32x4 *>(wa + 4);
                if (PART != 1) a1A = *reinterpret_cast<const f32x4 *>(wa + 16 * WS), a1B = *reinterpret_cast<const f32x4 *>(wa + 16 * WS + 4);
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const float bv = s8 < 4 ? bA[s8 & 3] : bB[s8 & 3];
                    if (PART != 2) g0 = mfma4(s8 < 4 ? a0A[s8 & 3] : a0B[s8 & 3], bv, g0);
                    if (PART != 1) g1 = mfma4(s8 < 4 ? a1A[s8 & 3] : a1B[s8 & 3], bv, g1);
                }
            }
            if (live) {
                if (PART != 2) *reinterpret_cast<f32x4 *>(G + (size_t)R * SX + 4 * q) = g0;
                if (PART != 1) *reinterpret_cast<f32x4 *>(G + (size_t)R * SX + 16 + 4 * q) = g1;
            }
        };
        for (int u = wave; u < units - (split3 ? 1 : 0); u += nw) phase3_tile(u, std::integral_constant<int, 0>{});
        if (split3) {  // (wave-uniform)
            if (wave == hw0) phase3_tile(units - 1, std::integral_constant<int, 1>{});
            else if (wave == hw1) phase3_tile(units - 1, std::integral_constant<int, 2>{});
        }
    }
    __syncthreads();
    lap(3);  // (phase 3 of every block but the last lands in lap 0 of the next)
    // ---- initial Conv1D(32,1): dW0[f][c] = sum_R x[R][f] g[R][c] ; db0[c] = sum_R g[R][c] -----------------------------
    const int fmt = (a.F + 15) >> 4;
    for (int job = wave; job < fmt * 2 + 2; job += nw) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool bias = job >= fmt * 2;
        const int mt = bias ? 0 : job >> 1, nt_ = bias ? job - fmt * 2 : job & 1;
        const int f = 16 * mt + j;
        // four k steps per iteration, the next iteration's loads of X (global) in flight under this one's products: one
        // load -> product per step left a memory round trip in every step (23 us of the kernel's 240 at one workgroup per CU)
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        const float *xcol = X + (size_t)n0 * T * a.F + (f < a.F ? f : 0);
        auto fetch = [&](int s, float (&av)[4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kr = 4 * s + e + 4 * q;  // (the row order of tile_jobs: two lanes per bank on the G reads)
                av[e] = bias ? 1.0f : ((kr < rows && f < a.F) ? xcol[(size_t)kr * a.F] : 0.f);
            }
        };
        float av[4];
        fetch(0, av);
        for (int s = 0; s < RPm / 4; s += 4) {  // RPm is a multiple of 16
            float an[4];
            fetch(min(s + 4, RPm / 4 - 4), an);  // (past the end: the last rows again, unused)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                acc = mfma4(av[e], G[(size_t)(4 * s + e + 4 * q) * SX + 16 * nt_ + j], acc);
                acc2 = mfma4(av[e + 1], G[(size_t)(4 * s + e + 1 + 4 * q) * SX + 16 * nt_ + j], acc2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = an[e];
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
    lap(4);
    if (stamping)
        printf("tcn_backward_mfma_kernel wg0 (x10 ns, summed over %d blocks): prologue %llu  park+phase3 %llu  phase1 %llu  phase2 %llu  tail %llu  layer0 %llu\n",
               a.n_blocks, tph[5], tph[0], tph[1] + tph[7], tph[2] + tph[6], tph[3], tph[4]);
    if (stamping) printf("   wave 0's own share: phase 1 tile %llu, phase 2 tiles %llu (the rest of each phase is its wait at the barrier)\n", tph[7], tph[6]);
}

}  // namespace

// tools / tests: workgroups of the MFMA backward kernel (T <= 128 instantiation) that fit one CU at patch size T -- two at the
// reference's T = 68 (126 VGPRs, 79 KB of LDS); one register class more and it is one, 345 -> 445 us per 510-patch step
extern "C" int smh_internal_bwd_residency(int T) {
    const int RPm = ((kMG * T + 15) / 16) * 16;
    const size_t lds_m = sizeof(float) * ((size_t)4 * RPm * SX + 4 * C * kWS + kMG * kPS + C + kZW);
    auto kern = tcn_backward_mfma_kernel<true, kMfmaMaxT>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m) != hipSuccess) return -1;
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, kMThreads, lds_m) != hipSuccess) return -1;
    return nb;
}

namespace {

// dWh[k][o] = sum_b relu(x_b)[k] dpre[b][o] for the '3C' kernel (grid.z = 0) and the Dense(16) kernel of every head
// (grid.z = 1 + h): one thread per element (k, o), o fastest, so the float atomics that combine the batch slices
// (grid.y, kDwhSlice patches each) hit contiguous addresses -- one lane per row of the matrix was 17x slower.
constexpr int kDwhSlice = 16;
__global__ void dwh_kernel(BwdArgs a, const float *__restrict__ acts, const float *__restrict__ dpre,
                           float *__restrict__ grad, int slice) {
    const int grp = blockIdx.z;
    const int ocount = grp == 0 ? a.n_classes : kHidden;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.D * ocount) return;
    const int k = i / ocount, o = i - k * ocount;
    const int nslot = a.n_blocks + 1;
    const int o0 = grp == 0 ? 0 : a.n_classes + (grp - 1) * kHidden;
    const int b0 = blockIdx.y * slice, b1 = min(a.N, b0 + slice);
    float acc = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float fl = fmaxf(acts[((size_t)b * nslot + a.n_blocks) * a.D + k], 0.f);
        acc = fmaf(fl, dpre[(size_t)b * kPS + o0 + o], acc);
    }
    gadd(grad, a.gq, (grp == 0 ? a.off.c3_k : a.off.head[grp - 1]) + i, acc);
}

// The same rank-N update on the matrix cores: dWh (D x 51) = relu(X)^T (D x N) . dpre (N x 51).  One wave = one 16 x 16 tile
// (16 trunk units k x the 16 outputs of one Dense(16) head, or the n_classes outputs of '3C') over one slice of the batch;
// A[i = k][kk = patch] are 64-byte runs of the saved trunk, B[kk = patch][j = output] 64-byte runs of dpre; eight product steps
// (32 patches) per iteration with their sixteen loads in flight together; batch slices combined by float atomics as above.
// 36 us -> see DESIGN 7 (the one-thread-per-element kernel is a dependent chain of N / slice loads + FMAs per thread).
constexpr int kDwhSplit = 8;  // batch slices (grid.z)
__global__ void __launch_bounds__(256) dwh_mfma_kernel(BwdArgs a, const float *__restrict__ acts, const float *__restrict__ dpre,
                                                       float *__restrict__ grad) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int k0 = (blockIdx.x * 4 + wave) * 16;
    if (k0 >= a.D) return;
    const int grp = blockIdx.y;  // 0: '3C', 1 + h: head h
    const int ocount = grp == 0 ? a.n_classes : kHidden;
    const int col0 = grp == 0 ? 0 : a.n_classes + (grp - 1) * kHidden;
    const int per = (((a.N + kDwhSplit - 1) / kDwhSplit) + 3) & ~3;  // patches per slice, a multiple of the 4 of a product step
    const int b0 = blockIdx.z * per, b1 = min(a.N, b0 + per);
    if (b0 >= b1) return;
    const int nslot = a.n_blocks + 1;
    const float *ap = acts + (size_t)a.n_blocks * a.D + k0 + i;  // + b * nslot * D
    const size_t astride = (size_t)nslot * a.D;
    const bool col_ok = i < ocount;
    const float *bp = dpre + col0 + (col_ok ? i : 0);  // + b * kPS
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int b = b0; b < b1; b += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int bb = b + 4 * u + kq;
            const bool ok = bb < b1;
            const int bc = ok ? bb : b0;
            const float x = ap[(size_t)bc * astride], d = bp[(size_t)bc * kPS];
            av[u] = ok ? fmaxf(x, 0.f) : 0.f;
            bv[u] = (ok && col_ok) ? d : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 = mfma4(av[u], bv[u], acc0);
            acc1 = mfma4(av[u + 1], bv[u + 1], acc1);
        }
    }
    acc0 += acc1;
    if (col_ok) {
        const size_t g = (grp == 0 ? a.off.c3_k : a.off.head[grp - 1]) + (size_t)(k0 + 4 * kq) * ocount + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) gadd(grad, a.gq, g + (size_t)r * ocount, acc0[r]);
    }
}

// l2(0.01) penalty of the Dense(16) kernels (the term Keras adds to the reported total loss), from the weights the
// step ran with: one workgroup, f64 accumulation.
// l2 = kL2 * sum over the heads' Dense(16) kernels of w^2, in f64: kL2Chunks workgroups per head write partial sums, the
// last one to arrive (ticket) adds them in a fixed order -- one workgroup over the 3 x 34 816 weights took 50 us.
constexpr int kL2Chunks = 16;
__global__ void __launch_bounds__(256) l2_penalty_kernel(BwdArgs a, const float *__restrict__ flatw, float *__restrict__ out,
                                                         double *__restrict__ part, unsigned *__restrict__ ticket) {
    __shared__ double sh[4];
    __shared__ bool last;
    const int h = blockIdx.x / kL2Chunks, c = blockIdx.x - h * kL2Chunks;
    const int n = a.D * kHidden, per = (n + kL2Chunks - 1) / kL2Chunks;
    const float *w = flatw + a.off.head[h];
    double s = 0.0;
    for (int i = c * per + threadIdx.x; i < min(n, (c + 1) * per); i += blockDim.x) s += (double)w[i] * (double)w[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        double total = 0.0;
        for (int hh = 0; hh < a.n_heads; ++hh) {
            double t = 0.0;
            for (int cc = 0; cc < kL2Chunks; ++cc) t += __builtin_nontemporal_load(part + hh * kL2Chunks + cc);
            out[1 + hh] = (float)((double)kL2 * t);  // per head: the sub-model Model(input, get_layer(h).output) carries only its own
            total += t;
        }
        out[0] = (float)((double)kL2 * total);
        *ticket = 0;  // ready for the next step (stream order)
    }
}

struct Segment {
    unsigned off, size;
    int kind;   // 0 plain, 1 l2-regularised Dense(16) kernel, 2 BN moving_mean, 3 BN moving_variance
    int aux;    // kinds 2/3: offset into the batch-statistics part of the gradient bucket
    int group;  // bit of `active_mask` this tensor belongs to: 0 trunk, 1 the '3C' Dense, 2 + h head h
    // A tensor larger than kSegChunk elements is cut into pieces (one workgroup each: the three Dense(16) kernels of 34 816
    // elements used to keep one workgroup busy for 75 us per pass while the chip idled); clipnorm is per TENSOR, so the
    // optimiser pass adds the pieces' sums of squares, sumsq[p0 .. p0 + pn), in that fixed order.
    int p0, pn;
};
constexpr unsigned kSegChunk = 4096;

// pass 1: g = grad * grad_scale (+ l2 term), written back; per-tensor sum of squares for clipnorm
__global__ void seg_sumsq_kernel(const Segment *__restrict__ segs, const float *__restrict__ w, float *__restrict__ grad,
                                 float grad_scale, unsigned active_mask, float *__restrict__ sumsq) {
    __shared__ float red[256];
    const Segment s = segs[blockIdx.x];
    float acc = 0.f;
    if (s.kind <= 1 && ((active_mask >> s.group) & 1u))
        for (unsigned i = threadIdx.x; i < s.size; i += blockDim.x) {
            float g = grad[s.off + i] * grad_scale;
            if (s.kind == 1) g += 2.0f * kL2 * w[s.off + i];  // d/dw of l2 * sum(w^2)
            grad[s.off + i] = g;
            acc = fmaf(g, g, acc);
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sumsq[blockIdx.x] = red[0];
}

// pass 2: the optimiser.  0 = SGD (Keras: v = momentum v - lr g; w += v), 1 = Adam, 2 = Nadam (tf.keras 2.x, see
// smh_trainer_apply_f32); BatchNorm moving statistics <- 0.99 old + 0.01 batch (batch statistics averaged over the
// data-parallel ranks by the same all-reduce as the gradient: they live behind it in one bucket).
struct OptArgs {
    int optimizer;
    float lr, b1, b2, eps, clipnorm, grad_scale;
    float alpha;                    // Adam: lr * sqrt(1 - b2^t) / (1 - b1^t)
    float n_g, n_m, n_v;            // Nadam: m_bar = n_g * g + n_m * m_t;  v' = v_t * n_v
    unsigned active_mask;
};
__global__ void seg_opt_kernel(const Segment *__restrict__ segs, OptArgs o, float *__restrict__ w, const float *__restrict__ grad,
                               float *__restrict__ s1, float *__restrict__ s2, const float *__restrict__ sumsq,
                               const float *__restrict__ bnstat) {
    const Segment s = segs[blockIdx.x];
    if (!((o.active_mask >> s.group) & 1u)) return;
    if (s.kind >= 2) {
        for (unsigned i = threadIdx.x; i < s.size; i += blockDim.x)
            w[s.off + i] = kBnMomentum * w[s.off + i] + (1.0f - kBnMomentum) * (bnstat[s.aux + i] * o.grad_scale);
        return;
    }
    float ss = 0.f;
    for (int c = 0; c < s.pn; ++c) ss += sumsq[s.p0 + c];
    const float nrm = sqrtf(ss);
    const float scale = (o.clipnorm > 0.f && nrm > o.clipnorm) ? o.clipnorm / nrm : 1.0f;
    for (unsigned i = threadIdx.x; i < s.size; i += blockDim.x) {
        const size_t k = (size_t)s.off + i;
        const float g = grad[k] * scale;
        if (o.optimizer == 0) {
            const float v = o.b1 * s1[k] - o.lr * g;
            s1[k] = v;
            w[k] += v;
        } else {
            const float m = o.b1 * s1[k] + (1.0f - o.b1) * g;
            const float v = o.b2 * s2[k] + (1.0f - o.b2) * g * g;
            s1[k] = m, s2[k] = v;
            if (o.optimizer == 1) w[k] -= o.alpha * m / (sqrtf(v) + o.eps);
            else w[k] -= o.lr * (o.n_g * g + o.n_m * m) / (sqrtf(v * o.n_v) + o.eps);
        }
    }
}

}  // namespace

int smh_tcn::launch_heads_train(const HeadsArgs &a, const float *pre, const float *y, const float *hp, const float *drop,
                                float *dpre, float *dxh, float *grad, float *bnstat, float *losses, unsigned *ticket,
                                hipStream_t st) {
    HeadsArgs ad = a;
    ad.stamps = getenv("SMH_HEADS_STAMPS") ? 1 : 0;
    const int NHc = a.n_classes + a.n_heads * kHidden;
    const size_t lds = sizeof(float) * ((size_t)a.N * (NHc | 1) + (size_t)a.N * (a.out_dim | 1));
    // 160 KB of LDS per CU minus the kernel's static 37 KB (per-wave accumulator rows): 3-class batches up to 528 patches
    const bool staged = lds <= 122 * 1024 && !getenv("SMH_HEADS_GLOBAL");
#define SMH_LAUNCH_HEADS(ST, TH)                                                                                        \
    do {                                                                                                                \
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)heads_train_kernel<ST, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)(ST ? lds : 0)));                                                        \
        hipLaunchKernelGGL((heads_train_kernel<ST, TH>), dim3(a.n_heads + 1), dim3(TH), ST ? lds : 0, st, ad, pre, y, hp, drop, \
                           dpre, dxh, grad, bnstat, losses, ticket);                                                    \
    } while (0)
    if (a.N <= 512) {
        if (staged) SMH_LAUNCH_HEADS(true, 512);
        else SMH_LAUNCH_HEADS(false, 512);
    } else {
        if (staged) SMH_LAUNCH_HEADS(true, 1024);
        else SMH_LAUNCH_HEADS(false, 1024);
    }
#undef SMH_LAUNCH_HEADS
    return smh::launch_status("heads_train_kernel");
}

struct smh_trainer {
    smh_model *m;
    int max_batch, nseg;
    float *d_acts = nullptr, *d_pre = nullptr, *d_dpre = nullptr, *d_dxh = nullptr;
    // ONE bucket [gradient (n_params) | BatchNorm batch statistics (kMaxHeads * 32)]: what data-parallel training all-reduces
    float *d_grad = nullptr, *d_bnstat = nullptr;
    float *d_vel = nullptr, *d_s2 = nullptr;  // optimiser state: momentum / first moment, second moment
    unsigned long long *d_gq = nullptr;       // deterministic mode: fixed-point gradient accumulators (n_params), else nullptr
    float *d_sumsq = nullptr, *d_scratch_out = nullptr;
    float *d_upre = nullptr;  // (max_batch, n_blocks, T, 32): TrainIO::upre
    float *d_gt = nullptr;       // (max_batch, T, 32): d loss / d (TCN output), dtrunk_kernel's product for the f32 backward
    void *d_bwd_pack = nullptr;  // the blocks' kernels as split bf16 A operands of the backward pass (smh_train_bf16.hip), dtype 1 only
    size_t bwd_pack_cap = 0;
    double *d_l2part = nullptr;   // l2_penalty_kernel: kL2Chunks partial sums per head, then its arrival ticket
    Segment *d_segs = nullptr;
    int dtype = 0;             // smh_trainer_set_dtype: 0 = exact-f32 matrix products, 1 = split-bf16 operands (f32 accumulators, f32 master weights)
    long step = 0;             // optimiser steps taken (Adam / Nadam bias corrections)
    double m_schedule = 1.0;   // Nadam's running product of the momentum schedule
};

extern "C" int smh_trainer_create(smh_model *m, int max_batch, smh_trainer **out) {
    SMH_REQUIRE(m && out && max_batch >= 1, "smh_trainer_create: bad argument");
    SMH_REQUIRE(m->cfg.block_variant == 0, "smh_trainer_create: training is built for block_variant 0 (keras-tcn 2.3.x) only");
    smh_trainer *t = new smh_trainer();
    t->m = m, t->max_batch = max_batch;
    const Offsets off = offsets(m);
    std::vector<Segment> segs;
    int group = 0;
    auto add = [&](size_t o, size_t n, int kind, int aux) {
        const int p0 = (int)segs.size(), pn = (int)((n + kSegChunk - 1) / kSegChunk);
        for (int c = 0; c < pn; ++c) {
            const size_t co = (size_t)c * kSegChunk, sz = std::min<size_t>(kSegChunk, n - co);
            segs.push_back(Segment{(unsigned)(o + co), (unsigned)sz, kind, kind >= 2 ? aux + (int)co : aux, group, p0, pn});
        }
    };
    add(off.w0_k, (size_t)m->cfg.n_feat * C, 0, 0);
    add(off.w0_b, C, 0, 0);
    for (int b = 0; b < m->n_blocks; ++b) {
        const size_t w = off.blk0 + (size_t)b * off.blk_stride;
        add(w, 3 * C * C, 0, 0);
        add(w + 3 * C * C, C, 0, 0);
        add(w + 3 * C * C + C, C * C, 0, 0);
        add(w + 3 * C * C + C + C * C, C, 0, 0);
    }
    group = 1;
    add(off.c3_k, (size_t)m->D * m->cfg.n_classes, 0, 0);
    add(off.c3_b, m->cfg.n_classes, 0, 0);
    for (int h = 0; h < m->n_heads; ++h) {
        group = 2 + h;
        size_t p = off.head[h];
        add(p, (size_t)m->D * kHidden, 1, 0), p += (size_t)m->D * kHidden;
        add(p, kHidden, 0, 0), p += kHidden;                 // dense bias
        add(p, kHidden, 0, 0), p += kHidden;                 // gamma
        add(p, kHidden, 0, 0), p += kHidden;                 // beta
        add(p, kHidden, 2, h * 32), p += kHidden;            // moving_mean
        add(p, kHidden, 3, h * 32 + 16), p += kHidden;       // moving_variance
        add(p, (size_t)kHidden * m->head_odim[h], 0, 0), p += (size_t)kHidden * m->head_odim[h];
        add(p, m->head_odim[h], 0, 0);
    }
    t->nseg = (int)segs.size();
    const size_t nact = (size_t)max_batch * (m->n_blocks + 1) * m->cfg.patch_size * C;
    hipError_t e = hipMalloc((void **)&t->d_acts, nact * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_pre, (size_t)max_batch * kPS * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_dpre, (size_t)max_batch * kPS * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_dxh, ((size_t)max_batch * kPS + 4) * sizeof(float));  // + the heads kernel's ticket
    if (e == hipSuccess) e = hipMemset(t->d_dxh + (size_t)max_batch * kPS, 0, 4 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_grad, (m->n_params + kMaxHeads * 32) * sizeof(float));
    if (e == hipSuccess) t->d_bnstat = t->d_grad + m->n_params;
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_vel, m->n_params * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_s2, m->n_params * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_sumsq, segs.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_scratch_out, (size_t)max_batch * m->out_dim * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_upre, (size_t)max_batch * m->n_blocks * m->cfg.patch_size * C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_l2part, (kMaxHeads * kL2Chunks + 1) * sizeof(double));
    if (e == hipSuccess) e = hipMemset(t->d_l2part, 0, (kMaxHeads * kL2Chunks + 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_segs, segs.size() * sizeof(Segment));
    if (e == hipSuccess) e = hipMemcpy(t->d_segs, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(t->d_vel, 0, m->n_params * sizeof(float));
    if (e == hipSuccess) e = hipMemset(t->d_s2, 0, m->n_params * sizeof(float));
    if (e == hipSuccess) e = hipMemset(t->d_grad, 0, (m->n_params + kMaxHeads * 32) * sizeof(float));
    if (e != hipSuccess) {
        smh_trainer_destroy(t);
        return smh::set_error(SMH_E_HIP, "smh_trainer_create: device allocation failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return SMH_OK;
}

extern "C" void smh_trainer_destroy(smh_trainer *t) {
    if (!t) return;
    for (float *p : {t->d_acts, t->d_pre, t->d_dpre, t->d_dxh, t->d_grad, t->d_vel, t->d_s2, t->d_sumsq, t->d_scratch_out, t->d_upre})
        (void)hipFree(p);
    (void)hipFree(t->d_segs);
    (void)hipFree(t->d_l2part);
    (void)hipFree(t->d_gq);
    (void)hipFree(t->d_bwd_pack);
    (void)hipFree(t->d_gt);
    delete t;
}

extern "C" float *smh_trainer_grad_ptr(smh_trainer *t) { return t ? t->d_grad : nullptr; }

// deterministic mode: fixed-point accumulators -> the float gradient (after the last backward kernel of a step)
static int det_finalize(smh_trainer *t, hipStream_t st) {
    if (!t->d_gq) return SMH_OK;
    const size_t n = t->m->n_params;
    hipLaunchKernelGGL(det_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t->d_gq, t->d_grad, n);
    return smh::launch_status("det_finalize_kernel");
}

extern "C" int smh_trainer_set_deterministic(smh_trainer *t, int on, void *stream) {
    SMH_REQUIRE(t, "smh_trainer_set_deterministic: null trainer");
    hipStream_t st = (hipStream_t)stream;
    if (on && !t->d_gq) {
        SMH_CHECK_HIP(hipMalloc((void **)&t->d_gq, t->m->n_params * sizeof(unsigned long long)));
        SMH_CHECK_HIP(hipMemsetAsync(t->d_gq, 0, t->m->n_params * sizeof(unsigned long long), st));
    } else if (!on && t->d_gq) {
        SMH_CHECK_HIP(hipStreamSynchronize(st));
        (void)hipFree(t->d_gq);
        t->d_gq = nullptr;
    }
    return SMH_OK;
}

extern "C" int smh_trainer_set_dtype(smh_trainer *t, int dtype) {
    SMH_REQUIRE(t, "smh_trainer_set_dtype: null trainer");
    SMH_REQUIRE(dtype == 0 || dtype == 1, "smh_trainer_set_dtype: dtype must be 0 (f32) or 1 (split bf16 operands)");
    t->dtype = dtype;
    return SMH_OK;
}

extern "C" int smh_train_step_f32(smh_trainer *t, const float *d_x, const float *d_y, int N, const float *d_drop_tcn,
                                  const float *d_drop_heads, const float *h_loss_weights, float *d_losses, void *stream) {
    SMH_REQUIRE(t && d_x && d_y && d_losses, "smh_train_step_f32: null argument");
    SMH_REQUIRE(N >= 1 && N <= t->max_batch, "smh_train_step_f32: batch %d outside [1, %d]", N, t->max_batch);
    smh_model *m = t->m;
    hipStream_t st = (hipStream_t)stream;
    SMH_CHECK_HIP(hipMemsetAsync(t->d_grad, 0, m->n_params * sizeof(float), st));
    // dtype 1: the residual blocks' backward on the bf16 matrix pipe (smh_train_bf16.hip); patch geometries outside its LDS plan, and
    // SMH_BWD_BF16=0 (A/B and tests), keep the exact-f32 kernel behind the bf16 forward -- which then has to save every block's input
    const bool bf16_bwd = t->dtype == 1 && !getenv("SMH_TRAIN_VALU") && !(getenv("SMH_BWD_BF16") && atoi(getenv("SMH_BWD_BF16")) == 0) &&
                          backward_bf16_supported(m->cfg.patch_size, m->cfg.n_dilations);
    TrainIO tio{t->d_acts, d_drop_tcn, t->d_pre, t->d_upre, bf16_bwd ? 1 : 0};
    // dtype 1: the training forward on the bf16 matrix pipe (split operands: f32-grade products, smh_tcn_bf16.hip); it saves the
    // same activations and gates in f32, so the backward pass below is unchanged
    int rc = t->dtype == 1 ? launch_forward_bf16_train(m, d_x, N, &tio, st) : launch_forward(m, d_x, N, t->d_scratch_out, nullptr, &tio, st);
    if (rc) return rc;
    const Offsets off = offsets(m);
    HeadsArgs ha;
    ha.N = N, ha.D = m->D, ha.NH = m->NH, ha.n_classes = m->cfg.n_classes, ha.n_heads = m->n_heads, ha.out_dim = m->out_dim;
    for (int i = 0; i < kMaxHeads; ++i) {
        ha.head_odim[i] = m->head_odim[i], ha.head_sigmoid[i] = m->head_sigmoid[i];
        ha.goff_head[i] = off.head[i];
    }
    for (int i = 0; i <= kMaxHeads; ++i) ha.lw[i] = 1.0f;
    if (h_loss_weights)
        for (int i = 0; i <= m->n_heads; ++i) ha.lw[i] = h_loss_weights[i];
    ha.goff_c3b = off.c3_b;
    ha.ext_losses = 1;
    size_t hpo = 0;
    for (int i = 0; i < m->n_heads; ++i) {  // d_hp: per head [gamma, beta, mean, var, out kernel, out bias], packed
        ha.hp_off[i] = hpo;
        hpo += 4 * kHidden + (size_t)kHidden * m->head_odim[i] + m->head_odim[i];
    }
    rc = launch_heads_train(ha, t->d_pre, d_y, m->d_hp, d_drop_heads, t->d_dpre, t->d_dxh, t->d_grad, t->d_bnstat, d_losses,
                            reinterpret_cast<unsigned *>(t->d_dxh + (size_t)t->max_batch * kPS), st);
    if (rc) return rc;
    BwdArgs ba;
    ba.gq = t->d_gq;  // nullptr unless smh_trainer_set_deterministic(t, 1)
    ba.split3 = 1;
    if (const char *ev = getenv("SMH_BWD_SPLIT")) ba.split3 = atoi(ev) != 0;
    ba.N = N, ba.T = m->cfg.patch_size, ba.F = m->cfg.n_feat, ba.n_blocks = m->n_blocks, ba.n_dil = m->cfg.n_dilations;
    ba.use_wt = 1;
    ba.stamps = getenv("SMH_BWD_STAMPS") ? 1 : 0;
    ba.D = m->D, ba.NH = m->NH, ba.n_classes = m->cfg.n_classes, ba.n_heads = m->n_heads, ba.off = off;
    hipLaunchKernelGGL(l2_penalty_kernel, dim3(m->n_heads * kL2Chunks), dim3(256), 0, st, ba, m->d_flat, d_losses + m->n_heads + 3,
                       t->d_l2part, reinterpret_cast<unsigned *>(t->d_l2part + kMaxHeads * kL2Chunks));
    rc = smh::launch_status("l2_penalty_kernel");
    if (rc) return rc;
    // MFMA backward (default); SMH_TRAIN_VALU=1 keeps the scalar reference kernel
    const int RPm = ((kMG * ba.T + 15) / 16) * 16;
    const size_t lds_m = sizeof(float) * ((size_t)4 * RPm * SX + 4 * C * kWS + kMG * kPS + C + kZW);
    const size_t lds_long = sizeof(float) * ((size_t)4 * RPm * SX + kMG * kPS + C + kZW);  // kernels stay in global memory
    const bool short_ok = lds_m <= 156 * 1024 && ba.T <= kMfmaMaxT, long_ok = lds_long <= 156 * 1024 && ba.T <= kMfmaLongT;
    bool bwd_done = false;
    if (bf16_bwd) {
        rc = launch_backward_bf16(ba, &t->d_bwd_pack, &t->bwd_pack_cap, d_x, m->d_flat, t->d_acts, d_drop_tcn, t->d_dpre, t->d_grad,
                                  (const float *)t->d_upre, st);
        if (rc) return rc;
        bwd_done = true;
    }
    if (bwd_done || ((short_ok || long_ok) && !getenv("SMH_TRAIN_VALU"))) {
        const dim3 grid((N + kMG - 1) / kMG);
        float *d_gt = nullptr;
        if (!bwd_done && !(getenv("SMH_DTRUNK") && atoi(getenv("SMH_DTRUNK")) == 0)) {  // (SMH_DTRUNK=0: the in-kernel loop, A/B and tests)
            if (!t->d_gt) SMH_CHECK_HIP(hipMalloc((void **)&t->d_gt, (size_t)t->max_batch * ba.D * sizeof(float)));
            d_gt = t->d_gt;
            rc = launch_dtrunk(ba, m->d_flat, t->d_acts, t->d_dpre, d_gt, st);
            if (rc) return rc;
        }
        if (bwd_done) {
        } else if (short_ok) {
            auto kern = tcn_backward_mfma_kernel<true, kMfmaMaxT>;
            SMH_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
            hipLaunchKernelGGL(kern, grid, dim3(kMThreads), lds_m, st, ba, d_x, m->d_flat, t->d_acts, d_drop_tcn, t->d_dpre, t->d_grad,
                               RPm, (const float *)t->d_upre, (const float *)d_gt);
        } else {
            auto kern = tcn_backward_mfma_kernel<false, kMfmaLongT>;
            SMH_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_long));
            hipLaunchKernelGGL(kern, grid, dim3(kMThreads), lds_long, st, ba, d_x, m->d_flat, t->d_acts, d_drop_tcn, t->d_dpre,
                               t->d_grad, RPm, (const float *)t->d_upre, (const float *)d_gt);
        }
        rc = smh::launch_status("tcn_backward_mfma_kernel");
        if (rc) return rc;
        if (getenv("SMH_DWH_VALU")) {  // the one-thread-per-element kernel (a second implementation for the tests)
            hipLaunchKernelGGL(dwh_kernel, dim3((ba.D * kHidden + 255) / 256, (N + kDwhSlice - 1) / kDwhSlice, 1 + ba.n_heads),
                               dim3(256), 0, st, ba, t->d_acts, t->d_dpre, t->d_grad, kDwhSlice);
            rc = smh::launch_status("dwh_kernel");
            return rc ? rc : det_finalize(t, st);
        }
        hipLaunchKernelGGL(dwh_mfma_kernel, dim3((ba.D / 16 + 3) / 4, 1 + ba.n_heads, kDwhSplit), dim3(256), 0, st, ba, t->d_acts,
                           t->d_dpre, t->d_grad);
        rc = smh::launch_status("dwh_mfma_kernel");
        return rc ? rc : det_finalize(t, st);
    }
    const int RP = kBG * ba.T;
    size_t lds = sizeof(float) * ((size_t)4 * RP * kBS + 2 * (3 * C * C + C * C) + C + 3 * RP + kBG * kPS);
    ba.use_wt = 1;
    if (lds > 156 * 1024) {  // long patches (the reference's W = 249): no room for the transposed kernel copies
        ba.use_wt = 0;
        lds -= sizeof(float) * (3 * C * C + C * C);
    }
    SMH_REQUIRE(lds <= 156 * 1024, "patch_size %d too long for the backward kernel (at most 264 frames)", ba.T);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)tcn_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(tcn_backward_kernel, dim3((N + kBG - 1) / kBG), dim3(kBThreads), lds, st, ba, d_x, m->d_flat,
                       t->d_acts, d_drop_tcn, t->d_dpre, t->d_grad);
    rc = smh::launch_status("tcn_backward_kernel");
    return rc ? rc : det_finalize(t, st);
}

extern "C" size_t smh_trainer_bucket_floats(const smh_trainer *t) { return t ? t->m->n_params + kMaxHeads * 32 : 0; }

extern "C" int smh_trainer_copy_state(smh_trainer *dst, const smh_trainer *src, void *stream) {
    SMH_REQUIRE(dst && src && dst->m == src->m, "smh_trainer_copy_state: both trainers must belong to the same model");
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = dst->m->n_params * sizeof(float);
    SMH_CHECK_HIP(hipMemcpyAsync(dst->d_vel, src->d_vel, nb, hipMemcpyDeviceToDevice, st));
    SMH_CHECK_HIP(hipMemcpyAsync(dst->d_s2, src->d_s2, nb, hipMemcpyDeviceToDevice, st));
    SMH_CHECK_HIP(hipStreamSynchronize(st));  // the caller destroys `src` next
    dst->step = src->step, dst->m_schedule = src->m_schedule;
    return SMH_OK;
}

extern "C" int smh_trainer_reset_state(smh_trainer *t, void *stream) {
    SMH_REQUIRE(t, "smh_trainer_reset_state: null trainer");
    hipStream_t st = (hipStream_t)stream;
    SMH_CHECK_HIP(hipMemsetAsync(t->d_vel, 0, t->m->n_params * sizeof(float), st));
    SMH_CHECK_HIP(hipMemsetAsync(t->d_s2, 0, t->m->n_params * sizeof(float), st));
    t->step = 0, t->m_schedule = 1.0;
    return SMH_OK;
}

extern "C" int smh_trainer_apply_f32(smh_trainer *t, int optimizer, float lr, float beta1, float beta2, float eps,
                                     float clipnorm, float grad_scale, unsigned active_mask, void *stream) {
    SMH_REQUIRE(t, "smh_trainer_apply_f32: null trainer");
    SMH_REQUIRE(optimizer >= 0 && optimizer <= 2, "smh_trainer_apply_f32: optimizer must be 0 (SGD), 1 (Adam) or 2 (Nadam)");
    smh_model *m = t->m;
    hipStream_t st = (hipStream_t)stream;
    t->step += 1;
    OptArgs o{};
    o.optimizer = optimizer, o.lr = lr, o.b1 = beta1, o.b2 = beta2, o.eps = eps, o.clipnorm = clipnorm, o.grad_scale = grad_scale;
    o.active_mask = active_mask;
    const double tt = (double)t->step;
    if (optimizer == 1) o.alpha = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, tt)) / (1.0 - std::pow((double)beta1, tt)));
    if (optimizer == 2) {  // tf.keras.optimizers.Nadam (2.x): momentum schedule u_t = b1 (1 - 0.5 * 0.96^(0.004 t))
        const double u_t = beta1 * (1.0 - 0.5 * std::pow(0.96, 0.004 * tt));
        const double u_t1 = beta1 * (1.0 - 0.5 * std::pow(0.96, 0.004 * (tt + 1.0)));
        const double ms_new = t->m_schedule * u_t, ms_next = ms_new * u_t1;
        t->m_schedule = ms_new;
        o.n_g = (float)((1.0 - u_t) / (1.0 - ms_new));   // (1 - u_t) * g / (1 - prod u)
        o.n_m = (float)(u_t1 / (1.0 - ms_next));         // u_{t+1} * m_t / (1 - prod u * u_{t+1})
        o.n_v = (float)(1.0 / (1.0 - std::pow((double)beta2, tt)));
    }
    hipLaunchKernelGGL(seg_sumsq_kernel, dim3(t->nseg), dim3(256), 0, st, t->d_segs, m->d_flat, t->d_grad, grad_scale, active_mask,
                       t->d_sumsq);
    int rc = smh::launch_status("seg_sumsq_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(seg_opt_kernel, dim3(t->nseg), dim3(256), 0, st, t->d_segs, o, m->d_flat, (const float *)t->d_grad, t->d_vel,
                       t->d_s2, (const float *)t->d_sumsq, (const float *)t->d_bnstat);
    rc = smh::launch_status("seg_opt_kernel");
    if (rc) return rc;
    return repack(m, st);
}

extern "C" int smh_trainer_apply_sgd_f32(smh_trainer *t, float lr, float momentum, float clipnorm, float grad_scale,
                                         void *stream) {
    return smh_trainer_apply_f32(t, 0, lr, momentum, 0.f, 0.f, clipnorm, grad_scale, 0xFFFFFFFFu, stream);
}
