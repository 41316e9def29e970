// a13 / a14: one training step of the Conv2D MTL baselines -- what `model.fit` runs per batch for the models compiled at
// lib/proposed_architectures.py:499-506 (Doukhan: Adam 1e-4), :572-580 (Papakostas: SGD, ExponentialDecay) and
// :750-757 (Jang: Adam 1e-3); losses S,M(,N): binary_crossentropy, R: mean_squared_error, 3C: categorical_crossentropy;
// l2(0.01) on the Dense(16) kernels of the heads (:46,60,73).  Arithmetic restated in oracle/cnn_mtl_train.py
// (torch autograd in float64 on the CPU).
//
// Every Conv2D / Dense costs three launches of the implicit-GEMM kernel of smh_cnn_impl.h on the f32 matrix cores:
//   forward   z  = im2col(x) W + b                         MODE 0
//   dgrad     dx = im2col'(dz) W^T   (stride 1: dz is the image, the tap table is mirrored, W transposed per (i,j))   MODE 0
//   wgrad     dW = im2col(x)^T dz    (rows = taps, reduction over the output pixels, ordered split partials)         MODE 1
// BatchNormalization runs on batch statistics (two-pass column sums over the (pixels x channels) matrix, ordered
// partials, f64 finish), ReLU and Dropout ride in the normalisation kernel; their backward is one column reduction
// plus one elementwise kernel, in place on the gradient buffer.  Max-pooling routes the gradient to the first maximum
// of each window (gather form, no atomics).  The heads reuse heads_train_kernel of the B3_MTL trainer.
// Papakostas adds LRN + ReLU backward (two passes), Conv2D without BatchNorm (gate + real bias gradients), a stride-2 data
// gradient (dz zero-stuffed, then the stride-1 GEMM) and overlapping pooling; Jang adds the mel-scale layer's weight
// gradient, a 3-channel data gradient (GEMM columns padded to 4), Dropout on feature maps and l2() on every kernel.
#include <cstdlib>

#include "smh_cnn_impl.h"
#include "smh_model.h"

namespace {

constexpr float kL2 = 0.01f;
constexpr float kBnMomentum = 0.99f;
constexpr int kPS = smh_tcn::kPS;
constexpr int kMaxRed = 1024;  // row chunks of a column reduction

// ---- column reductions over a row-major (M x C) matrix ---------------------------------------------------------
// F = 0: sum z           F = 1: sum (z - mean)^2          F = 2: g = relu'/dropout of dA; sums of g and g * xhat
struct RedArgs {
    const float *z, *dA, *mask, *mean, *rstd, *gamma, *beta;
    size_t M;
    int C, rows_per_block;
};
// the ReLU gate is re-derived from z with the forward's own expression (bitwise the same value), so the activation
// itself is not read again
template <int F>
__device__ __forceinline__ void red_item(const RedArgs &r, size_t i, int c, float &v0, float &v1) {
    if (F == 0) {
        v0 += r.z[i];
    } else if (F == 1) {
        const float d = r.z[i] - r.mean[c];
        v0 = fmaf(d, d, v0);
    } else {
        const float xh = (r.z[i] - r.mean[c]) * r.rstd[c];
        const float g = fmaf(xh, r.gamma[c], r.beta[c]) > 0.f ? r.dA[i] * (r.mask ? r.mask[i] : 1.f) : 0.f;
        v0 += g;
        v1 = fmaf(g, xh, v1);
    }
}
// four channels per thread (C % 4 == 0, 256 % (C/4) == 0): 16-byte loads, 256 / (C/4) rows per pass
template <int F>
__device__ __forceinline__ void red_item4(const RedArgs &r, size_t i, int c, f32x4 &v0, f32x4 &v1) {
    const f32x4 z = *reinterpret_cast<const f32x4 *>(r.z + i);
    if (F == 0) {
        v0 += z;
    } else {
        const f32x4 mean = *reinterpret_cast<const f32x4 *>(r.mean + c);
        const f32x4 d = z - mean;
        if (F == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v0[q] = fmaf(d[q], d[q], v0[q]);
        } else {
            const f32x4 rstd = *reinterpret_cast<const f32x4 *>(r.rstd + c), ga = *reinterpret_cast<const f32x4 *>(r.gamma + c),
                        be = *reinterpret_cast<const f32x4 *>(r.beta + c), dA = *reinterpret_cast<const f32x4 *>(r.dA + i);
            f32x4 mk = {1.f, 1.f, 1.f, 1.f};
            if (r.mask) mk = *reinterpret_cast<const f32x4 *>(r.mask + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xh = d[q] * rstd[q];
                const float g = fmaf(xh, ga[q], be[q]) > 0.f ? dA[q] * mk[q] : 0.f;
                v0[q] += g;
                v1[q] = fmaf(g, xh, v1[q]);
            }
        }
    }
}
template <int F>
__global__ void __launch_bounds__(256) colred_kernel(RedArgs r, float *__restrict__ partial) {
    constexpr int NV = F == 2 ? 2 : 1;
    __shared__ f32x4 sh[NV][256];
    const int tid = threadIdx.x, C = r.C;
    const size_t row0 = (size_t)blockIdx.x * r.rows_per_block;
    const size_t row1 = row0 + r.rows_per_block < r.M ? row0 + r.rows_per_block : r.M;
    float *out = partial + (size_t)blockIdx.x * NV * C;
    const int cq = C >> 2;
    if ((C & 3) == 0 && ((cq <= 256 && 256 % cq == 0) || cq % 256 == 0)) {
        // narrow matrices: 256 / cq rows per pass; wide ones (Dense 1024 / 2048 / 4096): blockIdx.y picks a 1024-column slab
        const int cqb = cq < 256 ? cq : 256;
        const int rp = 256 / cqb;  // rows per pass
        const int c = (blockIdx.y * 256 + tid % cqb) * 4, rsub = tid / cqb;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        for (size_t row = row0 + rsub; row < row1; row += rp) red_item4<F>(r, row * C + c, c, v0, v1);
        sh[0][tid] = v0;
        if (NV == 2) sh[NV - 1][tid] = v1;
        __syncthreads();
        if (tid < cqb) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < rp; ++q) {
                s0 += sh[0][q * cqb + tid];
                if (NV == 2) s1 += sh[NV - 1][q * cqb + tid];
            }
            *reinterpret_cast<f32x4 *>(out + c) = s0;
            if (NV == 2) *reinterpret_cast<f32x4 *>(out + C + c) = s1;
        }
    } else {
        for (int c = tid; c < C; c += 256) {
            float v0 = 0.f, v1 = 0.f;
            for (size_t row = row0; row < row1; ++row) red_item<F>(r, row * C + c, c, v0, v1);
            out[c] = v0;
            if (NV == 2) out[C + c] = v1;
        }
    }
}
// ordered f64 sum of the partials; mode 0: dst0 = s0 * scale (dst1 = s1 * scale when nv == 2)
//                                  mode 1: variance: dst0 = rstd, dst1[0..C) = mean copy, dst1[C..2C) = var * bessel
// 64 columns x 16 slices of the partial list per workgroup; the slices are combined in a fixed order.
__global__ void __launch_bounds__(1024) colred_finish_kernel(const float *__restrict__ partial, int nb, int C, int nv, float scale,
                                                             int mode, float bessel, const float *__restrict__ mean,
                                                             float *__restrict__ dst0, float *__restrict__ dst1) {
    __shared__ double sh[2][16][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0;
    if (c < C)
        for (int b = sub; b < nb; b += 16) {
            s0 += (double)partial[((size_t)b * nv) * C + c];
            if (nv == 2) s1 += (double)partial[((size_t)b * nv + 1) * C + c];
        }
    sh[0][sub][lane] = s0;
    sh[1][sub][lane] = s1;
    __syncthreads();
    if (sub != 0 || c >= C) return;
    s0 = s1 = 0.0;
    for (int q = 0; q < 16; ++q) s0 += sh[0][q][lane], s1 += sh[1][q][lane];
    if (mode == 0) {
        dst0[c] = (float)(s0 * (double)scale);
        if (nv == 2) dst1[c] = (float)(s1 * (double)scale);
    } else {
        const float var = (float)(s0 * (double)scale);
        dst0[c] = 1.0f / sqrtf(var + kBnEps);
        dst1[c] = mean[c];
        dst1[C + c] = var * bessel;
    }
}

// a = relu(gamma * (z - mean) * rstd + beta) * mask ; four channels per thread (C % 4 == 0)
__global__ void bn_apply_kernel(const float *__restrict__ z, size_t total4, int C, const float *__restrict__ mean,
                                const float *__restrict__ rstd, const float *__restrict__ gamma,
                                const float *__restrict__ beta, const float *__restrict__ mask, float *__restrict__ a) {
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= total4) return;
    const size_t i = i4 * 4;
    const int c = (int)(i % C);
    const f32x4 zz = *reinterpret_cast<const f32x4 *>(z + i), mu = *reinterpret_cast<const f32x4 *>(mean + c),
                rs = *reinterpret_cast<const f32x4 *>(rstd + c), ga = *reinterpret_cast<const f32x4 *>(gamma + c),
                be = *reinterpret_cast<const f32x4 *>(beta + c);
    f32x4 mk = {1.f, 1.f, 1.f, 1.f}, o;
    if (mask) mk = *reinterpret_cast<const f32x4 *>(mask + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = fmaxf(fmaf((zz[q] - mu[q]) * rs[q], ga[q], be[q]), 0.f) * mk[q];
    *reinterpret_cast<f32x4 *>(a + i) = o;
}
// in place on the gradient buffer: dA -> dz = gamma * rstd * (g - s1/M - xhat * s2/M); the ReLU gate comes from z
__global__ void bn_bwd_kernel(const float *__restrict__ z, const float *__restrict__ mask, size_t total4, int C, float invM,
                              const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                              const float *__restrict__ beta, const float *__restrict__ s12, float *__restrict__ g_io) {
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= total4) return;
    const size_t i = i4 * 4;
    const int c = (int)(i % C);
    const f32x4 zz = *reinterpret_cast<const f32x4 *>(z + i), mu = *reinterpret_cast<const f32x4 *>(mean + c),
                rs = *reinterpret_cast<const f32x4 *>(rstd + c), ga = *reinterpret_cast<const f32x4 *>(gamma + c),
                be = *reinterpret_cast<const f32x4 *>(beta + c), s1 = *reinterpret_cast<const f32x4 *>(s12 + c),
                s2 = *reinterpret_cast<const f32x4 *>(s12 + C + c);
    f32x4 mk = {1.f, 1.f, 1.f, 1.f}, g = *reinterpret_cast<const f32x4 *>(g_io + i);
    if (mask) mk = *reinterpret_cast<const f32x4 *>(mask + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float xh = (zz[q] - mu[q]) * rs[q];
        const float gg = fmaf(xh, ga[q], be[q]) > 0.f ? g[q] * mk[q] : 0.f;
        g[q] = ga[q] * rs[q] * (gg - s1[q] * invM - xh * s2[q] * invM);
    }
    *reinterpret_cast<f32x4 *>(g_io + i) = g;
}

// MaxPooling2D backward, gather form: the gradient of a window goes to its FIRST maximum in (dy, dx) scan order
// (TensorFlow's and torch's rule); post-ReLU ties at 0 are cut by the ReLU below anyway.
__global__ void maxpool_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy_,
                                   int H, int W, int C, int OH, int OW, int ph, int pw, int sh, int sw, int pt, int pl,
                                   size_t total, float *__restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H);
    const size_t img = r / H;
    const float v = x[i];
    float acc = 0.f;
    // windows (oy, ox) that contain (iy, ix): oy*sh - pt <= iy < oy*sh - pt + ph
    const int oy_hi = min(OH - 1, (iy + pt) / sh), ox_hi = min(OW - 1, (ix + pl) / sw);
    for (int oy = oy_hi; oy >= 0 && oy * sh - pt + ph > iy; --oy)
        for (int ox = ox_hi; ox >= 0 && ox * sw - pl + pw > ix; --ox) {
            const size_t o = ((img * OH + oy) * OW + ox) * C + c;
            if (y[o] != v) continue;
            bool first = true;  // no earlier tap of this window holds the same value
            const int y0 = oy * sh - pt, x0 = ox * sw - pl;
            for (int dy = 0; dy < ph && first; ++dy) {
                const int yy = y0 + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int dxx = 0; dxx < pw; ++dxx) {
                    const int xx = x0 + dxx;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    if (yy == iy && xx == ix) {
                        dy = ph;  // reached this tap: stop scanning
                        break;
                    }
                    if (x[((img * H + yy) * W + xx) * C + c] == v) {
                        first = false;
                        break;
                    }
                }
            }
            if (first) acc += dy_[o];
        }
    dx[i] = acc;
}

// stride == pool size (every pool of the Doukhan graph): one thread per window and channel finds the first maximum and
// writes all taps of the window; taps outside every window (valid-pooling remainder) are zeroed by the caller
__global__ void maxpool_bwd_tiles_kernel(const float *__restrict__ x, const float *__restrict__ dy_, int H, int W, int C, int OH,
                                         int OW, int ph, int pw, int pt, int pl, size_t total, float *__restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const size_t img = r / OH;
    const int y0 = oy * ph - pt, x0 = ox * pw - pl;
    float best = -__builtin_inff();
    int by = -1, bx = -1;
    for (int dy = 0; dy < ph; ++dy) {
        const int yy = y0 + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int dxx = 0; dxx < pw; ++dxx) {
            const int xx = x0 + dxx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float v = x[((img * H + yy) * W + xx) * C + c];
            if (v > best) best = v, by = yy, bx = xx;
        }
    }
    const float g = dy_[i];
    for (int dy = 0; dy < ph; ++dy) {
        const int yy = y0 + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int dxx = 0; dxx < pw; ++dxx) {
            const int xx = x0 + dxx;
            if ((unsigned)xx >= (unsigned)W) continue;
            dx[((img * H + yy) * W + xx) * C + c] = (yy == by && xx == bx) ? g : 0.f;
        }
    }
}

// ReLU behind a Conv2D without BatchNorm (Papakostas conv3): dz = a > 0 ? dA : 0, in place
__global__ void relu_bwd_kernel(const float *__restrict__ a, size_t total, float *__restrict__ g_io) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total && !(a[i] > 0.f)) g_io[i] = 0.f;
}
// tf.nn.local_response_normalization + ReLU backward.  y = relu(v), v = x * u^-beta, u = 1 + alpha * sum_{|d-c|<=r} x_d^2:
//   dx_c = dv_c * u_c^-beta - 2 alpha beta x_c * sum_{|d-c|<=r} dv_d x_d u_d^(-beta-1),   dv = y > 0 ? dy : 0.
// Pass 1 (in place on the gradient): g <- dv * u^-beta, t <- dv * x * u^(-beta-1); pass 2 gathers the window of t.
__global__ void lrn_bwd1_kernel(const float *__restrict__ x, const float *__restrict__ y, int C, int radius, float alpha,
                                float beta, size_t total, float *__restrict__ g_io, float *__restrict__ t) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float *px = x + (i - c);
    float s = 0.f;
    for (int d = max(0, c - radius); d <= min(C - 1, c + radius); ++d) s = fmaf(px[d], px[d], s);
    const float u = 1.f + alpha * s;
    const float dv = y[i] > 0.f ? g_io[i] : 0.f;
    const float ub = powf(u, -beta);
    g_io[i] = dv * ub;
    t[i] = dv * px[c] * ub / u;
}
__global__ void lrn_bwd2_kernel(const float *__restrict__ x, const float *__restrict__ p, const float *__restrict__ t, int C,
                                int radius, float alpha, float beta, size_t total, float *__restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float *pt = t + (i - c);
    float s = 0.f;
    for (int d = max(0, c - radius); d <= min(C - 1, c + radius); ++d) s += pt[d];
    dx[i] = p[i] - 2.f * alpha * beta * x[i] * s;
}
// strided Conv2D data gradient: dz (N, OH, OW, C) -> zero-stuffed (N, (OH-1)*sh+1, (OW-1)*sw+1, C), so that the
// stride-1 data-gradient GEMM applies (the buffer is zeroed first)
__global__ void stuff_kernel(const float *__restrict__ dz, int OH, int OW, int C, int sh, int sw, int SH, int SW, size_t total,
                             float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const size_t img = r / OH;
    out[((img * SH + (size_t)oy * sh) * SW + (size_t)ox * sw) * C + c] = dz[i];
}

// Wt[(i,j)][co][ci] = W[(i,j)][ci][co]; rows of Wt are `ld` >= Cin wide (zero padded: the GEMM wants a multiple of 4)
__global__ void transpose_taps_kernel(const float *__restrict__ w, int Cin, int Cout, int ld, size_t total,
                                      float *__restrict__ wt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % ld);
    const size_t r = i / ld;
    const int co = (int)(r % Cout);
    const size_t ij = r / Cout;
    wt[i] = ci < Cin ? w[(ij * Cin + ci) * Cout + co] : 0.f;
}
// Jang's mel-scale layer (melcl_kernel of smh_cnn_impl.h), weight gradient: y = tanh(sum_{b,d} x[top+b][t+d-half] w[b][d][ch])
// one workgroup per filter row, one thread per weight (b, d, ch), the batch and the frames summed in order.
// dA has `ldc` floats per pixel (the data gradient of the next Conv2D is written with a padded channel count).
__global__ void __launch_bounds__(256) melcl_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ dA, const MelCl *__restrict__ f, int N, int rows_in,
                                                        int W, int rows_out, int tdim, int ldc, float *__restrict__ grad) {
    const int r = blockIdx.x;
    const MelCl e = f[r];
    const int half = tdim / 2;
    for (int q = threadIdx.x; q < e.width * tdim * 3; q += blockDim.x) {
        const int ch = q % 3, bd = q / 3, d = bd % tdim, b = bd / tdim;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            const float *xr = x + ((size_t)n * rows_in + e.top + b) * W;
            const size_t o = ((size_t)n * rows_out + r) * W;
            for (int t = 0; t < W; ++t) {
                const int tt = t + d - half;
                if ((unsigned)tt >= (unsigned)W) continue;
                const float yv = y[(o + t) * 3 + ch];
                acc = fmaf(xr[tt], dA[(o + t) * ldc + ch] * (1.f - yv * yv), acc);
            }
        }
        grad[e.woff + q] = acc;
    }
}

__global__ void rowinfo_kernel(int H, int W, int C, int OH, int OW, int sh, int sw, int pt, int pl, size_t total,
                               int2 *__restrict__ out) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= total) return;
    const int ohw = OH * OW;
    const int img = (int)(m / ohw), r = (int)(m - (size_t)img * ohw);
    const int oy = r / OW, ox = r - oy * OW;
    const int iy0 = oy * sh - pt, ix0 = ox * sw - pl;
    out[m] = int2{(int)((((long)img * H + iy0) * W + ix0) * C), (iy0 & 0xffff) | (ix0 << 16)};
}

// dW[K x 64-wide column block] for a shallow first layer (K = kh*kw*Cin <= 32, e.g. conv1: 20): the MFMA tile would be
// 84 % padding, so this one runs on the VALU.  Workgroup = 64 output channels x 4 row groups; 64 output pixels at a time
// have their K taps staged in LDS (broadcast reads), every lane owns one channel and K accumulators.
constexpr int kSmallK = 32;
__global__ void __launch_bounds__(256) wgrad_smallk_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                           const int2 *__restrict__ lut, const int2 *__restrict__ rowinfo, int H,
                                                           int W, int K, int Cout, int M, int rows_per_block,
                                                           float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float xs[64][kSmallK];
    __shared__ float red[4][kSmallK][64];
    const int tid = threadIdx.x, lane = tid & 63, rg = tid >> 6;
    const int co = blockIdx.y * 64 + lane;
    const int m_begin = blockIdx.x * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float acc[kSmallK];
#pragma unroll
    for (int k = 0; k < kSmallK; ++k) acc[k] = 0.f;
    for (int m0 = m_begin; m0 < m_end; m0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * kSmallK; e += 256) {
            const int r = e / kSmallK, k = e - r * kSmallK;
            float v = 0.f;
            if (m0 + r < m_end && k < K) {
                const int2 ri = rowinfo[m0 + r], ek = lut[k];
                const int iy = (short)(ri.y & 0xffff) + (short)(ek.y & 0xffff), ix = (ri.y >> 16) + (ek.y >> 16);
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(long)ri.x + ek.x];
            }
            xs[r][k] = v;
        }
        __syncthreads();
        if (co < Cout)
            for (int r = rg; r < 64 && m0 + r < m_end; r += 4) {
                const float g = dz[(size_t)(m0 + r) * Cout + co];
#pragma unroll
                for (int k4 = 0; k4 < kSmallK / 4; ++k4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xs[r][4 * k4]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[4 * k4 + q] = fmaf(xv[q], g, acc[4 * k4 + q]);
                }
            }
    }
#pragma unroll
    for (int k = 0; k < kSmallK; ++k) red[rg][k][lane] = acc[k];
    __syncthreads();
    if (co < Cout)
        for (int k = rg; k < K; k += 4)
            partial[((size_t)blockIdx.x * K + k) * Cout + co] = (red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane]);
}

__global__ void partial_sum_kernel(const float *__restrict__ partial, int S, size_t n, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += partial[(size_t)s * n + i];  // fixed order
    out[i] = v;
}

// ---- heads ------------------------------------------------------------------------------------------------------
struct HeadPtrs {
    size_t c3k, c3b, hk[kMaxHeads], hb[kMaxHeads];
    int D, n_classes, n_heads;
};
// pre (N, kPS) = [feat @ 3C kernel + bias | feat @ Dense(16) kernel + bias per head]; one workgroup per sample
__global__ void __launch_bounds__(256) heads_pre_kernel(const float *__restrict__ feat, const float *__restrict__ F, HeadPtrs a,
                                                        float *__restrict__ pre) {
    constexpr int MAXV = 5 + kMaxHeads * kHidden;
    __shared__ float red[4][MAXV];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) acc[v] = 0.f;
    const float *f = feat + (size_t)n * a.D;
    for (int d = tid; d < a.D; d += 256) {
        const float x = f[d];
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (c < a.n_classes) acc[c] = fmaf(x, F[a.c3k + (size_t)d * a.n_classes + c], acc[c]);
#pragma unroll
        for (int h = 0; h < kMaxHeads; ++h)
            if (h < a.n_heads) {
                const float *kr = F + a.hk[h] + (size_t)d * kHidden;
#pragma unroll
                for (int j = 0; j < kHidden; ++j) acc[5 + h * kHidden + j] = fmaf(x, kr[j], acc[5 + h * kHidden + j]);
            }
    }
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        float s = acc[v];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave][v] = s;
    }
    __syncthreads();
    float *o = pre + (size_t)n * kPS;
    if (tid < kPS) {
        float v = 0.f;
        if (tid < a.n_classes) {
            v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]) + F[a.c3b + tid];
        } else if (tid < a.n_classes + a.n_heads * kHidden) {
            const int j = tid - a.n_classes, s = 5 + j;
            v = (red[0][s] + red[1][s]) + (red[2][s] + red[3][s]) + F[a.hb[j / kHidden] + j % kHidden];
        }
        o[tid] = v;
    }
}
// dfeat[n][d] = sum_c dpre[n][c] * Wcat[d][c]
__global__ void heads_dfeat_kernel(const float *__restrict__ dpre, const float *__restrict__ F, HeadPtrs a, int N,
                                   float *__restrict__ dfeat) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * a.D) return;
    const int d = (int)(i % a.D);
    const size_t n = i / a.D;
    const float *dp = dpre + n * kPS;
    float acc = 0.f;
    for (int c = 0; c < a.n_classes; ++c) acc = fmaf(dp[c], F[a.c3k + (size_t)d * a.n_classes + c], acc);
    for (int h = 0; h < a.n_heads; ++h) {
        const float *kr = F + a.hk[h] + (size_t)d * kHidden;
        const float *dh = dp + a.n_classes + h * kHidden;
#pragma unroll
        for (int j = 0; j < kHidden; ++j) acc = fmaf(dh[j], kr[j], acc);
    }
    dfeat[i] = acc;
}
// dW of the '3C' kernel (grid.y = 0) and of each head's Dense(16) kernel (grid.y = 1 + h): one thread per element,
// the batch summed in order
__global__ void heads_dw_kernel(const float *__restrict__ feat, const float *__restrict__ dpre, HeadPtrs a, int N,
                                float *__restrict__ grad) {
    const int grp = blockIdx.y;
    const int oc = grp == 0 ? a.n_classes : kHidden;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.D * oc) return;
    const int d = i / oc, o = i - d * oc;
    const int col = grp == 0 ? o : a.n_classes + (grp - 1) * kHidden + o;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(feat[(size_t)n * a.D + d], dpre[(size_t)n * kPS + col], acc);
    grad[(grp == 0 ? a.c3k : a.hk[grp - 1]) + i] = acc;
}

// ---- optimiser ----------------------------------------------------------------------------------------------------
struct Seg {
    unsigned off, size;
    int kind;  // 0 plain, 1 l2-regularised kernel, 2 BN moving_mean, 3 BN moving_variance
    unsigned aux;  // kinds 2/3: offset into the batch-statistics buffer
};
struct OptArgs {
    int optimizer;  // 0: SGD(momentum), 1: Adam
    float lr, b1, b2, eps, grad_scale, alpha;  // alpha = lr * sqrt(1 - b2^t) / (1 - b1^t) for Adam
};
__global__ void __launch_bounds__(256) opt_kernel(const Seg *__restrict__ segs, OptArgs o, float *__restrict__ w,
                                                  float *__restrict__ grad, float *__restrict__ s1, float *__restrict__ s2,
                                                  const float *__restrict__ bstat) {
    const Seg s = segs[blockIdx.x];
    if (s.kind >= 2) {
        for (unsigned i = threadIdx.x; i < s.size; i += blockDim.x)
            w[s.off + i] = kBnMomentum * w[s.off + i] + (1.0f - kBnMomentum) * (bstat[s.aux + i] * o.grad_scale);
        return;
    }
    for (unsigned i = threadIdx.x; i < s.size; i += blockDim.x) {
        const size_t k = (size_t)s.off + i;
        float g = grad[k] * o.grad_scale;
        if (s.kind == 1) g += 2.0f * kL2 * w[k];
        grad[k] = g;
        if (o.optimizer == 1) {
            const float m = o.b1 * s1[k] + (1.0f - o.b1) * g;
            const float v = o.b2 * s2[k] + (1.0f - o.b2) * g * g;
            s1[k] = m, s2[k] = v;
            w[k] -= o.alpha * m / (sqrtf(v) + o.eps);
        } else {
            const float v = o.b1 * s1[k] - o.lr * g;
            s1[k] = v;
            w[k] += v;
        }
    }
}
// l2(0.01) penalty of the regularised kernels: one workgroup per optimiser segment (<= 64 K weights) sums w^2 in f64,
// a second launch adds the per-segment sums in order.  (Jang regularises all 74 M weights: a single workgroup took 30 ms.)
__global__ void __launch_bounds__(256) l2_partial_kernel(const Seg *__restrict__ segs, const float *__restrict__ w,
                                                         double *__restrict__ partial) {
    __shared__ double sh[4];
    const Seg sg = segs[blockIdx.x];
    double s = 0.0;
    if (sg.kind == 1)
        for (unsigned i = threadIdx.x; i < sg.size; i += blockDim.x) s += (double)w[sg.off + i] * (double)w[sg.off + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void __launch_bounds__(1024) l2_finish_kernel(const double *__restrict__ partial, int nseg, float *__restrict__ out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int q = threadIdx.x; q < nseg; q += blockDim.x) s += partial[q];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sh[q];
        out[0] = (float)((double)kL2 * t);
    }
}
// inference epilogue y = acc * scale + shift of every Conv2D / Dense from the current weights (what
// smh_cnn_set_weights computes on the host)
struct FoldEnt {
    unsigned es_off, oc;
    long bias_off, bn_off;  // -1: absent
};
__global__ void refold_kernel(const FoldEnt *__restrict__ ents, const float *__restrict__ w, float *__restrict__ fold) {
    const FoldEnt e = ents[blockIdx.x];
    for (unsigned c = threadIdx.x; c < e.oc; c += blockDim.x) {
        float s = 1.f, t = 0.f;
        if (e.bn_off >= 0) {
            const float *g = w + e.bn_off;
            s = g[c] / sqrtf(g[3 * e.oc + c] + kBnEps);
            t = g[e.oc + c] - g[2 * e.oc + c] * s;
        }
        fold[e.es_off + c] = s;
        fold[e.es_off + e.oc + c] = (e.bias_off >= 0 ? w[e.bias_off + c] * s : 0.f) + t;
    }
}
__global__ void fill_kernel(float *p, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct LayerState {
    const float *in = nullptr;   // input activation (set per step)
    float *z = nullptr, *a = nullptr;
    size_t in_elems = 0, out_elems = 0;  // per image
    int2 *rowinfo = nullptr, *dlut = nullptr;
    int dK = 0, dKp = 0;          // dgrad GEMM depth kh*kw*Cout (padded)
    int SH = 0, SW = 0;           // height / width of dz as the data-gradient GEMM sees it (zero-stuffed when strided)
    float *mean = nullptr, *rstd = nullptr, *s12 = nullptr;  // per channel
    unsigned bstat_off = 0;       // [mean | var] of this layer's BN in the batch-statistics buffer
    size_t drop_off = 0;          // offset (in units of one image's floats) of this layer's mask block
};

inline unsigned nblk(size_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

struct smh_cnn_trainer {
    smh_cnn *m;
    int max_batch;
    std::vector<LayerState> ls;
    size_t drop_per_image = 0;  // floats of dropout mask per image, all layers
    int n_drop = 0;
    float *d_arena = nullptr, *d_g[2] = {nullptr, nullptr}, *d_partial = nullptr, *d_red = nullptr, *d_wt = nullptr;
    float *d_chan = nullptr, *d_ones = nullptr, *d_zeros = nullptr, *d_tmp = nullptr;
    float *d_pre = nullptr, *d_dpre = nullptr, *d_dxh = nullptr, *d_scratch = nullptr;
    float *d_grad = nullptr, *d_s1 = nullptr, *d_s2 = nullptr, *d_bstat = nullptr;
    int2 *d_tables = nullptr;
    double *d_l2part = nullptr;
    Seg *d_segs = nullptr;
    FoldEnt *d_foldents = nullptr;
    int nseg = 0, nfold = 0;
    unsigned head_bstat = 0;
    size_t partial_floats = 0, g_floats = 0, bstat_floats = 0;
    long step = 0;
};

namespace {

int wgrad_split(int blocks, int ksteps, size_t out_floats, size_t cap_floats) {
    int s = 1;
    if (blocks < 1024 && ksteps >= 32) {
        s = (1024 + blocks - 1) / blocks;
        if (s > ksteps / 16) s = ksteps / 16;
        if (s > 256) s = 256;
        while (s > 1 && (size_t)s * out_floats > cap_floats) --s;
        if (s < 1) s = 1;
    }
    return s;
}

template <int MODE>
void launch_gemm(const ConvArgs &a, int rows, int bn, hipStream_t st) {
    const dim3 grid((rows + BM - 1) / BM, (a.Cout + bn - 1) / bn, a.ksplit);
    if (bn == 64) hipLaunchKernelGGL((conv_gemm_kernel<64, MODE>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<128, MODE>), grid, dim3(256), 0, st, a);
}

// column sums of an (M x C) matrix into dst0 (and dst1): reduction + ordered finish
template <int F>
int col_reduce(smh_cnn_trainer *t, RedArgs r, int nv, float scale, int mode, float bessel, float *dst0, float *dst1,
               hipStream_t st) {
    const int C = r.C;
    const int cq = C >> 2;
    const bool vec = (C & 3) == 0 && ((cq <= 256 && 256 % cq == 0) || cq % 256 == 0);  // the kernel's own test
    const size_t rows_per_pass = vec ? (cq < 256 ? 256 / cq : 1) : 1;
    const unsigned slabs = vec && cq > 256 ? cq / 256 : 1;
    size_t rpb = vec ? rows_per_pass * 16 : 64;
    size_t nb = (r.M + rpb - 1) / rpb;
    if (nb > kMaxRed) {
        rpb = (r.M + kMaxRed - 1) / kMaxRed;
        rpb = (rpb + rows_per_pass - 1) / rows_per_pass * rows_per_pass;
        nb = (r.M + rpb - 1) / rpb;
    }
    r.rows_per_block = (int)rpb;
    hipLaunchKernelGGL(colred_kernel<F>, dim3((unsigned)nb, slabs), dim3(256), 0, st, r, t->d_red);
    hipLaunchKernelGGL(colred_finish_kernel, dim3(nblk(C, 64)), dim3(1024), 0, st, (const float *)t->d_red, (int)nb, C, nv, scale,
                       mode, bessel, r.mean, dst0, dst1);
    return smh::launch_status("colred_kernel");
}

}  // namespace

extern "C" void smh_cnn_trainer_destroy(smh_cnn_trainer *t) {
    if (!t) return;
    for (float *p : {t->d_arena, t->d_g[0], t->d_g[1], t->d_partial, t->d_red, t->d_wt, t->d_chan, t->d_ones, t->d_zeros, t->d_tmp, t->d_pre,
                     t->d_dpre, t->d_dxh, t->d_scratch, t->d_grad, t->d_s1, t->d_s2})
        (void)hipFree(p);
    (void)hipFree(t->d_tables);
    (void)hipFree(t->d_l2part);
    (void)hipFree(t->d_segs);
    (void)hipFree(t->d_foldents);
    delete t;
}

extern "C" int smh_cnn_trainer_create(smh_cnn *m, int max_batch, smh_cnn_trainer **out) {
    SMH_REQUIRE(m && out && max_batch >= 2, "smh_cnn_trainer_create: bad argument (a training batch needs at least 2 samples)");
    SMH_REQUIRE(m->cfg.kind == SMH_CNN_DOUKHAN || m->cfg.kind == SMH_CNN_PAPAKOSTAS || m->cfg.kind == SMH_CNN_JANG,
                "smh_cnn_trainer_create: unknown model kind");
    // owned until the end: every early return (SMH_REQUIRE included) releases what has been allocated so far
    struct Guard {
        smh_cnn_trainer *p;
        ~Guard() { smh_cnn_trainer_destroy(p); }
    } guard{new smh_cnn_trainer()};
    smh_cnn_trainer *t = guard.p;
    t->m = m, t->max_batch = max_batch;
    const size_t NB = (size_t)max_batch;
    const int nl = (int)m->layers.size();
    t->ls.resize(nl);
    // sizes
    size_t arena = 0, tables = 0, chan = 0, maxg = (size_t)m->cfg.in_h * m->cfg.in_w, wt = 0, maxC = 64, tmp = 0;
    unsigned bstat = 0;
    size_t in_elems = (size_t)m->cfg.in_h * m->cfg.in_w;
    for (int l = 0; l < nl; ++l) {
        const Layer &L = m->layers[l];
        LayerState &S = t->ls[l];
        S.in_elems = in_elems;
        S.out_elems = (size_t)L.OH * L.OW * L.OC;
        maxg = std::max(maxg, std::max(S.in_elems, S.out_elems));
        if (L.op == kConv) {
            SMH_REQUIRE(L.OC % 4 == 0 && (L.t_bn < 0 || L.act == kRelu) && (L.act == kRelu || L.act == kNone) &&
                            ((L.sh == 1 && L.sw == 1) || (L.pt == 0 && L.pl == 0)),
                        "smh_cnn_trainer_create: unsupported Conv2D variant");
            arena += (L.t_bn >= 0 ? 2 : 1) * S.out_elems * NB;  // z (in front of a BatchNorm) and a
            S.SH = (L.OH - 1) * L.sh + 1, S.SW = (L.OW - 1) * L.sw + 1;
            if (l > 0 && (L.sh > 1 || L.sw > 1)) tmp = std::max(tmp, (size_t)S.SH * S.SW * L.OC * NB);
            tables += (size_t)L.OH * L.OW * NB;       // rowinfo
            S.dK = L.kh * L.kw * L.OC, S.dKp = (S.dK + BK - 1) / BK * BK;
            const size_t ldc = ((size_t)L.C + 3) & ~(size_t)3;  // the data gradient is written with a padded channel count
            if (l > 0) tables += S.dKp, wt = std::max(wt, (size_t)L.kh * L.kw * L.OC * ldc), maxg = std::max(maxg, (size_t)L.H * L.W * ldc);
            chan += 4 * (size_t)L.OC;                 // mean, rstd, s1|s2
            S.bstat_off = bstat, bstat += 2 * L.OC;
            maxC = std::max(maxC, (size_t)L.OC);
            if (L.drop > 0.f) S.drop_off = t->drop_per_image, t->drop_per_image += S.out_elems, t->n_drop++;
        } else if (L.op == kPool) {
            arena += S.out_elems * NB;
        } else if (L.op == kLrnRelu) {
            arena += S.out_elems * NB;
            tmp = std::max(tmp, S.out_elems * NB);
        } else if (L.op == kMelCl && l == 0) {
            arena += S.out_elems * NB;
        } else {
            return smh::set_error(SMH_E_INVALID, "smh_cnn_trainer_create: layer kind %d has no backward", (int)L.op);
        }
        in_elems = S.out_elems;
    }
    t->head_bstat = bstat, bstat += kMaxHeads * 32;
    t->g_floats = maxg * NB;
    t->partial_floats = (size_t)48 << 20;  // 192 MB of split partials
    // forward split-K partials of the GEMMs at full batch also live in d_partial
    for (const Layer &L : m->layers)
        if (L.op == kConv) {
            const size_t M = NB * L.OH * L.OW;
            t->partial_floats = std::max(t->partial_floats, (size_t)16 * std::min<size_t>(M, 4096) * L.OC);
        }
    hipError_t e = hipMalloc((void **)&t->d_arena, std::max<size_t>(arena, 1) * sizeof(float));
    auto alloc = [&](float **p, size_t n) {
        if (e == hipSuccess) e = hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(float));
    };
    alloc(&t->d_g[0], t->g_floats), alloc(&t->d_g[1], t->g_floats);
    alloc(&t->d_partial, t->partial_floats);
    alloc(&t->d_red, (size_t)kMaxRed * 2 * maxC);
    alloc(&t->d_wt, wt);
    alloc(&t->d_tmp, tmp);
    alloc(&t->d_chan, chan);
    size_t maxN = maxC;  // widest GEMM output: the data gradient of a layer has its INPUT width
    for (const Layer &L : m->layers)
        if (L.op == kConv) maxN = std::max(maxN, (size_t)L.C);
    alloc(&t->d_ones, maxN), alloc(&t->d_zeros, maxN);
    alloc(&t->d_pre, NB * kPS), alloc(&t->d_dpre, NB * kPS), alloc(&t->d_dxh, NB * kPS + 4);  // + the heads kernel's ticket
    alloc(&t->d_scratch, NB * (size_t)m->out_dim);
    // ONE bucket [gradient (n_params) | BatchNorm batch statistics (bstat)]: what data-parallel training all-reduces
    alloc(&t->d_grad, m->n_params + bstat), alloc(&t->d_s1, m->n_params), alloc(&t->d_s2, m->n_params);
    t->bstat_floats = bstat;
    if (e == hipSuccess) t->d_bstat = t->d_grad + m->n_params;
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_tables, std::max<size_t>(tables, 1) * sizeof(int2));
    if (e != hipSuccess)
        return smh::set_error(SMH_E_HIP, "smh_cnn_trainer_create: device allocation failed: %s", hipGetErrorString(e));
    // carve + tables
    float *ap = t->d_arena, *cp = t->d_chan;
    int2 *tp = t->d_tables;
    std::vector<Seg> segs;
    std::vector<FoldEnt> folds;
    auto add_seg = [&](size_t off, size_t n, int kind, unsigned aux) {
        constexpr size_t kMax = 1 << 16;  // one workgroup per <= 64K parameters
        if (kind >= 2) {
            segs.push_back(Seg{(unsigned)off, (unsigned)n, kind, aux});
            return;
        }
        for (size_t o = 0; o < n; o += kMax) segs.push_back(Seg{(unsigned)(off + o), (unsigned)std::min(kMax, n - o), kind, 0});
    };
    SMH_REQUIRE(m->n_params < ((size_t)1 << 32), "smh_cnn_trainer_create: model too large");
    for (int l = 0; l < nl; ++l) {
        const Layer &L = m->layers[l];
        LayerState &S = t->ls[l];
        if (L.op == kConv) {
            if (L.t_bn >= 0) S.z = ap, ap += S.out_elems * NB;
            S.a = ap, ap += S.out_elems * NB;
            S.rowinfo = tp, tp += (size_t)L.OH * L.OW * NB;
            const size_t total = (size_t)L.OH * L.OW * NB;
            hipLaunchKernelGGL(rowinfo_kernel, dim3(nblk(total)), dim3(256), 0, 0, L.H, L.W, L.C, L.OH, L.OW, L.sh, L.sw, L.pt,
                               L.pl, total, S.rowinfo);
            if (l > 0) {
                std::vector<int2> dl(S.dKp);
                for (int k = 0; k < S.dKp; ++k) {
                    if (k < S.dK) {
                        const int co = k % L.OC, ij = k / L.OC, j = ij % L.kw, i = ij / L.kw;
                        dl[k] = int2{(-i * S.SW - j) * L.OC + co, ((-i) & 0xffff) | ((-j) << 16)};
                    } else {
                        dl[k] = int2{0, 0x7fff | (0x7fff << 16)};
                    }
                }
                S.dlut = tp, tp += S.dKp;
                e = hipMemcpy(S.dlut, dl.data(), dl.size() * sizeof(int2), hipMemcpyHostToDevice);
                if (e != hipSuccess) break;
            }
            S.mean = cp, S.rstd = cp + L.OC, S.s12 = cp + 2 * L.OC, cp += 4 * (size_t)L.OC;
            add_seg(m->tensors[L.t_kernel].off, m->tensors[L.t_kernel].count, L.l2 ? 1 : 0, 0);
            if (L.t_bias >= 0) add_seg(m->tensors[L.t_bias].off, L.OC, 0, 0);
            long g = -1;
            if (L.t_bn >= 0) {
                g = (long)m->tensors[L.t_bn].off;
                add_seg(g, 2 * (size_t)L.OC, 0, 0);  // gamma, beta
                add_seg(g + 2 * (size_t)L.OC, L.OC, 2, S.bstat_off);
                add_seg(g + 3 * (size_t)L.OC, L.OC, 3, S.bstat_off + L.OC);
            }
            folds.push_back(FoldEnt{(unsigned)L.es_off, (unsigned)L.OC, L.t_bias >= 0 ? (long)m->tensors[L.t_bias].off : -1, g});
        } else {
            S.a = ap, ap += S.out_elems * NB;
            if (L.op == kMelCl)  // 2 * n_mels trainable kernels, each with kernel_regularizer=l2() (:630, :639)
                for (int q = 0; q < m->mel_rows; ++q) add_seg(m->tensors[q].off, m->tensors[q].count, 1, 0);
        }
    }
    if (e == hipSuccess) {
        add_seg(m->tensors[m->t_c3].off, m->tensors[m->t_c3].count, m->c3_l2 ? 1 : 0, 0);
        add_seg(m->tensors[m->t_c3 + 1].off, m->cfg.n_classes, 0, 0);
        for (int h = 0; h < m->n_heads; ++h) {
            const int th = m->t_head[h];
            add_seg(m->tensors[th].off, m->tensors[th].count, 1, 0);           // Dense(16) kernel, l2()
            add_seg(m->tensors[th + 1].off, 3 * kHidden, 0, 0);                 // dense bias, gamma, beta
            add_seg(m->tensors[th + 4].off, kHidden, 2, t->head_bstat + h * 32);
            add_seg(m->tensors[th + 5].off, kHidden, 3, t->head_bstat + h * 32 + 16);
            add_seg(m->tensors[th + 6].off, (size_t)kHidden * m->odim[h] + m->odim[h], 0, 0);  // out kernel + bias
        }
        t->nseg = (int)segs.size(), t->nfold = (int)folds.size();
        e = hipMalloc((void **)&t->d_segs, segs.size() * sizeof(Seg));
        if (e == hipSuccess) e = hipMalloc((void **)&t->d_l2part, segs.size() * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&t->d_foldents, std::max<size_t>(folds.size(), 1) * sizeof(FoldEnt));
        if (e == hipSuccess) e = hipMemcpy(t->d_segs, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice);
        if (e == hipSuccess && !folds.empty()) e = hipMemcpy(t->d_foldents, folds.data(), folds.size() * sizeof(FoldEnt), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(t->d_s1, 0, m->n_params * sizeof(float));
        if (e == hipSuccess) e = hipMemset(t->d_s2, 0, m->n_params * sizeof(float));
        if (e == hipSuccess) e = hipMemset(t->d_grad, 0, (m->n_params + t->bstat_floats) * sizeof(float));
        if (e == hipSuccess) e = hipMemset(t->d_zeros, 0, maxN * sizeof(float));
        if (e == hipSuccess) e = hipMemset(t->d_dxh + NB * kPS, 0, 4 * sizeof(float));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(fill_kernel, dim3(nblk(maxN)), dim3(256), 0, 0, t->d_ones, maxN, 1.0f);
            e = hipDeviceSynchronize();
        }
    }
    if (e != hipSuccess) return smh::set_error(SMH_E_HIP, "smh_cnn_trainer_create: setup failed: %s", hipGetErrorString(e));
    guard.p = nullptr;
    *out = t;
    return SMH_OK;
}

extern "C" float *smh_cnn_trainer_grad_ptr(smh_cnn_trainer *t) { return t ? t->d_grad : nullptr; }
extern "C" size_t smh_cnn_trainer_bucket_floats(const smh_cnn_trainer *t) { return t ? t->m->n_params + t->bstat_floats : 0; }
extern "C" int smh_cnn_trainer_copy_state(smh_cnn_trainer *dst, const smh_cnn_trainer *src, void *stream) {
    SMH_REQUIRE(dst && src && dst->m == src->m, "smh_cnn_trainer_copy_state: both trainers must belong to the same model");
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = dst->m->n_params * sizeof(float);
    SMH_CHECK_HIP(hipMemcpyAsync(dst->d_s1, src->d_s1, nb, hipMemcpyDeviceToDevice, st));
    SMH_CHECK_HIP(hipMemcpyAsync(dst->d_s2, src->d_s2, nb, hipMemcpyDeviceToDevice, st));
    SMH_CHECK_HIP(hipStreamSynchronize(st));  // the caller destroys `src` next
    dst->step = src->step;
    return SMH_OK;
}
extern "C" int smh_cnn_trainer_num_dropouts(const smh_cnn_trainer *t) { return t ? t->n_drop : SMH_E_INVALID; }

extern "C" int smh_cnn_trainer_dropout_info(const smh_cnn_trainer *t, int i, size_t *dim, float *rate) {
    SMH_REQUIRE(t, "smh_cnn_trainer_dropout_info: null trainer");
    int k = 0;
    for (size_t l = 0; l < t->ls.size(); ++l) {
        const Layer &L = t->m->layers[l];
        if (L.op != kConv || L.drop <= 0.f) continue;
        if (k++ == i) {
            if (dim) *dim = t->ls[l].out_elems;
            if (rate) *rate = L.drop;
            return SMH_OK;
        }
    }
    return smh::set_error(SMH_E_INVALID, "smh_cnn_trainer_dropout_info: index %d out of range", i);
}

extern "C" int smh_cnn_train_step_f32(smh_cnn_trainer *t, const float *d_x, const float *d_y, int N, const float *d_drop,
                                      const float *d_drop_heads, const float *h_loss_weights, float *d_losses, void *stream) {
    SMH_REQUIRE(t && d_x && d_y && d_losses, "smh_cnn_train_step_f32: null argument");
    SMH_REQUIRE(N >= 2 && N <= t->max_batch, "smh_cnn_train_step_f32: batch %d outside [2, %d]", N, t->max_batch);
    smh_cnn *m = t->m;
    hipStream_t st = (hipStream_t)stream;
    const float *F = m->d_flat;
    const int nl = (int)m->layers.size();
    int rc;
    // ---------------- forward, training mode ----------------
    const float *src = d_x;
    size_t drop_base = 0;  // masks: per dropout layer one (N, dim) block, in graph order
    std::vector<const float *> masks(nl, nullptr);
    for (int l = 0; l < nl; ++l) {
        const Layer &L = m->layers[l];
        LayerState &S = t->ls[l];
        S.in = src;
        if (L.op == kConv) {
            ConvArgs a{};
            a.x = src, a.w = F + m->tensors[L.t_kernel].off;
            a.es = t->d_ones, a.eb = L.t_bias >= 0 ? F + m->tensors[L.t_bias].off : t->d_zeros;
            const bool bn_layer = L.t_bn >= 0;
            float *gemm_out = bn_layer ? S.z : S.a;  // without a BatchNorm the activation rides in the GEMM epilogue
            a.y = gemm_out, a.partial = t->d_partial, a.lut = m->d_lut + L.lut_off;
            a.H = L.H, a.W = L.W, a.Cin = L.C, a.OH = L.OH, a.OW = L.OW, a.Cout = L.OC, a.K = L.K;
            a.M = N * L.OH * L.OW;
            a.sh = L.sh, a.sw = L.sw, a.pt = L.pt, a.pl = L.pl, a.act = bn_layer ? kNone : L.act;
            a.ksteps = L.Kp / BK;
            a.vec4 = (L.C % 4 == 0) ? 1 : 0;
            const int bn = L.OC <= 64 ? 64 : 128;
            const int mt = (a.M + BM - 1) / BM, nt = (L.OC + bn - 1) / bn;
            a.ksplit = choose_split(mt, nt, a.ksteps);
            while (a.ksplit > 1 && (size_t)a.ksplit * a.M * L.OC > t->partial_floats) --a.ksplit;
            a.ksteps_per = (a.ksteps + a.ksplit - 1) / a.ksplit;
            launch_gemm<0>(a, a.M, bn, st);
            if (a.ksplit > 1) {
                const size_t MN = (size_t)a.M * L.OC;
                hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(nblk(MN)), dim3(256), 0, st, (const float *)t->d_partial, a.ksplit,
                                   MN, L.OC, a.es, a.eb, a.act, gemm_out);
            }
            if (!bn_layer) {
                src = S.a;
                continue;
            }
            // batch statistics: mean, then centred second moment
            RedArgs r{};
            r.z = S.z, r.M = (size_t)a.M, r.C = L.OC, r.mean = S.mean, r.rstd = S.rstd;
            rc = col_reduce<0>(t, r, 1, 1.0f / (float)a.M, 0, 1.f, S.mean, nullptr, st);
            if (rc) return rc;
            // 4-D inputs run Keras' fused BatchNorm, whose moving variance takes the unbiased batch variance
            const float bessel = (L.OH * L.OW > 1 || L.H * L.W > 1) ? (float)a.M / (float)(a.M - 1) : 1.0f;
            rc = col_reduce<1>(t, r, 1, 1.0f / (float)a.M, 1, bessel, S.rstd, t->d_bstat + S.bstat_off, st);
            if (rc) return rc;
            if (L.drop > 0.f && d_drop) {
                masks[l] = d_drop + drop_base;
                drop_base += (size_t)N * S.out_elems;
            }
            const float *g = F + m->tensors[L.t_bn].off;
            const size_t total = (size_t)a.M * L.OC;
            hipLaunchKernelGGL(bn_apply_kernel, dim3(nblk(total / 4)), dim3(256), 0, st, (const float *)S.z, total / 4, L.OC,
                               (const float *)S.mean, (const float *)S.rstd, g, g + L.OC, masks[l], S.a);
        } else if (L.op == kLrnRelu) {
            const size_t total = (size_t)N * L.H * L.W * L.C;
            hipLaunchKernelGGL(lrn_relu_kernel, dim3(nblk(total)), dim3(256), 0, st, src, L.C, 5, 1e-4f, 0.75f, total, S.a);
        } else if (L.op == kMelCl) {
            const size_t total = (size_t)N * L.OH * L.OW;
            hipLaunchKernelGGL(melcl_kernel, dim3(nblk(total)), dim3(256), 0, st, src, (const MelCl *)m->d_mel, F, L.H, L.W, L.OH,
                               L.kw, total, S.a);
        } else {  // kPool
            const size_t total = (size_t)N * L.OH * L.OW * L.C;
            hipLaunchKernelGGL(maxpool_kernel, dim3(nblk(total)), dim3(256), 0, st, src, L.H, L.W, L.C, L.OH, L.OW, L.kh, L.kw,
                               L.sh, L.sw, L.pt, L.pl, total, S.a);
        }
        src = S.a;
    }
    rc = smh::launch_status("smh_cnn training forward");
    if (rc) return rc;
    // ---------------- heads: pre-activations, losses, d loss / d pre ----------------
    const float *feat = src;
    HeadPtrs hp{};
    hp.c3k = m->tensors[m->t_c3].off, hp.c3b = m->tensors[m->t_c3 + 1].off;
    hp.D = m->feat_dim, hp.n_classes = m->cfg.n_classes, hp.n_heads = m->n_heads;
    smh_tcn::HeadsArgs ha{};
    ha.N = N, ha.D = m->feat_dim, ha.NH = m->n_heads * kHidden, ha.n_classes = m->cfg.n_classes, ha.n_heads = m->n_heads;
    ha.out_dim = m->out_dim;
    for (int i = 0; i <= kMaxHeads; ++i) ha.lw[i] = 1.0f;
    if (h_loss_weights)
        for (int i = 0; i <= m->n_heads; ++i) ha.lw[i] = h_loss_weights[i];
    for (int h = 0; h < m->n_heads; ++h) {
        const int th = m->t_head[h];
        hp.hk[h] = m->tensors[th].off, hp.hb[h] = m->tensors[th + 1].off;
        ha.head_odim[h] = m->odim[h], ha.head_sigmoid[h] = m->sigm[h];
        ha.goff_head[h] = m->tensors[th].off;
        ha.hp_off[h] = m->tensors[th + 2].off;  // gamma, beta, mean, var, out kernel, out bias: contiguous in the table
    }
    ha.goff_c3b = hp.c3b;
    hipLaunchKernelGGL(heads_pre_kernel, dim3(N), dim3(256), 0, st, feat, F, hp, t->d_pre);
    rc = smh::launch_status("heads_pre_kernel");
    if (rc) return rc;
    rc = smh_tcn::launch_heads_train(ha, t->d_pre, d_y, F, d_drop_heads, t->d_dpre, t->d_dxh, t->d_grad,
                                     t->d_bstat + t->head_bstat, d_losses,
                                     reinterpret_cast<unsigned *>(t->d_dxh + (size_t)t->max_batch * kPS), st);
    if (rc) return rc;
    hipLaunchKernelGGL(l2_partial_kernel, dim3(t->nseg), dim3(256), 0, st, (const Seg *)t->d_segs, F, t->d_l2part);
    hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(1024), 0, st, (const double *)t->d_l2part, t->nseg, d_losses + m->n_heads + 3);
    hipLaunchKernelGGL(heads_dw_kernel, dim3(nblk((size_t)m->feat_dim * kHidden), 1 + m->n_heads), dim3(256), 0, st, feat,
                       (const float *)t->d_dpre, hp, N, t->d_grad);
    int cur = 0;
    hipLaunchKernelGGL(heads_dfeat_kernel, dim3(nblk((size_t)N * m->feat_dim)), dim3(256), 0, st, (const float *)t->d_dpre, F, hp, N,
                       t->d_g[cur]);
    rc = smh::launch_status("smh_cnn heads backward");
    if (rc) return rc;
    // ---------------- backward through the layer list ----------------
    for (int l = nl - 1; l >= 0; --l) {
        const Layer &L = m->layers[l];
        LayerState &S = t->ls[l];
        float *G = t->d_g[cur], *Gn = t->d_g[cur ^ 1];
        if (L.op == kPool) {
            const size_t total = (size_t)N * L.H * L.W * L.C;
            if (L.sh == L.kh && L.sw == L.kw && !getenv("SMH_CNN_POOL_GATHER")) {
                if (L.OH * L.kh - L.pt < L.H || L.OW * L.kw - L.pl < L.W)  // rows / columns no window reaches
                    SMH_CHECK_HIP(hipMemsetAsync(Gn, 0, total * sizeof(float), st));
                const size_t nwin = (size_t)N * L.OH * L.OW * L.C;
                hipLaunchKernelGGL(maxpool_bwd_tiles_kernel, dim3(nblk(nwin)), dim3(256), 0, st, S.in, (const float *)G, L.H, L.W, L.C,
                                   L.OH, L.OW, L.kh, L.kw, L.pt, L.pl, nwin, Gn);
            } else {
                hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblk(total)), dim3(256), 0, st, S.in, (const float *)S.a, (const float *)G,
                                   L.H, L.W, L.C, L.OH, L.OW, L.kh, L.kw, L.sh, L.sw, L.pt, L.pl, total, Gn);
            }
            cur ^= 1;
            continue;
        }
        if (L.op == kMelCl) {  // first layer: weight gradients only; G holds dA with 4 floats per pixel (3 channels + pad)
            hipLaunchKernelGGL(melcl_bwd_kernel, dim3(L.OH), dim3(256), 0, st, S.in, (const float *)S.a, (const float *)G,
                               (const MelCl *)m->d_mel, N, L.H, L.W, L.OH, L.kw, 4, t->d_grad);
            continue;
        }
        if (L.op == kLrnRelu) {
            const size_t total = (size_t)N * L.H * L.W * L.C;
            hipLaunchKernelGGL(lrn_bwd1_kernel, dim3(nblk(total)), dim3(256), 0, st, S.in, (const float *)S.a, L.C, 5, 1e-4f, 0.75f,
                               total, G, t->d_tmp);
            hipLaunchKernelGGL(lrn_bwd2_kernel, dim3(nblk(total)), dim3(256), 0, st, S.in, (const float *)G, (const float *)t->d_tmp,
                               L.C, 5, 1e-4f, 0.75f, total, Gn);
            cur ^= 1;
            continue;
        }
        const int M = N * L.OH * L.OW;
        const size_t total = (size_t)M * L.OC;
        float *gr = t->d_grad;
        if (L.t_bn < 0) {  // bias + optional ReLU in the GEMM epilogue: gate, then the bias gradient is a plain column sum
            if (L.act == kRelu) hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblk(total)), dim3(256), 0, st, (const float *)S.a, total, G);
            if (L.t_bias >= 0) {
                RedArgs rb{};
                rb.z = G, rb.M = (size_t)M, rb.C = L.OC;
                rc = col_reduce<0>(t, rb, 1, 1.0f, 0, 1.f, gr + m->tensors[L.t_bias].off, nullptr, st);
                if (rc) return rc;
            }
        } else {
        const float *g = F + m->tensors[L.t_bn].off;
        // BN backward: dbeta = sum g, dgamma = sum g * xhat, then dz in place
        RedArgs r{};
        r.z = S.z, r.dA = G, r.mask = masks[l], r.mean = S.mean, r.rstd = S.rstd, r.gamma = g, r.beta = g + L.OC;
        r.M = (size_t)M, r.C = L.OC;
        rc = col_reduce<2>(t, r, 2, 1.0f, 0, 1.f, S.s12, S.s12 + L.OC, st);
        if (rc) return rc;
        SMH_CHECK_HIP(hipMemcpyAsync(gr + m->tensors[L.t_bn].off + L.OC, S.s12, L.OC * sizeof(float), hipMemcpyDeviceToDevice, st));  // beta
        SMH_CHECK_HIP(hipMemcpyAsync(gr + m->tensors[L.t_bn].off, S.s12 + L.OC, L.OC * sizeof(float), hipMemcpyDeviceToDevice, st));  // gamma
        hipLaunchKernelGGL(bn_bwd_kernel, dim3(nblk(total / 4)), dim3(256), 0, st, (const float *)S.z, masks[l], total / 4, L.OC,
                           1.0f / (float)M, (const float *)S.mean, (const float *)S.rstd, g, g + L.OC, (const float *)S.s12, G);
        // d bias = column sums of dz: behind a BatchNorm that is exactly zero (sum_m dz = gamma*rstd*(s1 - s1 - s2*sum xhat),
        // sum xhat = 0); Keras accumulates rounding noise there, this step leaves the zero the memset wrote
        if (L.t_bias >= 0) SMH_CHECK_HIP(hipMemsetAsync(gr + m->tensors[L.t_bias].off, 0, L.OC * sizeof(float), st));
        }
        if (L.K <= kSmallK && !getenv("SMH_CNN_WGRAD_MFMA")) {  // shallow first layer: VALU kernel
            const size_t outf = (size_t)L.K * L.OC;
            int nb = std::max(1, std::min(1024, M / 512));
            while (nb > 1 && (size_t)nb * outf > t->partial_floats) --nb;
            const int rpb = ((M + nb - 1) / nb + 63) / 64 * 64;
            nb = (M + rpb - 1) / rpb;
            hipLaunchKernelGGL(wgrad_smallk_kernel, dim3(nb, (L.OC + 63) / 64), dim3(256), 0, st, S.in, (const float *)G,
                               (const int2 *)(m->d_lut + L.lut_off), (const int2 *)S.rowinfo, L.H, L.W, L.K, L.OC, M, rpb, t->d_partial);
            hipLaunchKernelGGL(partial_sum_kernel, dim3(nblk(outf)), dim3(256), 0, st, (const float *)t->d_partial, nb, outf,
                               gr + m->tensors[L.t_kernel].off);
        } else {  // wgrad on the matrix cores
            ConvArgs a{};
            a.x = S.in, a.w = G, a.lut = m->d_lut + L.lut_off, a.rowinfo = S.rowinfo;
            a.H = L.H, a.W = L.W, a.Cin = L.C, a.OH = L.OH, a.OW = L.OW, a.Cout = L.OC, a.K = L.K, a.M = M;
            a.ksteps = (M + BK - 1) / BK;
            const int bn = L.OC <= 64 ? 64 : 128;
            const int mt = (L.K + BM - 1) / BM, nt = (L.OC + bn - 1) / bn;
            const size_t outf = (size_t)L.K * L.OC;
            a.ksplit = wgrad_split(mt * nt, a.ksteps, outf, t->partial_floats);
            if (const char *e = getenv("SMH_CNN_WSPLIT")) a.ksplit = std::max(1, std::min(atoi(e), a.ksteps));
            a.ksteps_per = (a.ksteps + a.ksplit - 1) / a.ksplit;
            a.ksplit = (a.ksteps + a.ksteps_per - 1) / a.ksteps_per;  // no empty slices
            a.y = gr + m->tensors[L.t_kernel].off, a.partial = t->d_partial;
            launch_gemm<1>(a, L.K, bn, st);
            if (a.ksplit > 1)
                hipLaunchKernelGGL(partial_sum_kernel, dim3(nblk(outf)), dim3(256), 0, st, (const float *)t->d_partial, a.ksplit, outf,
                                   a.y);
        }
        if (l > 0) {  // dgrad: dZ is the image, mirrored taps, kernel transposed per tap
            const int ldc = (L.C + 3) & ~3;  // columns of the data-gradient GEMM: Cin padded to a multiple of 4 (Jang conv1: 3 -> 4)
            const size_t wn = (size_t)L.kh * L.kw * L.OC * ldc;
            hipLaunchKernelGGL(transpose_taps_kernel, dim3(nblk(wn)), dim3(256), 0, st, F + m->tensors[L.t_kernel].off, L.C, L.OC, ldc,
                               wn, t->d_wt);
            const float *dzimg = G;
            if (L.sh > 1 || L.sw > 1) {  // strided layer: zero-stuff dz, then it is a stride-1 data gradient
                const size_t sn = (size_t)N * S.SH * S.SW * L.OC;
                SMH_CHECK_HIP(hipMemsetAsync(t->d_tmp, 0, sn * sizeof(float), st));
                hipLaunchKernelGGL(stuff_kernel, dim3(nblk(total)), dim3(256), 0, st, (const float *)G, L.OH, L.OW, L.OC, L.sh, L.sw,
                                   S.SH, S.SW, total, t->d_tmp);
                dzimg = t->d_tmp;
            }
            ConvArgs a{};
            a.x = dzimg, a.w = t->d_wt, a.es = t->d_ones, a.eb = t->d_zeros, a.y = Gn, a.partial = t->d_partial, a.lut = S.dlut;
            a.H = S.SH, a.W = S.SW, a.Cin = L.OC, a.OH = L.H, a.OW = L.W, a.Cout = ldc, a.K = S.dK;
            a.M = N * L.H * L.W;
            a.sh = a.sw = 1, a.pt = -L.pt, a.pl = -L.pl, a.act = kNone;
            a.ksteps = S.dKp / BK;
            a.vec4 = (L.OC % 4 == 0) ? 1 : 0;
            const int bn = ldc <= 64 ? 64 : 128;
            const int mt = (a.M + BM - 1) / BM, nt = (ldc + bn - 1) / bn;
            a.ksplit = choose_split(mt, nt, a.ksteps);
            if (const char *e = getenv("SMH_CNN_DSPLIT")) a.ksplit = std::max(1, std::min(atoi(e), a.ksteps));
            while (a.ksplit > 1 && (size_t)a.ksplit * a.M * ldc > t->partial_floats) --a.ksplit;
            a.ksteps_per = (a.ksteps + a.ksplit - 1) / a.ksplit;
            launch_gemm<0>(a, a.M, bn, st);
            if (a.ksplit > 1) {
                const size_t MN = (size_t)a.M * ldc;
                hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(nblk(MN)), dim3(256), 0, st, (const float *)t->d_partial, a.ksplit,
                                   MN, ldc, a.es, a.eb, (int)kNone, Gn);
            }
            cur ^= 1;
        }
        rc = smh::launch_status("smh_cnn backward");
        if (rc) return rc;
    }
    return SMH_OK;
}

extern "C" int smh_cnn_trainer_apply_f32(smh_cnn_trainer *t, int optimizer, float lr, float beta1, float beta2, float eps,
                                         float grad_scale, void *stream) {
    SMH_REQUIRE(t, "smh_cnn_trainer_apply_f32: null trainer");
    SMH_REQUIRE(optimizer == 0 || optimizer == 1, "smh_cnn_trainer_apply_f32: optimizer must be 0 (SGD) or 1 (Adam)");
    smh_cnn *m = t->m;
    hipStream_t st = (hipStream_t)stream;
    t->step += 1;
    m->wbf_valid = false;  // the bf16 operand cache of smh_cnn_forward_bf16 follows the weights
    OptArgs o{};
    o.optimizer = optimizer, o.lr = lr, o.b1 = beta1, o.b2 = beta2, o.eps = eps, o.grad_scale = grad_scale;
    if (optimizer == 1)
        o.alpha = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t->step)) /
                          (1.0 - std::pow((double)beta1, (double)t->step)));
    hipLaunchKernelGGL(opt_kernel, dim3(t->nseg), dim3(256), 0, st, (const Seg *)t->d_segs, o, m->d_flat, t->d_grad, t->d_s1, t->d_s2,
                       (const float *)t->d_bstat);
    if (t->nfold)
        hipLaunchKernelGGL(refold_kernel, dim3(t->nfold), dim3(256), 0, st, (const FoldEnt *)t->d_foldents, (const float *)m->d_flat,
                           m->d_fold);
    return smh::launch_status("smh_cnn_trainer_apply_f32");
}
