// Internals of the Conv2D MTL baselines shared by the inference (smh_cnn.hip) and training (smh_cnn_train.hip)
// translation units: GEMM kernel, layer graph, parameter table.  Everything lives in anonymous namespaces, so each
// unit gets its own copy of the kernels it launches.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "smh_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 128, BK = 16;
constexpr float kBnEps = 1e-3f;
constexpr int kMaxHeads = 4, kHidden = 16;
enum Act { kNone = 0, kRelu = 1, kTanh = 2 };

struct ConvArgs {
    const float *x, *w, *es, *eb;
    float *y, *partial;
    const int2 *lut;  // Kp entries: x = offset in floats from the pixel's (iy0, ix0, 0); y = dy | dx << 16
    const int2 *rowinfo;  // MODE 1 only, per output pixel m: x = offset of its (iy0, ix0, 0), y = iy0 | ix0 << 16
    int H, W, Cin, OH, OW, Cout, K, M;
    int sh, sw, pt, pl;
    int act, ksplit, ksteps, ksteps_per;
    int vec4;
    int Kp;  // bf16 kernel: K padded to a multiple of 32 = row length of the transposed bf16 weights
};

__device__ __forceinline__ float activate(float v, int act) {
    if (act == kRelu) return fmaxf(v, 0.f);
    if (act == kTanh) return tanhf(v);
    return v;
}

// MODE 0: y[M x Cout] = im2col(x)[M x K] * w[K x Cout]  (forward; also the stride-1 data gradient, with dZ as the
//         image, a mirrored table and the transposed kernel)
// MODE 1: dW[K x Cout] = im2col(x)^T[K x M] * dZ[M x Cout]  (weight gradient: tile rows = k, reduction over the
//         output pixels; a.w = dZ, a.ksteps = ceil(M / BK))
template <int BN, int MODE = 0>
__global__ void __launch_bounds__(256) conv_gemm_kernel(ConvArgs a) {
    constexpr int NT = BN / 64;        // 32-wide n tiles per wave
    constexpr int NB4 = BK * BN / 4 / 256;  // float4 B loads per thread and k-step
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;

    // the pixel this thread gathers for the A tile
    const int p = tid & (BM - 1);
    const int m = m0 + p;
    const int rows = MODE == 0 ? a.M : a.K;
    const bool valid_m = m < rows;
    int iy0 = 0, ix0 = 0;
    long base = 0;
    int2 ek = {0, 0};
    if (MODE == 1) {
        if (valid_m) ek = a.lut[m];
    } else if (valid_m) {
        const int ohw = a.OH * a.OW;
        const int img = m / ohw, r = m - img * ohw;
        const int oy = r / a.OW, ox = r - oy * a.OW;
        iy0 = oy * a.sh - a.pt;
        ix0 = ox * a.sw - a.pl;
        base = (((long)img * a.H + iy0) * a.W + ix0) * a.Cin;
    }
    const int kq0 = tid >> 7;  // wave-uniform

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[2], rb[NB4];
    auto gload = [&](int ks) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = __builtin_amdgcn_readfirstlane(ks * BK + 4 * (kq0 + 2 * it));
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k + q < a.M) {
                        const int2 ri = a.rowinfo[k + q];
                        const int iy = (short)(ri.y & 0xffff) + (short)(ek.y & 0xffff), ix = (ri.y >> 16) + (ek.y >> 16);
                        if (valid_m && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                            v[q] = a.x[(long)ri.x + ek.x];
                    }
            } else if (a.vec4) {
                const int2 e = a.lut[k];
                const int iy = iy0 + (short)(e.y & 0xffff), ix = ix0 + (e.y >> 16);
                if (valid_m && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    v = *reinterpret_cast<const f32x4 *>(a.x + base + e.x);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int2 e = a.lut[k + q];
                    const int iy = iy0 + (short)(e.y & 0xffff), ix = ix0 + (e.y >> 16);
                    if (valid_m && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v[q] = a.x[base + e.x];
                }
            }
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NB4; ++it) {
            const int idx = tid + it * 256;
            const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);
            const int k = ks * BK + row, n = n0 + c4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < (MODE == 0 ? a.K : a.M) && n < a.Cout) v = *reinterpret_cast<const f32x4 *>(a.w + (size_t)k * a.Cout + n);
            rb[it] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int kq = kq0 + 2 * it;
#pragma unroll
            for (int q = 0; q < 4; ++q) As[buf][4 * kq + q][p] = ra[it][q];
        }
#pragma unroll
        for (int it = 0; it < NB4; ++it) {
            const int idx = tid + it * 256;
            const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);
            *reinterpret_cast<f32x4 *>(&Bs[buf][row][c4 * 4]) = rb[it];
        }
    };

    const int ks_begin = z * a.ksteps_per;
    const int ks_end = min(a.ksteps, ks_begin + a.ksteps_per);
    if (ks_begin < ks_end) {
        gload(ks_begin);
        sstore(0);
        __syncthreads();
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const int cur = (ks - ks_begin) & 1;
            const bool more = ks + 1 < ks_end;
            if (more) gload(ks + 1);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int k = 2 * kk + (lane >> 5);
                float av[2], bv[NT];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = As[cur][k][wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = Bs[cur][k][wn * (BN / 2) + j * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (more) sstore(cur ^ 1);
            __syncthreads();
        }
    }

    // epilogue: register r of lane l holds D[8*(r>>2) + 4*(l>>5) + (r&3)][l & 31]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            if (n >= a.Cout) continue;
            const float es = (MODE == 0 && a.ksplit == 1) ? a.es[n] : 1.f, eb = (MODE == 0 && a.ksplit == 1) ? a.eb[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (mm >= rows) continue;
                const float v = acc[i][j][r];
                if (a.ksplit > 1) a.partial[((size_t)z * rows + mm) * a.Cout + n] = v;
                else if (MODE == 1) a.y[(size_t)mm * a.Cout + n] = v;
                else a.y[(size_t)mm * a.Cout + n] = activate(fmaf(v, es, eb), a.act);
            }
        }
}

__global__ void splitk_epilogue_kernel(const float *__restrict__ partial, int S, size_t MN, int Cout,
                                       const float *__restrict__ es, const float *__restrict__ eb, int act,
                                       float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MN) return;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += partial[(size_t)s * MN + i];  // fixed order
    const int n = (int)(i % Cout);
    y[i] = activate(fmaf(v, es[n], eb[n]), act);
}

// ---- mixed-precision variant: bf16 operands (v_mfma_f32_32x32x16_bf16), f32 accumulation and epilogue --------------------
// Same implicit GEMM; activations are rounded to bf16 while the im2col gather stages them in LDS, the kernels are read
// from a transposed bf16 copy [Cout][Kp] (k contiguous: one 16-byte load / LDS write / operand read per 8 k).
// LDS images are [m][k] and [n][k] with 40-element rows (80 bytes: 16-byte aligned, conflict-free per 16 lanes).
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
constexpr int BKB = 32, BKP = 40;

template <int BN>
__global__ void __launch_bounds__(256) conv_gemm_bf16_kernel(ConvArgs a, const __bf16 *__restrict__ wt) {
    constexpr int NT = BN / 64;
    constexpr int NBC = BN * 4 / 256;  // 16-byte B chunks per thread and k tile
    __shared__ __attribute__((aligned(16))) __bf16 As[2][BM][BKP];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BN][BKP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
    const int p = tid & (BM - 1);
    const int m = m0 + p;
    const bool valid_m = m < a.M;
    int iy0 = 0, ix0 = 0;
    long base = 0;
    if (valid_m) {
        const int ohw = a.OH * a.OW;
        const int img = m / ohw, r = m - img * ohw;
        const int oy = r / a.OW, ox = r - oy * a.OW;
        iy0 = oy * a.sh - a.pt;
        ix0 = ox * a.sw - a.pl;
        base = (((long)img * a.H + iy0) * a.W + ix0) * a.Cin;
    }
    const int kq0 = tid >> 7;  // wave-uniform
    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4];
    bf16x8 rb[NBC];
    auto gload = [&](int ks) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int k = __builtin_amdgcn_readfirstlane(ks * BKB + 4 * (kq0 + 2 * it));
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a.vec4) {
                const int2 e = a.lut[k];
                const int iy = iy0 + (short)(e.y & 0xffff), ix = ix0 + (e.y >> 16);
                if (valid_m && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    v = *reinterpret_cast<const f32x4 *>(a.x + base + e.x);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int2 e = a.lut[k + q];
                    const int iy = iy0 + (short)(e.y & 0xffff), ix = ix0 + (e.y >> 16);
                    if (valid_m && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v[q] = a.x[base + e.x];
                }
            }
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NBC; ++it) {
            const int idx = tid + it * 256;
            const int n = idx >> 2, k8 = idx & 3;
            bf16x8 v;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (__bf16)0.0f;
            if (n0 + n < a.Cout) v = *reinterpret_cast<const bf16x8 *>(wt + (size_t)(n0 + n) * a.Kp + ks * BKB + k8 * 8);
            rb[it] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const bf16x4 v = {(__bf16)ra[it][0], (__bf16)ra[it][1], (__bf16)ra[it][2], (__bf16)ra[it][3]};
            *reinterpret_cast<bf16x4 *>(&As[buf][p][4 * (kq0 + 2 * it)]) = v;
        }
#pragma unroll
        for (int it = 0; it < NBC; ++it) {
            const int idx = tid + it * 256;
            *reinterpret_cast<bf16x8 *>(&Bs[buf][idx >> 2][(idx & 3) * 8]) = rb[it];
        }
    };
    const int ks_begin = z * a.ksteps_per;
    const int ks_end = min(a.ksteps, ks_begin + a.ksteps_per);
    if (ks_begin < ks_end) {
        gload(ks_begin);
        sstore(0);
        __syncthreads();
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const int cur = (ks - ks_begin) & 1;
            const bool more = ks + 1 < ks_end;
            if (more) gload(ks + 1);
#pragma unroll
            for (int kk = 0; kk < BKB / 16; ++kk) {
                const int ko = kk * 16 + 8 * (lane >> 5);
                bf16x8 av[2], bv[NT];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8 *>(&As[cur][wm * 64 + i * 32 + (lane & 31)][ko]);
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const bf16x8 *>(&Bs[cur][wn * (BN / 2) + j * 32 + (lane & 31)][ko]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (more) sstore(cur ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            if (n >= a.Cout) continue;
            const float es = a.ksplit == 1 ? a.es[n] : 1.f, eb = a.ksplit == 1 ? a.eb[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (mm >= a.M) continue;
                const float v = acc[i][j][r];
                if (a.ksplit > 1) a.partial[((size_t)z * a.M + mm) * a.Cout + n] = v;
                else a.y[(size_t)mm * a.Cout + n] = activate(fmaf(v, es, eb), a.act);
            }
        }
}

// wt[n][k] = bf16(w[k][n]) for k < K, 0 up to Kp: one thread per element
__global__ void pack_wt_bf16_kernel(const float *__restrict__ w, int K, int Kp, int Cout, __bf16 *__restrict__ wt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cout * Kp) return;
    const int k = (int)(i % Kp);
    const size_t n = i / Kp;
    wt[i] = (__bf16)(k < K ? w[(size_t)k * Cout + n] : 0.f);
}

// MaxPooling2D, NHWC, 'valid' or 'same' (pt/pl = top/left padding; out-of-image taps are skipped = -inf padding)
__global__ void maxpool_kernel(const float *__restrict__ x, int H, int W, int C, int OH, int OW, int ph, int pw, int sh,
                               int sw, int pt, int pl, size_t total, float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const size_t img = r / OH;
    float v = -__builtin_inff();
    for (int dy = 0; dy < ph; ++dy) {
        const int iy = oy * sh - pt + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int dx = 0; dx < pw; ++dx) {
            const int ix = ox * sw - pl + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            v = fmaxf(v, x[((img * H + iy) * W + ix) * C + c]);
        }
    }
    y[i] = v;
}

// tf.nn.local_response_normalization(depth_radius, alpha, beta; bias = 1) followed by ReLU
__global__ void lrn_relu_kernel(const float *__restrict__ x, int C, int radius, float alpha, float beta, size_t total,
                                float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float *px = x + (i - c);
    float s = 0.f;
    for (int d = max(0, c - radius); d <= min(C - 1, c + radius); ++d) s = fmaf(px[d], px[d], s);
    const float v = px[c] / powf(1.f + alpha * s, beta);
    y[i] = fmaxf(v, 0.f);
}

// Jang's mel-scale layer: per mel filter r a (width_r x 5) kernel with 3 output channels over its band of bins,
// stride (width_r, 1), 'same' -> one row per filter; tanh.  x (N, 2K, W) -> y (N, 2*n_mels, W, 3).
struct MelCl {
    int top, width, woff;
};
__global__ void melcl_kernel(const float *__restrict__ x, const MelCl *__restrict__ f, const float *__restrict__ w,
                             int rows_in, int W, int rows_out, int tdim, size_t total, float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % W);
    const size_t q = i / W;
    const int r = (int)(q % rows_out);
    const size_t img = q / rows_out;
    const MelCl e = f[r];
    const float *xr = x + (img * rows_in + e.top) * W;
    const float *wr = w + e.woff;
    const int half = tdim / 2;  // 'same', stride 1: (tdim-1)/2 before -- tdim is odd
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int b = 0; b < e.width; ++b)
        for (int d = 0; d < tdim; ++d) {
            const int tt = t + d - half;
            if ((unsigned)tt >= (unsigned)W) continue;
            const float v = xr[(size_t)b * W + tt];
            const float *ww = wr + (b * tdim + d) * 3;
            a0 = fmaf(v, ww[0], a0);
            a1 = fmaf(v, ww[1], a1);
            a2 = fmaf(v, ww[2], a2);
        }
    float *o = y + i * 3;
    o[0] = tanhf(a0);
    o[1] = tanhf(a1);
    o[2] = tanhf(a2);
}

// '3C' softmax + MTL heads (Dense16 -> BN -> ReLU -> Dense -> sigmoid/linear) on a D-wide feature vector:
// one workgroup per sample.  out row = [head outputs ..., softmax].
struct HeadArgs {
    const float *c3k, *c3b;
    const float *hk[kMaxHeads], *hb[kMaxHeads], *hbn[kMaxHeads], *ok[kMaxHeads], *ob[kMaxHeads];
    int D, n_classes, n_heads, out_dim;
    int odim[kMaxHeads], sigm[kMaxHeads];
};
__global__ void __launch_bounds__(256) heads_kernel(const float *__restrict__ feat, HeadArgs a, float *__restrict__ out) {
    constexpr int MAXV = 5 + kMaxHeads * kHidden;
    __shared__ float red[4][MAXV];
    __shared__ float hid[MAXV];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NV = a.n_classes + a.n_heads * kHidden;
    float acc[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) acc[v] = 0.f;
    const float *f = feat + (size_t)n * a.D;
    for (int d = tid; d < a.D; d += 256) {
        const float x = f[d];
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (c < a.n_classes) acc[c] = fmaf(x, a.c3k[(size_t)d * a.n_classes + c], acc[c]);
#pragma unroll
        for (int h = 0; h < kMaxHeads; ++h)
            if (h < a.n_heads) {
                const float *kr = a.hk[h] + (size_t)d * kHidden;
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
    if (tid < MAXV) hid[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
    (void)NV;
    float *o = out + (size_t)n * a.out_dim;
    if (tid == 0) {  // softmax
        float mx = -__builtin_inff(), z[5], s = 0.f;
        for (int c = 0; c < a.n_classes; ++c) {
            z[c] = hid[c] + a.c3b[c];
            mx = fmaxf(mx, z[c]);
        }
        for (int c = 0; c < a.n_classes; ++c) {
            z[c] = expf(z[c] - mx);
            s += z[c];
        }
        for (int c = 0; c < a.n_classes; ++c) o[a.out_dim - a.n_classes + c] = z[c] / s;
    }
    if (tid >= 64 && tid < 64 + a.n_heads) {
        const int h = tid - 64;
        const float *bn = a.hbn[h];  // gamma, beta, mean, var
        float hv[kHidden];
        for (int j = 0; j < kHidden; ++j) {
            float v = hid[5 + h * kHidden + j] + a.hb[h][j];
            v = (v - bn[2 * kHidden + j]) / sqrtf(bn[3 * kHidden + j] + kBnEps) * bn[j] + bn[kHidden + j];
            hv[j] = fmaxf(v, 0.f);
        }
        int ooff = 0;
        for (int q = 0; q < h; ++q) ooff += a.odim[q];
        for (int c = 0; c < a.odim[h]; ++c) {
            float v = a.ob[h][c];
            for (int j = 0; j < kHidden; ++j) v = fmaf(hv[j], a.ok[h][j * a.odim[h] + c], v);
            o[ooff + c] = a.sigm[h] ? 1.f / (1.f + expf(-v)) : v;
        }
    }
}

// ---- host side: layer list + parameter table ------------------------------------------------------------------
struct Tensor {
    std::string name;
    int shape[4], ndim;
    size_t off, count;
};

enum Op { kConv, kPool, kLrnRelu, kMelCl };
struct Layer {
    Op op;
    int H, W, C, OH, OW, OC;            // input / output image geometry (Dense: H = W = 1)
    int kh, kw, sh, sw, pt, pl, act;
    int t_kernel = -1, t_bias = -1, t_bn = -1;  // parameter table indices (bn = gamma; beta, mean, var follow)
    size_t es_off = 0;                   // folded scale/shift in d_fold
    int K = 0, Kp = 0;
    size_t lut_off = 0;
    size_t wbf_off = 0;                  // this layer's transposed bf16 kernel [OC][Kp] in d_wbf (bf16 elements)
    float drop = 0.f;                    // Dropout rate behind this layer's activation (training only)
    int l2 = 0;                          // kernel_regularizer=l2() on this layer's kernel (Jang)
};

}  // namespace

struct smh_cnn {
    smh_cnn_cfg cfg;
    std::vector<Tensor> tensors;
    std::vector<Layer> layers;
    size_t n_params = 0, max_act = 0;
    int feat_dim = 0, out_dim = 0, n_heads = 0, odim[kMaxHeads], sigm[kMaxHeads];
    int t_c3 = -1, t_head[kMaxHeads];
    float *d_flat = nullptr;   // canonical weights (used in place by the GEMMs)
    float *d_fold = nullptr;   // per conv/dense layer: scale[OC], shift[OC]
    int2 *d_lut = nullptr;
    MelCl *d_mel = nullptr;
    size_t n_fold = 0;
    int mel_rows = 0, mel_in_rows = 0;
    int c3_l2 = 0;             // Jang: l2() on the '3C' kernel too (proposed_architectures.py:747)
    // bf16 operand cache of smh_cnn_forward_bf16: every Conv2D / Dense kernel transposed to [Cout][Kp], rebuilt lazily
    mutable void *d_wbf = nullptr;
    mutable bool wbf_valid = false;
    size_t n_wbf = 0;
    std::vector<float> h_flat;
};

namespace {

int add_tensor(smh_cnn *m, const std::string &name, std::initializer_list<int> shape) {
    Tensor t;
    t.name = name;
    t.ndim = (int)shape.size();
    t.count = 1;
    int i = 0;
    for (int s : shape) t.shape[i++] = s, t.count *= (size_t)s;
    for (; i < 4; ++i) t.shape[i] = 1;
    t.off = m->n_params;
    m->n_params += t.count;
    m->tensors.push_back(t);
    return (int)m->tensors.size() - 1;
}
int add_bn(smh_cnn *m, const std::string &p, int C) {
    const int g = add_tensor(m, p + "/gamma", {C});
    add_tensor(m, p + "/beta", {C});
    add_tensor(m, p + "/moving_mean", {C});
    add_tensor(m, p + "/moving_variance", {C});
    return g;
}
void same_pad(int n, int k, int s, int *out, int *before) {
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + k - n;
    if (total < 0) total = 0;
    *before = total / 2;
}

// geometry cursor while the graph is built
struct Cur {
    int H, W, C;
};

void push_conv(smh_cnn *m, Cur &c, const std::string &name, int kh, int kw, int oc, int sh, int sw, bool same, int act,
               const std::string &bn, bool bias = true, float drop = 0.f, int l2 = 0) {
    Layer L{};
    L.op = kConv;
    L.H = c.H, L.W = c.W, L.C = c.C, L.OC = oc, L.kh = kh, L.kw = kw, L.sh = sh, L.sw = sw, L.act = act;
    if (same) {
        same_pad(c.H, kh, sh, &L.OH, &L.pt);
        same_pad(c.W, kw, sw, &L.OW, &L.pl);
    } else {
        L.OH = (c.H - kh) / sh + 1, L.OW = (c.W - kw) / sw + 1, L.pt = L.pl = 0;
    }
    L.t_kernel = add_tensor(m, name + "/kernel", {kh, kw, c.C, oc});
    if (bias) L.t_bias = add_tensor(m, name + "/bias", {oc});
    if (!bn.empty()) L.t_bn = add_bn(m, bn, oc);
    L.K = kh * kw * c.C;
    L.Kp = (L.K + 31) / 32 * 32;  // a multiple of both k tiles (16 for the f32 kernel, 32 for the bf16 one)
    L.drop = drop, L.l2 = l2;
    m->layers.push_back(L);
    c.H = L.OH, c.W = L.OW, c.C = oc;
}
void push_dense(smh_cnn *m, Cur &c, const std::string &name, int oc, int act, const std::string &bn, float drop = 0.f,
                int l2 = 0) {
    Layer L{};
    L.op = kConv;
    L.H = L.W = 1, L.C = c.H * c.W * c.C, L.OH = L.OW = 1, L.OC = oc, L.kh = L.kw = L.sh = L.sw = 1, L.act = act;
    L.t_kernel = add_tensor(m, name + "/kernel", {L.C, oc});
    L.t_bias = add_tensor(m, name + "/bias", {oc});
    if (!bn.empty()) L.t_bn = add_bn(m, bn, oc);
    L.K = L.C;
    L.Kp = (L.K + 31) / 32 * 32;
    L.drop = drop, L.l2 = l2;
    m->layers.push_back(L);
    c.H = c.W = 1, c.C = oc;
}
void push_pool(smh_cnn *m, Cur &c, int ph, int pw, int sh, int sw, bool same) {
    Layer L{};
    L.op = kPool;
    L.H = c.H, L.W = c.W, L.C = c.C, L.OC = c.C, L.kh = ph, L.kw = pw, L.sh = sh, L.sw = sw;
    if (same) {
        same_pad(c.H, ph, sh, &L.OH, &L.pt);
        same_pad(c.W, pw, sw, &L.OW, &L.pl);
    } else {
        L.OH = (c.H - ph) / sh + 1, L.OW = (c.W - pw) / sw + 1;
    }
    m->layers.push_back(L);
    c.H = L.OH, c.W = L.OW;
}
void push_lrn(smh_cnn *m, Cur &c) {
    Layer L{};
    L.op = kLrnRelu;
    L.H = L.OH = c.H, L.W = L.OW = c.W, L.C = L.OC = c.C;
    m->layers.push_back(L);
}

void add_heads(smh_cnn *m, int D) {
    const int nc = m->cfg.n_classes;
    m->feat_dim = D;
    m->t_c3 = add_tensor(m, "3C/kernel", {D, nc});
    add_tensor(m, "3C/bias", {nc});
    static const char *names5[] = {"S", "M", "N", "R"};
    static const char *names3[] = {"S", "M", "R"};
    if (nc == 5) {  // 5_class_classification.py:150-215
        m->n_heads = 4;
        const int od[4] = {1, 1, 1, 3}, sg[4] = {1, 1, 1, 0};
        for (int i = 0; i < 4; ++i) m->odim[i] = od[i], m->sigm[i] = sg[i];
    } else {
        m->n_heads = 3;
        const int od[3] = {1, 1, 2}, sg[3] = {1, 1, 0};
        for (int i = 0; i < 3; ++i) m->odim[i] = od[i], m->sigm[i] = sg[i];
    }
    m->out_dim = nc;
    for (int i = 0; i < m->n_heads; ++i) {
        const std::string nm = nc == 5 ? names5[i] : names3[i];
        m->t_head[i] = add_tensor(m, nm + "/dense/kernel", {D, kHidden});
        add_tensor(m, nm + "/dense/bias", {kHidden});
        add_bn(m, nm + "/bn", kHidden);
        add_tensor(m, nm + "/out/kernel", {kHidden, m->odim[i]});
        add_tensor(m, nm + "/out/bias", {m->odim[i]});
        m->out_dim += m->odim[i];
    }
}

// librosa.filters.mel(fs, n_fft, n_mels, norm='slaney') support of every filter: first / last bin with weight > 0
// (proposed_architectures.py:681-691).  Same construction as the front end's mel table (smh_ctx.hip), in f64/f32.
void mel_filter_bins(double sr, int n_fft, int n_mels, std::vector<int> &lo, std::vector<int> &hi) {
    const int K = 1 + n_fft / 2;
    auto hz_to_mel = [](double f) {
        const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
        return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
    };
    auto mel_to_hz = [](double mm) {
        const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
        return mm >= min_log_mel ? min_log_hz * std::exp(logstep * (mm - min_log_mel)) : f_sp * mm;
    };
    std::vector<double> mel_f(n_mels + 2), fft(K);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(sr / 2);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    for (int k = 0; k < K; ++k) fft[k] = (sr / 2) * k / (K - 1);
    lo.assign(n_mels, -1), hi.assign(n_mels, -1);
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < K; ++k) {
            const double lower = -(mel_f[i] - fft[k]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fft[k]) / (mel_f[i + 2] - mel_f[i + 1]);
            const float wv = (float)std::fmax(0.0, std::fmin(lower, upper)) * (float)enorm;
            if (wv > 0.f) {
                if (lo[i] < 0) lo[i] = k;
                hi[i] = k;
            }
        }
    }
}

int build_graph(smh_cnn *m, std::vector<MelCl> &mel) {
    const smh_cnn_cfg &c = m->cfg;
    Cur cur{c.in_h, c.in_w, 1};
    if (c.kind == SMH_CNN_DOUKHAN) {  // proposed_architectures.py:448-492
        SMH_REQUIRE(c.in_h >= 24 && c.in_w >= 68, "Doukhan MTL needs an input of at least 24 x 68 (got %d x %d)", c.in_h, c.in_w);
        push_conv(m, cur, "conv1", 4, 5, 64, 1, 1, false, kRelu, "bn1");
        push_pool(m, cur, 2, 2, 2, 2, false);
        push_conv(m, cur, "conv2", 3, 3, 128, 1, 1, false, kRelu, "bn2");
        push_conv(m, cur, "conv3", 3, 3, 128, 1, 1, false, kRelu, "bn3");
        push_pool(m, cur, 2, 2, 2, 2, true);
        push_conv(m, cur, "conv4", 3, 3, 256, 1, 1, false, kRelu, "bn4");
        push_pool(m, cur, 1, 12, 1, 12, false);
        const float drop[4] = {0.2f, 0.3f, 0.4f, 0.5f};  // :477-492
        for (int i = 1; i <= 4; ++i)
            push_dense(m, cur, "fc" + std::to_string(i), 512, kRelu, "fc" + std::to_string(i) + "_bn", drop[i - 1]);
    } else if (c.kind == SMH_CNN_PAPAKOSTAS) {  // :539-571
        SMH_REQUIRE(c.in_h >= 29 && c.in_w >= 29, "Papakostas MTL needs an input of at least 29 x 29 (got %d x %d)", c.in_h, c.in_w);
        const int fc = c.fc_width > 0 ? c.fc_width : 4096;
        SMH_REQUIRE(fc % 4 == 0, "fc_width must be a multiple of 4");
        push_conv(m, cur, "conv1", 5, 5, 96, 2, 2, false, kNone, "");
        push_lrn(m, cur);
        push_pool(m, cur, 3, 3, 2, 2, true);
        push_conv(m, cur, "conv2", 3, 3, 384, 2, 2, false, kNone, "");
        push_lrn(m, cur);
        push_pool(m, cur, 3, 3, 2, 2, true);
        push_conv(m, cur, "conv3", 3, 3, 512, 1, 1, true, kRelu, "");
        push_pool(m, cur, 3, 3, 2, 2, true);
        push_dense(m, cur, "fc1", fc, kRelu, "fc1_bn", 0.5f);
        push_dense(m, cur, "fc2", fc, kRelu, "fc2_bn", 0.5f);
    } else if (c.kind == SMH_CNN_JANG) {  // :695-747
        const int n_fft = c.n_fft > 0 ? c.n_fft : 512, n_mels = c.n_mels > 0 ? c.n_mels : 120;
        const int Kh = n_fft / 2 + 1, tdim = 5;
        SMH_REQUIRE(c.in_h == 2 * Kh, "Jang MTL: input height %d is not 2 * (n_fft/2 + 1) = %d", c.in_h, 2 * Kh);
        SMH_REQUIRE(c.in_w >= 1, "Jang MTL: bad input width");
        std::vector<int> lo, hi;
        mel_filter_bins(c.fs > 0 ? c.fs : 16000.0, n_fft, n_mels, lo, hi);
        for (int half = 0; half < 2; ++half)
            for (int i = 0; i < n_mels; ++i) {
                SMH_REQUIRE(lo[i] >= 0, "Jang MTL: mel filter %d is empty (the reference fails here too)", i);
                const int width = hi[i] - lo[i] + 1;
                const int t = add_tensor(m, std::string(half ? "perc" : "harm") + "_melCl" + std::to_string(i) + "/kernel",
                                         {width, tdim, 1, 3});
                mel.push_back(MelCl{half * Kh + lo[i], width, (int)m->tensors[t].off});
            }
        Layer L{};
        L.op = kMelCl;
        L.H = c.in_h, L.W = c.in_w, L.C = 1, L.OH = 2 * n_mels, L.OW = c.in_w, L.OC = 3, L.kw = tdim;
        m->layers.push_back(L);
        m->mel_rows = 2 * n_mels, m->mel_in_rows = c.in_h;
        cur = Cur{2 * n_mels, c.in_w, 3};
        const int oc[3] = {32, 64, 128};
        for (int i = 0; i < 3; ++i) {
            push_conv(m, cur, "conv" + std::to_string(i + 1), 3, 3, oc[i], 1, 1, true, kRelu, "bn" + std::to_string(i + 1), true,
                      0.4f, 1);
            push_pool(m, cur, 2, 2, 2, 2, true);
        }
        push_dense(m, cur, "fc1", 2048, kRelu, "fc1_bn", 0.4f, 1);
        push_dense(m, cur, "fc2", 1024, kRelu, "fc2_bn", 0.4f, 1);
        m->c3_l2 = 1;
    } else {
        return smh::set_error(SMH_E_INVALID, "smh_cnn_create: unknown kind %d", c.kind);
    }
    for (const Layer &L : m->layers)
        if (L.OH < 1 || L.OW < 1) return smh::set_error(SMH_E_INVALID, "smh_cnn_create: input %d x %d is too small for this network", c.in_h, c.in_w);
    add_heads(m, cur.C);
    return SMH_OK;
}

constexpr int kChunk = 64;  // images per pass through the layer list (bounds the activation workspace)

struct Work {
    float *act[2];
    float *partial;
    size_t act_floats, partial_floats, bytes;
};
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int choose_split(int mtiles, int ntiles, int ksteps) {
    int s = 1;
    const int blocks = mtiles * ntiles;
    if (blocks >= 1 && blocks < 256 && ksteps >= 32) {
        s = (512 + blocks - 1) / blocks;
        if (s > 16) s = 16;
        if (s > ksteps / 8) s = ksteps / 8;
        if (s < 1) s = 1;
    }
    return s;
}

Work carve(const smh_cnn *m, void *base, int N) {
    Work w{};
    const int n = N < 1 ? 1 : (N < kChunk ? N : kChunk);
    w.act_floats = m->max_act * (size_t)n;
    size_t part = 0;
    for (const Layer &L : m->layers)
        if (L.op == kConv) {
            const int M = n * L.OH * L.OW;
            const int bn = L.OC <= 64 ? 64 : 128;
            const int s = choose_split((M + BM - 1) / BM, (L.OC + bn - 1) / bn, L.Kp / BK);
            if (s > 1) part = std::max(part, (size_t)s * M * L.OC);
        }
    if (N > kChunk && N % kChunk)  // the ragged last chunk may split K differently
        for (const Layer &L : m->layers)
            if (L.op == kConv) {
                const int M = (N % kChunk) * L.OH * L.OW;
                const int bn = L.OC <= 64 ? 64 : 128;
                const int s = choose_split((M + BM - 1) / BM, (L.OC + bn - 1) / bn, L.Kp / BK);
                if (s > 1) part = std::max(part, (size_t)s * M * L.OC);
            }
    w.partial_floats = part;
    char *p = (char *)base;
    size_t off = 0;
    for (int i = 0; i < 2; ++i) {
        w.act[i] = p ? (float *)(p + off) : nullptr;
        off += align256(w.act_floats * sizeof(float));
    }
    w.partial = p ? (float *)(p + off) : nullptr;
    off += align256(part * sizeof(float));
    w.bytes = off;
    return w;
}

}  // namespace
