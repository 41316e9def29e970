// a1: S = |STFT|  --  np.abs(librosa.core.stft(y, n_fft, win_length, hop_length, center=False))
// (call sites /root/reference/lib/preprocessing.py:407,417,429,439).
//
// Frame t = y[t*hop : t*hop + n_fft] * periodic-Hann.  The real frame of length n_fft is packed into a
// complex sequence of length M = n_fft/2 (z[m] = x[2m] + i x[2m+1]), transformed by a mixed-radix
// Stockham autosort FFT that lives entirely in LDS (ping-pong buffers, one butterfly per thread),
// un-tangled into the n_fft/2+1 real-FFT bins and written as magnitudes, frame-fastest, so that the
// (K, T) freq-major layout librosa returns is produced directly.  n_fft = 400 -> M = 200 = 8*5*5.
// Butterflies: radix 8 (= 2x4 in registers), radix 4, radix 2, Winograd radix 5, generic 3 / 7.
// LDS indices are padded (one float2 per 16) so that the strided Stockham writes spread over banks;
// work-item -> (frame, butterfly) maps use a reciprocal multiply instead of an integer division.
// HBM traffic per clip: 64,000 B audio in (re-reads of the 2.5x frame overlap are served by L2),
// 78,792 B magnitudes out.
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "smh_common.h"
#include "smh_rag.h"

namespace {

constexpr int kThreads = 256;

struct StftArgs {
    int n_samples, n_fft, hop, M, K, T, tt;  // tt = frames per workgroup
    int n_stages;
    int radix[smh::kMaxFftStages];
    float inv_nb[smh::kMaxFftStages];  // 1 / (M / radix)
    float inv_ns[smh::kMaxFftStages];  // 1 / Ns of the stage
    int tmul[smh::kMaxFftStages];      // M / (Ns * radix)
    float inv_tt;
    int B, xcd_tiles;  // xcd_tiles > 0: 1-D grid, neighbouring tiles on one XCD (see stft400_kernel); the value is the tiles per clip
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // * (-i)
__device__ __forceinline__ int pad(int i) { return i + (i >> 4); }
// exact floor(it / n) for 0 <= it < 2^20 given inv = 1/n
__device__ __forceinline__ int fdiv(int it, float inv) { return (int)(((float)it + 0.5f) * inv); }

template <int R>
__device__ __forceinline__ void dft_generic(float2 *v) {
    float2 o[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        float2 acc = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const int e = (r * q) % R;
            const double ang = -6.283185307179586476925 * (double)e / (double)R;
            const float2 w = make_float2((float)__builtin_cos(ang), (float)__builtin_sin(ang));  // constant-folded
            acc = cadd(acc, cmul(v[r], w));
        }
        o[q] = acc;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = o[q];
}

__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d) {
    const float2 s0 = cadd(a, c), d0 = csub(a, c);
    const float2 s1 = cadd(b, d), d1 = mul_mi(csub(b, d));
    a = cadd(s0, s1);
    b = cadd(d0, d1);
    c = csub(s0, s1);
    d = csub(d0, d1);
}

template <int R>
__device__ __forceinline__ void dft(float2 *v) {
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        dft4(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 8) {
        // 8 = 2 x 4: even/odd 4-point DFTs, odd outputs twisted by w8^k
        float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        dft4(e0, e1, e2, e3);
        dft4(o0, o1, o2, o3);
        const float h = 0.70710678118654752440f;
        o1 = make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));    // * (1 - i)/sqrt2
        o2 = mul_mi(o2);                                           // * (-i)
        o3 = make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));   // * (-1 - i)/sqrt2
        v[0] = cadd(e0, o0), v[4] = csub(e0, o0);
        v[1] = cadd(e1, o1), v[5] = csub(e1, o1);
        v[2] = cadd(e2, o2), v[6] = csub(e2, o2);
        v[3] = cadd(e3, o3), v[7] = csub(e3, o3);
    } else if constexpr (R == 5) {
        // Winograd-style 5-point DFT: 34 real operations
        const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
        const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
        const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
        const float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
        const float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
        const float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
        const float2 b1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
        const float2 b2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
        v[0] = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
        v[1] = make_float2(a1.x + b1.y, a1.y - b1.x);  // a1 - i b1
        v[4] = make_float2(a1.x - b1.y, a1.y + b1.x);  // a1 + i b1
        v[2] = make_float2(a2.x + b2.y, a2.y - b2.x);
        v[3] = make_float2(a2.x - b2.y, a2.y + b2.x);
    } else {
        dft_generic<R>(v);
    }
}

// One Stockham stage for butterfly j of one frame (Govindaraju et al. formulation).
template <int R, bool FIRST>
__device__ __forceinline__ void stage(int step, int j, int Ns, float inv_ns, int tmul, const float2 *__restrict__ src,
                                      float2 *__restrict__ dst, const float2 *__restrict__ tw,
                                      const float *__restrict__ audio, const float *__restrict__ win) {
    float2 v[R];
    if constexpr (FIRST) {
        // read the windowed real frame straight from global memory: z[m] = (x[2m], x[2m+1])
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = j + r * step;
            v[r] = make_float2(audio[2 * m] * win[2 * m], audio[2 * m + 1] * win[2 * m + 1]);
        }
    } else {
        const int k = j - fdiv(j, inv_ns) * Ns;
        const int tstep = k * tmul;  // exp(-2 pi i k r / (Ns R)) = twM[k r M/(Ns R)]
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float2 x = src[pad(j + r * step)];
            if (r > 0) x = cmul(x, tw[r * tstep]);
            v[r] = x;
        }
    }
    dft<R>(v);
    const int k = FIRST ? 0 : j - fdiv(j, inv_ns) * Ns;
    const int j0 = (j - k) * R + k;  // (j / Ns) * Ns * R + k
#pragma unroll
    for (int r = 0; r < R; ++r) dst[pad(j0 + r * Ns)] = v[r];
}

template <bool FIRST>
__device__ __forceinline__ void run_stage(int nb, int R, int j, int Ns, float inv_ns, int tmul, const float2 *src,
                                          float2 *dst, const float2 *tw, const float *audio, const float *win) {
    switch (R) {
        case 2: stage<2, FIRST>(nb, j, Ns, inv_ns, tmul, src, dst, tw, audio, win); break;
        case 3: stage<3, FIRST>(nb, j, Ns, inv_ns, tmul, src, dst, tw, audio, win); break;
        case 4: stage<4, FIRST>(nb, j, Ns, inv_ns, tmul, src, dst, tw, audio, win); break;
        case 5: stage<5, FIRST>(nb, j, Ns, inv_ns, tmul, src, dst, tw, audio, win); break;
        case 7: stage<7, FIRST>(nb, j, Ns, inv_ns, tmul, src, dst, tw, audio, win); break;
        case 8: stage<8, FIRST>(nb, j, Ns, inv_ns, tmul, src, dst, tw, audio, win); break;
        default: break;
    }
}

// |re + i im| on the hardware square root (v_sqrt_f32, 1 ulp): the correctly rounded expansion of sqrtf costs ~12 more
// instructions per bin and the STFT kernels are VALU-bound; the parity bound on |S| is 1e-5 of its maximum
__device__ __forceinline__ float mag(float re, float im) { return __builtin_amdgcn_sqrtf(re * re + im * im); }

__global__ void __launch_bounds__(kThreads)
stft_mag_kernel(StftArgs a, const float *__restrict__ audio, const float *__restrict__ window,
                const float2 *__restrict__ twM, const float2 *__restrict__ tw2M, float *__restrict__ S,
                const smh_rag::Clip *__restrict__ rag, const smh_rag::Item *__restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int M = a.M;
    const int MP = pad(M) + 2;  // padded frame stride (float2)
    float2 *tw = lds;           // M twiddles
    float2 *tw2 = lds + M;      // M + 1 untangle twiddles
    float2 *buf0 = tw2 + (M + 1);
    float2 *buf1 = buf0 + a.tt * MP;
    int b = blockIdx.y, tile = blockIdx.x;
    size_t clip_off, spec_off;
    if (rag) {  // ragged call (smh_rag.h): a.B items of clips of different lengths, in 8 contiguous ranges like the grid below
        const unsigned total = (unsigned)a.B, per_xcd = (total + 7u) >> 3;
        const unsigned j = blockIdx.x >> 3, n = (blockIdx.x & 7u) * per_xcd + j;
        if (j >= per_xcd || n >= total) return;
        const smh_rag::Item item = items[n];
        b = item.clip, tile = item.tile;
        a.T = rag[b].T;
        clip_off = (size_t)rag[b].audio_off, spec_off = (size_t)rag[b].spec_off;
    } else {
        if (a.xcd_tiles > 0) {  // the (clip, tile) items in 8 contiguous ranges, one per XCD: stft400_kernel has the reasoning
            const unsigned total = (unsigned)a.B * (unsigned)a.xcd_tiles, per_xcd = (total + 7u) >> 3;
            const unsigned j = blockIdx.x >> 3, n = (blockIdx.x & 7u) * per_xcd + j;
            if (j >= per_xcd || n >= total) return;
            b = (int)(n / (unsigned)a.xcd_tiles), tile = (int)(n - (unsigned)b * (unsigned)a.xcd_tiles);
        }
        clip_off = (size_t)b * a.n_samples, spec_off = (size_t)b * a.K * a.T;
    }
    const int t0 = tile * a.tt;
    const int nf = min(a.tt, a.T - t0);
    for (int i = threadIdx.x; i < M; i += blockDim.x) tw[i] = twM[i];
    for (int i = threadIdx.x; i <= M; i += blockDim.x) tw2[i] = tw2M[i];
    const float *clip = audio + clip_off;

    float2 *src = buf0, *dst = buf1;
    int Ns = 1;
    for (int s = 0; s < a.n_stages; ++s) {
        const int R = a.radix[s];
        const int nb = M / R;
        __syncthreads();
        for (int it = threadIdx.x; it < nf * nb; it += blockDim.x) {
            const int f = fdiv(it, a.inv_nb[s]), j = it - f * nb;
            if (s == 0)
                run_stage<true>(nb, R, j, Ns, 1.f, 0, nullptr, dst + f * MP, tw, clip + (size_t)(t0 + f) * a.hop, window);
            else
                run_stage<false>(nb, R, j, Ns, a.inv_ns[s], a.tmul[s], src + f * MP, dst + f * MP, tw, nullptr, nullptr);
        }
        float2 *tmp = src;
        src = dst;
        dst = tmp;
        Ns *= R;
    }
    __syncthreads();
    // real-FFT untangle + magnitude; frames fastest -> contiguous stores along t
    float *Sb = S + spec_off + t0;
    const float inv_nf = nf == a.tt ? a.inv_tt : 1.0f / (float)nf;
    for (int it = threadIdx.x; it < nf * a.K; it += blockDim.x) {
        const int k = fdiv(it, inv_nf), f = it - k * nf;
        const float2 *Z = src + f * MP;
        const float2 zk = Z[pad(k == M ? 0 : k)];
        float2 zc = Z[pad(k == 0 ? 0 : M - k)];
        zc.y = -zc.y;
        const float2 e = cadd(zk, zc), d = csub(zk, zc);
        const float2 wd = cmul(tw2[k], d);  // X = 0.5*e - 0.5*i*w*d
        const float re = 0.5f * (e.x + wd.y);
        const float im = 0.5f * (e.y - wd.x);
        Sb[(size_t)k * a.T + f] = mag(re, im);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// n_fft = 400 (the reference's only STFT size besides Jang's 512): M = 200 = 8 x 25 as a two-phase Cooley-Tukey
// with ONE exchange through LDS instead of two Stockham passes, and the real-FFT untangle done per pair (k, M-k):
//   phase 1  item (frame, n2 < 25): radix-8 DFT over n1 of z[25 n1 + n2] read straight from the windowed audio,
//            twiddle W_200^(n2 k1), store Y[k1][n2];
//   phase 2  item (frame, k1 < 8): 25-point DFT (5 x 5 Winograd butterflies in registers, constant twiddles) over
//            n2 of Y[k1][.], store Z[k1 + 8 k2];
//   phase 3  item (k <= 100, frame): |X[k]| and |X[200 - k]| from the pair Z[k], Z[200 - k]; frames fastest, so the
//            (K, T) stores are contiguous along t.
// ~225 wave-instructions per frame instead of ~510 in the generic kernel.  Frame stride 201 float2 (odd) keeps the
// frame-strided phase-3 reads and the 25-strided phase-2 reads conflict-free.
constexpr int kF400 = 32;    // upper bound of frames per workgroup: phase 2 has 8 items per frame
constexpr int kMP400 = 201;  // float2 stride of one frame in LDS

// Complex arithmetic on (re, im) register PAIRS for stft400_kernel.  hipcc packs float2 math into v_pk_* instructions but
// builds every swapped / negated operand ((-i) b, conj b, the cross terms of a complex product) with v_mov / v_pk_mov into a new
// pair first: 333 of the kernel's 1000 VALU instructions were such moves (round 2), and a VALU instruction costs the SIMD four
// cycles whatever it does.  The VOP3P encoding selects and negates the halves of each source itself (op_sel, op_sel_hi,
// neg_lo, neg_hi), so these patterns are single instructions -- spelled out here because the compiler does not form them:
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 sub_i(v2 a, v2 b) {  // a - i b = (a.x + b.y, a.y - b.x)
    v2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2 add_i(v2 a, v2 b) {  // a + i b = (a.x - b.y, a.y + b.x)
    v2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2 add_conj(v2 a, v2 b) {  // a + conj(b) = (a.x + b.x, a.y - b.y)
    v2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2 sub_conj(v2 a, v2 b) {  // a - conj(b) = (a.x - b.x, a.y + b.y)
    v2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2 cmulp(v2 a, v2 w) {  // a w: (a.x w.x, a.x w.y) then (- a.y w.y, + a.y w.x) on top
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// the same with a compile-time twiddle kept in a scalar register pair (one constant-bus operand per instruction)
__device__ __forceinline__ v2 cmulp_s(v2 a, v2 w) {
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}

__device__ __forceinline__ void pk_dft4(v2 &a, v2 &b, v2 &c, v2 &d) {
    const v2 s0 = a + c, d0 = a - c, s1 = b + d, bd = b - d;
    a = s0 + s1;
    b = sub_i(d0, bd);  // d0 + (-i)(b - d)
    c = s0 - s1;
    d = add_i(d0, bd);
}
__device__ __forceinline__ void pk_dft8(v2 *v) {  // 8 = 2 x 4: even / odd 4-point DFTs, odd outputs twisted by w8^k
    v2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    pk_dft4(e0, e1, e2, e3);
    pk_dft4(o0, o1, o2, o3);
    const float h = 0.70710678118654752440f;
    const v2 u1 = sub_i(o1, o1);  // o1 (1 - i)
    const v2 u3 = add_i(o3, o3);  // o3 (1 + i): w8^3 o3 = -h u3
    v[0] = e0 + o0, v[4] = e0 - o0;
    v[1] = e1 + h * u1, v[5] = e1 - h * u1;
    v[2] = sub_i(e2, o2), v[6] = add_i(e2, o2);  // w8^2 = -i
    v[3] = e3 - h * u3, v[7] = e3 + h * u3;
}
__device__ __forceinline__ void pk_dft5(v2 *v) {  // Winograd-style 5-point DFT
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const v2 t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
    const v2 a1 = v[0] + c1 * t1 + c2 * t2, a2 = v[0] + c2 * t1 + c1 * t2;
    const v2 b1 = s1 * t3 + s2 * t4, b2 = s2 * t3 - s1 * t4;
    v[0] = v[0] + t1 + t2;
    v[1] = sub_i(a1, b1), v[4] = add_i(a1, b1);
    v[2] = sub_i(a2, b2), v[3] = add_i(a2, b2);
}
__device__ __forceinline__ void pk_dft25(v2 *x) {
    // x[5a + b] -> X[c + 5d]:  inner DFT over a (output c), twiddle W_25^(b c), outer DFT over b (output d)
    v2 t[5][5];  // t[c][b]
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        v2 u[5] = {x[b], x[5 + b], x[10 + b], x[15 + b], x[20 + b]};
        pk_dft5(u);
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (b * c == 0) {
                t[c][b] = u[c];
            } else {
                const double ang = -6.283185307179586476925 * (double)(b * c) / 25.0;
                t[c][b] = cmulp_s(u[c], v2{(float)__builtin_cos(ang), (float)__builtin_sin(ang)});
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        pk_dft5(t[c]);
#pragma unroll
        for (int d = 0; d < 5; ++d) x[c + 5 * d] = t[c][d];
    }
}

__device__ __forceinline__ void dft25(float2 *x) {
    // x[5a + b] -> X[c + 5d]:  inner DFT over a (output c), twiddle W_25^(b c), outer DFT over b (output d)
    float2 t[5][5];  // t[c][b]
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        float2 u[5] = {x[b], x[5 + b], x[10 + b], x[15 + b], x[20 + b]};
        dft<5>(u);
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (b * c == 0) {
                t[c][b] = u[c];
            } else {
                const double ang = -6.283185307179586476925 * (double)(b * c) / 25.0;
                t[c][b] = cmul(u[c], make_float2((float)__builtin_cos(ang), (float)__builtin_sin(ang)));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        dft<5>(t[c]);
#pragma unroll
        for (int d = 0; d < 5; ++d) x[c + 5 * d] = t[c][d];
    }
}

// (at least four waves per SIMD: 128 VGPRs -- at 129 a fourth 256-thread workgroup per CU did not fit)
template <int ROW>  // float2 pitch of a phase-1 table row: 25, or 40 = the 25 entries and the first 15 again (below)
__global__ void __launch_bounds__(512, 4)
stft400_kernel(const float *__restrict__ audio, const float *__restrict__ window, const float2 *__restrict__ twM,
               const float2 *__restrict__ tw2M, float *__restrict__ S, int n_samples, int hop, int T, int F, int probe, int B,
               int ntiles, const smh_rag::Clip *__restrict__ rag, const smh_rag::Item *__restrict__ items) {
    // probe (SMH_STFT_PROBE_NOSTORE, tools/gpu/r2_fusion_bound.sh): magnitudes computed but not stored -- the cost of S's trip to HBM
    constexpr int M = 200, K = 201;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    v2 *tw = reinterpret_cast<v2 *>(lds);  // 8 rows of twiddles
    v2 *tw2 = tw + 8 * ROW;   // M + 1 untangle twiddles
    v2 *win2 = tw2 + (M + 2); // 8 rows: window as (w[2m], w[2m+1])
    v2 *Z = win2 + 8 * ROW;   // F frames of kMP400
    // (ROW == 40) table column of item `it` (= 25 f + n2): n2, + 25 behind a frame change inside the item's 16-lane group
    auto column = [&](int it, int f, int n2) { return ROW == 25 ? n2 : n2 + 25 * (f - (it & ~15) / 25); };
    // Workgroup -> (clip, frame tile).  ntiles > 0: a 1-D grid decoded so that NEIGHBOURING TILES RUN ON ONE XCD, dispatched next to
    // each other.  A row of S is T frames (98: 392 bytes) and a tile writes an 80-byte piece of it, so the lines of a row are completed
    // by up to three tiles -- in one L2 they merge before they leave for HBM; with consecutive tiles on consecutive XCDs (the plain
    // grid: workgroup i goes to XCD i % 8) every piece left as a partial line (WRITE_SIZE 105 MB for 80.7 MB of S), and the 240 samples
    // neighbouring tiles share were fetched twice.  The (clip, tile) items in clip-major order are cut into 8 contiguous ranges, one
    // per XCD: the same rule serves a batch of a thousand clips (an XCD owns 128 whole clips) and a single long file (an XCD owns an
    // eighth of its tiles).  ntiles == 0: the plain (tile, clip) grid.
    // rag != nullptr (smh_rag.h): B work items of clips of DIFFERENT lengths, item n = (clip, tile) from the list, clip shapes and
    // offsets from the descriptor table; the items are in clip-major order and cut into the same 8 ranges.
    int b = blockIdx.y, tile = blockIdx.x;
    size_t clip_off, spec_off;
    if (rag) {
        const unsigned total = (unsigned)B, per_xcd = (total + 7u) >> 3;
        const unsigned j = blockIdx.x >> 3, n = (blockIdx.x & 7u) * per_xcd + j;
        if (j >= per_xcd || n >= total) return;
        const smh_rag::Item item = items[n];
        b = item.clip, tile = item.tile;
        T = rag[b].T;
        clip_off = (size_t)rag[b].audio_off, spec_off = (size_t)rag[b].spec_off;
    } else {
        if (ntiles > 0) {
            const unsigned total = (unsigned)B * (unsigned)ntiles, per_xcd = (total + 7u) >> 3;  // (the launch keeps B * ntiles below 2^31)
            const unsigned j = blockIdx.x >> 3, n = (blockIdx.x & 7u) * per_xcd + j;
            if (j >= per_xcd || n >= total) return;
            b = (int)(n / (unsigned)ntiles), tile = (int)(n - (unsigned)b * (unsigned)ntiles);
        }
        clip_off = (size_t)b * n_samples, spec_off = (size_t)b * K * T;
    }
    const int t0 = tile * F, tid = threadIdx.x;
    const int nf = min(F, T - t0);
    const int nthr = blockDim.x;
    const float *clip = audio + clip_off + (size_t)t0 * hop;
    const int dq1 = nthr / 25, dr1 = nthr - dq1 * 25;
    // Phase 1's audio is requested FIRST, for all of this thread's items at once (up to kRounds1 rounds of (frame, n2) items,
    // 8 float2 each), then the twiddle / window tables travel to LDS: one trip to memory per workgroup where there were
    // one for the tables and one per round (a round's loads used to be issued when the previous round's butterflies were done).
    constexpr int kRounds1 = 3;
    const bool ahead = 25 * nf <= kRounds1 * nthr;
    v2 au[kRounds1][8];
    int fr_[kRounds1], n2_[kRounds1], col_[kRounds1];
    if (ahead) {
        int f = tid / 25, n2 = tid - (tid / 25) * 25;
#pragma unroll
        for (int r = 0; r < kRounds1; ++r) {
            fr_[r] = f, n2_[r] = n2, col_[r] = column(tid + r * nthr, f, n2);
            const v2 *fr = reinterpret_cast<const v2 *>(clip + (unsigned)(min(f, nf - 1) * hop));
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) au[r][n1] = fr[25 * n1 + n2];
            n2 += dr1;
            const int carry = n2 >= 25 ? 1 : 0;
            n2 -= 25 * carry;
            f += dq1 + carry;
        }
    }
    // tw is stored by output index: tw[25 * k1 + n2] = W_200^(n2 k1), so that phase 1's lanes (n2 fastest) read consecutive
    // entries (indexed n2 * k1 the reads were strided by k1: up to 4 lanes per bank)
    // ROW == 40: a row holds its 25 entries and the first 15 again.  The table reads are ds_read2_b64, served in groups of 16 lanes
    // over 32 banks, and a group in which the frame changes (n2 = a .. 24, 0 .. a - 10) meets itself in up to 7 banks -- 270 of the
    // kernel's 720 conflict cycles per workgroup (simulated per phase; matches SQ_LDS_BANK_CONFLICT / 5120); with the longer rows
    // such a group reads entries a .. a + 15.
    for (int i = tid; i < 8 * ROW; i += nthr) {
        const int k1 = i / ROW, j = i - ROW * k1, n2 = j < 25 ? j : j - 25;
        tw[i] = reinterpret_cast<const v2 *>(twM)[n2 * k1];
        win2[i] = reinterpret_cast<const v2 *>(window)[25 * k1 + n2];
    }
    for (int i = tid; i <= M; i += nthr) tw2[i] = reinterpret_cast<const v2 *>(tw2M)[i];
    __syncthreads();

    // phase 1
    auto butterfly1 = [&](v2 (&v)[8], int f, int n2, int j) {  // j: the item's table column
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = v[n1] * win2[ROW * n1 + j];
        pk_dft8(v);
        v2 *zf = Z + f * kMP400 + n2;
        zf[0] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) zf[k1 * 25] = cmulp(v[k1], tw[ROW * k1 + j]);
    };
    if (ahead) {
#pragma unroll
        for (int r = 0; r < kRounds1; ++r)
            if (fr_[r] < nf) butterfly1(au[r], fr_[r], n2_[r], col_[r]);
    } else {
        int it = tid;
        for (int f = tid / 25, n2 = tid - (tid / 25) * 25; f < nf; it += nthr) {
            const v2 *fr = reinterpret_cast<const v2 *>(clip + (unsigned)(f * hop));
            v2 v[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) v[n1] = fr[25 * n1 + n2];
            butterfly1(v, f, n2, column(it, f, n2));
            n2 += dr1;
            const int carry = n2 >= 25 ? 1 : 0;
            n2 -= 25 * carry;
            f += dq1 + carry;
        }
    }
    __syncthreads();
    // phase 2: read, barrier, compute, write in place (natural order); 8 * F <= blockDim.x items.
    // Banks: the 25 reads of an item compile to ds_read2_b64, which LDS serves in groups of 16 consecutive lanes over 32
    // dword banks.  Items are laid out FRAMES FASTEST (item = k1 * nf + f): a 16-lane group then holds 16 frames of one k1, float2
    // offsets 201 f + 25 k1 + n2, and 201 = 9 (mod 16) is odd -- 16 distinct bank pairs.  (Round 2 had k1 fastest: 2 frames x 8 k1
    // per group, offsets 9 f + 9 k1 (mod 16), the two frames' sets meeting in 6 of 8 -- two-way conflicts on every access, and no
    // frame stride serves that layout and phase 3's frame-fastest lanes at once: 8 (mod 16) here against odd there.)  Only the
    // groups that straddle two k1 (nf is not a multiple of 16) still meet, in at most 3 of their 16 lanes.
    {
        const int k1 = tid / nf, f = tid - k1 * nf;
        const bool live = k1 < 8;
        v2 x[25];
        if (live) {
            const v2 *zf = Z + f * kMP400 + k1 * 25;
#pragma unroll
            for (int n2 = 0; n2 < 25; ++n2) x[n2] = zf[n2];
        }
        __syncthreads();
        if (live) {
            pk_dft25(x);
            v2 *zo = Z + f * kMP400 + k1;
#pragma unroll
            for (int k2 = 0; k2 < 25; ++k2) zo[8 * k2] = x[k2];
        }
    }
    __syncthreads();
    // phase 3: untangle + magnitude, one pair (k, M - k) per item.  Item it = k * nf + f, frames fastest (contiguous stores);
    // (k, f) advance by (nthr / nf, nthr % nf) with a carry instead of a division per item, and every address is a 32-bit
    // offset from a uniform base: the item used to spend more instructions on 64-bit index arithmetic than on the butterfly.
    float *Sb = S + spec_off + t0;
    const int dq = nthr / nf, dr = nthr - dq * nf;
    int k = tid / nf, f = tid - k * nf;
    for (; k <= M / 2; ) {
        const v2 *zf = Z + f * kMP400;
        const v2 A = zf[k], Bz = zf[k == 0 ? 0 : M - k];
        // X[k] = (e - i g) / 2 and X[M-k] = conj(e + i g) / 2 with e = A + conj(B), g = tw2[k] (A - conj(B)):
        // one twiddle product serves both bins of the pair; the halving is exact, so it is taken on the magnitude
        const v2 e = add_conj(A, Bz), d = sub_conj(A, Bz);
        const v2 g = cmulp(d, tw2[k]);
        {
            const v2 x = sub_i(e, g);  // (e.x + g.y, e.y - g.x)
            const float v = 0.5f * mag(x.x, x.y);
            if (!probe || v == -1.f) Sb[(unsigned)(k * T + f)] = v;
        }
        if (k != M / 2) {
            const v2 x = add_i(e, g);  // (e.x - g.y, e.y + g.x)
            const float v = 0.5f * mag(x.x, x.y);
            if (!probe || v == -1.f) Sb[(unsigned)((M - k) * T + f)] = v;
        }
        f += dr;
        const int carry = f >= nf ? 1 : 0;
        f -= carry * nf;
        k += dq + carry;
    }
}

}  // namespace

namespace {
// the specialised 8 x 25 kernel serves n_fft = 400 with an even hop (frames start on 8-byte boundaries when the clip does)
bool stft400_ok(const smh_ctx *ctx) {
    return ctx->cfg.n_fft == 400 && ctx->M == 200 && ctx->cfg.win_length <= 400 && (ctx->cfg.hop % 2) == 0 && !getenv("SMH_STFT_GENERIC");
}
void generic_args(const smh_ctx *ctx, StftArgs &a) {
    a.n_fft = ctx->cfg.n_fft, a.hop = ctx->cfg.hop, a.M = ctx->M, a.K = ctx->K;
    a.n_stages = ctx->n_stages;
    for (int i = 0; i < smh::kMaxFftStages; ++i) {
        a.radix[i] = i < ctx->n_stages ? ctx->radix[i] : 1;
        a.inv_nb[i] = (float)a.radix[i] / (float)a.M;
    }
    for (int i = 0, ns = 1; i < smh::kMaxFftStages; ++i) {
        a.inv_ns[i] = 1.0f / (float)ns;
        a.tmul[i] = a.M / (ns * a.radix[i]) > 0 ? a.M / (ns * a.radix[i]) : 0;
        ns *= a.radix[i];
        if (ns > a.M) ns = a.M;
    }
}
constexpr int kRagFramesGeneric = 16;
}  // namespace

namespace smh_stft {
int rag_frames(const smh_ctx *ctx, bool aligned8) { return stft400_ok(ctx) && aligned8 ? kRagFrames : kRagFramesGeneric; }

// One launch for clips of different lengths (smh_rag.h).  A frame's transform does not depend on the frames it shares a workgroup
// with, so every clip gets the bits smh_stft_mag_f32 gives it alone.
int launch_rag(const smh_ctx *ctx, const float *d_audio, float *d_S, const smh_rag::Clip *d_clips, const smh_rag::Item *d_items,
               int n_items, bool aligned8, hipStream_t st) {
    if (n_items <= 0) return SMH_OK;
    const unsigned grid = (unsigned)(8 * (((long long)n_items + 7) / 8));
    if (stft400_ok(ctx) && aligned8) {
        const int F = kRagFrames;
        const size_t lds = sizeof(float2) * (8 * 25 + 202 + 8 * 25 + (size_t)F * kMP400);
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)stft400_kernel<25>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(stft400_kernel<25>, dim3(grid), dim3(256), lds, st, d_audio, ctx->d_window, ctx->d_twM, ctx->d_tw2M, d_S, 0,
                           ctx->cfg.hop, 0, F, 0, n_items, 0, d_clips, d_items);
        return smh::launch_status("stft400_kernel (ragged)");
    }
    StftArgs a;
    generic_args(ctx, a);
    a.n_samples = 0, a.T = 0, a.tt = kRagFramesGeneric, a.inv_tt = 1.0f / (float)a.tt;
    const int MP = a.M + (a.M >> 4) + 2;
    const size_t lds = sizeof(float2) * ((size_t)a.M + (a.M + 1) + 2 * (size_t)a.tt * MP);
    SMH_REQUIRE(lds <= 150 * 1024, "ragged STFT: n_fft=%d too large for the LDS FFT", a.n_fft);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)stft_mag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    a.B = n_items, a.xcd_tiles = 0;
    hipLaunchKernelGGL(stft_mag_kernel, dim3(grid), dim3(kThreads), lds, st, a, d_audio, ctx->d_window, ctx->d_twM, ctx->d_tw2M, d_S,
                       d_clips, d_items);
    return smh::launch_status("stft_mag_kernel (ragged)");
}
}  // namespace smh_stft

extern "C" int smh_stft_mag_f32(const smh_ctx *ctx, const float *d_audio, int B, int n_samples, float *d_S,
                                void *stream) {
    SMH_REQUIRE(ctx && (d_audio || B == 0) && (d_S || B == 0), "smh_stft_mag_f32: null argument");
    SMH_REQUIRE(B >= 0 && B <= 65535, "smh_stft_mag_f32: B=%d out of range", B);
    const int T = smh_num_frames(n_samples, ctx->cfg.n_fft, ctx->cfg.hop);
    SMH_REQUIRE(T >= 1, "smh_stft_mag_f32: clip of %d samples is shorter than n_fft=%d", n_samples, ctx->cfg.n_fft);
    if (B == 0) return SMH_OK;
    // n_fft = 400 with 8-byte aligned frames: the specialised 8 x 25 kernel (SMH_STFT_GENERIC=1 forces the generic one)
    // (an odd clip length only matters for where the NEXT clip starts: a single clip keeps the specialised kernel)
    if (stft400_ok(ctx) && ((n_samples % 2) == 0 || B == 1) && (reinterpret_cast<uintptr_t>(d_audio) % 8) == 0) {
        // frames per workgroup: <= 20 (37 KB of LDS, 128 VGPRs: FOUR 256-thread workgroups per CU), splitting T evenly (98 -> 5 x 20,
        // the last with 18).  25 frames (45 KB: three per CU) measured 69-71 us, 20 frames 64.6; 16 and fewer leave half of phase
        // 2's threads idle (8 items per frame) and are slower again.  tools/gpu/r2_stft_tune.sh sweeps it.
        int maxf = 20, nthreads = 256;
        if (const char *ev = getenv("SMH_STFT_FRAMES")) sscanf(ev, "%d,%d", &maxf, &nthreads);  // tuning override
        if (nthreads != 512) nthreads = 256;
        if (maxf < 1) maxf = 1;
        if (maxf > nthreads / 8) maxf = nthreads / 8;
        const int ntiles = (T + maxf - 1) / maxf;
        const int F = (T + ntiles - 1) / ntiles;
        int row = 25;
        if (const char *ev = smh::probe_env("SMH_STFT_ROW")) row = atoi(ev) == 40 ? 40 : 25;
        const size_t lds = sizeof(float2) * (8 * row + 202 + 8 * row + (size_t)F * kMP400);
        auto kernel = row == 40 ? stft400_kernel<40> : stft400_kernel<25>;
        SMH_CHECK_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int nt = (T + F - 1) / F;
        bool by_xcd = true;  // SMH_STFT_XCD=0: the plain (tile, clip) grid (A/B, tests)
        if (const char *ev = getenv("SMH_STFT_XCD")) by_xcd = atoi(ev) != 0;
        if ((long long)B * nt >= (1ll << 31) - 8) by_xcd = false;
        dim3 grid(nt, B), block(nthreads);
        if (by_xcd) grid = dim3((unsigned)(8 * (((long long)B * nt + 7) / 8)), 1);
        const int probe = smh::probe_env("SMH_STFT_PROBE_NOSTORE") ? 1 : 0;  // timing experiment, S is not written
        hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)stream, d_audio, ctx->d_window, ctx->d_twM,
                           ctx->d_tw2M, d_S, n_samples, ctx->cfg.hop, T, F, probe, B, by_xcd ? nt : 0, nullptr, nullptr);
        return smh::launch_status("stft400_kernel");
    }
    StftArgs a;
    generic_args(ctx, a);
    a.n_samples = n_samples, a.T = T;
    // frames per workgroup: <= 16, chosen to split T evenly (T=98 -> 7 tiles of 14)
    const int max_tt = 16;
    const int ntiles = (T + max_tt - 1) / max_tt;
    a.tt = (T + ntiles - 1) / ntiles;
    a.inv_tt = 1.0f / (float)a.tt;
    const int MP = a.M + (a.M >> 4) + 2;
    const size_t lds = sizeof(float2) * ((size_t)a.M + (a.M + 1) + 2 * (size_t)a.tt * MP);
    SMH_REQUIRE(lds <= 150 * 1024, "smh_stft_mag_f32: n_fft=%d too large for the LDS FFT", a.n_fft);
    SMH_REQUIRE((size_t)a.tt * a.K < (1u << 20), "smh_stft_mag_f32: tile too large");
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)stft_mag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    const int ntg = (T + a.tt - 1) / a.tt;
    bool xcd_grid = (long long)B * ntg < (1ll << 31) - 8;
    if (const char *ev = getenv("SMH_STFT_XCD")) xcd_grid = xcd_grid && atoi(ev) != 0;
    a.B = B, a.xcd_tiles = xcd_grid ? ntg : 0;
    dim3 grid(ntg, B), block(kThreads);
    if (xcd_grid) grid = dim3((unsigned)(8 * (((long long)B * ntg + 7) / 8)), 1);
    hipLaunchKernelGGL(stft_mag_kernel, grid, block, lds, (hipStream_t)stream, a, d_audio, ctx->d_window, ctx->d_twM,
                       ctx->d_tw2M, d_S, nullptr, nullptr);
    return smh::launch_status("stft_mag_kernel");
}
