// a1: S = |STFT|  --  np.abs(librosa.core.stft(y, n_fft, win_length, hop_length, center=False))
// (call sites /root/reference/lib/preprocessing.py:407,417,429,439).
//
// Frame t = y[t*hop : t*hop + n_fft] * periodic-Hann.  The real frame of length n_fft is packed into a
// complex sequence of length M = n_fft/2 (z[m] = x[2m] + i x[2m+1]), transformed by a mixed-radix
// Stockham autosort FFT that lives entirely in LDS (ping-pong buffers, one butterfly per thread),
// un-tangled into the n_fft/2+1 real-FFT bins and written as magnitudes, frame-fastest, so that the
// (K, T) freq-major layout librosa returns is produced directly.  n_fft = 400 -> M = 200 = 4*2*5*5.
// HBM traffic per clip: 64,000 B audio in (re-reads of the 2.5x frame overlap are served by L2),
// 78,792 B magnitudes out.
#include "smh_common.h"

namespace {

struct StftArgs {
    int n_samples, n_fft, hop, M, K, T, tt;  // tt = frames per workgroup
    int n_stages;
    int radix[smh::kMaxFftStages];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i  (forward DFT quarter turn)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// generic small DFT for odd prime radices: out[q] = sum_r v[r] * w^(r q),  w = exp(-2 pi i / R)
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

template <int R>
__device__ __forceinline__ void dft(float2 *v) {
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        const float2 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]);
        const float2 s1 = cadd(v[1], v[3]), d1 = mul_mi(csub(v[1], v[3]));
        v[0] = cadd(s0, s1);
        v[1] = cadd(d0, d1);
        v[2] = csub(s0, s1);
        v[3] = csub(d0, d1);
    } else {
        dft_generic<R>(v);
    }
}

// One Stockham stage for butterfly j of one frame (Govindaraju et al. formulation).
template <int R, bool FIRST>
__device__ __forceinline__ void stage(const StftArgs &a, int j, int Ns, const float2 *__restrict__ src,
                                      float2 *__restrict__ dst, const float2 *__restrict__ tw,
                                      const float *__restrict__ audio, const float *__restrict__ win) {
    const int M = a.M;
    const int step = M / R;
    float2 v[R];
    if constexpr (FIRST) {
        // read the windowed real frame straight from global memory: z[m] = (x[2m], x[2m+1])
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = j + r * step;
            v[r] = make_float2(audio[2 * m] * win[2 * m], audio[2 * m + 1] * win[2 * m + 1]);
        }
    } else {
        const int k = j % Ns;
        const int tstep = k * (M / (Ns * R));  // exp(-2 pi i k r / (Ns R)) = twM[k r M/(Ns R)]
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float2 x = src[j + r * step];
            if (r > 0) x = cmul(x, tw[r * tstep]);
            v[r] = x;
        }
    }
    dft<R>(v);
    const int k = FIRST ? 0 : j % Ns;
    const int j0 = (j - k) * R + k;  // (j / Ns) * Ns * R + k
#pragma unroll
    for (int r = 0; r < R; ++r) dst[j0 + r * Ns] = v[r];
}

template <bool FIRST>
__device__ __forceinline__ void run_stage(const StftArgs &a, int R, int j, int Ns, const float2 *src, float2 *dst,
                                          const float2 *tw, const float *audio, const float *win) {
    switch (R) {
        case 2: stage<2, FIRST>(a, j, Ns, src, dst, tw, audio, win); break;
        case 3: stage<3, FIRST>(a, j, Ns, src, dst, tw, audio, win); break;
        case 4: stage<4, FIRST>(a, j, Ns, src, dst, tw, audio, win); break;
        case 5: stage<5, FIRST>(a, j, Ns, src, dst, tw, audio, win); break;
        case 7: stage<7, FIRST>(a, j, Ns, src, dst, tw, audio, win); break;
        default: break;
    }
}

__global__ void __launch_bounds__(256)
stft_mag_kernel(StftArgs a, const float *__restrict__ audio, const float *__restrict__ window,
                const float2 *__restrict__ twM, const float2 *__restrict__ tw2M, float *__restrict__ S) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int M = a.M, MP = M + 1;  // +1 float2 row padding: frame-fastest post-processing reads
    float2 *tw = lds;               // M twiddles
    float2 *buf0 = lds + M;
    float2 *buf1 = buf0 + a.tt * MP;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * a.tt;
    const int nf = min(a.tt, a.T - t0);
    for (int i = threadIdx.x; i < M; i += blockDim.x) tw[i] = twM[i];
    const float *clip = audio + (size_t)b * a.n_samples;

    float2 *src = buf0, *dst = buf1;
    int Ns = 1;
    for (int s = 0; s < a.n_stages; ++s) {
        const int R = a.radix[s];
        const int nb = M / R;
        __syncthreads();
        for (int it = threadIdx.x; it < nf * nb; it += blockDim.x) {
            const int f = it / nb, j = it - f * nb;
            if (s == 0)
                run_stage<true>(a, R, j, Ns, nullptr, dst + f * MP, tw, clip + (size_t)(t0 + f) * a.hop, window);
            else
                run_stage<false>(a, R, j, Ns, src + f * MP, dst + f * MP, tw, nullptr, nullptr);
        }
        float2 *tmp = src;
        src = dst;
        dst = tmp;
        Ns *= R;
    }
    __syncthreads();
    // real-FFT untangle + magnitude; frames fastest -> contiguous stores along t
    float *Sb = S + (size_t)b * a.K * a.T + t0;
    for (int it = threadIdx.x; it < nf * a.K; it += blockDim.x) {
        const int k = it / nf, f = it - k * nf;
        const float2 *Z = src + f * MP;
        const float2 zk = Z[k == M ? 0 : k];
        float2 zc = Z[k == 0 ? 0 : M - k];
        zc.y = -zc.y;
        const float2 e = cadd(zk, zc), d = csub(zk, zc);
        const float2 w = tw2M[k];
        const float2 wd = cmul(w, d);  // X = 0.5*e - 0.5*i*w*d
        const float re = 0.5f * (e.x + wd.y);
        const float im = 0.5f * (e.y - wd.x);
        Sb[(size_t)k * a.T + f] = __builtin_sqrtf(re * re + im * im);
    }
}

}  // namespace

extern "C" int smh_stft_mag_f32(const smh_ctx *ctx, const float *d_audio, int B, int n_samples, float *d_S,
                                void *stream) {
    SMH_REQUIRE(ctx && d_audio && d_S, "smh_stft_mag_f32: null argument");
    SMH_REQUIRE(B >= 0 && B <= 65535, "smh_stft_mag_f32: B=%d out of range", B);
    const int T = smh_num_frames(n_samples, ctx->cfg.n_fft, ctx->cfg.hop);
    SMH_REQUIRE(T >= 1, "smh_stft_mag_f32: clip of %d samples is shorter than n_fft=%d", n_samples, ctx->cfg.n_fft);
    if (B == 0) return SMH_OK;
    StftArgs a;
    a.n_samples = n_samples, a.n_fft = ctx->cfg.n_fft, a.hop = ctx->cfg.hop, a.M = ctx->M, a.K = ctx->K, a.T = T;
    a.n_stages = ctx->n_stages;
    for (int i = 0; i < smh::kMaxFftStages; ++i) a.radix[i] = i < ctx->n_stages ? ctx->radix[i] : 1;
    // frames per workgroup: <= 16, chosen to split T evenly (T=98 -> 7 tiles of 14)
    const int max_tt = 16;
    const int ntiles = (T + max_tt - 1) / max_tt;
    a.tt = (T + ntiles - 1) / ntiles;
    const size_t lds = sizeof(float2) * ((size_t)a.M + 2 * (size_t)a.tt * (a.M + 1));
    SMH_REQUIRE(lds <= 150 * 1024, "smh_stft_mag_f32: n_fft=%d too large for the LDS FFT", a.n_fft);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)stft_mag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    dim3 grid((T + a.tt - 1) / a.tt, B), block(256);
    hipLaunchKernelGGL(stft_mag_kernel, grid, block, lds, (hipStream_t)stream, a, d_audio, ctx->d_window, ctx->d_twM,
                       ctx->d_tw2M, d_S);
    return smh::launch_status("stft_mag_kernel");
}
