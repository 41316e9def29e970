// Random draws of the training step, on the device and in ONE pass each (gfx950).
//   smh_noise_augment_f32 : batchData + N(0, scale)     -- Proposed_Work_Results.py:239-242 (np.random.normal + np.add)
//   smh_dropout_masks_f32 : the SpatialDropout1D / Dropout keep masks of one step, 0 or 1 / keep, two rates in one launch
//                           (lib/proposed_architectures.py: tcn.TCN(dropout_rate=...), Dropout(0.4) of the heads)
// Until round 3 these were torch kernels on the step's stream (normal_ 15.8 + add 15.1 + bernoulli 5.0 + mul 5.2 us per 510-clip
// step): two full passes over the 33 MB patch tensor where one does.  The generator is Philox4x32-10 (counter = element group,
// key = seed, second counter word = the caller's offset: a (seed, offset) pair never repeats a stream), normals by Box-Muller on
// v_log_f32 / v_sin_f32 / v_cos_f32 (the latter two take revolutions: no 2 pi product).  Both kernels are HBM-bound: 16 bytes in
// and 16 bytes out per lane and ~150 VALU instructions per four values.
#include "smh_common.h"

namespace {

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0, c[1] = lo1, c[2] = n2, c[3] = lo0;
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): four 32-bit words per (counter, key)
__device__ __forceinline__ void philox4x32_10(unsigned long long group, unsigned long long offset, unsigned long long seed,
                                              unsigned (&r)[4]) {
    r[0] = (unsigned)group, r[1] = (unsigned)(group >> 32), r[2] = (unsigned)offset, r[3] = (unsigned)(offset >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(r, k0, k1);
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}

// two standard normals from two 32-bit words: u1 in (0, 1], u2 in [0, 1) revolutions
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float &n0, float &n1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-8f;  // (k + 1) / 2^24: never 0, log finite
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;
    // -2 ln u1 = -2 ln2 * log2(u1)
    const float rad = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    n0 = rad * __builtin_amdgcn_cosf(u2), n1 = rad * __builtin_amdgcn_sinf(u2);
}

__global__ void __launch_bounds__(256) noise_augment_kernel(const float *__restrict__ x, float *__restrict__ out, size_t n,
                                                            float scale, unsigned long long seed, unsigned long long offset) {
    const size_t groups = (n + 3) / 4;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        unsigned r[4];
        philox4x32_10(g, offset, seed, r);
        float z[4];
        box_muller(r[0], r[1], z[0], z[1]);
        box_muller(r[2], r[3], z[2], z[3]);
        const size_t i = 4 * g;
        if (i + 4 <= n) {
            float4 v = *reinterpret_cast<const float4 *>(x + i);
            v.x += scale * z[0], v.y += scale * z[1], v.z += scale * z[2], v.w += scale * z[3];
            *reinterpret_cast<float4 *>(out + i) = v;
        } else {
            for (size_t j = i; j < n; ++j) out[j] = x[j] + scale * z[j - i];
        }
    }
}

__global__ void __launch_bounds__(256) dropout_masks_kernel(float *__restrict__ out, size_t n_a, float keep_a, size_t n_b,
                                                            float keep_b, unsigned long long seed, unsigned long long offset) {
    const size_t n = n_a + n_b, groups = (n + 3) / 4;
    const float inv_a = 1.0f / keep_a, inv_b = 1.0f / keep_b;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        unsigned r[4];
        philox4x32_10(g, offset, seed, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = 4 * g + j;
            if (i < n) {
                const float u = (float)(r[j] >> 8) * 5.9604644775390625e-8f;  // [0, 1)
                const bool a = i < n_a;
                out[i] = u < (a ? keep_a : keep_b) ? (a ? inv_a : inv_b) : 0.f;
            }
        }
    }
}

int grid_for(size_t groups) {
    const size_t blocks = (groups + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));  // grid-stride: 32 workgroups per CU at most
}

}  // namespace

extern "C" int smh_noise_augment_f32(const float *d_x, float *d_out, size_t n, float scale, unsigned long long seed,
                                     unsigned long long offset, void *stream) {
    SMH_REQUIRE((d_x && d_out) || n == 0, "smh_noise_augment_f32: null argument");
    SMH_REQUIRE(scale >= 0.f, "smh_noise_augment_f32: scale=%g is negative", (double)scale);
    SMH_REQUIRE((reinterpret_cast<uintptr_t>(d_x) % 16) == 0 && (reinterpret_cast<uintptr_t>(d_out) % 16) == 0,
                "smh_noise_augment_f32: buffers must be 16-byte aligned");
    if (n == 0) return SMH_OK;
    hipLaunchKernelGGL(noise_augment_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_x, d_out, n, scale,
                       seed, offset);
    return smh::launch_status("noise_augment_kernel");
}

extern "C" int smh_dropout_masks_f32(float *d_out, size_t n_a, float keep_a, size_t n_b, float keep_b, unsigned long long seed,
                                     unsigned long long offset, void *stream) {
    SMH_REQUIRE(d_out || n_a + n_b == 0, "smh_dropout_masks_f32: null argument");
    SMH_REQUIRE(keep_a > 0.f && keep_a <= 1.f && keep_b > 0.f && keep_b <= 1.f,
                "smh_dropout_masks_f32: keep probabilities %g, %g outside (0, 1]", (double)keep_a, (double)keep_b);
    if (n_a + n_b == 0) return SMH_OK;
    hipLaunchKernelGGL(dropout_masks_kernel, dim3(grid_for((n_a + n_b + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_out, n_a,
                       keep_a, n_b, keep_b, seed, offset);
    return smh::launch_status("dropout_masks_kernel");
}
