// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 per SIMD as a function of waves per SIMD and independent
// accumulator chains per wave.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_rate tools/ubench/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CH>
__global__ void k(float *out, int iters, float a0, float b0) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / CH; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
void run(int waves_per_simd) {
    const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU, waves spread over the 4 SIMDs
    const int iters = 4096, grid = 256;
    float *out;
    hipMalloc(&out, sizeof(float) * grid * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(threads), 0, 0, out, 16, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
    const double tflops = (double)grid * 4 * mfma_per_simd * 2048.0 / (ms * 1e-3) / 1e12;
    printf("chains %d  waves/SIMD %d : %.3f ms  %.1f ns per MFMA per SIMD  %.1f TFLOP/s\n", CH, waves_per_simd, ms,
           ms * 1e6 / mfma_per_simd, tflops);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 3, 4}) run<1>(w);
    for (int w : {1, 2, 3, 4}) run<2>(w);
    for (int w : {1, 2, 3, 4}) run<4>(w);
    for (int w : {1, 2}) run<8>(w);
    return 0;
}
