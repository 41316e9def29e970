// Probe of the operand / result layout of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks, k = 1).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mfma4x4_layout tools/ubench/mfma4x4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(float *out) {
    const int l = threadIdx.x;
    const float a = 100.f + l;   // A value of lane l
    const float b = 1000.f * (l + 1);  // B value of lane l
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // D[lane][r] = A[la] * B[lb]: recover la, lb
    for (int l = 0; l < 64; l += 1) {
        if (l % 4 == 0 || l < 8) {
            printf("lane %2d:", l);
            for (int r = 0; r < 4; ++r) {
                int la = -1, lb = -1;
                for (int x = 0; x < 64 && la < 0; ++x)
                    for (int y = 0; y < 64; ++y)
                        if (h[l * 4 + r] == (100.f + x) * (1000.f * (y + 1))) { la = x; lb = y; break; }
                printf("  r%d = A[%2d]*B[%2d]", r, la, lb);
            }
            printf("\n");
        }
    }
    return 0;
}
