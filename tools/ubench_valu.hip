// Micro-benchmark: cost of the VALU patterns used by the sliding-median update on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu tools/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// Every kernel runs `iters` iterations of a 64-"slot" body; prints cycles per slot per SIMD.
template <int V>
__global__ void __launch_bounds__(512) k(float *out, int iters, float a0) {
    float v0 = a0 + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, x = v0 * 0.5f, o = v0 * 0.25f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {  // independent v_med3 (1 VALU per slot)
            asm volatile(REP64("v_med3_f32 %0, %1, %4, %2\n v_med3_f32 %1, %2, %4, %3\n v_med3_f32 %2, %3, %4, %0\n v_med3_f32 %3, %0, %4, %1\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(x));
        } else if constexpr (V == 1) {  // cmp(vcc) ; s_nop 1 ; cndmask   (x4 per rep)
            asm volatile(REP64("v_cmp_nge_f32 vcc, %0, %4\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc\n"
                               "v_cmp_nge_f32 vcc, %1, %4\n s_nop 1\n v_cndmask_b32 %1, %2, %1, vcc\n"
                               "v_cmp_nge_f32 vcc, %2, %4\n s_nop 1\n v_cndmask_b32 %2, %3, %2, vcc\n"
                               "v_cmp_nge_f32 vcc, %3, %4\n s_nop 1\n v_cndmask_b32 %3, %0, %3, vcc\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(o) : "vcc");
        } else if constexpr (V == 2) {  // cmp ; med3 ; s_nop 0 ; cndmask  (what hipcc emitted first)
            asm volatile(REP64("v_cmp_nge_f32 vcc, %0, %4\n v_med3_f32 %5, %5, %6, %1\n s_nop 0\n v_cndmask_b32 %0, %1, %0, vcc\n"
                               "v_cmp_nge_f32 vcc, %1, %4\n v_med3_f32 %5, %5, %6, %2\n s_nop 0\n v_cndmask_b32 %1, %2, %1, vcc\n"
                               "v_cmp_nge_f32 vcc, %2, %4\n v_med3_f32 %5, %5, %6, %3\n s_nop 0\n v_cndmask_b32 %2, %3, %2, vcc\n"
                               "v_cmp_nge_f32 vcc, %3, %4\n v_med3_f32 %5, %5, %6, %0\n s_nop 0\n v_cndmask_b32 %3, %0, %3, vcc\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(o), "v"(v4), "v"(x) : "vcc");
        } else if constexpr (V == 3) {  // three SGPR pairs in flight, no nops: cmp,cmp,cmp,cnd,cnd,cnd + 3 med3
            asm volatile(REP64("v_cmp_nge_f32 s[20:21], %0, %4\n v_cmp_nge_f32 s[22:23], %1, %4\n v_cmp_nge_f32 s[24:25], %2, %4\n"
                               "v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %1, %2, %1, s[22:23]\n v_cndmask_b32 %2, %3, %2, s[24:25]\n"
                               "v_med3_f32 %5, %5, %6, %0\n v_med3_f32 %5, %5, %6, %1\n v_med3_f32 %5, %5, %6, %2\n"
                               "v_cmp_nge_f32 s[20:21], %3, %4\n v_med3_f32 %5, %5, %6, %3\n v_med3_f32 %5, %5, %6, %3\n v_cndmask_b32 %3, %0, %3, s[20:21]\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(o), "v"(v4), "v"(x) : "s20", "s21", "s22", "s23", "s24", "s25");
        } else if constexpr (V == 4) {  // v_cmp e64 only
            asm volatile(REP64("v_cmp_nge_f32 s[20:21], %0, %4\n v_cmp_nge_f32 s[22:23], %1, %4\n v_cmp_nge_f32 s[24:25], %2, %4\n v_cmp_nge_f32 s[26:27], %3, %4\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(o) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (V == 5) {  // v_cndmask e64 only (fixed mask)
            asm volatile(REP64("v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %1, %2, %1, s[20:21]\n v_cndmask_b32 %2, %3, %2, s[20:21]\n v_cndmask_b32 %3, %0, %3, s[20:21]\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(o) : "s20", "s21");
        } else if constexpr (V == 6) {  // v_max_f32 / v_min_f32 e32 pairs
            asm volatile(REP64("v_max_f32 %0, %1, %4\n v_min_f32 %1, %2, %4\n v_max_f32 %2, %3, %4\n v_min_f32 %3, %0, %4\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(x));
        } else if constexpr (V == 7) {  // integer variant: v_cmp_ge_u32 + cndmask, 3 in flight
            asm volatile(REP64("v_cmp_ge_u32 s[20:21], %0, %4\n v_cmp_ge_u32 s[22:23], %1, %4\n v_cmp_ge_u32 s[24:25], %2, %4\n"
                               "v_cndmask_b32 %0, %1, %0, s[20:21]\n v_cndmask_b32 %1, %2, %1, s[22:23]\n v_cndmask_b32 %2, %3, %2, s[24:25]\n"
                               "v_med3_u32 %5, %5, %6, %0\n v_med3_u32 %5, %5, %6, %1\n v_med3_u32 %5, %5, %6, %2\n"
                               "v_cmp_ge_u32 s[20:21], %3, %4\n v_med3_u32 %5, %5, %6, %3\n v_med3_u32 %5, %5, %6, %3\n v_cndmask_b32 %3, %0, %3, s[20:21]\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(o), "v"(v4), "v"(x) : "s20", "s21", "s22", "s23", "s24", "s25");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5;
}

template <int V>
void run(const char *name, int valu_per_rep, int slots_per_rep) {
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * 8 * 512);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int wps : {1, 2, 4, 8}) {          // waves per SIMD
        const int blocks = 256 * wps / 2;   // 512-thread blocks = 8 waves = 2 per SIMD; wps/2 blocks per CU
        const int threads = wps == 1 ? 256 : 512;
        const int nb = wps == 1 ? 256 : blocks;
        const int iters = 200;
        k<V><<<nb, threads>>>(d, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(a);
        k<V><<<nb, threads>>>(d, iters, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        // per SIMD: wps waves each executing iters*64 reps
        const double reps = (double)iters * 64 * wps;
        const double ns_per_rep = ms * 1e6 / reps;
        printf("%-34s waves/SIMD=%d  %.2f ns per rep per SIMD  => %.2f ns per VALU  (%.2f cycles @2.4GHz), %.2f ns per slot\n", name, wps,
               ns_per_rep, ns_per_rep / valu_per_rep, ns_per_rep / valu_per_rep * 2.4, ns_per_rep / slots_per_rep);
    }
    hipFree(d);
}

int main() {
    run<0>("med3 x4", 4, 4);
    run<6>("max/min e32 x4", 4, 4);
    run<4>("cmp e64 x4", 4, 4);
    run<5>("cndmask e64 x4", 4, 4);
    run<1>("cmp vcc; s_nop 1; cnd  x4", 8, 4);
    run<2>("cmp; med3; s_nop 0; cnd x4", 12, 4);
    run<3>("3 sgpr pairs, no nops, 4 slots", 13, 4);
    run<7>("same, u32 ops", 13, 4);
    return 0;
}
