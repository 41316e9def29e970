// Shared by the B3_MTL backward kernels (smh_train.hip: exact-f32 products; smh_train_bf16.hip: split-bf16 operands): the kernel
// arguments and the weight-gradient accumulation.
#pragma once
#include "smh_model.h"

namespace smh_tcn {

// Weight-gradient accumulation.  Default: hardware float atomics (global_atomic_add_f32, -munsafe-fp-atomics) -- fastest, but the
// order in which the workgroups' contributions meet is not fixed, so a gradient's last bits differ from run to run.
// DETERMINISTIC mode (smh_trainer_set_deterministic): every contribution is rounded to a 2^-36 grid and added to a 64-bit
// integer accumulator with an integer atomic.  Integer addition is associative, so the sum does not depend on arrival order:
// two runs of the same step give the same bits.  Range +-2^27 per tensor element, resolution 1.5e-11 (a float32 sum of these
// gradients resolves ~1e-8 at best); det_finalize_kernel converts the accumulators back into the float gradient and clears them.
constexpr float kDetScale = 68719476736.0f;          // 2^36
constexpr double kDetInvScale = 1.0 / 68719476736.0;
__device__ __forceinline__ void gadd(float *grad, unsigned long long *gq, size_t i, float v) {
    if (gq) atomicAdd(gq + i, (unsigned long long)__float2ll_rn(v * kDetScale));  // (uniform branch)
    else atomicAdd(grad + i, v);
}

struct BwdArgs {
    unsigned long long *gq;  // deterministic mode: the fixed-point accumulators (n_params), else nullptr
    int N, T, F, n_blocks, n_dil, D, NH, n_classes, n_heads;
    int stamps;  // tools only (SMH_BWD_STAMPS): workgroup 0 prints the time its phases took, summed over the blocks
    int split3;  // MFMA kernel: a lone last-round tile of phase 3 is shared by two waves (SMH_BWD_SPLIT=0 switches it off: A/B and tests)
    int use_wt;  // VALU kernel: keep transposed LDS copies of the block kernels (0 for patches so long that they do not fit)
    Offsets off;
};


// smh_train_bf16.hip: the residual blocks' backward pass on the bf16 matrix pipe (split operands, f32 accumulators), for the trainer's
// dtype 1.  *d_pack / *pack_cap: the trainer-owned buffer of the blocks' kernels as split A operands (grown on demand, rebuilt every
// call: the weights move every step).  Returns SMH_OK, an error, or kBwdBf16Unsupported when the patch geometry does not fit its LDS
// plan (the caller then runs the f32 kernel).
// d loss / d (TCN output before its final relu), (N, T, 32), as its own small product (dtrunk_kernel): both backward kernels start from it
int launch_dtrunk(const BwdArgs &ba, const float *d_flat, const float *d_acts, const float *d_dpre, float *d_gt, hipStream_t st);
constexpr int kBwdBf16Unsupported = -1000;
int launch_backward_bf16(const BwdArgs &ba, void **d_pack, size_t *pack_cap, const float *d_x, const float *d_flat,
                         const float *d_acts, const float *d_drop_tcn, const float *d_dpre, float *d_grad, const float *d_upre,
                         hipStream_t st);

}  // namespace smh_tcn
