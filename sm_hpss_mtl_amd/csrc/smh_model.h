// Internal definition of the B3_MTL model object shared by the inference (smh_tcn.hip) and training
// (smh_train.hip) translation units.
#pragma once
#include "smh_common.h"

namespace smh_tcn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int C = 32;          // nb_filters (fixed by the MFMA tiling)
constexpr int SX = 36;         // LDS row stride of x in floats (16-byte aligned rows)
constexpr int kMaxHeads = 4;
constexpr int kHidden = 16;    // Dense(16) of every MTL head
constexpr int kMaxG = 16;      // patches per workgroup <= MFMA N
constexpr int kPS = 80;        // row stride of the Dense-on-trunk outputs: up to 5 M-tiles (5-class: 69 outputs)
constexpr float kNormEps = 1e-5f;
constexpr float kBnEps = 1e-3f;
// Packed per-block weights: 64 A-operand slots per lane (48 dilated-conv + 16 1x1), four slots per lane contiguous
// ([slot / 4][lane][slot % 4], see load_block_lds), then [b1 32][b2 32]
constexpr int kBlockFloats = 24 * 2 * 64 + 8 * 2 * 64 + 32 + 32;

struct TcnArgs {
    int N, T, F, FQ, G, GRP, n_blocks, n_dil, vec_ok;
    int D, NH, n_mt, n_classes, n_heads, out_dim, skip_heads;
    int wlds;     // block weights staged through two LDS slots (0: read from L2 by every wave -- patches too long for the slots)
    unsigned long long *trace;  // tools only (tools/trace_model.py): per-task timestamps of workgroup 0, or nullptr
    int tune;     // experiment switches of the skewed schedule (SMH_TCN_TUNE; tools only)
    int *status;  // device error word of the model (smh_model_status): bit 0 = a wave of the skewed schedule gave up on a dependency
    int spin_limit;  // polls before a wave of the skewed schedule gives up (kSkewSpinLimit; lowered only by the debug knob of the test)
    int split_last;  // barrier schedule, two register sets: the last column tile is computed by TWO waves, 16 output channels each (smh_tcn.hip: half_tile_compute)
    int from_x0;  // X holds the two per-half partials of layer 0, (N, 2, T, 32) (smh_features_l0_f32), instead of patches
    int x0_shift, x0_T;  // from_x0 with x0_shift > 0 (smh_model_forward_dense_f32): X is (2, x0_T, 32), the partials of EVERY frame of a
                         // featuregram of x0_T frames; patch n is the window of T frames starting at min(n * x0_shift, x0_T - T)
    int head_odim[kMaxHeads];
    int head_sigmoid[kMaxHeads];
};

// training-mode extras of the forward kernel (all optional)
struct TrainIO {
    float *acts;            // (N, n_blocks + 1, T, 32): input of every block, then the pre-relu TCN output
    const float *drop_tcn;  // (N, n_blocks, 32) SpatialDropout1D masks (0 or 1/(1-rate)), or nullptr
    float *pre;             // (N, kPS): Dense-on-trunk outputs incl. bias (3C logits | head Dense(16)s)
    float *upre;            // (N, n_blocks, T, 32): every block's dilated-conv output incl. bias, before the relu -- the gates of
                            // the backward's relu / channel-max normalisation (it used to recompute them: 48 products per tile)
    int acts_last_only;     // (split-bf16 forward only) save slot n_blocks of `acts` alone: the split-bf16 backward rebuilds every
                            // block's input from its output (x_b = x_b+1 - W2 . y_b - b2), half of the saved bytes never move
};

// heads_train_kernel (smh_train.hip): batch-statistics BN, Dropout, the four losses and d loss / d pre for the '3C'
// softmax and the MTL heads, from the Dense-on-features outputs `pre` (N, kPS) = [3C logits | Dense(16) per head].
// Shared by the B3_MTL trainer and the Conv2D baselines' trainer (smh_cnn_train.hip).
struct HeadsArgs {
    int N, D, NH, n_classes, n_heads, out_dim;
    int head_odim[kMaxHeads], head_sigmoid[kMaxHeads];
    float lw[kMaxHeads + 1];
    size_t goff_head[kMaxHeads];  // canonical offset of each head's first tensor (dense kernel) in `grad`
    size_t goff_c3b;              // canonical offset of the 3C bias
    size_t hp_off[kMaxHeads];     // offset in `hp` of each head's [gamma, beta, mean, var (16 each), out kernel, out bias]
    int ext_losses;               // 1: `losses` has 3 n_heads + 4 floats and gets the per-head binary accuracies at [2 nh + 4 + h]
    int stamps;                   // tools only (SMH_HEADS_STAMPS): thread 0 prints the phase durations (100 MHz ticks)
};
// ticket: one zero-initialised device word the kernel's workgroups count themselves on (left at zero again)
int launch_heads_train(const HeadsArgs &a, const float *pre, const float *y, const float *hp, const float *drop, float *dpre,
                       float *dxh, float *grad, float *bnstat, float *losses, unsigned *ticket, hipStream_t st);

}  // namespace smh_tcn

struct smh_model {
    smh_model_cfg cfg;
    int n_blocks, n_heads, NH, n_mt, D, out_dim, FQ;
    int head_odim[smh_tcn::kMaxHeads], head_sigmoid[smh_tcn::kMaxHeads];
    size_t n_params;
    float *d_flat = nullptr;  // master weights, canonical (Keras-layout) order, n_params floats
    float *d_W0 = nullptr;    // layer-0 A operands + bias0
    float *d_Wb = nullptr;    // per-block packed weights
    float *d_WhA = nullptr;   // Dense-on-trunk weights in A-operand order + biases
    float *d_hp = nullptr;    // per-head BN / out params
    int *d_map = nullptr;     // gather map: packed[i] = map[i] ? flat[map[i]-1] : 0 for [W0 | Wb | WhA | hp]
    size_t nW0, nWb, nWhA, nhp;
    // bf16 operand cache of smh_model_forward_bf16 (smh_tcn_bf16.hip): rebuilt when `version` moves
    void *d_bf16 = nullptr;
    unsigned long long version = 1, bf16_version = 0;
    // Device error word (smh_model_status).  A kernel that cannot produce results -- today: a wave of the skewed schedule, or one
    // half of a split last-round tile of the barrier schedule, whose partner never arrived within its bounded spin -- ORs a bit in;
    // the skewed schedule also zero-fills its workgroup's outputs, the split tile leaves whatever it computed from the stale
    // exchange (NOT results either way).  The host reads and clears the word in smh_model_status.  Never a NaN payload:
    // smh_tcn.hip is compiled -fno-honor-nans.
    int *d_status = nullptr;
};

namespace smh_tcn {
// canonical offsets
struct Offsets {
    size_t w0_k, w0_b;                 // initial conv kernel / bias
    size_t blk0, blk_stride;           // first block; per block: k1 (3*C*C), b1 (C), k2 (C*C), b2 (C)
    size_t c3_k, c3_b;                 // 3C kernel (D x ncls), bias
    size_t head[kMaxHeads];            // per head: dense k (D x 16), dense b, gamma, beta, mean, var, out k, out b
};
Offsets offsets(const smh_model *m);
void fill_args(const smh_model *m, int N, TcnArgs *a, size_t *lds);
int repack(smh_model *m, hipStream_t st);  // d_flat -> packed operand buffers
int launch_forward_bf16_train(smh_model *m, const float *d_x, int N, const TrainIO *tio, hipStream_t st);  // smh_tcn_bf16.hip
bool backward_bf16_supported(int T, int n_dil);  // smh_train_bf16.hip: the patch geometry fits the split-bf16 backward's LDS plan
int launch_forward(const smh_model *m, const float *d_x, int N, float *d_out, float *d_trunk, const TrainIO *tio,
                   hipStream_t st, int from_x0 = 0, int x0_shift = 0, int x0_T = 0);
// smh_model_cfg.block_variant = 1 (smh_tcn_v2.hip): the two-convolution residual block of keras-tcn >= 2.8, inference only
int launch_forward_v2(const smh_model *m, const float *d_x, int N, float *d_out, float *d_trunk, hipStream_t st);
}  // namespace smh_tcn
