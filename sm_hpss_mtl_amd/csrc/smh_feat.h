// Internal interface between smh_feat.hip and smh_frontend.hip.
#pragma once
#include "smh_common.h"

namespace smh_feat {

constexpr int kMaxMels = 256;     // filters held in LDS by the fused feature kernel
constexpr int kMaxMelNnz = 2048;  // taps held in LDS

struct MelTable {
    int n_mels, nnz;
    const int *start, *count, *off;
    const float *w;
};

MelTable mel_table(const smh_ctx *c);

// the bin walk's per-context plan as a kernel argument (smh_ctx.hip builds it): row segments, their bin ranges, and where each
// segment's per-bin records {w0, w1, w2, w3, n_emit, -, -, -} start in `plan`
struct FeatPlan {
    int nseg, pend;  // pend: most filters pending at any bin (2 or 4 accumulators)
    int m0[smh_ctx::kMaxFeatSegs], m1[smh_ctx::kMaxFeatSegs], kbeg[smh_ctx::kMaxFeatSegs], kend[smh_ctx::kMaxFeatSegs],
        off[smh_ctx::kMaxFeatSegs];
    const float *plan;
    unsigned long long *trace;  // tools/trace_features.py: phase stamps (s_memrealtime) per workgroup and wave, or nullptr
};
FeatPlan feat_plan(const smh_ctx *c, int which /* 0: four segments, 1: eight */);

// (S, harm, perc) -> featuregram fv (B, 2*rows, T) with un-clipped dB values + per-array max keys
// harm_tmajor != 0: harm is (B, T, K) as written by smh_median::launch_hpss(want_tmajor = 1)
int launch_hp_feat(const smh_ctx *c, const float *S, const float *harm, const float *perc, int harm_tmajor, int B, int T,
                   float *fv, int *maxkeys, hipStream_t st);
// top_db clip (in place) + StandardScaler + time-major patches
// everything after the medians in one kernel per clip; harmb = harm in layout 2 (B, ceil(T/16), K, 16).
// Returns 1 if it ran, 0 if the shape does not qualify (the caller must then not have asked for layout 2), < 0 on error
int launch_features_clip(const smh_ctx *c, const float *S, const float *harmb, const float *perc, int B, int T, int W,
                         int shift, int nP, float *fv, float *patches, const float *w0, float *x0p, hipStream_t st);
// w0 / x0p non-null: also emit this half's share of the network's first Conv1D (see smh_features_l0_f32)
// scratch (scratch_bytes): device memory the long-clip path may use for its standardised copy of the featuregram (2 * B * rows * T
// floats) instead of a stream-ordered allocation per call -- smh_frontend_f32 hands it the S / perc part of its workspace, dead by then
int launch_std_patch(const smh_ctx *c, float *fv, const int *maxkeys, int B, int T, int W, int shift, int nP,
                     float *patches, hipStream_t st, const float *w0 = nullptr, float *x0p = nullptr, void *scratch = nullptr,
                     size_t scratch_bytes = 0);

}  // namespace smh_feat

namespace smh_median {
// both HPSS medians in one launch; returns 1 if harm was written time-major (B,T,K), 0 if (B,K,T), <0 on error
int launch_hpss(const float *S, int B, int K, int T, int lh, int lp, float *harm, float *perc, int want_tmajor,
                hipStream_t st);
}  // namespace smh_median
