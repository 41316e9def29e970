// Fused fast path: get_featuregram (from Xin) + get_feature_patches for a batch of equal-length clips.
// Mirrors the call sequence of the reference's generators (Proposed_Work_Results.py:92-95, 465-474):
//   stft -> hpss medians -> soft masks -> mel -> power_to_db -> [featuregram]
//        -> tile-if-short -> StandardScaler per half -> extract_patches -> transpose to (N, W, F).
// Four launches, all on the caller's stream, no host synchronisation (hipGraph-capturable).
#include "smh_common.h"
#include "smh_feat.h"

namespace {
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
}  // namespace

extern "C" int smh_features_ex_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc,
                                   int harm_layout, int B, int T, int W, int shift, float *d_fv, float *d_patches,
                                   int32_t *d_maxkeys, void *stream) {
    SMH_REQUIRE(ctx && d_S && d_harm && d_perc && d_fv && d_maxkeys, "smh_features_f32: null argument");
    SMH_REQUIRE(B >= 0 && B <= 65535 && T >= 1, "smh_features_f32: bad shape B=%d T=%d", B, T);
    SMH_REQUIRE(harm_layout == 0 || harm_layout == 1, "smh_features_f32: harm_layout must be 0 or 1");
    int nP = 0;
    if (d_patches) {
        SMH_REQUIRE(W >= 1 && shift >= 1, "smh_features_f32: bad patch geometry W=%d shift=%d", W, shift);
        nP = smh_num_patches(smh_tiled_frames(T, W), W, shift);
    }
    if (B == 0) return nP;
    hipStream_t st = (hipStream_t)stream;
    int rc = smh_feat::launch_hp_feat(ctx, d_S, d_harm, d_perc, harm_layout, B, T, d_fv, (int *)d_maxkeys, st);
    if (rc) return rc;
    // always run: it applies the top_db clip that completes the featuregram
    rc = smh_feat::launch_std_patch(ctx, d_fv, (const int *)d_maxkeys, B, T, W > 0 ? W : 1, shift > 0 ? shift : 1, nP,
                                    nP > 0 ? d_patches : nullptr, st);
    if (rc) return rc;
    return nP;
}

extern "C" int smh_features_l0_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc,
                                   int harm_layout, int B, int T, int W, int shift, float *d_fv, float *d_patches,
                                   const float *d_w0, float *d_x0p, int32_t *d_maxkeys, void *stream) {
    SMH_REQUIRE(ctx && d_S && d_harm && d_perc && d_fv && d_maxkeys && d_w0 && d_x0p, "smh_features_l0_f32: null argument");
    SMH_REQUIRE(B >= 0 && B <= 65535 && T >= 1, "smh_features_l0_f32: bad shape B=%d T=%d", B, T);
    SMH_REQUIRE(harm_layout == 0 || harm_layout == 1, "smh_features_l0_f32: harm_layout must be 0 or 1");
    SMH_REQUIRE(W >= 1 && shift >= 1, "smh_features_l0_f32: bad patch geometry W=%d shift=%d", W, shift);
    const int nP = smh_num_patches(smh_tiled_frames(T, W), W, shift);
    if (B == 0 || nP <= 0) return nP;
    hipStream_t st = (hipStream_t)stream;
    int rc = smh_feat::launch_hp_feat(ctx, d_S, d_harm, d_perc, harm_layout, B, T, d_fv, (int *)d_maxkeys, st);
    if (rc) return rc;
    rc = smh_feat::launch_std_patch(ctx, d_fv, (const int *)d_maxkeys, B, T, W, shift, nP, d_patches, st, d_w0, d_x0p);
    if (rc) return rc;
    return nP;
}

extern "C" int smh_features_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc, int B,
                                int T, int W, int shift, float *d_fv, float *d_patches, int32_t *d_maxkeys,
                                void *stream) {
    return smh_features_ex_f32(ctx, d_S, d_harm, d_perc, 0, B, T, W, shift, d_fv, d_patches, d_maxkeys, stream);
}

extern "C" size_t smh_frontend_workspace_bytes(const smh_ctx *ctx, int B, int n_samples) {
    if (!ctx || B < 0) return 0;
    const int T = smh_num_frames(n_samples, ctx->cfg.n_fft, ctx->cfg.hop);
    if (T < 1) return 0;
    const size_t spec = align_up((size_t)B * ctx->K * T * sizeof(float), 256);
    return 3 * spec + align_up((size_t)2 * B * sizeof(int), 256);
}

extern "C" int smh_frontend_f32(const smh_ctx *ctx, const float *d_audio, int B, int n_samples, int W, int shift,
                                float *d_fv, float *d_patches, void *d_work, size_t work_bytes, float *d_S,
                                float *d_harm, float *d_perc, void *stream) {
    SMH_REQUIRE(ctx && d_audio && d_fv && d_work, "smh_frontend_f32: null argument");
    SMH_REQUIRE(B >= 0 && B <= 65535, "smh_frontend_f32: B=%d out of range", B);
    const int T = smh_num_frames(n_samples, ctx->cfg.n_fft, ctx->cfg.hop);
    SMH_REQUIRE(T >= 1, "smh_frontend_f32: clip of %d samples is shorter than n_fft=%d", n_samples, ctx->cfg.n_fft);
    if (work_bytes < smh_frontend_workspace_bytes(ctx, B, n_samples))
        return smh::set_error(SMH_E_WORKSPACE, "smh_frontend_f32: workspace %zu < required %zu", work_bytes,
                              smh_frontend_workspace_bytes(ctx, B, n_samples));
    const size_t spec = align_up((size_t)B * ctx->K * T * sizeof(float), 256);
    char *w = (char *)d_work;
    float *S = d_S ? d_S : (float *)w;
    float *harm = d_harm ? d_harm : (float *)(w + spec);
    float *perc = d_perc ? d_perc : (float *)(w + 2 * spec);
    int32_t *maxkeys = (int32_t *)(w + 3 * spec);
    int nP = 0;
    if (d_patches) {
        SMH_REQUIRE(W >= 1 && shift >= 1, "smh_frontend_f32: bad patch geometry W=%d shift=%d", W, shift);
        nP = smh_num_patches(smh_tiled_frames(T, W), W, shift);
    }
    if (B == 0) return nP;
    hipStream_t st = (hipStream_t)stream;
    int rc = smh_stft_mag_f32(ctx, d_audio, B, n_samples, S, stream);
    if (rc) return rc;
    // the harmonic median is written time-major (coalesced stores) unless the caller taps it
    const int tm = smh_median::launch_hpss(S, B, ctx->K, T, ctx->cfg.l_harm, ctx->cfg.l_perc, harm, perc, d_harm ? 0 : 1, st);
    if (tm < 0) return tm;
    rc = smh_feat::launch_hp_feat(ctx, S, harm, perc, tm, B, T, d_fv, (int *)maxkeys, st);
    if (rc) return rc;
    rc = smh_feat::launch_std_patch(ctx, d_fv, (const int *)maxkeys, B, T, W > 0 ? W : 1, shift > 0 ? shift : 1, nP,
                                    nP > 0 ? d_patches : nullptr, st);
    if (rc) return rc;
    return nP;
}
