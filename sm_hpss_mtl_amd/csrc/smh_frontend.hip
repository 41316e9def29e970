// Fused fast path: get_featuregram (from Xin) + get_feature_patches for a batch of equal-length clips.
// Mirrors the call sequence of the reference's generators (Proposed_Work_Results.py:92-95, 465-474):
//   stft -> hpss medians -> soft masks -> mel -> power_to_db -> [featuregram]
//        -> tile-if-short -> StandardScaler per half -> extract_patches -> transpose to (N, W, F).
// Four launches, all on the caller's stream, no host synchronisation (hipGraph-capturable).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "smh_common.h"
#include "smh_feat.h"
#include "smh_rag.h"

namespace {
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
}  // namespace

namespace smh_rag {  // smh_ragged.hip: the streaming kernels that serve clips beyond the LDS image
size_t equal_overhead_bytes(const smh_ctx *ctx, int B, int T);
int run_equal(const smh_ctx *ctx, const float *d_audio, int B, int n_samples, int T, int W, int shift, int nP, float *d_fv,
              float *d_patches, void *d_work, size_t work_bytes, bool stft_aligned8, hipStream_t st);
}  // namespace smh_rag

extern "C" int smh_features_ex_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc,
                                   int harm_layout, int B, int T, int W, int shift, float *d_fv, float *d_patches,
                                   int32_t *d_maxkeys, void *stream) {
    SMH_REQUIRE(ctx && d_S && d_harm && d_perc && d_fv && d_maxkeys, "smh_features_f32: null argument");
    SMH_REQUIRE(B >= 0 && B <= 65535 && T >= 1, "smh_features_f32: bad shape B=%d T=%d", B, T);
    SMH_REQUIRE(harm_layout >= 0 && harm_layout <= 2, "smh_features_f32: harm_layout must be 0, 1 or 2");
    int nP = 0;
    if (d_patches) {
        SMH_REQUIRE(W >= 1 && shift >= 1, "smh_features_f32: bad patch geometry W=%d shift=%d", W, shift);
        nP = smh_num_patches(smh_tiled_frames(T, W), W, shift);
    }
    if (B == 0) return nP;
    hipStream_t st = (hipStream_t)stream;
    if (harm_layout == 2) {
        int rc2 = smh_feat::launch_features_clip(ctx, d_S, d_harm, d_perc, B, T, W > 0 ? W : 1, shift > 0 ? shift : 1, nP, d_fv,
                                                 nP > 0 ? d_patches : nullptr, nullptr, nullptr, st);
        if (rc2 < 0) return rc2;
        SMH_REQUIRE(rc2 == 1, "smh_features_ex_f32: harm_layout 2 needs smh_features_blocked_ok(ctx, T=%d, 0)", T);
        return nP;
    }
    int rc = smh_feat::launch_hp_feat(ctx, d_S, d_harm, d_perc, harm_layout, B, T, d_fv, (int *)d_maxkeys, st);
    if (rc) return rc;
    // always run: it applies the top_db clip that completes the featuregram
    rc = smh_feat::launch_std_patch(ctx, d_fv, (const int *)d_maxkeys, B, T, W > 0 ? W : 1, shift > 0 ? shift : 1, nP,
                                    nP > 0 ? d_patches : nullptr, st);
    if (rc) return rc;
    return nP;
}

extern "C" int smh_features_blocked_ok(const smh_ctx *ctx, int T, int with_l0) {
    if (!ctx || T < 1) return 0;
    const int rows = ctx->feat_rows;
    if (!ctx->feat_walk_ok || smh::lab_env("SMH_FEAT_TAPS") || getenv("SMH_FEAT_TWO_KERNELS")) return 0;
    if (with_l0 && (rows % 4 != 0 || rows > 128)) return 0;
    size_t lds = sizeof(float) * ((size_t)2 * rows * (T | 1) + 3 * (size_t)2 * rows) + 128;
    if (with_l0) lds += sizeof(float) * 2 * rows * 32;  // as launch_features_clip
    return lds <= 158 * 1024 ? 1 : 0;
}

extern "C" size_t smh_harm_buffer_floats(int K, int T) {  // room for every harm layout of one clip
    if (K < 1 || T < 1) return 0;
    return (size_t)((T + 15) / 16) * 16 * K;
}

extern "C" int smh_features_l0_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc,
                                   int harm_layout, int B, int T, int W, int shift, float *d_fv, float *d_patches,
                                   const float *d_w0, float *d_x0p, int32_t *d_maxkeys, void *stream) {
    SMH_REQUIRE(ctx && d_S && d_harm && d_perc && d_fv && d_maxkeys && d_w0 && d_x0p, "smh_features_l0_f32: null argument");
    SMH_REQUIRE(B >= 0 && B <= 65535 && T >= 1, "smh_features_l0_f32: bad shape B=%d T=%d", B, T);
    SMH_REQUIRE(harm_layout >= 0 && harm_layout <= 2, "smh_features_l0_f32: harm_layout must be 0, 1 or 2");
    SMH_REQUIRE(W >= 1 && shift >= 1, "smh_features_l0_f32: bad patch geometry W=%d shift=%d", W, shift);
    const int nP = smh_num_patches(smh_tiled_frames(T, W), W, shift);
    if (B == 0 || nP <= 0) return nP;
    hipStream_t st = (hipStream_t)stream;
    if (harm_layout == 2) {
        int rc2 = smh_feat::launch_features_clip(ctx, d_S, d_harm, d_perc, B, T, W, shift, nP, d_fv, d_patches, d_w0, d_x0p, st);
        if (rc2 < 0) return rc2;
        SMH_REQUIRE(rc2 == 1, "smh_features_l0_f32: harm_layout 2 needs smh_features_blocked_ok(ctx, T=%d, 1)", T);
        return nP;
    }
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
    const size_t hspec = align_up((size_t)B * smh_harm_buffer_floats(ctx->K, T) * sizeof(float), 256);
    size_t total = 2 * spec + hspec + align_up((size_t)2 * B * sizeof(int), 256);
    // clips beyond the LDS image take the length-array kernels of smh_ragged.hip: their tables and row statistics, and S / perc
    // clip by clip on 16-byte boundaries
    if (!smh_features_blocked_ok(ctx, T, 0)) total += smh_rag::equal_overhead_bytes(ctx, B, T) + (size_t)B * 32;
    return total;
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
    const size_t hspec = align_up((size_t)B * smh_harm_buffer_floats(ctx->K, T) * sizeof(float), 256);
    char *w = (char *)d_work;
    float *S = d_S ? d_S : (float *)w;
    float *perc = d_perc ? d_perc : (float *)(w + spec);
    float *harm = d_harm ? d_harm : (float *)(w + 2 * spec);
    int32_t *maxkeys = (int32_t *)(w + 2 * spec + hspec);
    int nP = 0;
    if (d_patches) {
        SMH_REQUIRE(W >= 1 && shift >= 1, "smh_frontend_f32: bad patch geometry W=%d shift=%d", W, shift);
        nP = smh_num_patches(smh_tiled_frames(T, W), W, shift);
    }
    if (B == 0) return nP;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (!d_S && !d_harm && !d_perc && !smh_features_blocked_ok(ctx, T, 0)) {
        // Clips beyond the LDS image (longer than ~1.6 s at 240 rows), no taps: the streaming kernels of the ragged front end, fed the
        // tables of B equal clips -- a file gets the same bits alone, in an equal-length batch and in a ragged one.  The STFT kernel is
        // the one smh_stft_mag_f32 would pick for this batch.
        const bool aligned8 = ((n_samples % 2) == 0 || B == 1) && (reinterpret_cast<uintptr_t>(d_audio) % 8) == 0;
        rc = smh_rag::run_equal(ctx, d_audio, B, n_samples, T, W, shift, nP, d_fv, nP > 0 ? d_patches : nullptr, d_work, work_bytes, aligned8, st);
        if (rc < 0) return rc;
        if (rc == 1) return nP;
    }
    rc = smh_stft_mag_f32(ctx, d_audio, B, n_samples, S, stream);
    if (rc) return rc;
    // the harmonic median is written time-major (coalesced stores) unless the caller taps it
    // the harmonic median is written in the layout its consumer reads best unless the caller taps it:
    // 16-frame blocks for the single-kernel feature path, time-major otherwise
    const int want = d_harm ? 0 : (smh_features_blocked_ok(ctx, T, 0) ? 2 : 1);
    const int tm = smh_median::launch_hpss(S, B, ctx->K, T, ctx->cfg.l_harm, ctx->cfg.l_perc, harm, perc, want, st);
    if (tm < 0) return tm;
    if (tm == 2) {
        rc = smh_feat::launch_features_clip(ctx, S, harm, perc, B, T, W > 0 ? W : 1, shift > 0 ? shift : 1, nP, d_fv,
                                            nP > 0 ? d_patches : nullptr, nullptr, nullptr, st);
        if (rc < 0) return rc;
        if (rc == 1) return nP;
        return smh::set_error(SMH_E_INVALID, "smh_frontend_f32: internal layout mismatch");
    }
    rc = smh_feat::launch_hp_feat(ctx, S, harm, perc, tm, B, T, d_fv, (int *)maxkeys, st);
    if (rc) return rc;
    // (the workspace's S and perc parts -- 2 * spec bytes at its start, also when the caller took the taps into buffers of its own --
    // are dead behind the feature kernel: the long-clip patch path standardises into them instead of allocating per call)
    rc = smh_feat::launch_std_patch(ctx, d_fv, (const int *)maxkeys, B, T, W > 0 ? W : 1, shift > 0 ? shift : 1, nP,
                                    nP > 0 ? d_patches : nullptr, st, nullptr, nullptr, (d_S || d_perc) ? nullptr : w, 2 * spec);
    if (rc) return rc;
    return nP;
}
