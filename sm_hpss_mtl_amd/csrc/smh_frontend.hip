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

namespace {
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
}  // namespace

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
    if (!ctx->feat_walk_ok || getenv("SMH_FEAT_TAPS") || getenv("SMH_FEAT_TWO_KERNELS")) return 0;
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
    return 2 * spec + hspec + align_up((size_t)2 * B * sizeof(int), 256);
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
    int rc = smh_stft_mag_f32(ctx, d_audio, B, n_samples, S, stream);
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

// ---- ragged batches: clips of different lengths in one call (the reference's generators process whole files of any length
// one by one: Proposed_Work_Results.py:92-95, 131-134, 189-192, 465-474) -------------------------------------------------------
// Host-side orchestration over the batched kernels above: consecutive clips of the same length that lie back to back in the
// audio buffer are one launch set, every other clip is a launch set of its own, all on the caller's stream.  Kernel choice
// depends on (length, pointer alignment) exactly as for smh_frontend_f32, so with 8-byte aligned clip starts every clip gets
// bit for bit what smh_frontend_f32 gives it alone or inside an equal-length batch.
namespace {
struct RaggedPlan {
    std::vector<int> T, nP;
    std::vector<long long> fv_off, patch_off;  // floats / patches in front of clip b
    size_t work = 0;
};
int plan_ragged(const smh_ctx *ctx, const long long *off, const int *len, int B, int W, int shift, bool patches, RaggedPlan &p) {
    p.T.assign(B, 0), p.nP.assign(B, 0), p.fv_off.assign(B + 1, 0), p.patch_off.assign(B + 1, 0);
    const int rows2 = 2 * ctx->feat_rows;
    for (int b = 0; b < B; ++b) {
        SMH_REQUIRE(off[b] >= 0 && len[b] >= 0, "ragged: clip %d has a negative offset or length", b);
        const int T = smh_num_frames(len[b], ctx->cfg.n_fft, ctx->cfg.hop);
        SMH_REQUIRE(T >= 1, "ragged: clip %d of %d samples is shorter than n_fft=%d", b, len[b], ctx->cfg.n_fft);
        p.T[b] = T;
        p.nP[b] = patches ? smh_num_patches(smh_tiled_frames(T, W), W, shift) : 0;
        p.fv_off[b + 1] = p.fv_off[b] + (long long)rows2 * T;
        p.patch_off[b + 1] = p.patch_off[b] + p.nP[b];
    }
    return SMH_OK;
}
// Lanes of a ragged call: a file's kernels form a dependent chain of small (B = 1) launches that leaves most of the chip idle, and
// the chains of different files have nothing to do with each other -- so consecutive launch sets go round-robin to kRaggedLanes
// streams of the library's own (forked from the caller's stream by an event at entry, joined back into it by one event per lane
// at exit: the call stays stream-ordered for the caller, and capturable), each with its own slice of the workspace.
// SMH_RAGGED_STREAMS=1 runs everything on the caller's stream as before.
constexpr int kRaggedLanes = 16;  // upper bound; ragged_lanes() is what a call uses
constexpr int kRaggedDefault = 16;  // 1 / 2 / 4 / 8 / 16 lanes, 256 files of 1-10 s: 21.1 / 11.9 / 12.9 / 9.6 / 8.0 ms per call (profiles/r03_ragged.txt)
constexpr size_t kRaggedWorkCap = (size_t)4 << 30;  // ... but no more lanes than slices fit in 4 GiB (hour-long files: one lane)
int ragged_lanes() {
    int n = kRaggedDefault;
    if (const char *ev = getenv("SMH_RAGGED_STREAMS")) n = std::max(1, std::min(atoi(ev), kRaggedLanes));
    return n;
}
struct LanePool {
    hipStream_t st[kRaggedLanes] = {};
    hipEvent_t fork = nullptr, join[kRaggedLanes] = {};
    int device = -1;
};
// one pool per host thread and device (a call forks and joins inside itself: pools are never shared between concurrent calls)
constexpr int kMaxDevices = 16;
int lane_pool(LanePool **out) {
    thread_local LanePool pools[kMaxDevices];
    int dev = 0;
    SMH_CHECK_HIP(hipGetDevice(&dev));
    SMH_REQUIRE(dev >= 0 && dev < kMaxDevices, "smh_frontend_ragged_f32: device index %d", dev);
    LanePool &pool = pools[dev];
    if (pool.device != dev) {
        for (int i = 0; i < kRaggedLanes; ++i) {
            SMH_CHECK_HIP(hipStreamCreateWithFlags(&pool.st[i], hipStreamNonBlocking));
            SMH_CHECK_HIP(hipEventCreateWithFlags(&pool.join[i], hipEventDisableTiming));
        }
        SMH_CHECK_HIP(hipEventCreateWithFlags(&pool.fork, hipEventDisableTiming));
        pool.device = dev;
    }
    *out = &pool;
    return SMH_OK;
}

// clips b .. e-1 form one launch set: same length, back to back
int run_end(const long long *off, const int *len, int B, int b) {
    int e = b + 1;
    while (e < B && len[e] == len[b] && off[e] == off[e - 1] + len[b] && e - b < 65535) ++e;
    return e;
}
}  // namespace

extern "C" int smh_frontend_ragged_sizes(const smh_ctx *ctx, const long long *h_offsets, const int *h_lengths, int B, int W,
                                         int shift, long long *h_fv_off, long long *h_patch_off, int *h_T, int *h_nP,
                                         size_t *work_bytes) {
    SMH_REQUIRE(ctx && (B == 0 || (h_offsets && h_lengths)) && B >= 0, "smh_frontend_ragged_sizes: bad argument");
    SMH_REQUIRE(W <= 0 || shift >= 1, "smh_frontend_ragged_sizes: bad patch geometry W=%d shift=%d", W, shift);
    RaggedPlan p;
    int rc = plan_ragged(ctx, h_offsets, h_lengths, B, W, shift, W > 0, p);
    if (rc) return rc;
    size_t work = 0;
    for (int b = 0; b < B;) {
        const int e = run_end(h_offsets, h_lengths, B, b);
        work = std::max(work, smh_frontend_workspace_bytes(ctx, e - b, h_lengths[b]));
        b = e;
    }
    for (int b = 0; b <= B; ++b) {
        if (h_fv_off) h_fv_off[b] = p.fv_off[b];
        if (h_patch_off) h_patch_off[b] = p.patch_off[b];
    }
    for (int b = 0; b < B; ++b) {
        if (h_T) h_T[b] = p.T[b];
        if (h_nP) h_nP[b] = p.nP[b];
    }
    if (work_bytes) {  // one slice per lane (see ragged_lanes), within kRaggedWorkCap
        const size_t slice = align_up(work, 256);
        size_t lanes = (size_t)ragged_lanes();
        if (slice) lanes = std::max<size_t>(1, std::min(lanes, kRaggedWorkCap / slice));
        *work_bytes = slice * lanes;
    }
    return SMH_OK;
}

extern "C" int smh_frontend_ragged_f32(const smh_ctx *ctx, const float *d_audio, const long long *h_offsets,
                                       const int *h_lengths, int B, int W, int shift, float *d_fv, float *d_patches,
                                       void *d_work, size_t work_bytes, void *stream) {
    SMH_REQUIRE(ctx && d_audio && d_fv && d_work && h_offsets && h_lengths && B >= 0, "smh_frontend_ragged_f32: bad argument");
    const bool patches = d_patches != nullptr;
    SMH_REQUIRE(!patches || (W >= 1 && shift >= 1), "smh_frontend_ragged_f32: bad patch geometry W=%d shift=%d", W, shift);
    RaggedPlan p;
    int rc = plan_ragged(ctx, h_offsets, h_lengths, B, W, shift, patches, p);
    if (rc) return rc;
    const size_t prow = (size_t)W * 2 * ctx->feat_rows;
    // workspace of the largest launch set = one slice; as many lanes as whole slices fit (ragged_sizes asks for ragged_lanes() of them)
    size_t slice = 0;
    int n_sets = 0;
    for (int b = 0; b < B; ++n_sets) {
        const int e = run_end(h_offsets, h_lengths, B, b);
        slice = std::max(slice, smh_frontend_workspace_bytes(ctx, e - b, h_lengths[b]));
        b = e;
    }
    slice = align_up(slice, 256);
    int lanes = slice ? (int)std::min<size_t>((size_t)ragged_lanes(), work_bytes / slice) : 1;
    if (lanes > n_sets) lanes = n_sets;
    if (lanes < 1) lanes = 1;  // (a workspace below one slice is reported by smh_frontend_f32)
    hipStream_t caller = (hipStream_t)stream;
    LanePool *pool = nullptr;
    if (lanes > 1) {
        rc = lane_pool(&pool);
        if (rc) return rc;
        SMH_CHECK_HIP(hipEventRecord(pool->fork, caller));
        for (int i = 0; i < lanes; ++i) SMH_CHECK_HIP(hipStreamWaitEvent(pool->st[i], pool->fork, 0));
    }
    int set = 0, err = SMH_OK;
    for (int b = 0; b < B; ++set) {
        const int e = run_end(h_offsets, h_lengths, B, b);
        float *pt = patches && p.nP[b] > 0 ? d_patches + (size_t)p.patch_off[b] * prow : nullptr;
        const int lane = lanes > 1 ? set % lanes : 0;
        rc = smh_frontend_f32(ctx, d_audio + h_offsets[b], e - b, h_lengths[b], pt ? W : 0, pt ? shift : 0, d_fv + p.fv_off[b], pt,
                              (char *)d_work + (lanes > 1 ? (size_t)lane * slice : 0), lanes > 1 ? slice : work_bytes, nullptr, nullptr,
                              nullptr, lanes > 1 ? (void *)pool->st[lane] : stream);
        if (rc < 0) {
            err = rc;
            break;
        }
        b = e;
    }
    if (lanes > 1)  // join also on an error: the caller's stream must not run ahead of what was enqueued
        for (int i = 0; i < lanes; ++i) {
            SMH_CHECK_HIP(hipEventRecord(pool->join[i], pool->st[i]));
            SMH_CHECK_HIP(hipStreamWaitEvent(caller, pool->join[i], 0));
        }
    return err;
}
