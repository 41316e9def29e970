// Internal helpers shared by the libsmh.so translation units (gfx950 only, no portability layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/smh.h"

namespace smh {

int set_error(int code, const char *fmt, ...);

#define SMH_CHECK_HIP(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return smh::set_error(SMH_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                                  __FILE__, __LINE__);                                               \
    } while (0)

#define SMH_REQUIRE(cond, ...)                                           \
    do {                                                                 \
        if (!(cond)) return smh::set_error(SMH_E_INVALID, __VA_ARGS__);  \
    } while (0)

inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(SMH_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return SMH_OK;
}

// Timing probes and experiment switches that make a launch's OUTPUTS INVALID (a phase skipped, a store or load left out):
// read only when SMH_ENABLE_PROBES=1 is set, and every affected launch says so on stderr.  Without that switch the variable is
// ignored (one notice per variable).  Selectors that keep results valid (kernel A/B choices such as SMH_TCN_SKEW) use getenv.
const char *probe_env(const char *name);
// Implementation variants that the measurements rejected (the 16-wave network schedule, the LDS copy of the layer-0 weights, the
// per-tap feature kernel as a forced choice) are only selectable in a lab build (python -m sm_hpss_mtl_amd.build --lab -> -DSMH_LAB):
// the production library ignores their switches and does not contain the instantiations they select.
#ifdef SMH_LAB
inline const char *lab_env(const char *name) { return getenv(name); }
#else
inline const char *lab_env(const char *) { return nullptr; }
#endif

constexpr int kMaxFftStages = 12;
constexpr int kLdsBytesPerCU = 160 * 1024;

}  // namespace smh

namespace smh_rag {
struct Staging;  // pinned host buffers of the ragged calls' descriptor uploads (smh_ragged.hip)
void destroy_staging(Staging *);
}  // namespace smh_rag

// Front-end context: immutable device tables + the config they were built for.
struct smh_ctx {
    smh_frontend_cfg cfg;
    int K;         // bins = 1 + n_fft/2
    int M;         // complex FFT length = n_fft/2 (real-input packing)
    int n_stages;  // Stockham stages
    int radix[smh::kMaxFftStages];
    int feat_rows;  // rows of one half of the featuregram
    // device tables
    float *d_window;   // (n_fft) periodic Hann, centre-padded
    float2 *d_twM;     // (M)   exp(-2 pi i j / M)
    float2 *d_tw2M;    // (M+1) exp(-2 pi i k / (2M))
    // mel filterbank, CSR by filter row
    int n_mels;
    int *d_mel_start;  // (n_mels) first bin with non-zero weight
    int *d_mel_count;  // (n_mels) number of taps
    int *d_mel_off;    // (n_mels) offset into d_mel_w
    float *d_mel_w;    // (nnz)
    int mel_nnz;
    int mel_max_taps;
    // bin-walk form of the fused feature kernel (smh_feat.hip): the feature rows are cut into segments of similar
    // bin counts; for every bin of a segment the plan holds how many finished filters to emit first and the weights
    // of the (at most four) pending ones -- {w0, w1, w2, w3, n_emit, -, -, -} per bin, 32 bytes
    static constexpr int kMaxFeatSegs = 8;
    int feat_walk_ok;
    int feat_pend;  // most filters pending at one bin of the walk (1..4)
    // two segmentations of the same rows: [0] four segments (two-kernel path, 8 waves per clip), [1] eight segments
    // (single-kernel paths: features_half_kernel, 8 waves per clip half; features_clip_kernel, 16 waves per clip)
    int feat_nseg[2];
    int feat_m0[2][kMaxFeatSegs], feat_m1[2][kMaxFeatSegs], feat_kbeg[2][kMaxFeatSegs], feat_kend[2][kMaxFeatSegs],
        feat_off[2][kMaxFeatSegs];
    float *d_feat_plan;
    std::vector<float> h_mel_dense;  // (n_mels, K) host copy
    // the one mutable part: staging buffers for the descriptor tables of ragged calls, created on first use, guarded by rag_mu
    mutable smh_rag::Staging *rag_staging = nullptr;
    mutable std::mutex rag_mu;
};
