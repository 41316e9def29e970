// Length-array ("ragged") form of the front end: what the kernels read when clips of DIFFERENT lengths share one launch.
// The reference's callers feed one file at a time (Proposed_Work_Results.py:92-95, 131-134, 189-192, 465-474 -> get_featuregram,
// lib/preprocessing.py:355-457, and get_feature_patches, :137-292); here the files of a batch are one launch per stage: every
// kernel takes a per-clip descriptor table and a flat (clip, tile) work list built on the host by smh_frontend_ragged_f32
// (smh_ragged.hip) and uploaded once per call.
#pragma once
#include <cstdint>

#include "smh_common.h"

namespace smh_rag {

// one clip of a ragged call (64 bytes; offsets in floats from the call's base pointers)
struct Clip {
    long long audio_off;  // d_audio
    long long spec_off;   // workspace S and perc: (K, T) each, rows of exactly T floats as in the equal-length layout; multiple of 4
    long long harm_off;   // workspace harm: the 16-frame blocked image (ceil(T/16), K, 16); multiple of 4
    long long fv_off;     // d_fv: (2*rows, T)
    long long patch_off;  // d_patches: patches in front of this clip's
    int T;                // frames, 1 + (n_samples - n_fft) / hop
    int Ttiled;           // frames after tile-if-short (lib/preprocessing.py:139-142)
    int nP;               // patches (0: none asked for)
    int row0;             // first row of this clip in the per-row statistics table (long clips only)
    int pad_[2];
};
static_assert(sizeof(Clip) == 64, "smh_rag::Clip is read with 16-byte scalar loads");

// one workgroup's share of a stage: tile / chunk `tile` of clip `clip`
struct Item {
    int clip, tile;
};

}  // namespace smh_rag

namespace smh_stft {
// frames per STFT item: 20 for the specialised n_fft = 400 kernel (every clip on an 8-byte boundary), 16 for the generic one
int rag_frames(const smh_ctx *ctx, bool aligned8);
// |STFT| of every (clip, frame tile) item into the workspace
int launch_rag(const smh_ctx *ctx, const float *d_audio, float *d_S, const smh_rag::Clip *d_clips, const smh_rag::Item *d_items,
               int n_items, bool aligned8, hipStream_t st);
constexpr int kRagFrames = 20;  // (the equal-length path splits 98 frames into 5 x 20 as well)
}  // namespace smh_stft

namespace smh_median {
// frames per median item and the LDS row stride that goes with it (two workgroups per CU); 0 when (lh, lp) has no block-split kernel
int rag_tile_frames(int K, int lh, int lp, int *stride);
// both HPSS medians of every (clip, frame tile) item; harm in the 16-frame blocked layout
int launch_rag(const float *d_S, float *d_harm, float *d_perc, int K, int lh, int lp, const smh_rag::Clip *d_clips,
               const smh_rag::Item *d_items, int n_items, hipStream_t st);
}  // namespace smh_median

namespace smh_feat {
// clips whose featuregram fits an LDS image (smh_features_blocked_ok): the kernels of the equal-length path with per-clip shapes.
// list: n clip indices, all of even T (even_T != 0: features_half_kernel) or all of odd T (features_clip_kernel); max_T over them.
int launch_features_rag(const smh_ctx *c, const float *S, const float *harmb, const float *perc, const smh_rag::Clip *d_clips,
                        const int *d_list, int n, int max_T, int even_T, int W, int shift, float *fv, float *patches, hipStream_t st);
}  // namespace smh_feat
