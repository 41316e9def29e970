/* smh.h -- C ABI of libsmh.so: the MI355X (gfx950) implementation of the SM_HPSS_MTL hot path.
 *
 * The reference (mrinmoy-iitg/SM_HPSS_MTL) has NO C ABI of its own: its hot path is a Python call
 * surface over librosa / scipy / sklearn / a Cython module / Keras.  Each entry point below therefore
 * cites the reference Python call it replaces (paths under /root/reference).  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer named `d_*` is DEVICE memory owned by the caller (e.g. torch tensors); the
 *     library never allocates or frees caller buffers and keeps no global state besides the
 *     immutable tables owned by an explicit `smh_ctx` / `smh_model`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is stream-ordered,
 *     nothing synchronises, so every call is hipGraph-capturable;
 *   - spectrogram-like tensors are (B, rows, T) row-major float32 -- per clip exactly the (K, T)
 *     arrays librosa returns;
 *   - return value: 0 = ok, <0 = error (SMH_E_*); smh_last_error() gives a thread-local message.
 *   - no CPU fallback exists: without a HIP device every compute entry point fails with SMH_E_HIP.
 */
#ifndef SMH_H
#define SMH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMH_OK 0
#define SMH_E_INVALID (-1)   /* bad argument / unsupported size            */
#define SMH_E_HIP (-2)       /* HIP runtime error (launch, alloc, no GPU)  */
#define SMH_E_WORKSPACE (-3) /* caller workspace too small                 */
#define SMH_E_DEVICE (-4)    /* a kernel reported through the object's device error word that its outputs are not results */

#define SMH_MAX_MEDIAN 63 /* largest supported (odd) median window        */

typedef struct smh_ctx smh_ctx;     /* front-end constant tables (window, twiddles, mel CSR)   */
typedef struct smh_model smh_model; /* B3_MTL descriptor + packed device weights              */

/* PARAMS keys the front end reads (Proposed_Work_Results.py:726-728,758-773,800-801). */
typedef struct smh_frontend_cfg {
    int32_t n_fft;      /* PARAMS['n_fft'][Model]: 400 (Jang: 512)                          */
    int32_t win_length; /* int(Tw*fs/1000): 400                                             */
    int32_t hop;        /* int(Ts*fs/1000): 160                                             */
    int32_t n_mels;     /* PARAMS['n_mels'][Model]: 120; <=0 -> no mel projection (rows=K)  */
    int32_t l_harm;     /* PARAMS['l_harm'][Model]: 21 (median along frames)                */
    int32_t l_perc;     /* PARAMS['l_perc'][Model]: 11 (median along bins)                  */
    int32_t log_db;     /* 1 -> librosa.power_to_db(x**2) ('Log*' feature names)            */
    float mel_sr;       /* librosa default 22050 (the reference never passes sr)            */
} smh_frontend_cfg;

const char *smh_last_error(void);
int smh_version(void);
int smh_device_count(void); /* 0 when no HIP device is visible */

int smh_ctx_create(const smh_frontend_cfg *cfg, smh_ctx **out);
void smh_ctx_destroy(smh_ctx *ctx);
/* rows of one half (harmonic or percussive) of the featuregram: n_mels, or K = 1+n_fft/2 */
int smh_ctx_feat_rows(const smh_ctx *ctx);
/* copies the (n_mels, K) float32 mel basis to HOST memory (for inspection / tests) */
int smh_ctx_mel_basis(const smh_ctx *ctx, float *h_out);

/* ---- integer contracts (host, exact) ------------------------------------------------------- */
/* librosa util.frame, center=False: 1 + (N - n_fft)/hop, 0 if N < n_fft */
int smh_num_frames(int n_samples, int n_fft, int hop);
/* get_feature_patches tiling (lib/preprocessing.py:139-142): frames after `while T<=W: append` */
int smh_tiled_frames(int T, int W);
/* tools.extract_patches (lib/cython_impl/tools.pyx:24-29): len(range(W/2, T-W/2, shift)) */
int smh_num_patches(int T, int W, int shift);
/* start frame of patch p (tools.pyx:30-34) */
int smh_patch_start(int T, int W, int shift, int p);

/* ---- a1: np.abs(librosa.core.stft(y, n_fft, win_length, hop_length, center=False)) ----------
 * lib/preprocessing.py:407,417,429,439.  d_audio (B, n_samples) -> d_S (B, K, T).               */
int smh_stft_mag_f32(const smh_ctx *ctx, const float *d_audio, int B, int n_samples, float *d_S, void *stream);

/* ---- a2: the two median filters inside librosa.decompose.hpss (preprocessing.py:408,418,...) --
 * harm = median_filter(S, (1,l_harm), 'reflect');  perc = median_filter(S, (l_perc,1), 'reflect').
 * Bit-exact selection.  smh_hpss_median_f32 computes both in ONE launch ("the median kernel").   */
int smh_hpss_median_f32(const smh_ctx *ctx, const float *d_S, int B, int K, int T, int l_harm, int l_perc,
                        float *d_harm, float *d_perc, void *stream);
int smh_median_time_f32(const smh_ctx *ctx, const float *d_S, int B, int K, int T, int l_harm, float *d_harm,
                        void *stream);
int smh_median_freq_f32(const smh_ctx *ctx, const float *d_S, int B, int K, int T, int l_perc, float *d_perc,
                        void *stream);

/* Layout-aware variants used by the fused pipeline (no reference counterpart: the reference never
 * materialises harm on a device).  harm_layout 0 = (B,K,T) as above; 1 = (B,T,K) time-major, which lets
 * the harmonic lanes (one per bin) store coalesced -- about 30 us per 1024 clips faster on MI355X.
 * smh_hpss_median_ex_f32 returns the layout it actually wrote (it falls back to 0 for window pairs or
 * tiny axes outside the fused kernel table); pass that value on to smh_features_ex_f32.              */
int smh_hpss_median_ex_f32(const smh_ctx *ctx, const float *d_S, int B, int K, int T, int l_harm, int l_perc,
                           float *d_harm, float *d_perc, int harm_layout, void *stream);
/* the harmonic median alone (= smh_median_time_f32) in any of those layouts; returns the layout written */
int smh_median_time_ex_f32(const smh_ctx *ctx, const float *d_S, int B, int K, int T, int l_harm, float *d_harm,
                           int harm_layout, void *stream);

/* ---- a3: H = S*softmask(harm,perc), P = S*softmask(perc,harm); power=2, split_zeros=True ---- */
int smh_softmask_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc, size_t n,
                     float *d_H, float *d_P, void *stream);

/* ---- a4: librosa.feature.melspectrogram(S=X, n_mels=) = mel_basis(22050, n_fft) @ X ---------
 * d_X (B, K, T) -> d_Y (B, n_mels, T)  (preprocessing.py:409-410,419,421)                       */
int smh_mel_f32(const smh_ctx *ctx, const float *d_X, int B, int T, float *d_Y, void *stream);

/* ---- a5: librosa.core.power_to_db(X**2): 10*log10(max(1e-10, x^2)), then max(., max-80) with the
 * max taken over each of the `n_arrays` arrays of `elems` values (preprocessing.py:420,422)      */
int smh_power_to_db_sq_f32(const smh_ctx *ctx, const float *d_X, int n_arrays, int elems, float *d_Y, void *stream);

/* ---- a7(iii): StandardScaler per row over frames (preprocessing.py:211-214,221-224) ----------
 * d_X (n_rows, T) -> d_Y (n_rows, T): (x-mean)/std, population std, std==0 -> 1                  */
int smh_standardize_rows_f32(const smh_ctx *ctx, const float *d_X, int n_rows, int T, float *d_Y, void *stream);

/* ---- a8: tools.extract_patches incl. the tile-if-short rule of get_feature_patches -----------
 * d_FV (B, F, T) -> patches.  layout 0: (B*nP, F, W) as the reference returns; layout 1:
 * (B*nP, W, F) time-major = the TCN input after np.transpose (Proposed_Work_Results.py:235-236).
 * Frames are taken modulo T when T < W (the tiled featuregram).  Returns nP per clip (>=0).     */
int smh_extract_patches_f32(const smh_ctx *ctx, const float *d_FV, int B, int F, int T, int W, int shift, int layout,
                            float *d_out, void *stream);

/* Layouts of the harmonic median between smh_hpss_median_ex_f32 and the feature stage (harm_layout):
 *   0 = (B, K, T) as the reference returns it, 1 = (B, T, K), 2 = (B, ceil(T/16), K, 16) -- 16-frame blocks, written
 *   by the block-split median kernels (windows <= 21; any clip length, tile by tile) and read by the single-kernel feature
 *   path.  A harm buffer of smh_harm_buffer_floats(K, T) floats per clip has room for every layout;
 *   smh_features_blocked_ok tells whether the feature stage accepts layout 2 for clips of T frames (with_l0: together
 *   with the layer-0 fusion of smh_features_l0_f32).  smh_hpss_median_ex_f32 returns the layout it actually wrote.  */
size_t smh_harm_buffer_floats(int K, int T);
int smh_features_blocked_ok(const smh_ctx *ctx, int T, int with_l0);

/* ---- a3..a9 in two launches: (S, harm, perc) -> featuregram -> standardised time-major patches ------
 * soft masks + mel + power_to_db (preprocessing.py:418-424) then tile / StandardScaler / patches /
 * transpose (preprocessing.py:137-142,208-234; Proposed_Work_Results.py:235-236).
 * d_fv (B, 2*rows, T) out; d_patches (B*nP, W, 2*rows) out or NULL; d_maxkeys: 2*B int32 scratch.
 * Returns nP per clip.                                                                            */
int smh_features_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc, int B, int T,
                     int W, int shift, float *d_fv, float *d_patches, int32_t *d_maxkeys, void *stream);
/* same, with d_harm in the layout returned by smh_hpss_median_ex_f32 */
int smh_features_ex_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc, int harm_layout,
                        int B, int T, int W, int shift, float *d_fv, float *d_patches, int32_t *d_maxkeys, void *stream);

/* ---- fused fast path: get_featuregram (from Xin) + get_feature_patches for a batch of clips ---
 * d_audio (B, n_samples) -> d_fv (B, 2*rows, T)  [the featuregram, = get_featuregram's return]
 *                        -> d_patches (B*nP, W, 2*rows) time-major, standardised  [may be NULL]
 * d_work: smh_frontend_workspace_bytes() bytes of scratch.  Optional parity taps (may be NULL):
 * d_S, d_harm, d_perc (B, K, T).  Returns nP per clip.                                          */
size_t smh_frontend_workspace_bytes(const smh_ctx *ctx, int B, int n_samples);
int smh_frontend_f32(const smh_ctx *ctx, const float *d_audio, int B, int n_samples, int W, int shift, float *d_fv,
                     float *d_patches, void *d_work, size_t work_bytes, float *d_S, float *d_harm, float *d_perc,
                     void *stream);

/* ---- ragged batches: B clips of DIFFERENT lengths in one call (the reference's generators take whole files of any length
 * one at a time: Proposed_Work_Results.py:92-95, 131-134, 189-192, 465-474).  d_audio holds the clips at sample offsets
 * h_offsets[b] with h_lengths[b] samples each (HOST arrays; keep every offset a multiple of 4 samples = 16 bytes: a clip that
 * starts off an 8-byte boundary is processed alone, through smh_frontend_f32).  ONE LAUNCH PER STAGE for all clips: the lengths
 * become a descriptor table + (clip, tile) work lists, uploaded in one copy from a pinned slot of the context, and the STFT, the
 * medians and the feature kernels each run once over every clip (smh_ragged.hip); a clip gets bit for bit what smh_frontend_f32
 * gives it alone or in an equal-length batch.  Outputs are concatenated: clip b's featuregram (2*rows, T_b) starts at float
 * h_fv_off[b] of d_fv, its nP_b standardised time-major patches at patch h_patch_off[b] of d_patches ((W, 2*rows) each).
 * smh_frontend_ragged_sizes fills the per-clip tables (each may be NULL) and the workspace requirement (tables + S / harm / perc of
 * every clip, at most 8 GiB: beyond that, or with a smaller workspace than asked for, the call runs in sub-batches; never less than
 * the largest single clip needs); W <= 0: no patches.  d_work must start on a 16-byte boundary.  Stream-ordered on `stream`; not
 * capturable in a hipGraph (the table upload reads a staging buffer that the next call rewrites). */
int smh_frontend_ragged_sizes(const smh_ctx *ctx, const long long *h_offsets, const int *h_lengths, int B, int W, int shift,
                              long long *h_fv_off /* B+1 */, long long *h_patch_off /* B+1 */, int *h_T /* B */,
                              int *h_nP /* B */, size_t *work_bytes);
int smh_frontend_ragged_f32(const smh_ctx *ctx, const float *d_audio, const long long *h_offsets, const int *h_lengths, int B,
                            int W, int shift, float *d_fv, float *d_patches /* or NULL */, void *d_work, size_t work_bytes,
                            void *stream);

/* ---- 8f rank 1: load_and_preprocess_signal after the decode (lib/preprocessing.py:330-350) ------------------
 * Batched over B clips of N samples each, resident on the device.  Floating point: the mean is accumulated in
 * f64 in a fixed order (numpy: pairwise f32), everything else is the same f32 arithmetic as numpy.           */
/* bytes of device workspace for smh_normalize_f32 / for the silence functions (hop = frame shift in samples) */
size_t smh_normalize_workspace_bytes(int B, int N);
size_t smh_silence_workspace_bytes(int B, int N, int hop);
/* Xin -= mean(Xin); Xin /= max|Xin| per clip (preprocessing.py:332-333,348-349).  d_out may alias d_x.        */
int smh_normalize_f32(const float *d_x, int B, int N, float *d_out, void *d_work, size_t work_bytes, void *stream);
/* librosa.feature.rms(y=, frame_length=, hop_length=)[0] (preprocessing.py:338): center=True, reflect padding.
 * d_y (B, N) -> d_energy (B, nFrames); returns nFrames = 1 + (N + 2*(frame_length/2) - frame_length)/hop.      */
int smh_rms_f32(const float *d_y, int B, int N, int frame_length, int hop, float *d_energy, void *stream);
/* tools.removeSilence(Xin, nSamples, energy, nFrames, fs, Tw, Ts, alpha, beta) (lib/cython_impl/tools.pyx:42-134),
 * integer-exact given the energies: float threshold alpha*max(energy), scipy medfilt(marker, 5) with zero padding,
 * the literal run loop, removal only when at least two runs exceed beta seconds.  d_out (B, N) keeps the input
 * length: retained samples first, then the reference's tail of 1.0 (unchanged copy when nothing is removed).
 * Optional outputs (null to skip): d_sample_marker (B, N) u8, d_frame_marker (B, nFrames) i32, d_n_keep (B) i32 =
 * number of retained samples (N when nothing was removed).  d_out must not alias d_x.                         */
int smh_remove_silence_f32(const float *d_x, int B, int N, const float *d_energy, int nFrames, int fs, int Tw, int Ts,
                           double alpha, double beta, float *d_out, unsigned char *d_sample_marker, int *d_frame_marker,
                           int *d_n_keep, void *d_work, size_t work_bytes, void *stream);
/* preprocessing.py:332-349 in one call: normalise, rms(frameSize, frameShift), removeSilence (alpha=0.025,
 * beta=0.075), normalise again (tail of ones included, as in the reference).  The "< 0.1 s: duplicate" rule of
 * :343-346 depends on N alone and stays with the host caller.                                                  */
int smh_preprocess_signal_f32(const float *d_x, int B, int N, int fs, int Tw, int Ts, float *d_out, int *d_n_keep,
                              void *d_work, size_t work_bytes, void *stream);

/* ---- 8f rank 2: mix_signals(Xin_sp, Xin_mu, target_dB) (lib/preprocessing.py:297-325) for B pairs of clips ------
 * d_sp (B, N) speech, d_mu (B, N_mu) music (looped to N when shorter, cut when longer), d_target_db (B) SMR in dB ->
 * d_out (B, N): music scaled to the target speech-to-music ratio, both weighted so the factors sum to one, then the
 * reference's normalisation.  Energies are f64 ordered sums (numpy: pairwise f32).  Workspace:
 * smh_normalize_workspace_bytes(B, N).                                                                          */
int smh_mix_signals_f32(const float *d_sp, const float *d_mu, int B, int N, int N_mu, const float *d_target_db,
                        float *d_out, void *d_work, size_t work_bytes, void *stream);

/* ---- 8f rank 4: scipy.signal.medfilt(x, kernel_size) on B tracks of n values (zero padded, odd window <= 8191),
 * the smoothing of the frame-level probability track (DAFx12_Speech_Music_Detection_B3_MTL_v2.py:94-98, window 501).
 * Bit-exact selection.  d_y must not alias d_x.                                                                 */
int smh_medfilt1d_f32(const float *d_x, int B, int n, int kernel_size, float *d_y, void *stream);

/* ---- the other two functions of lib/cython_impl/tools.pyx (off by default on the hot path), float64 like the reference:
 * scale_data(FV, mean, stdev) :138-165 -> (FV - mean[f]) / (stdev[f] + 1e-10), FV (F, T);
 * get_data_statistics(FV, stat_type, axis) :169-215 for patches (N, F, T): stat 0 mean, 1 variance (population),
 * 2 skew, 3 kurtosis (scipy.stats, biased, Fisher); axis 0 reduces over the rows -> (N, T), axis 1 over the frames ->
 * (N, F).                                                                                                            */
int smh_scale_data_f64(const double *d_FV, int F, int T, const double *d_mean, const double *d_stdev, double *d_out,
                       void *stream);
int smh_data_statistics_f64(const double *d_FV, int N, int F, int T, int stat, int axis, double *d_out, void *stream);

/* ---- a10-a12: B3_MTL = get_Lemaire_MTL_model (lib/proposed_architectures.py:85-170, 25-80) ---- */
typedef struct smh_model_cfg {
    int32_t n_feat;     /* N_MELS argument = input_shape[1]: 240                       */
    int32_t patch_size; /* W: 68 / 99 / 249                                            */
    int32_t n_classes;  /* 3 (heads S,M,R[2],3C) or 5 (heads S,M,N,R[3],3C)            */
    int32_t nb_filters; /* 32                                                          */
    int32_t kernel_size;/* 3                                                           */
    int32_t nb_stacks;  /* 3                                                           */
    int32_t n_dilations;/* 8 -> dilations 1,2,...,128                                  */
    int32_t block_variant; /* residual block of the third-party `tcn.TCN` (lib/proposed_architectures.py:124-144), which the
                            * reference does not pin: 0 = keras-tcn 2.3.x (initial 1x1 conv; per block dilated conv -> relu ->
                            * channel-max normalisation ('norm_relu') -> dropout -> 1x1 conv -> + input; final relu) -- the API
                            * the reference's positional call binds under, the default, training and the fused bench path;
                            * 1 = keras-tcn >= 2.8 (no initial conv; per block two dilated convs, relu each, shortcut = input or
                            * a 1x1 'matching' conv, relu of the sum) -- inference forward only (smh_model_forward_f32).
                            * Canonical weight order for 1: per block [conv0 kernel (3,Cin,32), bias, conv1 kernel (3,32,32),
                            * bias, (first block: matching kernel (1,n_feat,32), bias)], then '3C' and the heads as for 0. */
} smh_model_cfg;
#define SMH_TCN_BLOCK_2_3 0
#define SMH_TCN_BLOCK_2_8 1

int smh_model_create(const smh_model_cfg *cfg, smh_model **out);
void smh_model_destroy(smh_model *m);
/* number of float32 parameters in canonical (Keras-layout) order, see DESIGN.md "weight order" */
size_t smh_model_num_params(const smh_model *m);
/* upload weights given as ONE flat host float32 array in canonical order */
int smh_model_set_weights(smh_model *m, const float *h_flat, size_t n, void *stream);
/* total width of the concatenated head outputs per patch: 3-class: 1+1+2+3 = 7; 5-class: 11 */
int smh_model_out_dim(const smh_model *m);
/* inference forward.  d_x (N, W, n_feat) float32 time-major -> d_out (N, out_dim) float32 holding
 * [S | M | (N) | R | 3C-softmax] per row, i.e. model.predict's list concatenated on axis 1
 * (Proposed_Work_Results.py:520,586).  d_trunk (N, W, nb_filters) optional TCN output tap.      */
int smh_model_forward_f32(const smh_model *m, const float *d_x, int N, float *d_out, float *d_trunk, void *stream);
/* Error contract of the stream-ordered forwards (SURVEY 8(b) "Errors": the reference raises; a launch cannot).  Every
 * smh_model_forward_* (and the training forward inside smh_train_step_f32) only enqueues work, so a condition a kernel meets on the
 * device -- today one: a wave whose dependency (a tile flag of the barrier-free block schedule, the partner half of a split tile of
 * the barrier schedule) did not arrive within its bounded spin -- is recorded in the model's device error word; the affected
 * outputs are NOT results (zero-filled by the barrier-free schedule, computed from a stale exchange by the split tile).
 * smh_model_status waits for `stream`, returns SMH_OK or SMH_E_DEVICE (+ smh_last_error) and clears the word.  The Python side calls
 * it wherever results leave the device or decide something: `predict`, the device `evaluate`, `patch_probabilities`, `train_on_batch`
 * (sync=True) and once per epoch in `fit`. */
int smh_model_status(smh_model *m, void *stream);

/* download the (device-resident, possibly trained) weights in canonical order */
/* Fusion of the network's first layer into the feature stage (bench fast path; same logits within f32 tolerance):
 * smh_features_l0_f32 = smh_features_ex_f32 that also emits, per clip half, its share of B3_MTL's initial Conv1D(32, 1)
 * on the standardised patches, d_x0p (B*nP, 2, W, 32) float32, from the canonical kernel d_w0 = smh_model_w0_ptr(model)
 * ((n_feat, 32), n_feat = 2 * feat_rows); d_patches may be null.  smh_model_forward_x0_f32 = smh_model_forward_f32
 * that starts from those partials (adds them and the bias) instead of reading (N, W, n_feat) patches.              */
const float *smh_model_w0_ptr(const smh_model *m);
int smh_features_l0_f32(const smh_ctx *ctx, const float *d_S, const float *d_harm, const float *d_perc, int harm_layout,
                        int B, int T, int W, int shift, float *d_fv, float *d_patches, const float *d_w0, float *d_x0p,
                        int32_t *d_maxkeys, void *stream);
int smh_model_forward_x0_f32(const smh_model *m, const float *d_x0p, int N, float *d_out, float *d_trunk, void *stream);
/* model.evaluate's arithmetic for one batch (Proposed_Work_Results.py:160-165 losses, 683-687 evaluate): from d_out (N, out_dim) as
 * smh_model_forward_f32 wrote it and d_targets (N, out_dim) in the same column order [S | M | (N) | R | 3C], the batch means of the
 * inference-mode losses -- binary cross-entropy on the sigmoid heads, mean squared error on 'R', categorical cross-entropy on '3C', all
 * with Keras' 1e-7 clipping, float64 arithmetic --, the '3C' accuracy and the total loss sum_i h_loss_weights[i] * loss_i + l2_penalty
 * (h_loss_weights: n_heads + 1 host doubles in output order; l2_penalty: the heads' kernel-regulariser term, a function of the weights
 * only), each multiplied by `weight` and ADDED to d_sums[n_heads + 3] = [total | head losses in output order | 3C loss | 3C accuracy]
 * (float64, zeroed by the caller): a validation pass accumulates on the device and reads back once. */
int smh_model_eval_losses_f32(const smh_model *m, const float *d_out, const float *d_targets, int N, double weight,
                              const double *h_loss_weights, double l2_penalty, double *d_sums, void *stream);
/* Dense file-level inference: the per-batch body of `patch_probability_generator`,
 * DAFx12_Speech_Music_Detection_B3_MTL_v2.py:634-665 -- every hop-`shift` patch of a 10 000-frame batch of one file's featuregram
 * through model.predict.  d_fv (n_feat, Tc) float32: the batch as get_feature_patches leaves it in front of the patch extraction
 * (each half standardised over the batch, lib/preprocessing.py:208-224); patch p = frames [min(p*shift, Tc - W), + W), p <
 * smh_num_patches(Tc, W, shift) -- tools.extract_patches' grid (tools.pyx:24-34).  Returns the number of patches (>= 0) or an error;
 * d_out (nP, out_dim) as smh_model_forward_f32 writes it, to the same f32 tolerance.  Nothing of size nP x W x n_feat is built: the
 * first layer (a 1x1 convolution) is evaluated once per FRAME into d_work ((2, Tc, 32) float32 =
 * smh_model_dense_workspace_bytes) and every patch is read as a window of it.  Needs Tc >= W (shorter batches are tiled by
 * get_feature_patches: build the patches and call smh_model_forward_f32), n_feat a multiple of 8, block_variant 0. */
size_t smh_model_dense_workspace_bytes(const smh_model *m, int Tc);
int smh_model_forward_dense_f32(const smh_model *m, const float *d_fv, int Tc, int shift, void *d_work, size_t work_bytes,
                                float *d_out, void *stream);
/* Same forward with bf16 matrix-core operands and f32 accumulation / residual stream / normalisation (BASELINE config 5,
 * "mixed bf16 CNN + fp32 HPSS").  Weights are split once per weight version, activations right before each product.
 *   split = 1 (what smh_model_forward_bf16 runs): every operand is hi + lo (two bf16 values, 16 mantissa bits), every
 *     product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 -- three bf16 products per f32 product, still 5x fewer
 *     matrix-core cycles than the exact-f32 MFMA.  Outputs within 2e-2 of smh_model_forward_f32 (tests state the measured
 *     distance, ~1e-3), which SURVEY 8(d') asks of a bf16 variant.
 *   split = 0: operands rounded to one bf16.  Faster again, but 24 blocks of "divide by the channel maximum" amplify the
 *     8-bit operand rounding to 3.5-5e-2 at the outputs: outside the tolerance, kept for measurement only.
 * Neither is the parity path: smh_model_forward_f32 is. */
int smh_model_forward_bf16(smh_model *m, const float *d_x, int N, float *d_out, void *stream);
int smh_model_forward_bf16_ex(smh_model *m, const float *d_x, int N, float *d_out, int split, void *stream);
/* The same from the layer-0 partials of smh_features_l0_f32 (the bench fast path; layer 0 is then exact f32): d_x0p
 * (N, 2, patch_size, 32) float32 as for smh_model_forward_x0_f32. */
int smh_model_forward_x0_bf16(smh_model *m, const float *d_x0p, int N, float *d_out, int split, void *stream);
int smh_model_get_weights(const smh_model *m, float *h_flat, size_t n, void *stream);

/* ---- a13: Conv2D MTL baselines, inference forward (lib/proposed_architectures.py:425-511 Doukhan, :516-588
 * Papakostas, :650-764 Jang; heads :25-80).  Input: (N, in_h, in_w) float32 images = the (nP, 2F, W, 1) patches of
 * get_feature_patches for the non-Lemaire models (lib/preprocessing.py:216-217,226-227).  Output row =
 * [S | M | (N) | R | 3C], like smh_model_forward_f32.  Weights: one flat float32 vector, tensors in the order
 * reported by smh_cnn_tensor_info (Keras layouts: Conv2D (kh,kw,Cin,Cout), Dense (in,out), BN gamma/beta/mean/var). */
enum { SMH_CNN_DOUKHAN = 0, SMH_CNN_PAPAKOSTAS = 1, SMH_CNN_JANG = 2 };
typedef struct smh_cnn_cfg {
    int32_t kind;      /* SMH_CNN_* */
    int32_t in_h;      /* 2*n_mels (Doukhan), 2*(n_fft/2+1) (Papakostas, Jang) */
    int32_t in_w;      /* patch width W */
    int32_t n_classes; /* 3 or 5 */
    int32_t n_mels;    /* Jang: mel-scale kernels per half (0 = 120) */
    int32_t n_fft;     /* Jang: 0 = 512 */
    int32_t fc_width;  /* Papakostas: width of the two Dense layers (0 = 4096) */
    float fs;          /* Jang: sampling rate of the mel filter bank (0 = 16000) */
} smh_cnn_cfg;
typedef struct smh_cnn smh_cnn;
int smh_cnn_create(const smh_cnn_cfg *cfg, smh_cnn **out);
void smh_cnn_destroy(smh_cnn *m);
size_t smh_cnn_num_params(const smh_cnn *m);
int smh_cnn_out_dim(const smh_cnn *m);
int smh_cnn_feat_dim(const smh_cnn *m); /* width of the feature vector the heads read */
int smh_cnn_num_tensors(const smh_cnn *m);
/* name (<= name_cap-1 chars), shape (4 ints, unused dims 1), rank and offset (in floats) of parameter tensor i */
int smh_cnn_tensor_info(const smh_cnn *m, int i, char *name, int name_cap, int *shape4, int *ndim, size_t *offset);
int smh_cnn_set_weights(smh_cnn *m, const float *h_flat, size_t n, void *stream);
int smh_cnn_get_weights(const smh_cnn *m, float *h_flat, size_t n, void *stream);
size_t smh_cnn_workspace_bytes(const smh_cnn *m, int N);
/* d_x (N, in_h, in_w) -> d_out (N, out_dim); d_feat (N, feat_dim) optional (null to skip) */
int smh_cnn_forward_f32(const smh_cnn *m, const float *d_x, int N, float *d_out, float *d_feat, void *d_work,
                        size_t work_bytes, void *stream);
/* Mixed-precision variant ("second-priority MFMA target"): the Conv2D / Dense products take bf16
 * operands (v_mfma_f32_32x32x16_bf16; activations rounded while they are staged, kernels from a transposed bf16 copy
 * rebuilt after every weight change), f32 accumulation, epilogues, pooling, LRN, mel-scale layer and heads as in f32.
 * NOT the parity path: outputs differ from smh_cnn_forward_f32 by the bf16 rounding of the operands (tests state it). */
int smh_cnn_forward_bf16(const smh_cnn *m, const float *d_x, int N, float *d_out, float *d_feat, void *d_work,
                         size_t work_bytes, void *stream);

/* ---- a13 / a14: training step of the Conv2D MTL baselines (what model.fit runs per batch for the models compiled at
 * lib/proposed_architectures.py:499-506 / :572-580 / :750-757), all three kinds.  The trainer owns its activations,
 * gradients and optimiser state (sized for max_batch >= 2: BatchNorm needs a batch).
 *   d_x (N, H, W) images; d_y (N, out_dim) targets laid out like the forward output [S | M | (N) | R | 3C one-hot]
 *   d_drop: Dropout masks (0 or 1/(1-rate)) of the trunk, one (N, dim_i) block per Dropout layer in graph order
 *           (smh_cnn_trainer_num_dropouts / _dropout_info give dim_i and rate_i), or NULL (no dropout)
 *   d_drop_heads (N, n_heads, 16) Dropout(0.4) masks of the heads or NULL; h_loss_weights: n_heads + 1 host floats or NULL
 *   d_losses: n_heads + 4 floats out = [per-head losses..., 3C loss, weighted sum (without l2), 3C accuracy, l2 penalty]
 * smh_cnn_trainer_apply_f32: g = grad * grad_scale (+ l2 term); optimizer 0 = SGD (beta1 = momentum), 1 = Adam
 * (Keras: w -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)); BatchNorm moving statistics <- 0.99*old + 0.01*batch;
 * the inference epilogues of smh_cnn_forward_f32 are re-folded from the new weights.                              */
typedef struct smh_cnn_trainer smh_cnn_trainer;
int smh_cnn_trainer_create(smh_cnn *m, int max_batch, smh_cnn_trainer **out);
void smh_cnn_trainer_destroy(smh_cnn_trainer *t);
float *smh_cnn_trainer_grad_ptr(smh_cnn_trainer *t);
/* floats of the data-parallel bucket that starts at smh_cnn_trainer_grad_ptr: [gradient | BatchNorm batch statistics]
 * (all-reduce all of it: the moving statistics then follow the mean over the ranks) */
size_t smh_cnn_trainer_bucket_floats(const smh_cnn_trainer *t);
/* copy the optimiser state (Adam moments / momentum, step counter) of `src` into `dst` (same model): a trainer that has
 * to grow for a larger batch keeps its state (create the larger one, copy, destroy the old one) */
int smh_cnn_trainer_copy_state(smh_cnn_trainer *dst, const smh_cnn_trainer *src, void *stream);
int smh_cnn_trainer_num_dropouts(const smh_cnn_trainer *t);
int smh_cnn_trainer_dropout_info(const smh_cnn_trainer *t, int i, size_t *dim, float *rate);
int smh_cnn_train_step_f32(smh_cnn_trainer *t, const float *d_x, const float *d_y, int N, const float *d_drop,
                           const float *d_drop_heads, const float *h_loss_weights, float *d_losses, void *stream);
int smh_cnn_trainer_apply_f32(smh_cnn_trainer *t, int optimizer, float lr, float beta1, float beta2, float eps,
                              float grad_scale, void *stream);

/* ---- a15: the random draws of a training batch, one pass each (csrc/smh_rng.hip) ------------------------------------
 * Philox4x32-10 keyed by `seed`; `offset` selects an independent stream under the same seed (pass a per-call counter):
 * equal (seed, offset, n) give equal output, on any launch geometry.
 * smh_noise_augment_f32: d_out[i] = d_x[i] + N(0, scale) -- the generator's noise augmentation,
 *   Proposed_Work_Results.py:239-242 (np.random.normal(0, scale, shape) + np.add; the scale is drawn by the caller).
 *   d_out may be d_x (in place).  Both 16-byte aligned.
 * smh_dropout_masks_f32: d_out[0 .. n_a) = (u < keep_a ? 1 / keep_a : 0), d_out[n_a .. n_a + n_b) likewise with keep_b:
 *   the SpatialDropout1D masks (N, n_blocks, 32) and the heads' Dropout(0.4) masks (N, n_heads, 16) that
 *   smh_train_step_f32 takes, in one launch.                                                                          */
int smh_noise_augment_f32(const float *d_x, float *d_out, size_t n, float scale, unsigned long long seed,
                          unsigned long long offset, void *stream);
int smh_dropout_masks_f32(float *d_out, size_t n_a, float keep_a, size_t n_b, float keep_b, unsigned long long seed,
                          unsigned long long offset, void *stream);

/* ---- a14: one training step = what model.fit runs per batch (Proposed_Work_Results.py:298-307) for the
 * model compiled at lib/proposed_architectures.py:156-165: BCE (S, M[, N]) + MSE (R) + CCE (3C) with optional
 * loss_weights, l2(0.01) on the Dense(16) kernels, SGD(momentum, clipnorm, lr from ExponentialDecay).
 * The trainer owns activations, gradients (canonical order) and momentum; weights stay on the device.  */
typedef struct smh_trainer smh_trainer;
int smh_trainer_create(smh_model *m, int max_batch, smh_trainer **out);
void smh_trainer_destroy(smh_trainer *t);
/* device pointer to the flat gradient (smh_model_num_params floats, canonical order): all-reduce it over
 * RCCL between smh_train_step_f32 and smh_trainer_apply_sgd_f32 for data-parallel training. */
float *smh_trainer_grad_ptr(smh_trainer *t);
/* forward (training mode) + losses + backward for one batch.
 *   d_x (N, W, n_feat); d_y (N, out_dim) targets laid out like the forward output [S | M | (N) | R | 3C one-hot]
 *   d_drop_tcn  (N, n_blocks, 32) SpatialDropout1D masks (0 or 1/(1-rate)) or NULL (no dropout)
 *   d_drop_heads (N, n_heads, 16) Dropout(0.4) masks (0 or 1/0.6) or NULL
 *   h_loss_weights: n_heads + 1 host floats in output order (NULL = all 1)
 *   d_losses: 3 * n_heads + 4 floats out = [per-head losses..., 3C loss, weighted sum (without the l2 term), 3C accuracy,
 *             l2(0.01) penalty of the Dense(16) kernels = the term Keras adds to the reported total,
 *             that penalty per head (n_heads), binary accuracy (threshold 0.5) of every sigmoid head (n_heads)] */
int smh_train_step_f32(smh_trainer *t, const float *d_x, const float *d_y, int N, const float *d_drop_tcn,
                       const float *d_drop_heads, const float *h_loss_weights, float *d_losses, void *stream);
/* g = grad * grad_scale (+ l2 term); per-tensor clip to `clipnorm` (<= 0: off); v = momentum*v - lr*g; w += v;
 * BN moving statistics <- 0.99*old + 0.01*batch; operand buffers re-packed on the device.              */
int smh_trainer_apply_sgd_f32(smh_trainer *t, float lr, float momentum, float clipnorm, float grad_scale, void *stream);
/* floats of the data-parallel bucket that starts at smh_trainer_grad_ptr: [gradient (num_params) | BatchNorm batch
 * statistics of the heads]; all-reduce all of it so the moving statistics follow the mean over the ranks. */
size_t smh_trainer_bucket_floats(const smh_trainer *t);
/* copy momentum / Adam moments / step counters from `src` to `dst` (same model): growing a trainer keeps its state */
int smh_trainer_copy_state(smh_trainer *dst, const smh_trainer *src, void *stream);
/* zero the optimiser state (what compiling a Keras model with a new optimiser does) */
int smh_trainer_reset_state(smh_trainer *t, void *stream);
/* Deterministic weight gradients (off by default).  The backward kernels combine the workgroups' contributions with float
 * atomics, whose arrival order -- hence the last bits of a gradient -- changes from run to run.  on = 1: contributions are
 * rounded to a 2^-36 grid and summed in 64-bit integers (integer atomics: order-independent), so two runs of the same step give
 * bit-identical gradients, data-parallel replicas stay comparable run to run, and a result can be reproduced.  Costs one extra
 * pass over the gradient and 8-byte atomics (tools/bench_train.py --deterministic states the step time). */
int smh_trainer_set_deterministic(smh_trainer *t, int on, void *stream);
/* Arithmetic of the training step's matrix products (BASELINE config 5: "mixed bf16 CNN").  dtype 0 (default): exact-f32 MFMA.
 * dtype 1: the residual blocks' FORWARD and BACKWARD run on the bf16 matrix pipe with split operands -- every f32 operand as hi + lo
 * bf16, three bf16 products per f32 product, f32 accumulators; master weights, gates, losses, heads, the Dense gradients and the
 * optimiser stay f32.  The forward's outputs are within 1e-4 of the f32 forward's; the backward agrees with the f32 backward on the same
 * forward to 2e-4 (relative L2 per tensor); against the float64 oracle the gradients sit where the forward's 1e-5 puts the relu /
 * channel-maximum gates (2-3e-2 of a tensor's norm).  Patches longer than 128 frames keep the f32 backward. */
int smh_trainer_set_dtype(smh_trainer *t, int dtype);
/* General optimiser step.  optimizer 0 = SGD (beta1 = momentum; what smh_trainer_apply_sgd_f32 calls), 1 = Adam,
 * 2 = Nadam as tf.keras 2.x implements it (momentum schedule u_t = beta1 (1 - 0.5 * 0.96^(0.004 t)); the optimiser of the
 * single-head fine-tuning in DAFx12_Speech_Music_Detection_B3_MTL_v2.py:524-526).  active_mask selects the tensors that
 * belong to the (sub-)model being trained: bit 0 the TCN trunk, bit 1 the '3C' Dense, bit 2 + h head h (dense, BatchNorm
 * incl. moving statistics, output layer); tensors outside the mask are left untouched (Model(input,
 * get_layer('M').output) owns the trunk and head M only).  clipnorm <= 0: off.                                       */
#define SMH_TRAIN_TRUNK 1u
#define SMH_TRAIN_3C 2u
#define SMH_TRAIN_HEAD(h) (4u << (h))
#define SMH_TRAIN_ALL 0xFFFFFFFFu
int smh_trainer_apply_f32(smh_trainer *t, int optimizer, float lr, float beta1, float beta2, float eps, float clipnorm,
                          float grad_scale, unsigned active_mask, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SMH_H */
