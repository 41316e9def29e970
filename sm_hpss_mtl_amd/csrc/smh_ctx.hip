// Context (constant tables), error reporting and the integer contracts of the hot path.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "smh_common.h"

namespace smh {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *probe_env(const char *name) {
    const char *v = getenv(name);
    if (!v) return nullptr;
    const char *sw = getenv("SMH_ENABLE_PROBES");
    if (!sw || atoi(sw) != 1) {
        static thread_local char noticed[16][48];
        static thread_local int n_noticed = 0;
        for (int i = 0; i < n_noticed; ++i)
            if (!strncmp(noticed[i], name, 47)) return nullptr;
        if (n_noticed < 16) strncpy(noticed[n_noticed++], name, 47);
        fprintf(stderr, "libsmh: %s is a probe that invalidates outputs; ignored (set SMH_ENABLE_PROBES=1 to use it)\n", name);
        return nullptr;
    }
    fprintf(stderr, "libsmh: PROBE %s=%s active -- this launch's outputs are NOT results\n", name, v);
    return v;
}

}  // namespace smh

extern "C" const char *smh_last_error(void) { return smh::g_err; }
extern "C" int smh_version(void) { return 100; }
extern "C" int smh_internal_lab(void) {  // 1: built with -DSMH_LAB (the rejected variants are compiled in and selectable)
#ifdef SMH_LAB
    return 1;
#else
    return 0;
#endif
}

extern "C" int smh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---- integer contracts -------------------------------------------------------------------------
// librosa util.frame(center=False)
extern "C" int smh_num_frames(int n_samples, int n_fft, int hop) {
    if (n_fft <= 0 || hop <= 0 || n_samples < n_fft) return 0;
    return 1 + (n_samples - n_fft) / hop;
}

// lib/preprocessing.py:139-142: if T < W: while T <= W: FV = [FV, FV1]
extern "C" int smh_tiled_frames(int T, int W) {
    if (T <= 0) return 0;
    int Tt = T;
    if (Tt < W)
        while (Tt <= W) Tt += T;
    return Tt;
}

// lib/cython_impl/tools.pyx:24-25: len(range(int(W/2), T - int(W/2), shift))
extern "C" int smh_num_patches(int T, int W, int shift) {
    if (shift <= 0 || W <= 0) return -1;
    int half = W / 2;
    int lo = half, hi = T - half;
    if (hi <= lo) return 0;
    return (hi - lo + shift - 1) / shift;
}

// lib/cython_impl/tools.pyx:29-34
extern "C" int smh_patch_start(int T, int W, int shift, int p) {
    int half = W / 2;
    int i = half + p * shift;
    int s = i - half;
    int e = s + W < T ? s + W : T;
    if (e - s < W) s = e - W;
    return s;
}

// ---- mel filterbank: librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, 'slaney')
static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

static void build_mel_dense(double sr, int n_fft, int n_mels, std::vector<float> &W) {
    const int K = 1 + n_fft / 2;
    const double fmax = sr / 2;
    std::vector<double> fftfreqs(K), mel_f(n_mels + 2);
    // np.linspace(0, sr/2, K): step*i, last point exact
    for (int k = 0; k < K; ++k) fftfreqs[k] = (k == K - 1) ? fmax : (fmax / (K - 1)) * k;
    const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i) {
        double m = (i == n_mels + 1) ? m1 : m0 + ((m1 - m0) / (n_mels + 1)) * i;
        mel_f[i] = mel_to_hz(m);
    }
    W.assign((size_t)n_mels * K, 0.f);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < K; ++k) {
            const double lower = -(mel_f[i] - fftfreqs[k]) / fd0;
            const double upper = (mel_f[i + 2] - fftfreqs[k]) / fd1;
            const double tri = std::fmax(0.0, std::fmin(lower, upper));
            const float tri32 = (float)tri;                       // weights[i] = ... (float32 array)
            W[(size_t)i * K + k] = (float)((double)tri32 * enorm);  // weights *= enorm (f64 math, f32 store)
        }
    }
}

static int factor_radices(int M, int *radix) {
    int n = 0;
    const int pref[] = {8, 4, 2, 5, 3, 7};  // radices smh_stft.hip implements
    for (int p : pref) {
        while (M % p == 0 && M > 1) {
            if (n >= smh::kMaxFftStages) return -1;
            radix[n++] = p;
            M /= p;
        }
    }
    return M == 1 ? n : -1;
}

template <typename T>
static hipError_t upload(T **d, const std::vector<T> &h) {
    *d = nullptr;
    if (h.empty()) return hipSuccess;
    hipError_t e = hipMalloc((void **)d, h.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

extern "C" int smh_ctx_create(const smh_frontend_cfg *cfg, smh_ctx **out) {
    SMH_REQUIRE(cfg && out, "smh_ctx_create: null argument");
    SMH_REQUIRE(cfg->n_fft >= 8 && cfg->n_fft % 2 == 0, "n_fft must be even and >= 8 (got %d)", cfg->n_fft);
    SMH_REQUIRE(cfg->win_length >= 1 && cfg->win_length <= cfg->n_fft, "win_length must be in [1, n_fft]");
    SMH_REQUIRE(cfg->hop >= 1, "hop must be >= 1");
    SMH_REQUIRE(cfg->l_harm >= 1 && cfg->l_harm <= SMH_MAX_MEDIAN && (cfg->l_harm & 1), "l_harm must be odd in [1,%d]",
                SMH_MAX_MEDIAN);
    SMH_REQUIRE(cfg->l_perc >= 1 && cfg->l_perc <= SMH_MAX_MEDIAN && (cfg->l_perc & 1), "l_perc must be odd in [1,%d]",
                SMH_MAX_MEDIAN);
    SMH_REQUIRE(smh_device_count() > 0, "no HIP device visible: libsmh has no CPU path");

    smh_ctx *c = new smh_ctx();
    c->cfg = *cfg;
    if (c->cfg.mel_sr <= 0) c->cfg.mel_sr = 22050.f;
    c->K = 1 + cfg->n_fft / 2;
    c->M = cfg->n_fft / 2;
    c->n_stages = factor_radices(c->M, c->radix);
    if (c->n_stages < 0) {
        delete c;
        return smh::set_error(SMH_E_INVALID, "n_fft/2 = %d has a prime factor outside {2,3,5,7}", cfg->n_fft / 2);
    }
    c->n_mels = cfg->n_mels > 0 ? cfg->n_mels : 0;
    c->feat_rows = c->n_mels > 0 ? c->n_mels : c->K;

    // periodic Hann, zero-padded centred to n_fft (librosa get_window + pad_center)
    std::vector<float> win(cfg->n_fft, 0.f);
    const int lpad = (cfg->n_fft - cfg->win_length) / 2;
    for (int n = 0; n < cfg->win_length; ++n)
        win[lpad + n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / cfg->win_length));
    std::vector<float2> twM(c->M), tw2M(c->M + 1);
    for (int j = 0; j < c->M; ++j) {
        const double a = -2.0 * M_PI * j / c->M;
        twM[j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int k = 0; k <= c->M; ++k) {
        const double a = -2.0 * M_PI * k / (2.0 * c->M);
        tw2M[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    std::vector<int> mstart, mcount, moff;
    std::vector<float> mw;
    c->mel_max_taps = 0;
    if (c->n_mels > 0) {
        build_mel_dense(c->cfg.mel_sr, cfg->n_fft, c->n_mels, c->h_mel_dense);
        for (int i = 0; i < c->n_mels; ++i) {
            int first = -1, last = -2;
            for (int k = 0; k < c->K; ++k)
                if (c->h_mel_dense[(size_t)i * c->K + k] != 0.f) {
                    if (first < 0) first = k;
                    last = k;
                }
            if (first < 0) first = 0, last = -1;
            mstart.push_back(first);
            mcount.push_back(last - first + 1);
            moff.push_back((int)mw.size());
            for (int k = first; k <= last; ++k) mw.push_back(c->h_mel_dense[(size_t)i * c->K + k]);
            if (last - first + 1 > c->mel_max_taps) c->mel_max_taps = last - first + 1;
        }
        if (mw.empty()) mw.push_back(0.f);
    }
    c->mel_nnz = (int)mw.size();
    std::vector<float> plan;
    {   // plans of the bin-walk feature kernels
        const int R = c->feat_rows;
        std::vector<int> st(R), en(R);
        for (int i = 0; i < R; ++i) {
            st[i] = c->n_mels > 0 ? mstart[i] : i;
            en[i] = st[i] + (c->n_mels > 0 ? mcount[i] : 1);
        }
        auto weight = [&](int m, int k) { return c->n_mels > 0 ? mw[moff[m] + (k - st[m])] : 1.0f; };
        bool ok = true;
        c->feat_pend = 1;
        for (int v = 0; v < 2; ++v) {
            int NS = v == 0 ? 4 : 8;  // 4 measured best of 3..8 for the two-kernel path at K = 201, T = 98
            if (const char *ev = getenv(v == 0 ? "SMH_FEAT_SEGS" : "SMH_FEAT_SEGS1"))
                NS = std::max(1, std::min(atoi(ev), (int)smh_ctx::kMaxFeatSegs));  // tuning
            int bound[smh_ctx::kMaxFeatSegs + 1];
            bound[0] = 0;
            for (int sgm = 1; sgm < NS; ++sgm) {  // boundaries where the filter start crosses sgm/NS of the bins
                int m = bound[sgm - 1];
                while (m < R && st[m] < (long)c->K * sgm / NS) ++m;
                bound[sgm] = m;
            }
            bound[NS] = R;
            c->feat_nseg[v] = 0;
            for (int sgm = 0; sgm < NS; ++sgm) {
                const int m0 = bound[sgm], m1 = bound[sgm + 1];
                if (m0 >= m1) continue;
                int kbeg = c->K, kend = 0;
                for (int m = m0; m < m1; ++m)
                    if (en[m] > st[m]) kbeg = std::min(kbeg, st[m]), kend = std::max(kend, en[m]);
                if (kend <= kbeg) kbeg = kend = 0;
                const int q = c->feat_nseg[v]++;
                c->feat_m0[v][q] = m0, c->feat_m1[v][q] = m1, c->feat_kbeg[v][q] = kbeg, c->feat_kend[v][q] = kend;
                c->feat_off[v][q] = (int)plan.size();
                int mcur = m0;
                for (int k = kbeg; k < kend; ++k) {
                    int nemit = 0;
                    while (mcur < m1 && k >= en[mcur]) ++mcur, ++nemit;
                    float w4[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int m = m0; m < m1; ++m)
                        if (st[m] <= k && k < en[m]) {
                            if (m < mcur || m > mcur + 3) ok = false;  // more than four pending filters, or out of order
                            else w4[m - mcur] = weight(m, k), c->feat_pend = std::max(c->feat_pend, m - mcur + 1);
                        }
                    for (int e = 0; e < 4; ++e) plan.push_back(w4[e]);
                    float ne;
                    std::memcpy(&ne, &nemit, sizeof(float));
                    plan.push_back(ne);
                    plan.push_back(0.f), plan.push_back(0.f), plan.push_back(0.f);
                }
            }
        }
        c->feat_walk_ok = ok ? 1 : 0;
        if (plan.empty()) plan.push_back(0.f);
    }

    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = upload(&c->d_window, win);
    if (e == hipSuccess) e = upload(&c->d_twM, twM);
    if (e == hipSuccess) e = upload(&c->d_tw2M, tw2M);
    if (e == hipSuccess) e = upload(&c->d_mel_start, mstart);
    if (e == hipSuccess) e = upload(&c->d_mel_count, mcount);
    if (e == hipSuccess) e = upload(&c->d_mel_off, moff);
    if (e == hipSuccess) e = upload(&c->d_mel_w, mw);
    if (e == hipSuccess) e = upload(&c->d_feat_plan, plan);
    if (e != hipSuccess) {
        smh_ctx_destroy(c);
        return smh::set_error(SMH_E_HIP, "smh_ctx_create: table upload failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return SMH_OK;
}

extern "C" void smh_ctx_destroy(smh_ctx *c) {
    if (!c) return;
    (void)hipFree(c->d_window);
    (void)hipFree(c->d_twM);
    (void)hipFree(c->d_tw2M);
    (void)hipFree(c->d_feat_plan);
    (void)hipFree(c->d_mel_start);
    (void)hipFree(c->d_mel_count);
    (void)hipFree(c->d_mel_off);
    (void)hipFree(c->d_mel_w);
    smh_rag::destroy_staging(c->rag_staging);
    delete c;
}

extern "C" int smh_ctx_feat_rows(const smh_ctx *c) { return c ? c->feat_rows : SMH_E_INVALID; }

extern "C" int smh_ctx_mel_basis(const smh_ctx *c, float *h_out) {
    SMH_REQUIRE(c && h_out, "smh_ctx_mel_basis: null argument");
    SMH_REQUIRE(c->n_mels > 0, "context has no mel filterbank (n_mels <= 0)");
    std::memcpy(h_out, c->h_mel_dense.data(), c->h_mel_dense.size() * sizeof(float));
    return SMH_OK;
}
