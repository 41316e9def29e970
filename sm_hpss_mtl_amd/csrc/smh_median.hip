// Entry points of the HPSS median filters (kernel template: smh_median_kernel.h).
#include <cstdlib>

#include "smh_median_kernel.h"
#include "smh_median_split.h"

namespace smh_median {

const Entry kPairs[] = {
    // both filters in ONE launch: the reference's configuration (21,11), BASELINE config 2 (17,17)
    // and the reference's sweep {11,21,31,41,51}^2 (Hyperparameter_Selection.py:543-544)
    SMH_MEDIAN_E2(21, 11) SMH_MEDIAN_E2(17, 17)
    SMH_MEDIAN_E2(11, 11) SMH_MEDIAN_E2(11, 21) SMH_MEDIAN_E2(11, 31) SMH_MEDIAN_E2(11, 41) SMH_MEDIAN_E2(11, 51)
    SMH_MEDIAN_E2(21, 21) SMH_MEDIAN_E2(21, 31) SMH_MEDIAN_E2(21, 41) SMH_MEDIAN_E2(21, 51)
    SMH_MEDIAN_E2(31, 11) SMH_MEDIAN_E2(31, 21) SMH_MEDIAN_E2(31, 31) SMH_MEDIAN_E2(31, 41) SMH_MEDIAN_E2(31, 51)
    SMH_MEDIAN_E2(41, 11) SMH_MEDIAN_E2(41, 21) SMH_MEDIAN_E2(41, 31) SMH_MEDIAN_E2(41, 41) SMH_MEDIAN_E2(41, 51)
    SMH_MEDIAN_E2(51, 11) SMH_MEDIAN_E2(51, 21) SMH_MEDIAN_E2(51, 31) SMH_MEDIAN_E2(51, 41) SMH_MEDIAN_E2(51, 51)};

KernelFn find_pair_kernel(int lh, int lp) {
    for (const Entry &e : kPairs)
        if (e.lh == lh && e.lp == lp) return e.fn;
    return nullptr;
}

}  // namespace smh_median

namespace {

using namespace smh_median;

// Slow, fully general path for axes not longer than half the window (multiple reflections):
// one thread per output, rank-counting selection.  Only tiny inputs ever reach it.
__global__ void median_small_kernel(const float *__restrict__ S, float *__restrict__ out, int K, int T, int w,
                                    int along_t, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % T);
    const int k = (int)((i / T) % K);
    const size_t base = i - (size_t)k * T - t;
    const int n = along_t ? T : K;
    const int pos = along_t ? t : k;
    const int h = w / 2;
    auto at = [&](int j) {
        int q = pos - h + j;
        const int p = 2 * n;
        int r = q % p;
        if (r < 0) r += p;
        if (r >= n) r = p - 1 - r;
        return along_t ? S[base + (size_t)k * T + r] : S[base + (size_t)r * T + t];
    };
    // the median is the element with exactly h elements before it in the stable (value, index) order
    for (int a = 0; a < w; ++a) {
        const float va = at(a);
        int rank = 0;
        for (int c = 0; c < w; ++c) {
            const float vc = at(c);
            rank += (vc < va) || (vc == va && c < a);
        }
        if (rank == h) {
            out[i] = va;
            return;
        }
    }
}

struct Plan {
    int TT, ntiles, stride, nsh, nsp, nwh, nwp;
    size_t lds;
};

// Tile so that two workgroups fit one CU's 160 KiB LDS whenever the clip allows it.
int make_plan(int K, int T, int lh, int lp, Plan *p) {
    const int hh = lh / 2;
    const int budget_words = (smh::kLdsBytesPerCU / 2 - 1024) / 4;
    auto odd = [](int x) { return x | 1; };
    int TT, stride;
    if ((size_t)K * odd(T) <= (size_t)budget_words) {
        TT = T;
        stride = odd(T);
    } else {
        int maxcols = budget_words / K;
        if ((maxcols & 1) == 0) maxcols -= 1;
        TT = maxcols - 2 * hh;
        if (TT < 8) {
            // very tall spectrograms: give one workgroup the whole 160 KiB
            maxcols = ((smh::kLdsBytesPerCU - 1024) / 4) / K;
            if ((maxcols & 1) == 0) maxcols -= 1;
            TT = maxcols - 2 * hh;
            if (TT < 1) return smh::set_error(SMH_E_INVALID, "K=%d too large for an LDS tile with l_harm=%d", K, lh);
        }
        if (TT > T) TT = T;
        stride = odd(TT + 2 * hh);
    }
    p->TT = TT;
    p->ntiles = (T + TT - 1) / TT;
    p->stride = stride;
    p->lds = (size_t)K * stride * sizeof(float);
    // segments: aim at similar lane-task lengths and <= 16 waves per workgroup
    const int nt = TT;
    int nsh = lh ? 2 : 0, nsp = lp ? 3 : 0;
    // tuning override for experiments (tools/tune_median.py): SMH_MEDIAN_SEG="nsh,nsp"
    if (const char *ev = getenv("SMH_MEDIAN_SEG")) {
        int a = 0, b = 0;
        if (sscanf(ev, "%d,%d", &a, &b) == 2) {
            if (lh && a >= 1) nsh = a;
            if (lp && b >= 1) nsp = b;
        }
    }
    if (lh && nt < 2 * lh) nsh = 1;
    if (lp && K < 6 * lp) nsp = 1;
    auto waves = [&](int n, int seg) { return seg ? (n * seg + 63) / 64 : 0; };
    int nwh = waves(K, nsh), nwp = waves(nt, nsp);
    while (nwh + nwp > 16) {
        if (nsh > 1 && nwh >= nwp) nsh--;
        else if (nsp > 1) nsp--;
        else if (nsh > 1) nsh--;
        else break;
        nwh = waves(K, nsh), nwp = waves(nt, nsp);
    }
    if (nwh + nwp > 16)
        return smh::set_error(SMH_E_INVALID, "tile %dx%d needs more than 16 waves per workgroup", K, nt);
    p->nsh = nsh, p->nsp = nsp, p->nwh = nwh, p->nwp = nwp;
    return SMH_OK;
}

// Roles of the block-split kernel: per-lane cost ~ (outputs + one block of start-up) x (VALU per output ~ W + 6);
// pick the segment counts that minimise the longest lane within the workgroup's wave budget.
bool make_split_roles(int K, int nt, int lh, int lp, int maxwaves, Plan *p) {
    auto waves = [](int n, int seg) { return seg ? (n * seg + 63) / 64 : 0; };
    long best = -1;
    int bh = 0, bp = 0;
    for (int nsh = lh ? 1 : 0; nsh <= (lh ? 4 : 0); ++nsh) {
        if (lh && nsh > 1 && nt < 2 * lh * nsh) break;
        for (int nsp = lp ? 1 : 0; nsp <= (lp ? 6 : 0); ++nsp) {
            if (lp && nsp > 1 && K < 2 * lp * nsp) break;
            if (waves(K, nsh) + waves(nt, nsp) > maxwaves) continue;
            const long ch = lh ? (long)((nt + nsh - 1) / nsh + lh) * (lh + 6) : 0;
            const long cp = lp ? (long)((K + nsp - 1) / nsp + lp) * (lp + 6) : 0;
            const long c = ch > cp ? ch : cp;
            if (best < 0 || c < best) best = c, bh = nsh, bp = nsp;
        }
    }
    if (best < 0) return false;
    if (const char *ev = getenv("SMH_MEDIAN_SEG")) {  // tuning override (tools/tune_median.py): "nsh,nsp"
        int a = 0, b = 0;
        if (sscanf(ev, "%d,%d", &a, &b) == 2 && (!lh || a >= 1) && (!lp || b >= 1) &&
            waves(K, lh ? a : 0) + waves(nt, lp ? b : 0) <= maxwaves)
            bh = lh ? a : 0, bp = lp ? b : 0;
    }
    p->nsh = bh, p->nsp = bp, p->nwh = waves(K, bh), p->nwp = waves(nt, bp);
    return true;
}

int launch_small(const float *S, int B, int K, int T, int w, int along_t, float *out, hipStream_t st) {
    const size_t total = (size_t)B * K * T;
    const int bs = 256;
    hipLaunchKernelGGL(median_small_kernel, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, st, S, out, K, T, w,
                       along_t, total);
    return smh::launch_status("median_small_kernel");
}

// harm_tmajor: 0 = (B,K,T), 1 = (B,T,K), 2 = (B, ceil(T/16), K, 16) (block-split kernels; any other kernel writes layout 1
// instead).  *written (optional) receives the layout that was produced.
int launch(const float *S, int B, int K, int T, int lh, int lp, float *harm, float *perc, hipStream_t st,
           int harm_tmajor = 0, int *written = nullptr) {
    if (written) *written = harm_tmajor;
    KernelFn fn = (lh && lp) ? find_pair_kernel(lh, lp) : find_single_kernel(lh, lp);
    if (!fn) return smh::set_error(SMH_E_INVALID, "no median kernel for (l_harm,l_perc)=(%d,%d)", lh, lp);
    Plan p;
    int rc = make_plan(K, T, lh, lp, &p);
    if (rc) return rc;
    // windows up to 21: the block-split kernel (smh_median_split.h) when its wave budget covers the tile
    static const bool no_split = getenv("SMH_MEDIAN_NOSPLIT") != nullptr;  // A/B switches for tools/tune_median.py
    // Whole clip per tile and at least two clips per CU: windows above 17 (history > 128 VGPRs, so only 6 waves per
    // workgroup otherwise) take the persistent double-buffered kernel, one 8-wave workgroup per CU with 256 VGPRs;
    // up to 17 two ordinary 8-wave workgroups per CU measure 3-5 % faster (4 waves per SIMD).  SMH_MEDIAN_PERSIST=0/1
    // forces either, SMH_MEDIAN_PTHREADS picks the 512 / 768 / 1024-thread build (tools/gpu/tune_split.sh).
    bool want_persist = (lh > 17 || lp > 17);
    if (const char *ev = getenv("SMH_MEDIAN_PERSIST")) want_persist = atoi(ev) != 0;
    int pthreads = 512;
    if (const char *ev = getenv("SMH_MEDIAN_PTHREADS")) pthreads = atoi(ev);
    if (const PersistEntry *pe = (no_split || !want_persist || !(lh && lp)) ? nullptr : find_persist_kernel(lh, lp, pthreads)) {
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0;
            SMH_CHECK_HIP(hipGetDevice(&dev));
            SMH_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        const int tile_bytes = persist_tile_bytes(K, T);
        const int g = T & 63;
        const bool conflict_free = (T & 1) || (g % 4 == 2);  // gcd(T, 64) <= 2
        Plan q = p;
        if (p.ntiles == 1 && conflict_free && 2 * tile_bytes <= smh::kLdsBytesPerCU && B >= 2 * n_cu &&
            make_split_roles(K, T, lh, lp, pe->threads / 64, &q)) {
            const int lds = 2 * tile_bytes;
            SMH_CHECK_HIP(hipFuncSetAttribute((const void *)pe->fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(pe->fn, dim3(n_cu), dim3((q.nwh + q.nwp) * 64), lds, st, S, harm, perc, B, K, T, q.nsh,
                               q.nsp, q.nwh, harm_tmajor);
            return smh::launch_status("hpss_median_persist_kernel");
        }
    }
    // (layout 2 is also written tile by tile: the blocked walker addresses harm[t / 16][k][t % 16] by absolute frame)
    if (const SplitEntry *se = no_split ? nullptr : find_split_kernel(lh, lp)) {
        Plan q = p;
        if (make_split_roles(K, p.TT, lh, lp, se->threads / 64, &q)) {
            SMH_CHECK_HIP(hipFuncSetAttribute((const void *)se->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds));
            dim3 grid(q.ntiles, B), block((q.nwh + q.nwp) * 64);
            const bool probe_noload = smh::probe_env("SMH_MEDIAN_PROBE_NOLOAD") != nullptr;  // timing experiment, outputs invalid
            hipLaunchKernelGGL(se->fn, grid, block, q.lds, st, S, harm, perc, K, T, q.TT, q.stride, q.nsh, q.nsp, q.nwh,
                               harm_tmajor, probe_noload ? 1.f : -__builtin_inff(), __builtin_inff(), nullptr, nullptr, 0);
            return smh::launch_status("hpss_median_split_kernel");
        }
    }
    if (harm_tmajor == 2) {  // the delete/insert kernel has no blocked store
        harm_tmajor = 1;
        if (written) *written = 1;
    }
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    dim3 grid(p.ntiles, B), block((p.nwh + p.nwp) * 64);
    hipLaunchKernelGGL(fn, grid, block, p.lds, st, S, harm, perc, K, T, p.TT, p.stride, p.nsh, p.nsp, p.nwh,
                       harm_tmajor, -__builtin_inff(), __builtin_inff());
    return smh::launch_status("hpss_median_kernel");
}

int check_args(const void *S, int B, int K, int T, int w, const char *name) {
    SMH_REQUIRE(S != nullptr || B == 0, "%s: null input", name);
    SMH_REQUIRE(B >= 0 && K >= 1 && T >= 1, "%s: bad shape B=%d K=%d T=%d", name, B, K, T);
    SMH_REQUIRE(w >= 1 && w <= SMH_MAX_MEDIAN && (w & 1), "%s: window must be odd in [1,%d], got %d", name,
                SMH_MAX_MEDIAN, w);
    SMH_REQUIRE(B <= 65535, "%s: B=%d exceeds the grid limit; split the batch", name, B);
    return SMH_OK;
}

// the register-window kernel folds once per side: it needs window/2 < axis length
bool fast_ok(int n, int w) { return w >= 3 && w / 2 + 4 < n; }

}  // namespace

namespace smh_median {
// ---- clips of different lengths in one launch (smh_rag.h) ----
// Frame tile of the ragged launch: the widest tile -- a multiple of 4 frames, so that the blocked harm rows start on float4
// boundaries -- whose halo columns fit the LDS budget of two workgroups per CU (K = 201, l_harm = 21: 76 frames, 96 columns).
int rag_tile_frames(int K, int lh, int lp, int *stride) {
    static const bool no_split = getenv("SMH_MEDIAN_NOSPLIT") != nullptr;
    if (no_split || !(lh && lp) || !find_split_kernel(lh, lp)) return 0;
    const int hh = lh / 2;
    const int budget_words = (smh::kLdsBytesPerCU / 2 - 1024) / 4;
    int maxcols = budget_words / K;
    if ((maxcols & 1) == 0) maxcols -= 1;
    int TT = (maxcols - 2 * hh) & ~3;
    if (TT < 2 * hh + 8) return 0;
    if (stride) *stride = (TT + 2 * hh) | 1;
    return TT;
}

int launch_rag(const float *d_S, float *d_harm, float *d_perc, int K, int lh, int lp, const smh_rag::Clip *d_clips,
               const smh_rag::Item *d_items, int n_items, hipStream_t st) {
    if (n_items <= 0) return SMH_OK;
    int stride = 0;
    const int TT = rag_tile_frames(K, lh, lp, &stride);
    const SplitEntry *se = TT ? find_split_kernel(lh, lp) : nullptr;
    Plan q;
    if (!se || !make_split_roles(K, TT, lh, lp, se->threads / 64, &q))
        return smh::set_error(SMH_E_INVALID, "ragged medians: no block-split kernel for (l_harm,l_perc)=(%d,%d), K=%d", lh, lp, K);
    const size_t lds = (size_t)K * stride * sizeof(float);
    SMH_CHECK_HIP(hipFuncSetAttribute((const void *)se->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)(8 * (((long long)n_items + 7) / 8));
    hipLaunchKernelGGL(se->fn, dim3(grid), dim3((q.nwh + q.nwp) * 64), lds, st, d_S, d_harm, d_perc, K, 0, TT, stride, q.nsh, q.nsp,
                       q.nwh, 2, -__builtin_inff(), __builtin_inff(), d_clips, d_items, n_items);
    return smh::launch_status("hpss_median_split_kernel (ragged)");
}

// Fused-pipeline entry: both filters in one launch, harm optionally time-major (B,T,K).
// Returns 1 if the time-major layout was produced, 0 if the reference layout was used, <0 on error.
int launch_hpss(const float *S, int B, int K, int T, int lh, int lp, float *harm, float *perc, int want_tmajor,
                hipStream_t st) {
    if (fast_ok(T, lh) && fast_ok(K, lp) && find_pair_kernel(lh, lp)) {
        int written = want_tmajor;
        int rc = launch(S, B, K, T, lh, lp, harm, perc, st, want_tmajor, &written);
        return rc ? rc : written;
    }
    int rc = smh_hpss_median_f32(nullptr, S, B, K, T, lh, lp, harm, perc, (void *)st);
    return rc ? rc : 0;
}
}  // namespace smh_median

extern "C" int smh_median_time_f32(const smh_ctx *, const float *d_S, int B, int K, int T, int l_harm, float *d_harm,
                                   void *stream) {
    int rc = check_args(d_S, B, K, T, l_harm, "smh_median_time_f32");
    if (rc) return rc;
    SMH_REQUIRE(d_harm || B == 0, "smh_median_time_f32: null output");
    if (B == 0) return SMH_OK;
    hipStream_t st = (hipStream_t)stream;
    if (l_harm == 1) {
        SMH_CHECK_HIP(hipMemcpyAsync(d_harm, d_S, (size_t)B * K * T * sizeof(float), hipMemcpyDeviceToDevice, st));
        return SMH_OK;
    }
    if (!fast_ok(T, l_harm)) return launch_small(d_S, B, K, T, l_harm, 1, d_harm, st);
    return launch(d_S, B, K, T, l_harm, 0, d_harm, nullptr, st);
}

// the harmonic median alone in any of the three layouts of smh_hpss_median_ex_f32 (returns the layout written)
extern "C" int smh_median_time_ex_f32(const smh_ctx *, const float *d_S, int B, int K, int T, int l_harm, float *d_harm,
                                      int harm_layout, void *stream) {
    int rc = check_args(d_S, B, K, T, l_harm, "smh_median_time_ex_f32");
    if (rc) return rc;
    SMH_REQUIRE(harm_layout >= 0 && harm_layout <= 2, "smh_median_time_ex_f32: harm_layout must be 0, 1 or 2");
    SMH_REQUIRE(d_harm || B == 0, "smh_median_time_ex_f32: null output");
    if (B == 0) return harm_layout;
    if (l_harm == 1 || !fast_ok(T, l_harm) || harm_layout == 0) {
        rc = smh_median_time_f32(nullptr, d_S, B, K, T, l_harm, d_harm, stream);
        return rc ? rc : 0;
    }
    int written = harm_layout;
    rc = launch(d_S, B, K, T, l_harm, 0, d_harm, nullptr, (hipStream_t)stream, harm_layout, &written);
    return rc ? rc : written;
}

extern "C" int smh_median_freq_f32(const smh_ctx *, const float *d_S, int B, int K, int T, int l_perc, float *d_perc,
                                   void *stream) {
    int rc = check_args(d_S, B, K, T, l_perc, "smh_median_freq_f32");
    if (rc) return rc;
    SMH_REQUIRE(d_perc || B == 0, "smh_median_freq_f32: null output");
    if (B == 0) return SMH_OK;
    hipStream_t st = (hipStream_t)stream;
    if (l_perc == 1) {
        SMH_CHECK_HIP(hipMemcpyAsync(d_perc, d_S, (size_t)B * K * T * sizeof(float), hipMemcpyDeviceToDevice, st));
        return SMH_OK;
    }
    if (!fast_ok(K, l_perc)) return launch_small(d_S, B, K, T, l_perc, 0, d_perc, st);
    return launch(d_S, B, K, T, 0, l_perc, nullptr, d_perc, st);
}

extern "C" int smh_hpss_median_f32(const smh_ctx *, const float *d_S, int B, int K, int T, int l_harm, int l_perc,
                                   float *d_harm, float *d_perc, void *stream) {
    int rc = check_args(d_S, B, K, T, l_harm, "smh_hpss_median_f32");
    if (rc) return rc;
    rc = check_args(d_S, B, K, T, l_perc, "smh_hpss_median_f32");
    if (rc) return rc;
    SMH_REQUIRE((d_harm && d_perc) || B == 0, "smh_hpss_median_f32: null output");
    if (B == 0) return SMH_OK;
    if (fast_ok(T, l_harm) && fast_ok(K, l_perc) && find_pair_kernel(l_harm, l_perc))
        return launch(d_S, B, K, T, l_harm, l_perc, d_harm, d_perc, (hipStream_t)stream);
    // window pairs outside the fused table, or tiny axes: one launch per filter
    rc = smh_median_time_f32(nullptr, d_S, B, K, T, l_harm, d_harm, stream);
    if (rc) return rc;
    return smh_median_freq_f32(nullptr, d_S, B, K, T, l_perc, d_perc, stream);
}

extern "C" int smh_hpss_median_ex_f32(const smh_ctx *, const float *d_S, int B, int K, int T, int l_harm, int l_perc,
                                      float *d_harm, float *d_perc, int harm_layout, void *stream) {
    int rc = check_args(d_S, B, K, T, l_harm, "smh_hpss_median_ex_f32");
    if (rc) return rc;
    rc = check_args(d_S, B, K, T, l_perc, "smh_hpss_median_ex_f32");
    if (rc) return rc;
    SMH_REQUIRE(harm_layout >= 0 && harm_layout <= 2, "smh_hpss_median_ex_f32: harm_layout must be 0, 1 or 2");
    SMH_REQUIRE((d_harm && d_perc) || B == 0, "smh_hpss_median_ex_f32: null output");
    if (B == 0) return harm_layout;
    return smh_median::launch_hpss(d_S, B, K, T, l_harm, l_perc, d_harm, d_perc, harm_layout, (hipStream_t)stream);
}
