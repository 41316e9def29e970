// median_lab: A/B laboratory for the HPSS median kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o tools/median_lab tools/median_lab.hip && ./tools/median_lab
// Variants of the walker are template parameters; every variant is checked against a CPU reference on
// two clips and timed in interleaved rounds in ONE process (B=1024 clips of 201x98).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ int reflect_lo(int i) { return i ^ (i >> 31); }
__device__ __forceinline__ int reflect_hi(int i, int n) { return min(i, 2 * n - 1 - i); }

template <int W>
struct SortedWindow {
    float s[W];
    float ninf, pinf;
    __device__ __forceinline__ void clear(float ni, float pi) {
        ninf = ni, pinf = pi;
#pragma unroll
        for (int i = 0; i < W; ++i) s[i] = pi;
    }
    template <int N>
    __device__ __forceinline__ void insert(float x) {
        float prev = ninf;
#pragma unroll
        for (int i = 0; i <= N; ++i) {
            const float cur = s[i];
            s[i] = med3(prev, x, cur);
            prev = cur;
        }
    }
    __device__ __forceinline__ void replace(float out_v, float in_v) {
        float rprev = ninf;
#pragma unroll
        for (int i = 0; i < W - 1; ++i) {
            const float ri = (s[i] >= out_v) ? s[i + 1] : s[i];
            s[i] = med3(rprev, in_v, ri);
            rprev = ri;
        }
        s[W - 1] = med3(rprev, in_v, pinf);
    }
    // same update, written so that three compare masks are in flight (SGPR pairs) and the selects run
    // in place: r[i] overwrites s[i]; then the insertion runs top-down in place.
    __device__ __forceinline__ void replace2(float out_v, float in_v) {
        bool f[W - 1];
#pragma unroll
        for (int i = 0; i < W - 1; ++i) f[i] = s[i] >= out_v;
#pragma unroll
        for (int i = 0; i < W - 1; ++i) s[i] = f[i] ? s[i + 1] : s[i];  // r[i] in place
        // s[0..W-2] = r ; insertion: s'[i] = med3(r[i-1], in, r[i]), s'[W-1] = med3(r[W-2], in, +inf)
        s[W - 1] = med3(s[W - 2], in_v, pinf);
#pragma unroll
        for (int i = W - 2; i >= 1; --i) s[i] = med3(s[i - 1], in_v, s[i]);
        s[0] = med3(ninf, in_v, s[0]);
    }
    __device__ __forceinline__ float median() const { return s[W / 2]; }
};

template <int W, int N>
struct Filler {
    template <typename F>
    static __device__ __forceinline__ void run(SortedWindow<W> &w, F &&fetch) {
        w.template insert<N>(fetch(N));
        if constexpr (N + 1 < W) Filler<W, N + 1>::run(w, fetch);
    }
};

// VAR 0: uniform generic loop, one step per iteration, scalar stores
// VAR 1: uniform generic loop, 4 steps per iteration, vec4 stores for rows          (first GPU version)
// VAR 2: step-index phases, pointer-increment steady loop                          (current product)
// VAR 3: as 2 with replace2 (masks in flight)
// VAR 4: as 1 with replace2
template <int W, int ES, bool VEC4, int VAR, bool NOSTORE>
__device__ __forceinline__ void walk(const float *line, int es_rt, int p0, int n_out, int n, float ninf, float pinf,
                                     char *obase, unsigned boff, unsigned ostep) {
    constexpr int H = W / 2;
    const int es = ES > 0 ? ES : es_rt;
    auto at = [&](int pos) { return line[pos * es]; };
    float sink = 0.f;
    auto emit1 = [&](float v) {
        if constexpr (NOSTORE) { sink = fmaxf(sink, v); return; }
        if constexpr (VEC4) *reinterpret_cast<float *>(obase + boff) = v;
        else __builtin_nontemporal_store(v, reinterpret_cast<float *>(obase + boff));
        boff += ostep;
    };
    auto emit4 = [&](float *o) {
        if constexpr (NOSTORE) { sink = fmaxf(fmaxf(sink, o[0]), fmaxf(o[1], fmaxf(o[2], o[3]))); return; }
        if constexpr (VEC4) {
            float4u v = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<float4u *>(obase + boff) = v;
            boff += 4 * ostep;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) emit1(o[u]);
        }
    };
    SortedWindow<W> win;
    win.clear(ninf, pinf);
    Filler<W, 0>::run(win, [&](int j) { return at(reflect_hi(reflect_lo(p0 - H + j), n)); });
    const int p_end = p0 + n_out;
    int p = p0;
    auto rep = [&](float o, float i) {
        if constexpr (VAR == 3 || VAR == 4) win.replace2(o, i);
        else win.replace(o, i);
    };
    if constexpr (VAR == 0) {
        for (; p < p_end; ++p) {
            emit1(win.median());
            if (p + 1 < p_end) rep(at(reflect_lo(p - H)), at(reflect_hi(p + H + 1, n)));
        }
    } else if constexpr (VAR == 1 || VAR == 4) {
        for (; p + 4 < p_end; p += 4) {
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                o[u] = win.median();
                rep(at(reflect_lo(p + u - H)), at(reflect_hi(p + u + H + 1, n)));
            }
            emit4(o);
        }
        for (; p < p_end; ++p) {
            emit1(win.median());
            if (p + 1 < p_end) rep(at(reflect_lo(p - H)), at(reflect_hi(p + H + 1, n)));
        }
    } else {
        const int head_end = min(p_end, p0 + H);
        for (; p < head_end; ++p) {
            emit1(win.median());
            if (p + 1 < p_end) rep(at(reflect_lo(p - H)), at(reflect_hi(p + H + 1, n)));
        }
        if (p < p_end) {
            const float *pin = line + (p + H + 1) * es;
            const float *pout = line + (p - H) * es;
            const int steady_end = p_end - (H + 4);
            for (; p + 4 <= steady_end; p += 4) {
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    o[u] = win.median();
                    rep(pout[u * es], pin[u * es]);
                }
                pin += 4 * es;
                pout += 4 * es;
                emit4(o);
            }
        }
        for (; p < p_end; ++p) {
            emit1(win.median());
            if (p + 1 < p_end) rep(at(reflect_lo(p - H)), at(reflect_hi(p + H + 1, n)));
        }
    }
    if constexpr (NOSTORE) *reinterpret_cast<float *>(obase + boff) = sink;
}

// LOADV 0: one wave per row, simple loop; 1: rows in batches of 6 (loads first)
template <int LH, int LP, int VARH, int VARP, int LOADV, bool NOSTORE>
__global__ void __launch_bounds__(1024)
lab_kernel(const float *__restrict__ S, float *__restrict__ harm, float *__restrict__ perc, int K, int T, int stride,
           int nsh, int nsp, int nwh, int do_harm, int do_perc, float ninf, float pinf) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.x;
    const float *Sb = S + (size_t)b * K * T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwaves = blockDim.x >> 6;
    const int n2 = T >> 1;
    if constexpr (LOADV == 0) {
        for (int k = wave; k < K; k += nwaves) {
            const float2v *src = reinterpret_cast<const float2v *>(Sb + (size_t)k * T);
            float *dst = tile + k * stride;
            for (int c = lane; c < n2; c += 64) {
                const float2v v = __builtin_nontemporal_load(src + c);
                dst[2 * c] = v.x;
                dst[2 * c + 1] = v.y;
            }
        }
    } else {
        constexpr int kRowBatch = 6;
        for (int k0 = wave; k0 < K; k0 += nwaves * kRowBatch) {
            float2v v[kRowBatch];
#pragma unroll
            for (int r = 0; r < kRowBatch; ++r) {
                const int k = min(k0 + r * nwaves, K - 1);
                v[r] = __builtin_nontemporal_load(reinterpret_cast<const float2v *>(Sb + (size_t)k * T) + min(lane, n2 - 1));
            }
#pragma unroll
            for (int r = 0; r < kRowBatch; ++r) {
                const int k = k0 + r * nwaves;
                if (k < K && lane < n2) {
                    tile[k * stride + 2 * lane] = v[r].x;
                    tile[k * stride + 2 * lane + 1] = v[r].y;
                }
            }
        }
    }
    __syncthreads();
    if (wave < nwh) {
        if (!do_harm) return;
        const int id = wave * 64 + lane;
        if (id < K * nsh) {
            const int sg = id / K, k = id - sg * K;
            const int seglen = (T + nsh - 1) / nsh;
            const int ts = sg * seglen, te = min(T, ts + seglen);
            if (ts < te)
                walk<LH, 1, true, VARH, NOSTORE>(tile + k * stride, 1, ts, te - ts, T, ninf, pinf,
                                       reinterpret_cast<char *>(harm + (size_t)b * K * T), (unsigned)(k * T + ts) * 4u, 4u);
        }
    } else {
        if (!do_perc) return;
        const int id = (wave - nwh) * 64 + lane;
        if (id < T * nsp) {
            const int sg = id / T, tt = id - sg * T;
            const int seglen = (K + nsp - 1) / nsp;
            const int ks = sg * seglen, ke = min(K, ks + seglen);
            if (ks < ke)
                walk<LP, 0, false, VARP, NOSTORE>(tile + tt, stride, ks, ke - ks, K, ninf, pinf,
                                        reinterpret_cast<char *>(perc + (size_t)b * K * T), (unsigned)(ks * T + tt) * 4u,
                                        (unsigned)T * 4u);
        }
    }
}

static void cpu_median(const float *S, float *out, int K, int T, int w, bool along_t) {
    const int h = w / 2;
    std::vector<float> win(w);
    for (int k = 0; k < K; ++k)
        for (int t = 0; t < T; ++t) {
            for (int j = -h; j <= h; ++j) {
                int n = along_t ? T : K, q = (along_t ? t : k) + j;
                int p2 = 2 * n, r = ((q % p2) + p2) % p2;
                if (r >= n) r = p2 - 1 - r;
                win[j + h] = along_t ? S[k * T + r] : S[r * T + t];
            }
            std::nth_element(win.begin(), win.begin() + h, win.end());
            out[k * T + t] = win[h];
        }
}

struct Result {
    double med, mn;
    bool ok;
};

template <int LH, int LP, int VARH, int VARP, int LOADV, bool NOSTORE>
Result run(const float *dS, float *dH, float *dP, const std::vector<float> &hS, int B, int K, int T, int nsh, int nsp,
           int do_harm, int do_perc, const std::vector<float> &refH, const std::vector<float> &refP) {
    const int stride = T | 1;
    const size_t lds = (size_t)K * stride * 4;
    auto fn = lab_kernel<LH, LP, VARH, VARP, LOADV, NOSTORE>;
    hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nwh = (K * nsh + 63) / 64, nwp = (T * nsp + 63) / 64;
    hipMemset(dH, 0, (size_t)B * K * T * 4);
    hipMemset(dP, 0, (size_t)B * K * T * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ts;
    for (int it = 0; it < 14; ++it) {
        hipEventRecord(a);
        fn<<<B, (nwh + nwp) * 64, lds>>>(dS, dH, dP, K, T, stride, nsh, nsp, nwh, do_harm, do_perc, -INFINITY, INFINITY);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it >= 4) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    Result r{ts[ts.size() / 2], ts[0], true};
    std::vector<float> h((size_t)2 * K * T), p((size_t)2 * K * T);
    hipMemcpy(h.data(), dH, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(p.data(), dP, p.size() * 4, hipMemcpyDeviceToHost);
    if (do_harm && !NOSTORE) r.ok = r.ok && std::equal(h.begin(), h.end(), refH.begin());
    if (do_perc && !NOSTORE) r.ok = r.ok && std::equal(p.begin(), p.end(), refP.begin());
    return r;
}

template <int LH, int LP>
void suite(const float *dS, float *dH, float *dP, const std::vector<float> &hS, int B, int K, int T) {
    std::vector<float> refH((size_t)2 * K * T), refP((size_t)2 * K * T);
    for (int c = 0; c < 2; ++c) {
        cpu_median(hS.data() + (size_t)c * K * T, refH.data() + (size_t)c * K * T, K, T, LH, true);
        cpu_median(hS.data() + (size_t)c * K * T, refP.data() + (size_t)c * K * T, K, T, LP, false);
    }
    const double bytes = 3.0 * K * T * 4 * B;
    auto show = [&](const char *name, Result r) {
        printf("(%2d,%2d) %-44s median %.4f ms  min %.4f ms  %5.1f%% of 8TB/s  %s\n", LH, LP, name, r.med, r.mn,
               100 * bytes / (r.med * 1e-3) / 8e12, r.ok ? "ok" : "MISMATCH");
        fflush(stdout);
    };
#define RUN(VH, VP, NS, nsh, nsp, dh, dp, name) show(name, run<LH, LP, VH, VP, 1, NS>(dS, dH, dP, hS, B, K, T, nsh, nsp, dh, dp, refH, refP))
    RUN(1, 1, false, 2, 3, 1, 1, "H:gen4          P:gen4");
    RUN(4, 4, false, 2, 3, 1, 1, "H:gen4+r2       P:gen4+r2");
    RUN(4, 3, false, 2, 3, 1, 1, "H:gen4+r2       P:phases+r2");
    RUN(4, 2, false, 2, 3, 1, 1, "H:gen4+r2       P:phases");
    RUN(4, 3, false, 2, 2, 1, 1, "H:gen4+r2       P:phases+r2  seg 2,2");
    RUN(4, 3, false, 1, 2, 1, 1, "H:gen4+r2       P:phases+r2  seg 1,2");
    RUN(4, 3, false, 1, 3, 1, 1, "H:gen4+r2       P:phases+r2  seg 1,3");
    RUN(4, 3, false, 2, 4, 1, 1, "H:gen4+r2       P:phases+r2  seg 2,4");
    RUN(4, 3, true, 2, 3, 1, 1, "NOSTORE H:gen4+r2 P:phases+r2");
    RUN(4, 3, false, 2, 3, 1, 0, "harm only gen4+r2");
    RUN(4, 3, true, 2, 3, 1, 0, "harm only gen4+r2 NOSTORE");
    RUN(3, 3, false, 2, 3, 1, 0, "harm only phases+r2");
    RUN(3, 3, true, 2, 3, 1, 0, "harm only phases+r2 NOSTORE");
    RUN(0, 3, false, 2, 3, 1, 0, "harm only gen1 (scalar stores)");
    RUN(4, 3, false, 1, 3, 1, 0, "harm only gen4+r2 seg1");
    RUN(4, 3, false, 2, 3, 0, 1, "perc only phases+r2");
    RUN(4, 3, true, 2, 3, 0, 1, "perc only phases+r2 NOSTORE");
#undef RUN
}

int main() {
    const int B = 1024, K = 201, T = 98;
    std::vector<float> hS((size_t)B * K * T);
    std::mt19937 rng(1);
    std::exponential_distribution<float> ex(1.0f);
    for (auto &v : hS) v = ex(rng);
    for (size_t i = 0; i < (size_t)K * T; i += 7) hS[i] = 0.5f;  // ties in clip 0
    float *dS, *dH, *dP;
    hipMalloc(&dS, hS.size() * 4);
    hipMalloc(&dH, hS.size() * 4);
    hipMalloc(&dP, hS.size() * 4);
    hipMemcpy(dS, hS.data(), hS.size() * 4, hipMemcpyHostToDevice);
    suite<17, 17>(dS, dH, dP, hS, B, K, T);
    suite<21, 11>(dS, dH, dP, hS, B, K, T);
    return 0;
}
