// median_lab2: prototype of the LDS-free "streaming" HPSS median kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o tools/median_lab2 tools/median_lab2.hip && ./tools/median_lab2
// Rows (harmonic) and columns (percussive) are flattened across clips so every wave is full; each lane
// streams its axis from global memory, the window is sorted in registers and the outgoing values come
// from a statically indexed register ring (the loop is unrolled by the ring size R >= W, R % 4 == 0).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

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
        bool f[W - 1];
#pragma unroll
        for (int i = 0; i < W - 1; ++i) f[i] = s[i] >= out_v;
#pragma unroll
        for (int i = 0; i < W - 1; ++i) s[i] = f[i] ? s[i + 1] : s[i];
        s[W - 1] = med3(s[W - 2], in_v, pinf);
#pragma unroll
        for (int i = W - 2; i >= 1; --i) s[i] = med3(s[i - 1], in_v, s[i]);
        s[0] = med3(ninf, in_v, s[0]);
    }
    __device__ __forceinline__ float median() const { return s[W / 2]; }
};

template <int W>
constexpr int ring_size() { return (W + 3) / 4 * 4; }

// initialise window + ring from element fetcher: element j of the window (j = 0..W-1)
template <int W, int N>
struct Init {
    template <typename F>
    static __device__ __forceinline__ void run(SortedWindow<W> &w, float *ring, F &&fetch) {
        constexpr int R = ring_size<W>();
        const float v = fetch(N);
        ring[(R - W + N) % R] = v;
        w.template insert<N>(v);
        if constexpr (N + 1 < W) Init<W, N + 1>::run(w, ring, fetch);
    }
};

// Harmonic role: lane <-> flattened row r = b*K + k; axis = T frames, contiguous in memory.
template <int W, bool TMAJOR>
__device__ __forceinline__ void harm_row(const float *__restrict__ row, int T, float *__restrict__ out, unsigned ooff,
                                         unsigned ostep, float ninf, float pinf) {
    constexpr int H = W / 2, R = ring_size<W>();
    SortedWindow<W> win;
    float ring[R];
    win.clear(ninf, pinf);
    Init<W, 0>::run(win, ring, [&](int j) { return row[reflect_hi(reflect_lo(j - H), T)]; });
    const int n_steps = T - 1;
    // steps served by un-clamped float4 groups: group of step s4 (multiple of 4) starts at H+1+s4 <= T-4
    const int n_vec = (T - 5 - H) >= 0 ? ((T - 5 - H) / 4 + 1) * 4 : 0;
    // stream of incoming samples: float4 groups starting at index H+1, one group prefetched
    const int last4 = T - 4;
    float4u nxt = *reinterpret_cast<const float4u *>(row + min(H + 1, last4));
    float o4[4];
    char *ob = reinterpret_cast<char *>(out);
    for (int s0 = 0; s0 < n_steps; s0 += R) {
        float4u cur;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int s = s0 + i;
            if (s >= n_steps) break;  // wave-uniform
            if (i % 4 == 0) {
                cur = nxt;
                nxt = *reinterpret_cast<const float4u *>(row + min(H + 1 + s + 4, last4));
            }
            const float med = win.median();
            if constexpr (TMAJOR) {
                *reinterpret_cast<float *>(ob + ooff) = med;
                ooff += ostep;
            } else {
                o4[i % 4] = med;
                if (i % 4 == 3) {
                    float4u v = {o4[0], o4[1], o4[2], o4[3]};
                    *reinterpret_cast<float4u *>(ob + ooff) = v;
                    ooff += 16;
                }
            }
            float in_v = cur[i % 4];
            // steps whose float4 group was clamped at the row end, and folded steps: scalar fetch
            if (s >= n_vec) in_v = row[reflect_hi(s + H + 1, T)];
            const float out_v = ring[(i + R - W) % R];
            ring[i] = in_v;
            win.replace(out_v, in_v);
        }
    }
    // last output (position T-1) and, for the row-major layout, the pending partial group
    const float med = win.median();
    if constexpr (TMAJOR) {
        *reinterpret_cast<float *>(ob + ooff) = med;
    } else {
        const int done = (n_steps / 4) * 4;  // outputs already stored as full groups
        // outputs done .. T-2 are still in o4 (their slots are (idx % 4) because groups are 4-aligned)
        for (int idx = done; idx < n_steps; ++idx) {
            *reinterpret_cast<float *>(ob + ooff) = o4[idx % 4];
            ooff += 4;
        }
        *reinterpret_cast<float *>(ob + ooff) = med;
    }
}

// Percussive role: lane <-> flattened column c = b*T + t; axis = K bins at stride T; segment [ks, ke).
template <int W>
__device__ __forceinline__ void perc_col(const float *__restrict__ col, int K, int T, int ks, int ke,
                                         float *__restrict__ out, float ninf, float pinf) {
    constexpr int H = W / 2, R = ring_size<W>();
    SortedWindow<W> win;
    float ring[R];
    win.clear(ninf, pinf);
    Init<W, 0>::run(win, ring, [&](int j) { return col[(size_t)reflect_hi(reflect_lo(ks - H + j), K) * T]; });
    const int n_steps = ke - ks - 1;
    float *o = out + (size_t)ks * T;
    // incoming samples, four steps prefetched
    float pf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pf[u] = col[(size_t)reflect_hi(ks + H + 1 + u, K) * T];
    for (int s0 = 0; s0 < n_steps; s0 += R) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int s = s0 + i;
            if (s >= n_steps) break;  // wave-uniform
            __builtin_nontemporal_store(win.median(), o);
            o += T;
            const float in_v = pf[i % 4];
            pf[i % 4] = col[(size_t)reflect_hi(ks + H + 1 + s + 4, K) * T];
            const float out_v = ring[(i + R - W) % R];
            ring[i] = in_v;
            win.replace(out_v, in_v);
        }
    }
    __builtin_nontemporal_store(win.median(), o);
}

template <int LH, int LP, bool TMAJOR>
__global__ void __launch_bounds__(256)
stream_kernel(const float *__restrict__ S, float *__restrict__ harm, float *__restrict__ perc, int B, int K, int T,
              int n_harm_blocks, int nsp, float ninf, float pinf, int roles) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < n_harm_blocks ? !(roles & 1) : !(roles & 2)) return;
    if ((int)blockIdx.x < n_harm_blocks) {
        const long r = (long)blockIdx.x * 256 + tid;
        if (r >= (long)B * K) return;
        const int b = (int)(r / K), k = (int)(r - (long)b * K);
        const float *row = S + (size_t)r * T;
        if (TMAJOR)
            harm_row<LH, true>(row, T, harm + (size_t)b * K * T, (unsigned)k * 4u, (unsigned)K * 4u, ninf, pinf);
        else
            harm_row<LH, false>(row, T, harm + (size_t)r * T, 0u, 4u, ninf, pinf);
    } else {
        const long id = (long)(blockIdx.x - n_harm_blocks) * 256 + tid;
        const long ncol = (long)B * T;
        if (id >= ncol * nsp) return;
        const int sg = (int)(id / ncol);
        const long c = id - (long)sg * ncol;
        const int b = (int)(c / T), t = (int)(c - (long)b * T);
        const int seglen = (K + nsp - 1) / nsp;
        const int ks = sg * seglen, ke = min(K, ks + seglen);
        if (ks >= ke) return;
        perc_col<LP>(S + (size_t)b * K * T + t, K, T, ks, ke, perc + (size_t)b * K * T + t, ninf, pinf);
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

template <int LH, int LP, bool TMAJOR>
void run(const char *name, const float *dS, float *dH, float *dP, const std::vector<float> &hS, int B, int K, int T,
         int nsp, int roles = 3) {
    std::vector<float> refH((size_t)2 * K * T), refP((size_t)2 * K * T);
    for (int c = 0; c < 2; ++c) {
        cpu_median(hS.data() + (size_t)c * K * T, refH.data() + (size_t)c * K * T, K, T, LH, true);
        cpu_median(hS.data() + (size_t)c * K * T, refP.data() + (size_t)c * K * T, K, T, LP, false);
    }
    const int nhb = (int)(((long)B * K + 255) / 256);
    const int npb = (int)(((long)B * T * nsp + 255) / 256);
    hipMemset(dH, 0, (size_t)B * K * T * 4);
    hipMemset(dP, 0, (size_t)B * K * T * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ts;
    for (int it = 0; it < 14; ++it) {
        hipEventRecord(a);
        stream_kernel<LH, LP, TMAJOR><<<nhb + npb, 256>>>(dS, dH, dP, B, K, T, nhb, nsp, -INFINITY, INFINITY, roles);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it >= 4) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    std::vector<float> h((size_t)2 * K * T), p((size_t)2 * K * T);
    hipMemcpy(h.data(), dH, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(p.data(), dP, p.size() * 4, hipMemcpyDeviceToHost);
    bool okH = true, okP = !(roles & 2) || std::equal(p.begin(), p.end(), refP.begin());
    if (!(roles & 1)) {
    } else if (TMAJOR) {
        for (int c = 0; c < 2 && okH; ++c)
            for (int k = 0; k < K && okH; ++k)
                for (int t = 0; t < T; ++t)
                    if (h[(size_t)c * K * T + (size_t)t * K + k] != refH[(size_t)c * K * T + (size_t)k * T + t]) {
                        okH = false;
                        break;
                    }
    } else {
        okH = std::equal(h.begin(), h.end(), refH.begin());
    }
    const double bytes = 3.0 * K * T * 4 * B;
    printf("(%2d,%2d) %-40s median %.4f ms  min %.4f ms  %5.1f%% of 8TB/s  harm %s perc %s\n", LH, LP, name,
           ts[ts.size() / 2], ts[0], 100 * bytes / (ts[ts.size() / 2] * 1e-3) / 8e12, okH ? "ok" : "MISMATCH",
           okP ? "ok" : "MISMATCH");
    fflush(stdout);
}

int main() {
    const int B = 1024, K = 201, T = 98;
    std::vector<float> hS((size_t)B * K * T);
    std::mt19937 rng(1);
    std::exponential_distribution<float> ex(1.0f);
    for (auto &v : hS) v = ex(rng);
    for (size_t i = 0; i < (size_t)K * T; i += 7) hS[i] = 0.5f;  // ties in clip 0
    float *dS, *dH, *dP;
    hipMalloc(&dS, hS.size() * 4 + 64);
    hipMalloc(&dH, hS.size() * 4 + 64);
    hipMalloc(&dP, hS.size() * 4 + 64);
    hipMemcpy(dS, hS.data(), hS.size() * 4, hipMemcpyHostToDevice);
    run<17, 17, true>("stream tmajor nsp=2", dS, dH, dP, hS, B, K, T, 2);
    run<17, 17, true>("harm only", dS, dH, dP, hS, B, K, T, 2, 1);
    run<17, 17, true>("perc only nsp=2", dS, dH, dP, hS, B, K, T, 2, 2);
    run<17, 17, true>("perc only nsp=1", dS, dH, dP, hS, B, K, T, 1, 2);
    run<11, 11, true>("harm only", dS, dH, dP, hS, B, K, T, 2, 1);
    run<11, 11, true>("perc only nsp=2", dS, dH, dP, hS, B, K, T, 2, 2);
    run<17, 17, true>("stream tmajor nsp=3", dS, dH, dP, hS, B, K, T, 3);
    run<17, 17, false>("stream rowmajor nsp=2", dS, dH, dP, hS, B, K, T, 2);
    run<21, 11, true>("stream tmajor nsp=2", dS, dH, dP, hS, B, K, T, 2);
    run<21, 11, true>("stream tmajor nsp=1", dS, dH, dP, hS, B, K, T, 1);
    run<21, 11, false>("stream rowmajor nsp=2", dS, dH, dP, hS, B, K, T, 2);
    run<11, 11, true>("stream tmajor nsp=2", dS, dH, dP, hS, B, K, T, 2);
    run<31, 31, true>("stream tmajor nsp=2", dS, dH, dP, hS, B, K, T, 2);
    return 0;
}
