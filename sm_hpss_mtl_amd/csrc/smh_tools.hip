// The remaining two functions of the reference's native module (lib/cython_impl/tools.pyx) -- off by default on the hot
// path (frame_level_scaling False, skewness_vector None) but part of the module's surface:
//   scale_data(FV, mean, stdev)                      tools.pyx:138-165   float64 out, (FV - mean) / (stdev + 1e-10) per row
//   get_data_statistics(FV, stat_type, axis)         tools.pyx:169-215   mean / variance / skew / kurtosis of every patch
// float64 arithmetic like the reference; the statistics are scipy.stats.skew / kurtosis (biased, Fisher) and
// numpy mean / var restated -- sums in index order here, pairwise in numpy.
#include "smh_common.h"

namespace {

__global__ void scale_data_kernel(const double *__restrict__ FV, int F, int T, const double *__restrict__ mean,
                                  const double *__restrict__ stdev, double *__restrict__ out) {
    const size_t n = (size_t)F * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / T);
        out[i] = (FV[i] - mean[f]) / (stdev[f] + 1e-10);  // np.subtract, then np.divide by S + 1e-10
    }
}

// one thread per output element; `len` values at stride `step` starting at `base`
__device__ __forceinline__ double statistic(const double *base, int len, size_t step, int stat) {
    double s = 0.0;
    for (int i = 0; i < len; ++i) s += base[(size_t)i * step];
    const double mean = s / len;
    if (stat == 0) return mean;
    double m2 = 0.0, m3 = 0.0, m4 = 0.0;
    for (int i = 0; i < len; ++i) {
        const double d = base[(size_t)i * step] - mean, d2 = d * d;
        m2 += d2;
        m3 += d2 * d;
        m4 += d2 * d2;
    }
    m2 /= len, m3 /= len, m4 /= len;
    if (stat == 1) return m2;                            // np.var (population)
    if (stat == 2) return m2 == 0.0 ? 0.0 : m3 / (m2 * sqrt(m2));  // scipy.stats.skew (bias=True); constant input -> 0
    return m2 == 0.0 ? -3.0 : m4 / (m2 * m2) - 3.0;      // scipy.stats.kurtosis (fisher=True, bias=True)
}

__global__ void data_statistics_kernel(const double *__restrict__ FV, int N, int F, int T, int stat, int axis,
                                       double *__restrict__ out) {
    const int per = axis == 0 ? T : F;
    const size_t total = (size_t)N * per;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / per;
    const int e = (int)(i - n * per);
    const double *p = FV + n * (size_t)F * T;
    out[i] = axis == 0 ? statistic(p + e, F, (size_t)T, stat)       // over the rows (percussive direction): (N, T)
                       : statistic(p + (size_t)e * T, T, 1, stat);  // over the frames (harmonic direction): (N, F)
}

}  // namespace

extern "C" int smh_scale_data_f64(const double *d_FV, int F, int T, const double *d_mean, const double *d_stdev,
                                  double *d_out, void *stream) {
    SMH_REQUIRE(F >= 0 && T >= 0, "smh_scale_data_f64: bad shape");
    if (F == 0 || T == 0) return SMH_OK;
    SMH_REQUIRE(d_FV && d_mean && d_stdev && d_out, "smh_scale_data_f64: null argument");
    size_t nb = ((size_t)F * T + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(scale_data_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, d_FV, F, T, d_mean, d_stdev,
                       d_out);
    return smh::launch_status("scale_data_kernel");
}

extern "C" int smh_data_statistics_f64(const double *d_FV, int N, int F, int T, int stat, int axis, double *d_out,
                                       void *stream) {
    SMH_REQUIRE(N >= 0 && F >= 1 && T >= 1, "smh_data_statistics_f64: bad shape");
    SMH_REQUIRE(stat >= 0 && stat <= 3, "smh_data_statistics_f64: stat must be 0 (mean), 1 (variance), 2 (skew) or 3 (kurtosis)");
    SMH_REQUIRE(axis == 0 || axis == 1, "smh_data_statistics_f64: axis must be 0 or 1");
    if (N == 0) return SMH_OK;
    SMH_REQUIRE(d_FV && d_out, "smh_data_statistics_f64: null argument");
    const size_t total = (size_t)N * (axis == 0 ? T : F);
    hipLaunchKernelGGL(data_statistics_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream, d_FV, N,
                       F, T, stat, axis, d_out);
    return smh::launch_status("data_statistics_kernel");
}
