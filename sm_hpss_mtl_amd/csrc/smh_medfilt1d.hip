// scipy.signal.medfilt(x, kernel_size) on 1-D tracks: the smoothing of the per-frame probability
// track in the dense file-level inference, /root/reference/DAFx12_Speech_Music_Detection_B3_MTL_v2.py:94-98 with
// smoothing_win_size = 501 (:802).  Zero padding at both ends, odd window, bit-exact selection.
//
// A 501-wide window does not fit a register-resident sorted run, so the selection is a bitwise binary search over
// an order-preserving integer key: the k-th smallest key is the largest v with #{keys < v} <= k, found bit by bit
// (32 counting passes over the window).  One thread per output, the span of a 256-output workgroup plus its halo
// lives in LDS as keys (consecutive lanes read consecutive words: conflict-free).  32 * W compare-accumulates per
// output: an hour of audio at 100 frames/s (360 000 outputs, W = 501) is ~0.3 ms -- not worth a cleverer scheme.
#include "smh_common.h"

namespace {

constexpr int kOut = 256;

__device__ __forceinline__ unsigned to_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_key(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void __launch_bounds__(kOut) medfilt1d_kernel(const float *__restrict__ x, int n, int W, float *__restrict__ y) {
    extern __shared__ unsigned keys[];  // kOut + W - 1
    const int b = blockIdx.y;
    const int o0 = blockIdx.x * kOut;
    const int h = W / 2;
    const float *xb = x + (size_t)b * n;
    const int span = kOut + W - 1;
    for (int i = threadIdx.x; i < span; i += kOut) {
        const int p = o0 - h + i;
        keys[i] = to_key((p >= 0 && p < n) ? xb[p] : 0.0f);  // zero padding
    }
    __syncthreads();
    const int o = o0 + threadIdx.x;
    if (o >= n) return;
    const unsigned *win = keys + threadIdx.x;
    unsigned res = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = res | (1u << bit);
        int cnt = 0;
        for (int i = 0; i < W; ++i) cnt += win[i] < cand ? 1 : 0;
        if (cnt <= h) res = cand;
    }
    y[(size_t)b * n + o] = from_key(res);
}

}  // namespace

extern "C" int smh_medfilt1d_f32(const float *d_x, int B, int n, int kernel_size, float *d_y, void *stream) {
    SMH_REQUIRE(B >= 0 && B <= 65535 && n >= 0, "smh_medfilt1d_f32: bad shape B=%d n=%d", B, n);
    SMH_REQUIRE(kernel_size >= 1 && (kernel_size & 1) && kernel_size <= 8191,
                "smh_medfilt1d_f32: kernel_size must be odd in [1, 8191], got %d", kernel_size);
    if (B == 0 || n == 0) return SMH_OK;
    SMH_REQUIRE(d_x && d_y, "smh_medfilt1d_f32: null argument");
    SMH_REQUIRE(d_x != d_y, "smh_medfilt1d_f32: in-place operation is not supported");
    const size_t lds = (size_t)(kOut + kernel_size - 1) * sizeof(unsigned);
    hipLaunchKernelGGL(medfilt1d_kernel, dim3((n + kOut - 1) / kOut, B), dim3(kOut), lds, (hipStream_t)stream, d_x, n,
                       kernel_size, d_y);
    return smh::launch_status("medfilt1d_kernel");
}
