// svo_msa.hip - stages of the reference's MSA dense stereo built so far.
//
//  svo_ctmf : Thirdparty/MB/ctmf.h:7 `ctmf(src, dst, width, height, src_step, dst_step, r, channels, memsize)`,
//             the constant-time median filter MSA runs on the colour images (r = 1, 3 channels,
//             MSA.cpp:58-59) and on the winner-take-all disparity map (r = 2, 1 channel, MSA.cpp:1006).
//             What ctmf.c computes (ctmf.c:222-320): per channel the true median (rank 2r^2+2r of the
//             (2r+1)^2 window) with the window clamped to the image - rows and columns are REPLICATED at the
//             borders (`MAX(0, ..)`/`MIN(m-1, ..)` and the `r`-fold first column), not zero-padded as its
//             comment says.  The stripe decomposition (`memsize`) is a cache optimisation of the CPU code
//             and does not change results; it has no counterpart here.
//             GPU: one thread per (pixel, channel); the median is found by bisection on the 8-bit value
//             range (8 counting passes over the window held in registers) - O(r^2) per pixel, which for the
//             radii MSA uses (1, 2) is 9 / 25 elements.
// The rest of MSA (gradient graph, Tarjan arborescence, tree DP; MSA.cpp:152-990) is not built.
#include "svo_internal.h"

namespace {

template <int R>
__global__ __launch_bounds__(256) void k_ctmf(const uint8_t* src, uint8_t* dst, int width, int height,
                                              int src_step, int dst_step, int cn) {
  const int xc = blockIdx.x * 256 + threadIdx.x;   // column * cn + channel
  const int y = blockIdx.y;
  if (xc >= width * cn) return;
  const int x = xc / cn, c = xc - x * cn;
  constexpr int N = (2 * R + 1) * (2 * R + 1);
  uint8_t v[N];
#pragma unroll
  for (int dy = -R; dy <= R; ++dy) {
    const uint8_t* row = src + (size_t)min(max(y + dy, 0), height - 1) * src_step;
#pragma unroll
    for (int dx = -R; dx <= R; ++dx) v[(dy + R) * (2 * R + 1) + dx + R] = row[min(max(x + dx, 0), width - 1) * cn + c];
  }
  // smallest value m with #{v <= m} > t, t = 2r^2 + 2r  (ctmf.c:271-312)
  int lo = 0, hi = 255;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int mid = (lo + hi) >> 1;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) cnt += v[k] <= mid;
    if (cnt > 2 * R * R + 2 * R) hi = mid; else lo = mid + 1;
  }
  dst[(size_t)y * dst_step + xc] = (uint8_t)lo;
}

}  // namespace

extern "C" int svo_ctmf(svo_ctx* ctx, const uint8_t* src, uint8_t* dst, int width, int height, int src_step,
                        int dst_step, int r, int channels) {
  if (!ctx) return SVO_E_INVALID;
  if (!src || !dst || width < 1 || height < 1 || channels < 1 || channels > 4 || r < 1 || r > 3 ||
      src_step < width * channels || dst_step < width * channels) {
    ctx->last_error = "svo_ctmf: invalid argument (radius 1..3, 1..4 channels)";
    return SVO_E_INVALID;
  }
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sbytes = (size_t)src_step * height, dbytes = (size_t)dst_step * height;
  uint8_t *d_src = nullptr, *d_dst = nullptr;
  SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d_src), sbytes));
  if (hipMalloc(reinterpret_cast<void**>(&d_dst), dbytes) != hipSuccess) { hipFree(d_src); ctx->last_error = "svo_ctmf: hipMalloc"; return SVO_E_HIP; }
  int rc = SVO_OK;
  hipStream_t s = ctx->stream;
  const dim3 grid((width * channels + 255) / 256, height);
  if (hipMemcpyAsync(d_src, src, sbytes, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_dst, dst, dbytes, hipMemcpyHostToDevice, s) != hipSuccess) rc = SVO_E_HIP;   // keep dst padding
  if (rc == SVO_OK) {
    SvoTimer t(ctx, "k_ctmf");
    if (r == 1) hipLaunchKernelGGL(k_ctmf<1>, grid, dim3(256), 0, s, d_src, d_dst, width, height, src_step, dst_step, channels);
    else if (r == 2) hipLaunchKernelGGL(k_ctmf<2>, grid, dim3(256), 0, s, d_src, d_dst, width, height, src_step, dst_step, channels);
    else hipLaunchKernelGGL(k_ctmf<3>, grid, dim3(256), 0, s, d_src, d_dst, width, height, src_step, dst_step, channels);
  }
  if (rc == SVO_OK && (hipMemcpyAsync(dst, d_dst, dbytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
                       hipStreamSynchronize(s) != hipSuccess)) rc = SVO_E_HIP;
  hipFree(d_src); hipFree(d_dst);
  if (rc) ctx->last_error = "svo_ctmf: HIP error";
  return rc;
}
