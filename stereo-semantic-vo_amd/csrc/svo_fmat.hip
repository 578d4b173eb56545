// svo_fmat.hip - svo_fundamental_8point: cv::findFundamentalMat(cur_pts, last_pts, CV_FM_8POINT) of
// pnpmatch::poseEstimation2D_2D (reference src/pnpmatch.cc:336) for point pairs given by the caller, on the device
// (one wavefront: svo_fmat_dev.h).  The tracker's own frames never come through here: its index chain runs the same
// wave routine on the brute-force matches it finds itself (svo_track.hip, k_tg_fmat).
#include "svo_internal.h"
#include "svo_fmat_dev.h"

__global__ __launch_bounds__(64) void k_fmat_points(const double* pts1, const double* pts2, int n, double* F_out) {
  __shared__ EpnpWaveLds S;
  const int lane = threadIdx.x;
  uint32_t keep = 0;
  double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int i = lane + 64 * t;
    x1[t] = y1[t] = x2[t] = y2[t] = 0.0;
    if (i < n) {
      keep |= 1u << t;
      x1[t] = pts1[2 * i]; y1[t] = pts1[2 * i + 1]; x2[t] = pts2[2 * i]; y2[t] = pts2[2 * i + 1];
    }
  }
  double F[9];
  fmat8_wave(S, keep, x1, y1, x2, y2, F);
  if (lane < 9) {
    double v = F[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) v = lane == k ? F[k] : v;
    F_out[lane] = v;
  }
}

extern "C" int svo_fundamental_8point(svo_ctx* ctx, const double* pts1, const double* pts2, int n, double F[9]) {
  if (!ctx || !F || n < 0 || (n > 0 && (!pts1 || !pts2))) return SVO_E_INVALID;
  if (n > 512) return SVO_E_CAPACITY;   // one pair per keypoint at most (src/frame.cc:54: 500)
  hipSetDevice(ctx->device);
  const size_t pb = sizeof(double) * 2 * (size_t)(n > 0 ? n : 1);
  if (2 * pb + 128 > ctx->scratch_bytes) return SVO_E_CAPACITY;
  double* d1 = reinterpret_cast<double*>(ctx->d_scratch);
  double* d2 = d1 + 2 * (size_t)(n > 0 ? n : 1);
  double* dF = d2 + 2 * (size_t)(n > 0 ? n : 1);
  if (n > 0) {
    SVO_HIP(ctx, hipMemcpyAsync(d1, pts1, pb, hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(ctx, hipMemcpyAsync(d2, pts2, pb, hipMemcpyHostToDevice, ctx->stream));
  }
  {
    SvoTimer t(ctx, "k_fmat_points");
    hipLaunchKernelGGL(k_fmat_points, dim3(1), dim3(64), 0, ctx->stream, d1, d2, n, dF);
  }
  SVO_HIP(ctx, hipMemcpyAsync(F, dF, 72, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}
