// svo_fmat_dev.h - normalised 8-point fundamental matrix on ONE WAVE (gfx950, float64).
//
// Stands in for cv::findFundamentalMat(cur_pts, last_pts, CV_FM_8POINT) in pnpmatch::poseEstimation2D_2D (reference
// src/pnpmatch.cc:302-337 -> OpenCV 3.2 modules/calib3d/src/fundam.cpp run8Point): isotropic normalisation of both
// point sets (centroid, mean distance sqrt 2), the 9 x 9 normal matrix of the epipolar constraints, the eigenvector of
// its smallest eigenvalue, rank-2 projection, de-normalisation, F(2,2) = 1.
//
// Mapping to a wavefront: lane l holds the point pairs l, l + 64, ... (eight at most: <= 512 keypoints); centroids, mean
// distances and the 45 distinct entries of the normal matrix are per-lane partial sums folded by DPP wave reductions; the
// eigenproblem is padded to 12 x 12 (three decoupled diagonal entries) and handed to the wave-parallel Jacobi of
// svo_epnp_dev.h; the 3 x 3 rank-2 step is the scalar one-sided Jacobi SVD of the same header.  The result is the same
// in every lane.
#pragma once
#include "svo_epnp_dev.h"
#include "svo_wave.h"

// keep: bit t set = pair (lane + 64 t) takes part.  x1, y1: the current frame's point, x2, y2: the last frame's
// (p2^T F p1 = 0, as src/pnpmatch.cc:336 passes them).  Fewer than 8 pairs, or a degenerate set: F = 0 (OpenCV returns an
// empty matrix; a zero F makes every gate distance NaN: no veto).
__device__ inline void fmat8_wave(EpnpWaveLds& S, uint32_t keep, const double (&x1)[8], const double (&y1)[8],
                                  const double (&x2)[8], const double (&y2)[8], double F[9]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 9; ++k) F[k] = 0.0;
  const int n = wave_sum_i32_dpp(__popc(keep));
  if (n < 8) return;
  double s[4] = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if ((keep >> t) & 1u) { s[0] += x1[t]; s[1] += y1[t]; s[2] += x2[t]; s[3] += y2[t]; }
  const double c1x = wave_sum_f64_dpp(s[0]) / n, c1y = wave_sum_f64_dpp(s[1]) / n;
  const double c2x = wave_sum_f64_dpp(s[2]) / n, c2y = wave_sum_f64_dpp(s[3]) / n;
  double d1 = 0, d2 = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if ((keep >> t) & 1u) {
      d1 += sqrt((x1[t] - c1x) * (x1[t] - c1x) + (y1[t] - c1y) * (y1[t] - c1y));
      d2 += sqrt((x2[t] - c2x) * (x2[t] - c2x) + (y2[t] - c2y) * (y2[t] - c2y));
    }
  d1 = wave_sum_f64_dpp(d1) / n; d2 = wave_sum_f64_dpp(d2) / n;
  if (d1 < 2.220446049250313e-16 || d2 < 2.220446049250313e-16) return;
  const double s1 = sqrt(2.0) / d1, s2 = sqrt(2.0) / d2;
  // normal matrix: upper triangle, per-lane partial sums first
  double acc[45];
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if ((keep >> t) & 1u) {
      const double a1 = (x1[t] - c1x) * s1, b1 = (y1[t] - c1y) * s1, a2 = (x2[t] - c2x) * s2, b2 = (y2[t] - c2y) * s2;
      const double r[9] = {a2 * a1, a2 * b1, a2, b2 * a1, b2 * b1, b2, a1, b1, 1.0};
      int k = 0;
#pragma unroll
      for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int b = a; b < 9; ++b) acc[k++] += r[a] * r[b];
    }
  EPNP_WAVE_SYNC();
  {
    double dmax = 0;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int b = a; b < 9; ++b) {
        const double v = wave_sum_f64_dpp(acc[k++]);
        if (a == b) dmax = fmax(dmax, v);
        if (lane == 0) { S.A[12 * a + b] = v; S.A[12 * b + a] = v; }
      }
    // padding: three decoupled coordinates whose eigenvalue (the largest diagonal entry) is never the smallest
    for (int e = lane; e < 144; e += 64) {
      const int a = e / 12, b = e % 12;
      if (a >= 9 || b >= 9) S.A[e] = a == b ? dmax : 0.0;
      S.V[e] = a == b ? 1.0 : 0.0;
    }
  }
  EPNP_WAVE_SYNC();
  epnp_eig12_wave(S);
  EPNP_WAVE_SYNC();
  int kmin = 0;
  double wmin = S.A[0];
#pragma unroll
  for (int k = 1; k < 12; ++k) {
    const double w = S.A[13 * k];
    if (w < wmin) { wmin = w; kmin = k; }
  }
  double F0[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) F0[k] = S.V[12 * k + kmin];
  // rank 2: F0 (I - v3 v3^T), v3 = right singular vector of the smallest singular value
  double w3[3], Ut[9], Vt[9];
  epnp_svd3(F0, w3, Ut, Vt);
  const double v3[3] = {Vt[6], Vt[7], Vt[8]};
  double F1[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double dot = F0[3 * r] * v3[0] + F0[3 * r + 1] * v3[1] + F0[3 * r + 2] * v3[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) F1[3 * r + c] = F0[3 * r + c] - dot * v3[c];
  }
  // F = T2^T F1 T1,  T = [s 0 -s cx; 0 s -s cy; 0 0 1]
  const double T1[9] = {s1, 0, -s1 * c1x, 0, s1, -s1 * c1y, 0, 0, 1};
  const double T2[9] = {s2, 0, -s2 * c2x, 0, s2, -s2 * c2y, 0, 0, 1};
  double M[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[3 * r + c] = F1[3 * r] * T1[c] + F1[3 * r + 1] * T1[3 + c] + F1[3 * r + 2] * T1[6 + c];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) F[3 * r + c] = T2[r] * M[c] + T2[3 + r] * M[3 + c] + T2[6 + r] * M[6 + c];
  if (fabs(F[8]) > 1.1920929e-07) {
    const double f8 = F[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) F[k] /= f8;
    F[8] = 1.0;
  }
  EPNP_WAVE_SYNC();
}
