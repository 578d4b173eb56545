/* orc_fmat.c - CPU restatement of the fundamental-matrix stage used for semantic gating.
 *
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 *
 * pnpmatch::poseEstimation2D_2D (reference src/pnpmatch.cc:302-337): brute-force matches between
 * the current and the last frame (find_feature_matches, :253-300), matches whose CURRENT point
 * lies inside a detection box padded by 10 px are dropped (:318-328), and
 * cv::findFundamentalMat(cur_pts, last_pts, CV_FM_8POINT) gives F (:336).  OpenCV is absent, so
 * the normalised 8-point algorithm is restated from its published form [upstream-memory of
 * cv::run8Point]: isotropic normalisation (centroid, mean distance sqrt(2)), 9x9 normal matrix,
 * eigenvector of the smallest eigenvalue, rank-2 projection, de-normalisation, F(2,2) = 1.
 * PARITY UNPINNED (eigen/SVD round-off of OpenCV is not reproducible here); the gate only uses
 * point-to-line distances, which are invariant to the scale and sign of F.
 */
#include <math.h>
#include <string.h>

#include "svo_oracle.h"

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9): A = V diag(w) V^T */
static void jacobi_eig(double* A, int n, double* w, double* V) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[n * i + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) off += A[n * p + q] * A[n * p + q];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[n * p + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[n * q + q] - A[n * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[n * k + p], akq = A[n * k + q];
          A[n * k + p] = c * akp - s * akq;
          A[n * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[n * p + k], aqk = A[n * q + k];
          A[n * p + k] = c * apk - s * aqk;
          A[n * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[n * k + p], vkq = V[n * k + q];
          V[n * k + p] = c * vkp - s * vkq;
          V[n * k + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[n * i + i];
}

/* F (row-major 3x3) with  p2^T F p1 = 0,  p1 = pts1[i] (current frame), p2 = pts2[i] (last frame).
 * Returns 0 (and F = 0) with fewer than 8 points. */
int orc_fundamental_8point(const double* pts1, const double* pts2, int n, double F[9]) {
  memset(F, 0, 9 * sizeof(double));
  if (n < 8) return 0;
  double c1[2] = {0, 0}, c2[2] = {0, 0};
  for (int i = 0; i < n; ++i) { c1[0] += pts1[2 * i]; c1[1] += pts1[2 * i + 1]; c2[0] += pts2[2 * i]; c2[1] += pts2[2 * i + 1]; }
  c1[0] /= n; c1[1] /= n; c2[0] /= n; c2[1] /= n;
  double d1 = 0, d2 = 0;
  for (int i = 0; i < n; ++i) {
    d1 += sqrt((pts1[2 * i] - c1[0]) * (pts1[2 * i] - c1[0]) + (pts1[2 * i + 1] - c1[1]) * (pts1[2 * i + 1] - c1[1]));
    d2 += sqrt((pts2[2 * i] - c2[0]) * (pts2[2 * i] - c2[0]) + (pts2[2 * i + 1] - c2[1]) * (pts2[2 * i + 1] - c2[1]));
  }
  d1 /= n; d2 /= n;
  if (d1 < 2.220446049250313e-16 || d2 < 2.220446049250313e-16) return 0;
  const double s1 = sqrt(2.0) / d1, s2 = sqrt(2.0) / d2;
  double A[81];
  memset(A, 0, sizeof A);
  for (int i = 0; i < n; ++i) {
    const double x1 = (pts1[2 * i] - c1[0]) * s1, y1 = (pts1[2 * i + 1] - c1[1]) * s1;
    const double x2 = (pts2[2 * i] - c2[0]) * s2, y2 = (pts2[2 * i + 1] - c2[1]) * s2;
    const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) A[9 * a + b] += r[a] * r[b];
  }
  double w[9], V[81];
  jacobi_eig(A, 9, w, V);
  int kmin = 0;
  for (int k = 1; k < 9; ++k) if (w[k] < w[kmin]) kmin = k;
  double F0[9];
  for (int k = 0; k < 9; ++k) F0[k] = V[9 * k + kmin];
  /* rank-2 projection: remove the component along the smallest singular direction.
   * F0^T F0 = V S^2 V^T ; F0' = F0 (I - v3 v3^T) with v3 the eigenvector of the smallest eigenvalue */
  double G[9], gw[3], GV[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) G[3 * a + b] = F0[a] * F0[b] + F0[3 + a] * F0[3 + b] + F0[6 + a] * F0[6 + b];
  jacobi_eig(G, 3, gw, GV);
  int gmin = 0;
  for (int k = 1; k < 3; ++k) if (gw[k] < gw[gmin]) gmin = k;
  const double v3[3] = {GV[gmin], GV[3 + gmin], GV[6 + gmin]};
  double F1[9];
  for (int r = 0; r < 3; ++r) {
    const double dot = F0[3 * r] * v3[0] + F0[3 * r + 1] * v3[1] + F0[3 * r + 2] * v3[2];
    for (int c = 0; c < 3; ++c) F1[3 * r + c] = F0[3 * r + c] - dot * v3[c];
  }
  /* F = T2^T F1 T1,  T = [s 0 -s cx; 0 s -s cy; 0 0 1] */
  const double T1[9] = {s1, 0, -s1 * c1[0], 0, s1, -s1 * c1[1], 0, 0, 1};
  const double T2[9] = {s2, 0, -s2 * c2[0], 0, s2, -s2 * c2[1], 0, 0, 1};
  double M[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M[3 * r + c] = F1[3 * r] * T1[c] + F1[3 * r + 1] * T1[3 + c] + F1[3 * r + 2] * T1[6 + c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) F[3 * r + c] = T2[r] * M[c] + T2[3 + r] * M[3 + c] + T2[6 + r] * M[6 + c];
  if (fabs(F[8]) > 1.1920929e-07)
    for (int k = 0; k < 8; ++k) F[k] /= F[8], (void)0;
  if (fabs(F[8]) > 1.1920929e-07) F[8] = 1.0;
  return 1;
}

/* src/pnpmatch.cc:103-121: is the current point inside a box padded by `pad`? */
int orc_point_in_boxes(float x, float y, const int32_t* boxes, int n_boxes, int pad) {
  for (int k = 0; k < n_boxes; ++k) {
    const int left = boxes[4 * k], right = boxes[4 * k + 1], top = boxes[4 * k + 2], bottom = boxes[4 * k + 3];
    if (x > left - pad && x < right + pad && y > top - pad && y < bottom + pad) return 1;
  }
  return 0;
}

/* src/pnpmatch.cc:110-114: distance of `cur` to the line F * [last.x, last.y, 1] (as written in
 * the reference - F is applied to the LAST point). */
double orc_epipolar_distance(const double F[9], float last_x, float last_y, float cur_x, float cur_y) {
  const double A = F[0] * last_x + F[1] * last_y + F[2];
  const double B = F[3] * last_x + F[4] * last_y + F[5];
  const double C = F[6] * last_x + F[7] * last_y + F[8];
  return fabs(A * cur_x + B * cur_y + C) / sqrt(A * A + B * B);
}
