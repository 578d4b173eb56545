// svo_fmat.hip - normalised 8-point fundamental matrix (host side, float64).
// Stands in for cv::findFundamentalMat(cur_pts, last_pts, CV_FM_8POINT) in
// pnpmatch::poseEstimation2D_2D (reference src/pnpmatch.cc:302-337).  It is a 9x9 symmetric
// eigenproblem over <= 500 point pairs, evaluated once per frame and only when the frame carries
// detection boxes, so it runs on the host between two device phases of svo_track_frame.
#include <math.h>
#include <string.h>

#include "svo_internal.h"

namespace {
// symmetric eigen-decomposition by cyclic Jacobi rotations; columns of V are eigenvectors
template <int N>
void sym_eig(double (&A)[N][N], double (&w)[N], double (&V)[N][N]) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) V[i][j] = i == j;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) off += A[p][q] * A[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = A[p][q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
        for (int k = 0; k < N; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
        for (int k = 0; k < N; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
      }
  }
  for (int i = 0; i < N; ++i) w[i] = A[i][i];
}
}  // namespace

extern "C" int svo_fundamental_8point(const double* pts1, const double* pts2, int n, double F[9]) {
  if (!F || n < 0 || (n > 0 && (!pts1 || !pts2))) return SVO_E_INVALID;
  memset(F, 0, 9 * sizeof(double));
  if (n < 8) return SVO_OK;   // cv returns an empty matrix; F = 0 makes every gate distance NaN (no veto)
  double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
  for (int i = 0; i < n; ++i) { c1x += pts1[2 * i]; c1y += pts1[2 * i + 1]; c2x += pts2[2 * i]; c2y += pts2[2 * i + 1]; }
  c1x /= n; c1y /= n; c2x /= n; c2y /= n;
  double d1 = 0, d2 = 0;
  for (int i = 0; i < n; ++i) {
    d1 += hypot(pts1[2 * i] - c1x, pts1[2 * i + 1] - c1y);
    d2 += hypot(pts2[2 * i] - c2x, pts2[2 * i + 1] - c2y);
  }
  d1 /= n; d2 /= n;
  if (d1 < 2.220446049250313e-16 || d2 < 2.220446049250313e-16) return SVO_OK;
  const double s1 = sqrt(2.0) / d1, s2 = sqrt(2.0) / d2;
  double A[9][9] = {};
  for (int i = 0; i < n; ++i) {
    const double x1 = (pts1[2 * i] - c1x) * s1, y1 = (pts1[2 * i + 1] - c1y) * s1;
    const double x2 = (pts2[2 * i] - c2x) * s2, y2 = (pts2[2 * i + 1] - c2y) * s2;
    const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) A[a][b] += r[a] * r[b];
  }
  double w[9], V[9][9];
  sym_eig<9>(A, w, V);
  int kmin = 0;
  for (int k = 1; k < 9; ++k) if (w[k] < w[kmin]) kmin = k;
  double F0[3][3];
  for (int k = 0; k < 9; ++k) F0[k / 3][k % 3] = V[k][kmin];
  // rank 2: F0 (I - v3 v3^T), v3 = right singular vector of the smallest singular value
  double G[3][3], gw[3], GV[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) G[a][b] = F0[0][a] * F0[0][b] + F0[1][a] * F0[1][b] + F0[2][a] * F0[2][b];
  sym_eig<3>(G, gw, GV);
  int gmin = 0;
  for (int k = 1; k < 3; ++k) if (gw[k] < gw[gmin]) gmin = k;
  const double v3[3] = {GV[0][gmin], GV[1][gmin], GV[2][gmin]};
  double F1[3][3];
  for (int r = 0; r < 3; ++r) {
    const double dot = F0[r][0] * v3[0] + F0[r][1] * v3[1] + F0[r][2] * v3[2];
    for (int c = 0; c < 3; ++c) F1[r][c] = F0[r][c] - dot * v3[c];
  }
  const double T1[3][3] = {{s1, 0, -s1 * c1x}, {0, s1, -s1 * c1y}, {0, 0, 1}};
  const double T2[3][3] = {{s2, 0, -s2 * c2x}, {0, s2, -s2 * c2y}, {0, 0, 1}};
  double M[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M[r][c] = F1[r][0] * T1[0][c] + F1[r][1] * T1[1][c] + F1[r][2] * T1[2][c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) F[3 * r + c] = T2[0][r] * M[0][c] + T2[1][r] * M[1][c] + T2[2][r] * M[2][c];
  if (fabs(F[8]) > 1.1920929e-07) {
    const double s = 1.0 / F[8];
    for (int k = 0; k < 9; ++k) F[k] *= s;
    F[8] = 1.0;
  }
  return SVO_OK;
}
