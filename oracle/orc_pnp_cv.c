/* orc_pnp_cv.c - CPU restatement of cv::solvePnPRansac as the reference calls it
 * (src/pnpmatch.cc:227: solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 8.0, 0.99, inliers)).
 *
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 *
 * The algorithm lives in a third-party dependency that is NOT under /root/reference: OpenCV, pinned to 3.2 by the
 * `NEEDED libopencv_*.so.3.2` entries of the reference's prebuilt Stereo/stereo_kitti.  OpenCV
 * is not on this machine, so what follows restates the published 3.2 sources [upstream-memory]; PARITY UNPINNED.
 *   modules/calib3d/src/solvepnp.cpp   solvePnPRansac, PnPRansacCallback (runKernel = solvePnP(..., SOLVEPNP_EPNP) on
 *                                      5-point minimal sets, computeError = squared reprojection error in float)
 *   modules/calib3d/src/ptsetreg.cpp   RANSACPointSetRegistrator::run / getSubset / findInliers, RANSACUpdateNumIters
 *   modules/calib3d/src/epnp.cpp       epnp::compute_pose and everything it calls (Lepetit, Moreno-Noguer, Fua 2009)
 *   modules/core/src/lapack.cpp        JacobiSVDImpl_ (one-sided Jacobi), SVBkSbImpl_ (cvSolve / cvInvert with CV_SVD)
 *   modules/core/include/.../core.hpp  cv::RNG (multiply-with-carry, coefficient 4164903690)
 *   modules/calib3d/src/calibration.cpp cvRodrigues2, cvProjectPoints2 (no distortion)
 * Semantics restated:
 *   - NO extrinsic guess: every hypothesis is solved from its five points alone (EPnP).
 *   - RNG rng((uint64)-1) is constructed inside run(), i.e. the sample sequence is the same for every call with the same
 *     point count; getSubset redraws an index until it differs from the ones already drawn.
 *   - a hypothesis replaces the best one iff its inlier count is larger (and > 4); the iteration bound is then lowered
 *     to log(1 - 0.99) / log(1 - w^5), w = inlier ratio (RANSACUpdateNumIters), so good data stops after a handful
 *     of hypotheses.
 *   - inliers: squared reprojection error, evaluated in float as PnPRansacCallback::computeError does, <= 8^2.
 *   - output: OpenCV 3.2 returns the best MINIMAL-SET model (`_local_model`); it also runs solvePnP(ITERATIVE) on the
 *     inliers but discards that pose (the refit only reaches the caller from 3.3 on).  `refine` != 0 selects the later
 *     behaviour here: a Gauss-Newton refit of the reprojection error on the inliers, started from the RANSAC model.
 *   - the image points go through undistortPoints with zero distortion; EPnP on pixel coordinates with K is the same
 *     estimator (for fx == fy the two differ by a uniform scaling of M^T M), which is what is restated.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

/* ---- cv::RNG ------------------------------------------------------------------------------- */
typedef struct { uint64_t state; } cvrng_t;
static unsigned cvrng_next(cvrng_t* r) {
  r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
  return (unsigned)r->state;
}
static int cvrng_uniform(cvrng_t* r, int a, int b) { return a == b ? a : (int)(cvrng_next(r) % (unsigned)(b - a) + a); }

/* ---- cv::hypot (lapack.cpp, the template JacobiSVDImpl_ calls - not libm's): scaled, IEEE operations only ------ */
static double cv_hypot(double a, double b) {
  a = fabs(a);
  b = fabs(b);
  if (a > b) {
    b /= a;
    return a * sqrt(1 + b * b);
  }
  if (b > 0) {
    a /= b;
    return b * sqrt(1 + a * a);
  }
  return 0;
}

/* ---- JacobiSVDImpl_<double> (lapack.cpp): one-sided Jacobi on the n rows (length m) of At; Vt n x n ---------- */
static void jacobi_svd(double* At, int astep, double* _W, double* Vt, int vstep, int m, int n, int n1) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  double W[16];
  int i, j, k, iter, max_iter = m > 30 ? m : 30;
  double c, s, sd;
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) { double t = At[i * astep + k]; sd += t * t; }
    W[i] = sd;
    if (Vt) {
      for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
      Vt[i * vstep + i] = 1;
    }
  }
  for (iter = 0; iter < max_iter; iter++) {
    int changed = 0;
    for (i = 0; i < n - 1; i++)
      for (j = i + 1; j < n; j++) {
        double *Ai = At + i * astep, *Aj = At + j * astep;
        double a = W[i], p = 0, b = W[j];
        for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = cv_hypot(p, beta);
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (k = 0; k < m; k++) {
          double t0 = c * Ai[k] + s * Aj[k];
          double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = 1;
        if (Vt) {
          double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
          for (k = 0; k < n; k++) {
            double t0 = c * Vi[k] + s * Vj[k];
            double t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0; Vj[k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) { double t = At[i * astep + k]; sd += t * t; }
    W[i] = sqrt(sd);
  }
  for (i = 0; i < n - 1; i++) {
    j = i;
    for (k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
      if (Vt) {
        for (k = 0; k < m; k++) { t = At[i * astep + k]; At[i * astep + k] = At[j * astep + k]; At[j * astep + k] = t; }
        for (k = 0; k < n; k++) { t = Vt[i * vstep + k]; Vt[i * vstep + k] = Vt[j * vstep + k]; Vt[j * vstep + k] = t; }
      }
    }
  }
  for (i = 0; i < n; i++) _W[i] = W[i];
  if (!Vt) return;
  {
    cvrng_t rng = {0x12345678};
    for (i = 0; i < n1; i++) {
      sd = i < n ? W[i] : 0;
      for (int ii = 0; ii < 100 && sd <= minval; ii++) {
        /* a zero singular value: a random vector, orthogonalised against the rows found so far */
        const double val0 = 1. / m;
        for (k = 0; k < m; k++) At[i * astep + k] = (cvrng_next(&rng) & 256) != 0 ? val0 : -val0;
        for (iter = 0; iter < 2; iter++) {
          for (j = 0; j < i; j++) {
            sd = 0;
            for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
            double asum = 0;
            for (k = 0; k < m; k++) {
              double t = At[i * astep + k] - sd * At[j * astep + k];
              At[i * astep + k] = t;
              asum += fabs(t);
            }
            asum = asum > eps * 100 ? 1 / asum : 0;
            for (k = 0; k < m; k++) At[i * astep + k] *= asum;
          }
          sd = 0;
          for (k = 0; k < m; k++) { double t = At[i * astep + k]; sd += t * t; }
          sd = sqrt(sd);
        }
      }
      s = sd > minval ? 1 / sd : 0.;
      for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
  }
}

/* cv::SVD::compute for an m x n matrix with m >= n (row-major A): w[n], Ut = rows u_i (n x m), Vt = rows v_i (n x n). */
static void svd_compute(const double* A, int m, int n, double* w, double* Ut, double* Vt) {
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < m; ++k) Ut[i * m + k] = A[k * n + i];   /* temp_a = A^T */
  jacobi_svd(Ut, m, w, Vt, n, m, n, n);
}
/* SVBkSbImpl_: x = V diag(1/w) U^T b (singular values <= 2 eps sum(w) dropped); b == NULL: b = identity (m x m). */
static void svd_backsubst(int m, int n, const double* w, const double* Ut, const double* Vt, const double* b, int nb,
                          double* x) {
  double threshold = 0;
  if (!b) nb = m;
  for (int i = 0; i < n * nb; ++i) x[i] = 0;
  for (int i = 0; i < n; ++i) threshold += w[i];
  threshold *= DBL_EPSILON * 2;
  for (int i = 0; i < n; ++i) {
    double wi = w[i];
    if (fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    for (int c = 0; c < nb; ++c) {
      double s = 0;
      if (b) for (int j = 0; j < m; ++j) s += Ut[i * m + j] * b[j * nb + c];
      else s = Ut[i * m + c];
      s *= wi;
      for (int j = 0; j < n; ++j) x[j * nb + c] = x[j * nb + c] + s * Vt[i * n + j];
    }
  }
}
/* cvSolve(A, b, x, CV_SVD) for an m x n system (m >= n, one right-hand side) */
static void solve_svd(const double* A, int m, int n, const double* b, double* x) {
  double w[8], Ut[8 * 8], Vt[8 * 8];
  svd_compute(A, m, n, w, Ut, Vt);
  svd_backsubst(m, n, w, Ut, Vt, b, 1, x);
}

/* ---- cvRodrigues2 ----------------------------------------------------------------------------- */
static void rodrigues_to_matrix(const double r_in[3], double R[9]) {
  double r[3] = {r_in[0], r_in[1], r_in[2]};
  const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < DBL_EPSILON) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1. : 0.;
    return;
  }
  const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
  r[0] *= itheta; r[1] *= itheta; r[2] *= itheta;
  const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1], r[1] * r[2],
                         r[0] * r[2], r[1] * r[2], r[2] * r[2]};
  const double r_x[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  for (int i = 0; i < 9; ++i) R[i] = c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i] + s * r_x[i];
}
static void rodrigues_to_vector(const double Rin[9], double r[3]) {
  double w[3], Ut[9], Vt[9], R[9];
  svd_compute(Rin, 3, 3, w, Ut, Vt);           /* R = U * Vt: the closest rotation */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = Ut[0 * 3 + i] * Vt[0 * 3 + j] + Ut[1 * 3 + i] * Vt[1 * 3 + j] + Ut[2 * 3 + i] * Vt[2 * 3 + j];
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  const double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) { rx = ry = rz = 0; }
    else {
      double t = (R[0] + 1) * 0.5;
      rx = sqrt(t > 0. ? t : 0.);
      t = (R[4] + 1) * 0.5;
      ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5;
      rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
      if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      const double nr = sqrt(rx * rx + ry * ry + rz * rz);
      const double k = theta / nr;
      rx *= k; ry *= k; rz *= k;
    }
  } else {
    const double vth = 1 / (2 * s) * theta;
    rx *= vth; ry *= vth; rz *= vth;
  }
  r[0] = rx; r[1] = ry; r[2] = rz;
}

/* ---- epnp (epnp.cpp) --------------------------------------------------------------------------- */
typedef struct {
  double uc, vc, fu, fv;
  int n;
  double pws[5 * 3], us[5 * 2], alphas[5 * 4], pcs[5 * 3];
  double cws[4][3], ccs[4][3];
} epnp_t;

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double* p1, const double* p2) {
  return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

static void choose_control_points(epnp_t* e) {
  const int n = e->n;
  e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) e->cws[0][j] += e->pws[3 * i + j];
  for (int j = 0; j < 3; j++) e->cws[0][j] /= n;
  double PW0[5 * 3], pw0tpw0[9], dc[3], uct[9], vt[9];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) PW0[3 * i + j] = e->pws[3 * i + j] - e->cws[0][j];
  for (int a = 0; a < 3; ++a)                       /* cvMulTransposed(PW0, &PW0tPW0, 1): PW0^T PW0 */
    for (int b = 0; b < 3; ++b) {
      double s = 0;
      for (int i = 0; i < n; ++i) s += PW0[3 * i + a] * PW0[3 * i + b];
      pw0tpw0[3 * a + b] = s;
    }
  svd_compute(pw0tpw0, 3, 3, dc, uct, vt);          /* cvSVD(..., CV_SVD_MODIFY_A | CV_SVD_U_T) */
  for (int i = 1; i < 4; i++) {
    const double k = sqrt(dc[i - 1] / n);
    for (int j = 0; j < 3; j++) e->cws[i][j] = e->cws[0][j] + k * uct[3 * (i - 1) + j];
  }
}

static void compute_barycentric_coordinates(epnp_t* e) {
  double cc[9], cc_inv[9], w[3], Ut[9], Vt[9];
  for (int i = 0; i < 3; i++)
    for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
  svd_compute(cc, 3, 3, w, Ut, Vt);                 /* cvInvert(&CC, &CC_inv, CV_SVD) */
  svd_backsubst(3, 3, w, Ut, Vt, NULL, 3, cc_inv);
  const double* ci = cc_inv;
  for (int i = 0; i < e->n; i++) {
    const double* pi = e->pws + 3 * i;
    double* a = e->alphas + 4 * i;
    for (int j = 0; j < 3; j++)
      a[1 + j] = ci[3 * j] * (pi[0] - e->cws[0][0]) + ci[3 * j + 1] * (pi[1] - e->cws[0][1]) +
                 ci[3 * j + 2] * (pi[2] - e->cws[0][2]);
    a[0] = 1.0f - a[1] - a[2] - a[3];
  }
}

static void fill_M(const epnp_t* e, double* M, int row, const double* as, double u, double v) {
  double* M1 = M + row * 12;
  double* M2 = M1 + 12;
  for (int i = 0; i < 4; i++) {
    M1[3 * i] = as[i] * e->fu; M1[3 * i + 1] = 0.0; M1[3 * i + 2] = as[i] * (e->uc - u);
    M2[3 * i] = 0.0; M2[3 * i + 1] = as[i] * e->fv; M2[3 * i + 2] = as[i] * (e->vc - v);
  }
}

static void compute_L_6x10(const double* ut, double* l_6x10) {
  const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
  double dv[4][6][3];
  for (int i = 0; i < 4; i++) {
    int a = 0, b = 1;
    for (int j = 0; j < 6; j++) {
      dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
      dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
      dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
      b++;
      if (b > 3) { a++; b = a + 1; }
    }
  }
  for (int i = 0; i < 6; i++) {
    double* row = l_6x10 + 10 * i;
    row[0] = dot3(dv[0][i], dv[0][i]);
    row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
    row[2] = dot3(dv[1][i], dv[1][i]);
    row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
    row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
    row[5] = dot3(dv[2][i], dv[2][i]);
    row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
    row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
    row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
    row[9] = dot3(dv[3][i], dv[3][i]);
  }
}
static void compute_rho(const epnp_t* e, double* rho) {
  rho[0] = dist2(e->cws[0], e->cws[1]); rho[1] = dist2(e->cws[0], e->cws[2]); rho[2] = dist2(e->cws[0], e->cws[3]);
  rho[3] = dist2(e->cws[1], e->cws[2]); rho[4] = dist2(e->cws[1], e->cws[3]); rho[5] = dist2(e->cws[2], e->cws[3]);
}

/* betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44] */
static void find_betas_approx_1(const double* L, const double* rho, double* betas) {   /* [B11 B12 B13 B14] */
  double l[6 * 4], b4[4];
  for (int i = 0; i < 6; i++) { l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6]; }
  solve_svd(l, 6, 4, rho, b4);
  if (b4[0] < 0) {
    betas[0] = sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0];
  } else {
    betas[0] = sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0];
  }
}
static void find_betas_approx_2(const double* L, const double* rho, double* betas) {   /* [B11 B12 B22] */
  double l[6 * 3], b3[3];
  for (int i = 0; i < 6; i++) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
  solve_svd(l, 6, 3, rho, b3);
  if (b3[0] < 0) { betas[0] = sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
  else { betas[0] = sqrt(b3[0]); betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
  if (b3[1] < 0) betas[0] = -betas[0];
  betas[2] = 0.0; betas[3] = 0.0;
}
static void find_betas_approx_3(const double* L, const double* rho, double* betas) {   /* [B11 B12 B22 B13 B23] */
  double l[6 * 5], b5[5];
  for (int i = 0; i < 6; i++)
    for (int c = 0; c < 5; ++c) l[5 * i + c] = L[10 * i + c];
  solve_svd(l, 6, 5, rho, b5);
  if (b5[0] < 0) { betas[0] = sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
  else { betas[0] = sqrt(b5[0]); betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
  if (b5[1] < 0) betas[0] = -betas[0];
  betas[2] = b5[3] / betas[0];
  betas[3] = 0.0;
}

static void compute_A_and_b_gauss_newton(const double* l_6x10, const double* rho, const double betas[4], double* A, double* b) {
  for (int i = 0; i < 6; i++) {
    const double* rowL = l_6x10 + i * 10;
    double* rowA = A + i * 4;
    rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
    rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
    rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
    rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
    b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                     rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                     rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                     rowL[9] * betas[3] * betas[3]);
  }
}
/* epnp::qr_solve, literally (including its `eta` scan, which starts at the diagonal element twice and never looks at
 * the last row) */
static void qr_solve(double* pA, double* pb, double* pX, int nr, int nc) {
  double A1[6], A2[6];
  double* ppAkk = pA;
  for (int k = 0; k < nc; k++) {
    double* ppAik1 = ppAkk;
    double eta = fabs(*ppAik1);
    for (int i = k + 1; i < nr; i++) {
      const double elt = fabs(*ppAik1);
      if (eta < elt) eta = elt;
      ppAik1 += nc;
    }
    if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
    double* ppAik2 = ppAkk;
    double sum2 = 0.0;
    const double inv_eta = 1. / eta;
    for (int i = k; i < nr; i++) {
      *ppAik2 *= inv_eta;
      sum2 += *ppAik2 * *ppAik2;
      ppAik2 += nc;
    }
    double sigma = sqrt(sum2);
    if (*ppAkk < 0) sigma = -sigma;
    *ppAkk += sigma;
    A1[k] = sigma * *ppAkk;
    A2[k] = -eta * sigma;
    for (int j = k + 1; j < nc; j++) {
      double* ppAik = ppAkk;
      double sum = 0;
      for (int i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
      const double tau = sum / A1[k];
      ppAik = ppAkk;
      for (int i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
    }
    ppAkk += nc + 1;
  }
  double* ppAjj = pA;                               /* b <- Qt b */
  for (int j = 0; j < nc; j++) {
    double* ppAij = ppAjj;
    double tau = 0;
    for (int i = j; i < nr; i++) { tau += *ppAij * pb[i]; ppAij += nc; }
    tau /= A1[j];
    ppAij = ppAjj;
    for (int i = j; i < nr; i++) { pb[i] -= tau * *ppAij; ppAij += nc; }
    ppAjj += nc + 1;
  }
  pX[nc - 1] = pb[nc - 1] / A2[nc - 1];             /* X = R-1 b */
  for (int i = nc - 2; i >= 0; i--) {
    const double* ppAij = pA + i * nc + (i + 1);
    double sum = 0;
    for (int j = i + 1; j < nc; j++) { sum += *ppAij * pX[j]; ppAij++; }
    pX[i] = (pb[i] - sum) / A2[i];
  }
}
static void gauss_newton(const double* L_6x10, const double* rho, double betas[4]) {
  double a[6 * 4], b[6], x[4];
  for (int k = 0; k < 5; k++) {
    compute_A_and_b_gauss_newton(L_6x10, rho, betas, a, b);
    x[0] = x[1] = x[2] = x[3] = 0;
    qr_solve(a, b, x, 6, 4);
    for (int i = 0; i < 4; i++) betas[i] += x[i];
  }
}

static void estimate_R_and_t(epnp_t* e, double R[3][3], double t[3]) {
  const int n = e->n;
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
  for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
  double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, d[3], Ut[9], Vt[9];
  for (int i = 0; i < n; i++) {
    const double* pc = e->pcs + 3 * i;
    const double* pw = e->pws + 3 * i;
    for (int j = 0; j < 3; j++) {
      abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
      abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
      abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
    }
  }
  svd_compute(abt, 3, 3, d, Ut, Vt);                /* cvSVD(&ABt, &ABt_D, &ABt_U, &ABt_V, CV_SVD_MODIFY_A): U, V as matrices */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)                     /* R[i][j] = dot(row i of U, row j of V) */
      R[i][j] = Ut[0 * 3 + i] * Vt[0 * 3 + j] + Ut[1 * 3 + i] * Vt[1 * 3 + j] + Ut[2 * 3 + i] * Vt[2 * 3 + j];
  const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                     R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
  if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
  t[0] = pc0[0] - dot3(R[0], pw0); t[1] = pc0[1] - dot3(R[1], pw0); t[2] = pc0[2] - dot3(R[2], pw0);
}
static double reprojection_error(const epnp_t* e, double R[3][3], const double t[3]) {
  double sum2 = 0.0;
  for (int i = 0; i < e->n; i++) {
    const double* pw = e->pws + 3 * i;
    const double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1], inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
    const double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
    const double u = e->us[2 * i], v = e->us[2 * i + 1];
    sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  return sum2 / e->n;
}
static double compute_R_and_t(epnp_t* e, const double* ut, const double* betas, double R[3][3], double t[3]) {
  for (int i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0;      /* compute_ccs */
  for (int i = 0; i < 4; i++) {
    const double* v = ut + 12 * (11 - i);
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3 * j + k];
  }
  for (int i = 0; i < e->n; i++) {                                                     /* compute_pcs */
    const double* a = e->alphas + 4 * i;
    double* pc = e->pcs + 3 * i;
    for (int j = 0; j < 3; j++) pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
  }
  if (e->pcs[2] < 0.0) {                                                               /* solve_for_sign */
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
    for (int i = 0; i < e->n; i++) { e->pcs[3 * i] = -e->pcs[3 * i]; e->pcs[3 * i + 1] = -e->pcs[3 * i + 1]; e->pcs[3 * i + 2] = -e->pcs[3 * i + 2]; }
  }
  estimate_R_and_t(e, R, t);
  return reprojection_error(e, R, t);
}

double orc_epnp_last_rep[3];   /* reprojection errors of the three candidates of the last orc_epnp5 call (tests) */
/* epnp::compute_pose on n = 5 correspondences (object points as floats, image points as floats, K as doubles of floats) */
void orc_epnp5(const double Xw5[15], const double uv5[10], const double K[4], double R_out[9], double t_out[3]) {
  epnp_t e;
  e.fu = K[0]; e.fv = K[1]; e.uc = K[2]; e.vc = K[3];
  e.n = 5;
  memcpy(e.pws, Xw5, sizeof e.pws);
  memcpy(e.us, uv5, sizeof e.us);
  choose_control_points(&e);
  compute_barycentric_coordinates(&e);
  double M[10 * 12], mtm[144], d[12], ut[144], vt[144];
  for (int i = 0; i < 5; i++) fill_M(&e, M, 2 * i, e.alphas + 4 * i, e.us[2 * i], e.us[2 * i + 1]);
  for (int a = 0; a < 12; ++a)                      /* cvMulTransposed(M, &MtM, 1) */
    for (int b = 0; b < 12; ++b) {
      double s = 0;
      for (int r = 0; r < 10; ++r) s += M[12 * r + a] * M[12 * r + b];
      mtm[12 * a + b] = s;
    }
  svd_compute(mtm, 12, 12, d, ut, vt);              /* cvSVD(&MtM, &D, &Ut, 0, CV_SVD_MODIFY_A | CV_SVD_U_T) */
  double l_6x10[60], rho[6];
  compute_L_6x10(ut, l_6x10);
  compute_rho(&e, rho);
  double Betas[4][4], rep_errors[4], Rs[4][3][3], ts[4][3];
  find_betas_approx_1(l_6x10, rho, Betas[1]);
  gauss_newton(l_6x10, rho, Betas[1]);
  rep_errors[1] = compute_R_and_t(&e, ut, Betas[1], Rs[1], ts[1]);
  find_betas_approx_2(l_6x10, rho, Betas[2]);
  gauss_newton(l_6x10, rho, Betas[2]);
  rep_errors[2] = compute_R_and_t(&e, ut, Betas[2], Rs[2], ts[2]);
  find_betas_approx_3(l_6x10, rho, Betas[3]);
  gauss_newton(l_6x10, rho, Betas[3]);
  rep_errors[3] = compute_R_and_t(&e, ut, Betas[3], Rs[3], ts[3]);
  int N = 1;
  if (rep_errors[2] < rep_errors[1]) N = 2;
  if (rep_errors[3] < rep_errors[N]) N = 3;
  orc_epnp_last_rep[0] = rep_errors[1]; orc_epnp_last_rep[1] = rep_errors[2]; orc_epnp_last_rep[2] = rep_errors[3];
  for (int i = 0; i < 3; ++i) {
    t_out[i] = ts[N][i];
    for (int j = 0; j < 3; ++j) R_out[3 * i + j] = Rs[N][i][j];
  }
}

/* ---- PnPRansacCallback::computeError + findInliers ------------------------------------------ */
/* projectPoints (double, rounded once to float) against the float image points; error and threshold in float */
static int find_inliers(const double* Xw, const double* obs, int n, const double K[4], const double rvec[3],
                        const double tvec[3], uint8_t* mask) {
  double R[9];
  rodrigues_to_matrix(rvec, R);
  const float thr = (float)(8.0 * 8.0);
  int good = 0;
  for (int i = 0; i < n; ++i) {
    const double X = Xw[3 * i], Y = Xw[3 * i + 1], Z = Xw[3 * i + 2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    const float px = (float)(x * K[0] + K[2]), py = (float)(y * K[1] + K[3]);
    const float dx = (float)obs[2 * i] - px, dy = (float)obs[2 * i + 1] - py;
    const float err = dx * dx + dy * dy;
    const int in = err <= thr;
    if (mask) mask[i] = (uint8_t)in;
    good += in;
  }
  return good;
}

static int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
  ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
  double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
  double denom = 1. - pow(1. - ep, modelPoints);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

/* reprojection Gauss-Newton on a subset (only with refine != 0: the behaviour of OpenCV >= 3.3 in spirit) */
static void refine_on_inliers(const double* Xw, const double* obs, int n, const uint8_t* mask, const double K[4], double R[9],
                              double t[3]) {
  for (int it = 0; it < 10; ++it) {
    double H[36], b[6];
    memset(H, 0, sizeof H); memset(b, 0, sizeof b);
    for (int i = 0; i < n; ++i) {
      if (!mask[i]) continue;
      const double* X = Xw + 3 * i;
      const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
                   z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
      const double iz = 1.0 / z, iz2 = iz * iz;
      const double e0 = obs[2 * i] - (x * iz * K[0] + K[2]), e1 = obs[2 * i + 1] - (y * iz * K[1] + K[3]);
      /* d(error)/d(omega, upsilon) for the left perturbation T <- exp(x) T (the tracker's own Jacobian layout) */
      const double J[12] = {x * y * iz2 * K[0], -(1 + x * x * iz2) * K[0], y * iz * K[0], -iz * K[0], 0, x * iz2 * K[0],
                            (1 + y * y * iz2) * K[1], -x * y * iz2 * K[1], -x * iz * K[1], 0, -iz * K[1], y * iz2 * K[1]};
      for (int r = 0; r < 6; ++r) {
        b[r] -= J[r] * e0 + J[6 + r] * e1;
        for (int c = 0; c < 6; ++c) H[6 * r + c] += J[r] * J[c] + J[6 + r] * J[6 + c];
      }
    }
    double x6[6];
    solve_svd(H, 6, 6, b, x6);
    double dR[9], Rn[9], tn[3];
    rodrigues_to_matrix(x6, dR);                    /* small-step update: R <- exp(w) R, t <- exp(w) t + v */
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) Rn[3 * r + c] = dR[3 * r] * R[c] + dR[3 * r + 1] * R[3 + c] + dR[3 * r + 2] * R[6 + c];
      tn[r] = dR[3 * r] * t[0] + dR[3 * r + 1] * t[1] + dR[3 * r + 2] * t[2] + x6[3 + r];
    }
    memcpy(R, Rn, sizeof Rn); memcpy(t, tn, sizeof tn);
    double mx = 0;
    for (int r = 0; r < 6; ++r) mx = fabs(x6[r]) > mx ? fabs(x6[r]) : mx;
    if (mx < 1e-10) break;
  }
}

/* cv::solvePnPRansac(pts3d, pts2d, K, noDist, rvec, tvec, false, 100, 8.0, 0.99, inliers).
 * Xw: n x 3 (float values), obs: n x 2 (float values), K = {fx, fy, cx, cy} (float values).  T_fallback: what T gets
 * when OpenCV returns false (fewer than 5 points, no consensus of at least 5) - the reference would go on with
 * uninitialised rvec / tvec there.  rng_state: 0 = (uint64)-1, OpenCV's.  Returns 1 on success. */
int orc_solvepnp_ransac(const double* Xw, const double* obs, int n, const double K[4], const double T_fallback[16],
                        uint64_t rng_state, int refine, double T[16], uint8_t* inlier_mask, orc_pnp_stats* stats) {
  const int modelPoints = 5;
  int niters = 100, maxGoodCount = 0, best_iter = -1, iters_run = 0;
  double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
  uint8_t* mask = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
  uint8_t* bestMask = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
  cvrng_t rng = {rng_state ? rng_state : (uint64_t)-1};
  if (n >= modelPoints) {
    for (int iter = 0; iter < niters; iter++) {
      ++iters_run;
      int idx[5];
      if (n > modelPoints) {                        /* getSubset */
        for (int i = 0; i < modelPoints; ++i) {
          for (;;) {
            const int c = idx[i] = cvrng_uniform(&rng, 0, n);
            int j = 0;
            for (; j < i; j++)
              if (c == idx[j]) break;
            if (j == i) break;
          }
        }
      } else {
        for (int i = 0; i < modelPoints; ++i) idx[i] = i;
      }
      double X5[15], u5[10], R[9], t[3], rvec[3];
      for (int i = 0; i < 5; ++i) {
        memcpy(X5 + 3 * i, Xw + 3 * idx[i], 3 * sizeof(double));
        memcpy(u5 + 2 * i, obs + 2 * idx[i], 2 * sizeof(double));
      }
      orc_epnp5(X5, u5, K, R, t);                   /* runKernel: solvePnP(..., SOLVEPNP_EPNP), then Rodrigues(R, rvec) */
      rodrigues_to_vector(R, rvec);
      if (!(isfinite(rvec[0]) && isfinite(rvec[1]) && isfinite(rvec[2]) && isfinite(t[0]) && isfinite(t[1]) && isfinite(t[2])))
        continue;
      if (n == modelPoints) {                       /* run(): count == modelPoints -> the model, mask all ones */
        memset(bestMask, 1, (size_t)n);
        memcpy(best_r, rvec, sizeof best_r); memcpy(best_t, t, sizeof best_t);
        maxGoodCount = n; best_iter = 0;
        break;
      }
      const int goodCount = find_inliers(Xw, obs, n, K, rvec, t, mask);
      if (goodCount > (maxGoodCount > modelPoints - 1 ? maxGoodCount : modelPoints - 1)) {
        uint8_t* tmp = mask; mask = bestMask; bestMask = tmp;
        memcpy(best_r, rvec, sizeof best_r); memcpy(best_t, t, sizeof best_t);
        maxGoodCount = goodCount;
        best_iter = iter;
        niters = ransac_update_num_iters(0.99, (double)(n - goodCount) / n, modelPoints, niters);
      }
    }
  }
  const int ok = maxGoodCount > 0;
  if (ok) {
    double R[9];
    rodrigues_to_matrix(best_r, R);                 /* pnpmatch.cc:238 Rodrigues(rvec, Cur_Rcw) */
    if (refine) refine_on_inliers(Xw, obs, n, bestMask, K, R, best_t);
    T[0] = R[0]; T[1] = R[1]; T[2] = R[2]; T[3] = best_t[0];
    T[4] = R[3]; T[5] = R[4]; T[6] = R[5]; T[7] = best_t[1];
    T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = best_t[2];
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
    if (inlier_mask) memcpy(inlier_mask, bestMask, (size_t)n);
  } else {
    memcpy(T, T_fallback, 16 * sizeof(double));
    if (inlier_mask && n > 0) memset(inlier_mask, 0, (size_t)n);
  }
  if (stats) {
    stats->n_points = n; stats->n_inliers = ok ? maxGoodCount : 0; stats->best_hypothesis = ok ? best_iter : -1;
    stats->ok = ok; stats->iterations = iters_run;
  }
  free(mask); free(bestMask);
  return ok;
}

int orc_pnp_ransac(const double* Xw, const double* obs, int n, const double K[4], const double T_fallback[16],
                   uint64_t rng_state, double T[16], uint8_t* inlier_mask, orc_pnp_stats* stats) {
  return orc_solvepnp_ransac(Xw, obs, n, K, T_fallback, rng_state, 0, T, inlier_mask, stats);
}
