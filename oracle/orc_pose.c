/* orc_pose.c - CPU restatement of the pose stages (float64 throughout).
 *
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 *
 * orc_pose_opt restates Optimizer::PoseOptimization (reference src/Optimizer.cc:15-86)
 * through the g2o code it executes:
 *   SE3Quat ctor / normalizeRotation / operator* / map / exp / to_homogeneous_matrix
 *       Thirdparty/g2o/g2o/types/se3quat.h:58-60,98-104,217-257,270-285
 *   skew                      Thirdparty/g2o/g2o/types/se3_ops.hpp:27-38
 *   EdgeSE3ProjectXYZOnlyPose computeError / linearizeOplus / cam_project
 *       Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:153-157, .cpp:266-296
 *   VertexSE3Expmap::oplusImpl types_six_dof_expmap.h:73-76
 *   RobustKernelHuber::robustify core/robust_kernel_impl.cpp:77-91
 *   BaseUnaryEdge::constructQuadraticForm core/base_unary_edge.hpp:43-72
 *   robustInformation core/base_edge.h:96-102; chi2 :58-61
 *   activeRobustChi2  core/sparse_optimizer.cpp:100-114; optimize :354-419
 *   OptimizationAlgorithmLevenberg::solve / computeLambdaInit / computeScale
 *       core/optimization_algorithm_levenberg.cpp:61-189
 *   LinearSolverDense::solve (Eigen LDLT) solvers/linear_solver_dense.h:65-113
 * Eigen itself is un-vendored (version >= 3.1 only, Thirdparty/g2o/CMakeLists.txt:70):
 * Quaterniond(Matrix3d), toRotationMatrix, quaternion product and q*v follow
 * Eigen's published formulas; LDLT is restated without pivoting, so solutions
 * agree with the reference's to round-off, not bitwise (PARITY UNPINNED at that
 * boundary).  The HIP path reproduces THIS file bit for bit (tests/test_gpu_parity.py,
 * tests/test_full_length.py): same operations, same order of the sums over the edges.
 *
 * cv::solvePnPRansac (src/pnpmatch.cc:227) is restated in orc_pnp_cv.c.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

typedef struct { double q[4]; /* x y z w */ double t[3]; } se3_t;

/* Eigen Quaternion(Matrix3) - quaternionbase_assign_impl<Other,3,3>. */
static void quat_from_R(const double m[9], double q[4]) {
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
}
/* se3quat.h:280-285 */
static void normalize_rotation(se3_t* s) {
  if (s->q[3] < 0) for (int i = 0; i < 4; ++i) s->q[i] *= -1;
  double n = sqrt(s->q[0] * s->q[0] + s->q[1] * s->q[1] + s->q[2] * s->q[2] + s->q[3] * s->q[3]);
  for (int i = 0; i < 4; ++i) s->q[i] /= n;
}
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
/* Eigen QuaternionBase::_transformVector */
static void quat_rot(const double q[4], const double v[3], double o[3]) {
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_R(const double q[4], double R[9]) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
/* convert::toSE3Quat (src/convert.cc:6-16) -> SE3Quat(R,t) */
static void se3_from_T(const double T[16], se3_t* s) {
  double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  quat_from_R(R, s->q);
  s->t[0] = T[3]; s->t[1] = T[7]; s->t[2] = T[11];
  normalize_rotation(s);
}
/* SE3Quat::to_homogeneous_matrix (se3quat.h:270-278) */
static void se3_to_T(const se3_t* s, double T[16]) {
  double R[9];
  quat_to_R(s->q, R);
  T[0] = R[0]; T[1] = R[1]; T[2] = R[2]; T[3] = s->t[0];
  T[4] = R[3]; T[5] = R[4]; T[6] = R[5]; T[7] = s->t[1];
  T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = s->t[2];
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}
static void mat3_mul(const double a[9], const double b[9], double o[9]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
/* sin(t) / t, (1 - cos t) / t^2, (t - sin t) / t^3 for |t| < 0.5.  se3quat.h:240-250 takes them from libm's sin / cos, whose
 * last bit differs from one libm to the next (glibc's are not correctly rounded either).  To have ONE definition that a CPU and
 * a GPU evaluate to the same bits, both sides use these alternating series in t^2 (truncated below 1e-17 relative) as a Horner
 * chain of fma() calls - exact IEEE operations on either machine.  They are accurate to 1 ulp (tests/test_oracle_golden.py checks
 * them against exact rational arithmetic), which the reference's own formulas are not: (1 - cos t) / t^2 and (t - sin t) / t^3
 * cancel for small t (relative error ~1e-16 / t^2 and / t^3), so the two agree to within the LIBM formulas' error.  A tracking
 * step's rotation is ~0.01 rad; |t| >= 0.5 still takes libm. */
static void exp_coeffs_series(double t, double* a, double* b, double* c) {
  const double x = t * t;
  double pa = -1.0 / 121645100408832000.0, pb = 1.0 / 6402373705728000.0, pc = 1.0 / 121645100408832000.0; /* 1/19!, 1/18!, 1/19! */
  pa = fma(pa, x, 1.0 / 355687428096000.0);  /* 1/17! */
  pa = fma(pa, x, -1.0 / 1307674368000.0);   /* 1/15! */
  pa = fma(pa, x, 1.0 / 6227020800.0);       /* 1/13! */
  pa = fma(pa, x, -1.0 / 39916800.0);        /* 1/11! */
  pa = fma(pa, x, 1.0 / 362880.0);           /* 1/9!  */
  pa = fma(pa, x, -1.0 / 5040.0);            /* 1/7!  */
  pa = fma(pa, x, 1.0 / 120.0);              /* 1/5!  */
  pa = fma(pa, x, -1.0 / 6.0);               /* 1/3!  */
  pa = fma(pa, x, 1.0);
  pb = fma(pb, x, -1.0 / 20922789888000.0);  /* 1/16! */
  pb = fma(pb, x, 1.0 / 87178291200.0);      /* 1/14! */
  pb = fma(pb, x, -1.0 / 479001600.0);       /* 1/12! */
  pb = fma(pb, x, 1.0 / 3628800.0);          /* 1/10! */
  pb = fma(pb, x, -1.0 / 40320.0);           /* 1/8!  */
  pb = fma(pb, x, 1.0 / 720.0);              /* 1/6!  */
  pb = fma(pb, x, -1.0 / 24.0);              /* 1/4!  */
  pb = fma(pb, x, 0.5);
  pc = fma(pc, x, -1.0 / 355687428096000.0); /* 1/17! */
  pc = fma(pc, x, 1.0 / 1307674368000.0);    /* 1/15! */
  pc = fma(pc, x, -1.0 / 6227020800.0);      /* 1/13! */
  pc = fma(pc, x, 1.0 / 39916800.0);         /* 1/11! */
  pc = fma(pc, x, -1.0 / 362880.0);          /* 1/9!  */
  pc = fma(pc, x, 1.0 / 5040.0);             /* 1/7!  */
  pc = fma(pc, x, -1.0 / 120.0);             /* 1/5!  */
  pc = fma(pc, x, 1.0 / 6.0);                /* 1/3!  */
  *a = pa; *b = pb; *c = pc;
}
void orc_exp_coeffs(double t, double abc[3], int use_libm) {
  if (use_libm) {
    abc[0] = sin(t) / t; abc[1] = (1 - cos(t)) / (t * t); abc[2] = (t - sin(t)) / (t * t * t);
  } else {
    exp_coeffs_series(t, &abc[0], &abc[1], &abc[2]);
  }
}

/* SE3Quat::exp (se3quat.h:223-257); update = [omega ; upsilon]. */
static void se3_exp(const double u[6], se3_t* out) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double Om2[9], R[9], V[9];
  mat3_mul(Om, Om, Om2);
  if (theta < 0.00001) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
    memcpy(V, R, sizeof R);
  } else {
    double a, b, c;
    if (theta < 0.5) {
      exp_coeffs_series(theta, &a, &b, &c);
    } else {
      a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta); c = (theta - sin(theta)) / (theta * theta * theta);
    }
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  quat_from_R(R, out->q);
  for (int r = 0; r < 3; ++r) out->t[r] = V[3 * r] * up[0] + V[3 * r + 1] * up[1] + V[3 * r + 2] * up[2];
  normalize_rotation(out);
}
/* VertexSE3Expmap::oplusImpl: est <- exp(update) * est  (se3quat.h:98-104) */
static void se3_oplus(const double u[6], se3_t* est) {
  se3_t e, r;
  se3_exp(u, &e);
  double rt[3];
  quat_rot(e.q, est->t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = e.t[i] + rt[i];
  quat_mul(e.q, est->q, r.q);
  normalize_rotation(&r);
  *est = r;
}

void orc_se3_exp(const double upd[6], double q_xyzw[4], double t[3]) {
  se3_t s;
  se3_exp(upd, &s);
  memcpy(q_xyzw, s.q, sizeof s.q);
  memcpy(t, s.t, sizeof s.t);
}
void orc_se3_exp_matrix(const double upd[6], double T[16]) {
  se3_t s;
  se3_exp(upd, &s);
  se3_to_T(&s, T);
}
void orc_se3_update(const double upd[6], double T[16]) {
  se3_t s;
  se3_from_T(T, &s);
  se3_oplus(upd, &s);
  se3_to_T(&s, T);
}

/* RobustKernelHuber::robustify (robust_kernel_impl.cpp:77-91) */
void orc_huber(double e, double delta, double rho[3]) {
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    rho[0] = e; rho[1] = 1.; rho[2] = 0.;
  } else {
    const double sqrte = sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

/* 6x6 LDL^T solve; returns 0 if a pivot is not positive (isPositive() false). */
static int ldlt6_solve(const double Hin[36], const double b[6], double x[6]) {
  double L[36], D[6];
  memset(L, 0, sizeof L);
  for (int j = 0; j < 6; ++j) {
    double d = Hin[6 * j + j];
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
    if (!(d > 0.0)) return 0;
    D[j] = d;
    L[6 * j + j] = 1.0;
    for (int i = j + 1; i < 6; ++i) {
      double s = Hin[6 * i + j];
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k] * D[k];
      L[6 * i + j] = s / d;
    }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < 6; ++i) y[i] /= D[i];
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
    x[i] = s;
  }
  return 1;
}

/* error of one edge at `est` (types_six_dof_expmap.h:153-157) */
static inline void edge_error(const se3_t* est, const double* Xw, const double* obs,
                              const double K[4], double e[2], double pc[3]) {
  quat_rot(est->q, Xw, pc);
  pc[0] += est->t[0]; pc[1] += est->t[1]; pc[2] += est->t[2];
  e[0] = obs[0] - (pc[0] / pc[2] * K[0] + K[2]);
  e[1] = obs[1] - (pc[1] / pc[2] * K[1] + K[3]);
}
/* Jacobian (types_six_dof_expmap.cpp:266-288) */
static inline void edge_jacobian(const double pc[3], const double K[4], double J[12]) {
  const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz;
  J[0] = x * y * invz_2 * K[0];
  J[1] = -(1 + (x * x * invz_2)) * K[0];
  J[2] = y * invz * K[0];
  J[3] = -invz * K[0];
  J[4] = 0;
  J[5] = x * invz_2 * K[0];
  J[6] = (1 + y * y * invz_2) * K[1];
  J[7] = -x * y * invz_2 * K[1];
  J[8] = -x * invz * K[1];
  J[9] = 0;
  J[10] = -invz * K[1];
  J[11] = y * invz_2 * K[1];
}
static double robust_chi2(const se3_t* est, const double* Xw, const double* obs, int n,
                          const double K[4], double delta) {
  double chi = 0.0, rho[3];
  for (int i = 0; i < n; ++i) {
    double e[2], pc[3];
    edge_error(est, Xw + 3 * i, obs + 2 * i, K, e, pc);
    orc_huber(e[0] * e[0] + e[1] * e[1], delta, rho);
    chi += rho[0];
  }
  return chi;
}

/* trace (optional): 8 doubles per LM trial:
 * {iteration, trial, lambda_used, currentChi, tempChi, rho, accepted, solve_ok}. */
int orc_pose_opt(const double* Xw, const double* obs, int n, const double K[4], double T[16],
                 orc_lm_stats* stats, double* trace, int trace_cap) {
  const double delta = (double)(float)sqrt(5.991); /* const float deltaMono, Optimizer.cc:38 */
  se3_t est;
  se3_from_T(T, &est);
  double lambda = -1., ni = 2.;
  int nBad = 0, ntrace = 0, iters = 0, trials_total = 0, terminated = 0;
  double x[6] = {0, 0, 0, 0, 0, 0};
  double chi_init = 0, currentChi = 0;
  if (n <= 0) {
    if (stats) memset(stats, 0, sizeof *stats);
    return 0;
  }
  for (int it = 0; it < 10; ++it) {
    /* computeActiveErrors + activeRobustChi2 + buildSystem */
    double H[36], b[6];
    memset(H, 0, sizeof H);
    memset(b, 0, sizeof b);
    currentChi = 0;
    for (int i = 0; i < n; ++i) {
      double e[2], pc[3], J[12], rho[3];
      edge_error(&est, Xw + 3 * i, obs + 2 * i, K, e, pc);
      orc_huber(e[0] * e[0] + e[1] * e[1], delta, rho);
      currentChi += rho[0];
      edge_jacobian(pc, K, J);
      for (int r = 0; r < 6; ++r) {
        b[r] -= rho[1] * (J[r] * e[0] + J[6 + r] * e[1]);
        for (int c = 0; c < 6; ++c) H[6 * r + c] += rho[1] * (J[r] * J[c] + J[6 + r] * J[6 + c]);
      }
    }
    const double iniChi = currentChi;
    if (it == 0) {
      chi_init = currentChi;
      double maxDiag = 0;
      for (int j = 0; j < 6; ++j) maxDiag = fmax(fabs(H[7 * j]), maxDiag);
      lambda = 1e-5 * maxDiag;
      ni = 2;
      nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const se3_t backup = est;
      const double chi_before = currentChi;
      double Hl[36];
      memcpy(Hl, H, sizeof H);
      for (int j = 0; j < 6; ++j) Hl[7 * j] += lambda;
      const int ok2 = ldlt6_solve(Hl, b, x); /* x keeps its old value on failure */
      se3_oplus(x, &est);
      double tempChi = robust_chi2(&est, Xw, obs, n, K, delta);
      if (!ok2) tempChi = DBL_MAX;
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      const double lambda_used = lambda;
      int accepted = 0;
      if (rho > 0 && isfinite(tempChi)) {
        const double v = 2 * rho - 1;
        double alpha = 1. - v * v * v;
        alpha = fmin(alpha, 2. / 3.);
        const double scaleFactor = fmax(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
        accepted = 1;
      } else {
        lambda *= ni;
        ni *= 2;
        est = backup;
      }
      if (trace && ntrace < trace_cap) {
        double* tr = trace + 8 * ntrace++;
        tr[0] = it; tr[1] = qmax; tr[2] = lambda_used; tr[3] = chi_before;
        tr[4] = tempChi; tr[5] = rho; tr[6] = accepted; tr[7] = ok2;
      }
      ++qmax;
      ++trials_total;
    } while (rho < 0 && qmax < 10);
    ++iters;
    if (qmax == 10 || rho == 0) { terminated = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) ++nBad; else nBad = 0;
    if (nBad >= 3) { terminated = 1; break; }
  }
  se3_to_T(&est, T);
  if (stats) {
    stats->n_edges = n; stats->iterations = iters; stats->trials_total = trials_total;
    stats->terminated = terminated; stats->chi2_initial = chi_init;
    stats->chi2_final = currentChi; stats->lambda_final = lambda;
  }
  return ntrace;
}

/* cv::solvePnPRansac is restated in orc_pnp_cv.c. */
