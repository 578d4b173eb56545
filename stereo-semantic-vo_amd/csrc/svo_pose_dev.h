// svo_pose_dev.h - device-side pose stages for gfx950, float64 throughout; shared by the stand-alone kernels of
// svo_pose.hip (svo_pose_opt, svo_pnp_ransac) and by the tracker's fused per-frame pose kernel (svo_track.hip).
//
//   pose_opt_block   <- Optimizer::PoseOptimization (reference src/Optimizer.cc:15-86) as g2o executes it: one SE3
//                       vertex, n unary reprojection edges, Huber delta=(double)(float)sqrt(5.991), Levenberg-
//                       Marquardt x10 (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-164).
//   pnp_ransac_block <- the cv::solvePnPRansac call of pnpmatch::poseEstimationPnP (src/pnpmatch.cc:212-247).
//
// Both are called by ALL 256 threads of a workgroup (they contain barriers).  Per-edge residuals / Jacobians live in
// registers; the 6x6 J^T W J, J^T W e and chi2 are reduced on f64 MFMA (or wave shuffles + one LDS hop); lane 0 runs
// the scalar LM logic (LDLT, exp map, lambda schedule) so every accept/reject branch is taken in float64 exactly once.
#pragma once
#include "svo_internal.h"
#include "svo_wave.h"

struct Se3 { double q[4]; double t[3]; };

// BIT-IDENTICAL to the CPU restatement (oracle of the tests; g2o's loops with their operation order): every IEEE operation of
// base_unary_edge.hpp:43-72 / block_solver.hpp:503-561 / optimization_algorithm_levenberg.cpp:61-189 is kept and rounded once -
// what runs in parallel are INDEPENDENT operations only.  Three ingredients:
//  * `/` and sqrt on the serial path (thread 0: LDL^T, exp map, quaternion updates: ~20 divisions, 5 square roots per trial) are
//    the compiler's correctly rounded sequences WITHOUT their range scaling and special-case fix-up (pose_ndiv / pose_nsqrt: the
//    same reciprocal / rsq estimate, Newton steps and final correction instruction for instruction - bit-identical to IEEE for
//    operands a few hundred binades inside the range, which every quantity of the LM is);
//  * the sums over the edges (H += rho1 J^T J, b -= rho1 J^T e, chi2 += rho) run in INSERTION ORDER on the matrix core:
//    v_mfma_f64_4x4x4_4b_f64 with A = 1.0 is the IEEE sum ((((C + b0) + b1) + b2) + b3) of the four 16-lane groups' values per
//    column (svo_epnp_ord_dev.h; tools/microbench/mfma_f64_4x4.hip), so four edges are added per instruction to 16 running sums;
//  * sin / cos of SE3Quat::exp: for |theta| < 0.5 the fixed series below (FMA Horner), which the CPU restatement evaluates with the
//    same fma() calls (beyond that libm on both sides: not bit-guaranteed, never reached by a tracking step).
__device__ __forceinline__ double pose_ndiv(double a, double b) {   // a / b
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = a * r;
  const double rem = __builtin_fma(-b, q, a);
  return __builtin_fma(rem, r, q);
}
// a / b for several a and one b: the refined reciprocal depends on b only (pose_ndiv's first five instructions), the last three
// are per dividend - the same instructions per quotient as pose_ndiv, hence the same bits
__device__ __forceinline__ double pose_rcp_refined(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double pose_ndiv_r(double a, double b, double r) {   // a / b, r = pose_rcp_refined(b)
  const double q = a * r;
  const double rem = __builtin_fma(-b, q, a);
  return __builtin_fma(rem, r, q);
}
__device__ __forceinline__ double pose_nsqrt(double x) {   // sqrt(x), x >= 0
  if (!(x > 0.0)) return 0.0;
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}
// sin(t) / t, (1 - cos t) / t^2, (t - sin t) / t^3 for |t| < 0.5 (every LM step; larger rotations take libm's sin / cos):
// alternating series in t^2, truncated below 1e-17 relative
__device__ __forceinline__ void pose_exp_coeffs(double t, double* a, double* b, double* c) {
  const double x = t * t;
  // a = 1 - x/3! + x^2/5! - ... ;  b = 1/2! - x/4! + x^2/6! - ... ;  c = 1/3! - x/5! + x^2/7! - ...
  double pa = -1.0 / 121645100408832000.0, pb = 1.0 / 6402373705728000.0, pc = 1.0 / 121645100408832000.0;   // 1/19!, 1/18!, 1/19!
  pa = __builtin_fma(pa, x, 1.0 / 355687428096000.0);      // 1/17!
  pa = __builtin_fma(pa, x, -1.0 / 1307674368000.0);       // 1/15!
  pa = __builtin_fma(pa, x, 1.0 / 6227020800.0);           // 1/13!
  pa = __builtin_fma(pa, x, -1.0 / 39916800.0);            // 1/11!
  pa = __builtin_fma(pa, x, 1.0 / 362880.0);               // 1/9!
  pa = __builtin_fma(pa, x, -1.0 / 5040.0);                // 1/7!
  pa = __builtin_fma(pa, x, 1.0 / 120.0);                  // 1/5!
  pa = __builtin_fma(pa, x, -1.0 / 6.0);                   // 1/3!
  pa = __builtin_fma(pa, x, 1.0);
  pb = __builtin_fma(pb, x, -1.0 / 20922789888000.0);      // 1/16!
  pb = __builtin_fma(pb, x, 1.0 / 87178291200.0);          // 1/14!
  pb = __builtin_fma(pb, x, -1.0 / 479001600.0);           // 1/12!
  pb = __builtin_fma(pb, x, 1.0 / 3628800.0);              // 1/10!
  pb = __builtin_fma(pb, x, -1.0 / 40320.0);               // 1/8!
  pb = __builtin_fma(pb, x, 1.0 / 720.0);                  // 1/6!
  pb = __builtin_fma(pb, x, -1.0 / 24.0);                  // 1/4!
  pb = __builtin_fma(pb, x, 0.5);
  pc = __builtin_fma(pc, x, -1.0 / 355687428096000.0);     // 1/17!
  pc = __builtin_fma(pc, x, 1.0 / 1307674368000.0);        // 1/15!
  pc = __builtin_fma(pc, x, -1.0 / 6227020800.0);          // 1/13!
  pc = __builtin_fma(pc, x, 1.0 / 39916800.0);             // 1/11!
  pc = __builtin_fma(pc, x, -1.0 / 362880.0);              // 1/9!
  pc = __builtin_fma(pc, x, 1.0 / 5040.0);                 // 1/7!
  pc = __builtin_fma(pc, x, -1.0 / 120.0);                 // 1/5!
  pc = __builtin_fma(pc, x, 1.0 / 6.0);                    // 1/3!
  *a = pa; *b = pb; *c = pc;
}

// Eigen Quaternion(Matrix3): the three "largest diagonal" cases are written out so that nothing is
// indexed dynamically (dynamic indices would push the matrix into scratch memory).
__device__ __forceinline__ void quat_from_R(const double m[9], double q[4]) {
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = pose_nsqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = pose_ndiv(0.5, t);
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > (i == 1 ? m[4] : m[0])) i = 2;
    if (i == 0) {            // j = 1, k = 2
      t = pose_nsqrt(m[0] - m[4] - m[8] + 1.0);
      q[0] = 0.5 * t; t = pose_ndiv(0.5, t);
      q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
    } else if (i == 1) {     // j = 2, k = 0
      t = pose_nsqrt(m[4] - m[8] - m[0] + 1.0);
      q[1] = 0.5 * t; t = pose_ndiv(0.5, t);
      q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
    } else {                 // j = 0, k = 1
      t = pose_nsqrt(m[8] - m[0] - m[4] + 1.0);
      q[2] = 0.5 * t; t = pose_ndiv(0.5, t);
      q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
    }
  }
}
__device__ __forceinline__ void normalize_rotation(Se3& s) {
  if (s.q[3] < 0) { s.q[0] = -s.q[0]; s.q[1] = -s.q[1]; s.q[2] = -s.q[2]; s.q[3] = -s.q[3]; }
  const double n = pose_nsqrt(s.q[0] * s.q[0] + s.q[1] * s.q[1] + s.q[2] * s.q[2] + s.q[3] * s.q[3]);
  const double rn = pose_rcp_refined(n);
  s.q[0] = pose_ndiv_r(s.q[0], n, rn); s.q[1] = pose_ndiv_r(s.q[1], n, rn); s.q[2] = pose_ndiv_r(s.q[2], n, rn); s.q[3] = pose_ndiv_r(s.q[3], n, rn);
}
__device__ __forceinline__ void quat_mul(const double a[4], const double b[4], double o[4]) {
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
__device__ __forceinline__ void quat_rot(const double q[4], const double v[3], double o[3]) {
  double uv0 = q[1] * v[2] - q[2] * v[1], uv1 = q[2] * v[0] - q[0] * v[2], uv2 = q[0] * v[1] - q[1] * v[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  o[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  o[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  o[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}
__device__ __forceinline__ void quat_to_R(const double q[4], double R[9]) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void se3_from_T(const double* T, Se3& s) {
  const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  quat_from_R(R, s.q);
  s.t[0] = T[3]; s.t[1] = T[7]; s.t[2] = T[11];
  normalize_rotation(s);
}
__device__ __forceinline__ void se3_to_T(const Se3& s, double* T) {
  double R[9];
  quat_to_R(s.q, R);
  T[0] = R[0]; T[1] = R[1]; T[2] = R[2]; T[3] = s.t[0];
  T[4] = R[3]; T[5] = R[4]; T[6] = R[5]; T[7] = s.t[1];
  T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = s.t[2];
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}
// SE3Quat::exp (se3quat.h:223-257)
__device__ void se3_exp(const double u[6], Se3& out) {
  const double om0 = u[0], om1 = u[1], om2 = u[2];
  const double theta = pose_nsqrt(om0 * om0 + om1 * om1 + om2 * om2);
  const double Om[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
  double Om2[9], R[9], V[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Om2[3 * r + c] = Om[3 * r] * Om[c] + Om[3 * r + 1] * Om[3 + c] + Om[3 * r + 2] * Om[6 + c];
  if (theta < 0.00001) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    double a, b, c;
    if (theta < 0.5) {
      pose_exp_coeffs(theta, &a, &b, &c);
    } else {
      a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta); c = (theta - sin(theta)) / (theta * theta * theta);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  quat_from_R(R, out.q);
#pragma unroll
  for (int r = 0; r < 3; ++r) out.t[r] = V[3 * r] * u[3] + V[3 * r + 1] * u[4] + V[3 * r + 2] * u[5];
  normalize_rotation(out);
}
__device__ void se3_oplus(const double u[6], Se3& est) {
  Se3 e, r;
  se3_exp(u, e);
  double rt[3];
  quat_rot(e.q, est.t, rt);
  r.t[0] = e.t[0] + rt[0]; r.t[1] = e.t[1] + rt[1]; r.t[2] = e.t[2] + rt[2];
  quat_mul(e.q, est.q, r.q);
  normalize_rotation(r);
  est = r;
}
__device__ __forceinline__ void huber(double e, double delta, double dsqr, double& rho0, double& rho1) {
  if (e <= dsqr) { rho0 = e; rho1 = 1.; }
  else { const double sq = pose_nsqrt(e); rho0 = 2 * sq * delta - dsqr; rho1 = pose_ndiv(delta, sq); }
}
// 6x6 LDL^T solve, fully unrolled so L, D and y stay in registers (dynamically indexed local arrays
// would be placed in scratch memory, and this sits on the serial path of every LM / RANSAC step).
// Returns 0 if a pivot is not positive (Eigen LDLT::isPositive() false).
__device__ __forceinline__ int ldlt6_solve(const double* Hin, const double* b, double* x) {
  double L[6][6], D[6], Dr[6], y[6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = Hin[6 * j + j];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (k < j) d -= L[j][k] * L[j][k] * D[k];
    ok = ok && (d > 0.0);
    D[j] = d;
    const double dr = pose_rcp_refined(d);     // (one refined reciprocal per pivot: every division by it shares it)
    Dr[j] = dr;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i > j) {
        double sacc = Hin[6 * i + j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (k < j) sacc -= L[i][k] * L[j][k] * D[k];
        L[i][j] = pose_ndiv_r(sacc, d, dr);
      }
    }
  }
  if (!ok) return 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double sacc = b[i];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (k < i) sacc -= L[i][k] * y[k];
    y[i] = sacc;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = pose_ndiv_r(y[i], D[i], Dr[i]);
  double xs[6];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double sacc = y[i];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (k > i) sacc -= L[k][i] * xs[k];
    xs[i] = sacc;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = xs[i];
  return 1;
}
// error and Jacobian of one edge (types_six_dof_expmap.h:153-157, .cpp:266-288).  The three divisions by the point's depth - x / z,
// y / z in the error, 1 / z in the Jacobian - share ONE refined reciprocal (pose_ndiv_r: the compiler's division without its
// range scaling, bit-identical for a depth a few hundred binades inside the range; a depth of exactly 0 gives NaN where IEEE
// gives infinity - either way the LM rejects every trial and keeps its pose).
__device__ __forceinline__ void edge_error(const Se3& est, const double* Xw, const double* obs,
                                           const double* K, double e[2], double pc[3], double* rz_out = nullptr) {
  quat_rot(est.q, Xw, pc);
  pc[0] += est.t[0]; pc[1] += est.t[1]; pc[2] += est.t[2];
  const double rz = pose_rcp_refined(pc[2]);
  e[0] = obs[0] - (pose_ndiv_r(pc[0], pc[2], rz) * K[0] + K[2]);
  e[1] = obs[1] - (pose_ndiv_r(pc[1], pc[2], rz) * K[1] + K[3]);
  if (rz_out) *rz_out = rz;
}
__device__ __forceinline__ void edge_jacobian(const double pc[3], const double* K, double J[12], double rz) {
  const double x = pc[0], y = pc[1], invz = pose_ndiv_r(1.0, pc[2], rz), invz_2 = invz * invz;
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

// ---- the sums over the edges, in insertion order, on the matrix core -------------------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 with A = 1.0: D(lane 16 g + c) = ((((C + B(c)) + B(16 + c)) + B(32 + c)) + B(48 + c)) for every group g:
// the four lane groups' values of column c are added to the running sum in group order, each addition rounded (IEEE), and the
// result lands in all four lanes of the column.  Lane group g supplies edge 4 s + g at step s; column c is one of up to 16
// independent running sums.
__device__ __forceinline__ double pose_gsum(double term, double acc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, term, acc, 0, 0, 0); }

// `cnt` values per column (value e of this lane's column at p[e * pitch]), added to `acc` in order, four per MFMA (lane group g
// supplies value 4 s + g at step s).  Four steps at a time: their LDS loads are issued together, THEN the dependent MFMAs run;
// no per-step branches or selects (a load + a branch in front of every MFMA cost 165 cycles per step instead of ~40: measured,
// tools/lm_phases.py).  The caller has ZERO-FILLED the values from `cnt` up to the next multiple of 16 and the columns that carry
// no quantity: adding +0.0 leaves every partial sum as it is.
__device__ __forceinline__ double pose_chain_sum(const double* p, int pitch, int cnt, int g, double acc) {
  for (int s0 = 0; 4 * s0 < cnt; s0 += 4) {      // (one uniform branch per four steps = sixteen values; the caller zero-filled up to there)
    double v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = p[(4 * (s0 + s) + g) * pitch];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = pose_gsum(v[s], acc);
  }
  return acc;
}

// The same sum by plain additions, ONE LANE PER COLUMN: the lane adds its column's values one after the other (loads sixteen at a
// time, then sixteen dependent v_add_f64: ~16 cycles per value for a lone wave against ~10 per value on the matrix core, which
// takes four values per dependent step) - "pose_mfma" = 2, the second witness of the order.
// The caller has ZERO-FILLED the values from `cnt` up to the next multiple of 16 (adding +0.0 changes nothing): no clamping,
// no selects - per value half a two-value LDS load and the addition (a lone wave issues an instruction every ~4.4 cycles, and
// with address arithmetic and selects per value the chain took 44 cycles per value instead of ~9).
__device__ __forceinline__ double pose_lane_sum(const double* p, int pitch, int cnt, double acc) {
  for (int e0 = 0; e0 < cnt; e0 += 16) {         // (one uniform branch per sixteen values)
    double v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = p[(e0 + s) * pitch];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc += v[s];
  }
  return acc;
}

// One wave's LDS: the terms of up to 64 edges (one per lane), POSE_TW doubles apart (an odd pitch: conflict-free column reads)
#define POSE_TW 9
#define POSE_NQ 28          // 21 entries of the upper triangle of H (row-major, r <= c), 6 of b, chi2
#define POSE_QW 7           // of them per wave (four waves)
#define POSE_MAXN 512
#define POSE_CHUNK 128      // edges a wave turns into terms per pass (two per lane)

// wave-local visibility of LDS writes (LDS instructions of a wave execute in issue order; the fence keeps the compiler in line)
#define POSE_WSYNC()                                       \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

// term k (0..27) of an edge, with the CPU loop's operations (base_unary_edge.hpp:43-72 as g2o's Eigen expressions evaluate:
// H(r, c) += rho1 * (J(0, r) * J(0, c) + J(1, r) * J(1, c)); b(r) -= rho1 * (J(0, r) * e0 + J(1, r) * e1); chi2 += rho0).
// The b terms come out NEGATED: adding -t is subtracting t, bit for bit.
constexpr int pose_tri_row(int k) { int r = 0; while (k >= 6 - r) { k -= 6 - r; ++r; } return r; }
constexpr int pose_tri_col(int k) { int r = 0; while (k >= 6 - r) { k -= 6 - r; ++r; } return r + k; }
template <int K>
__device__ __forceinline__ double edge_term(const double J[12], const double e[2], double rho0, double rho1) {
  if constexpr (K < 21) {
    constexpr int r = pose_tri_row(K), c = pose_tri_col(K);
    return rho1 * (J[r] * J[c] + J[6 + r] * J[6 + c]);
  } else if constexpr (K < 27) {
    constexpr int r = K - 21;
    return -(rho1 * (J[r] * e[0] + J[6 + r] * e[1]));
  } else {
    return rho0;
  }
}
template <int W>
__device__ __forceinline__ void edge_terms(const double J[12], const double e[2], double rho0, double rho1, double out[POSE_QW]) {
  out[0] = edge_term<W * POSE_QW + 0>(J, e, rho0, rho1); out[1] = edge_term<W * POSE_QW + 1>(J, e, rho0, rho1);
  out[2] = edge_term<W * POSE_QW + 2>(J, e, rho0, rho1); out[3] = edge_term<W * POSE_QW + 3>(J, e, rho0, rho1);
  out[4] = edge_term<W * POSE_QW + 4>(J, e, rho0, rho1); out[5] = edge_term<W * POSE_QW + 5>(J, e, rho0, rho1);
  out[6] = edge_term<W * POSE_QW + 6>(J, e, rho0, rho1);
}

// One edge's share of wave W's quantities (error, Huber weight, Jacobian, then the wave's seven terms)
template <int W>
__device__ __forceinline__ void edge_wave_terms(const Se3& est, const double* Xw, const double* obs, int i, const double* K,
                                                double delta, double dsqr, double t[POSE_QW]) {
  double e[2], pc[3], J[12], rz;
  edge_error(est, Xw + 3 * i, obs + 2 * i, K, e, pc, &rz);
  double rho0 = e[0] * e[0] + e[1] * e[1], rho1 = 1.;
  huber(rho0, delta, dsqr, rho0, rho1);
  edge_jacobian(pc, K, J, rz);
  edge_terms<W>(J, e, rho0, rho1, t);
}
// wave W's terms of the edges e0 + lane and (TWO) e0 + 64 + lane into its slab: straight-line code for both edges (indices clamped,
// stores masked), so that the two dependent chains - IEEE divisions, the rotation - fill each other's issue gaps
template <int W, bool TWO>
__device__ __forceinline__ void wave_terms_to_lds(const Se3& est, const double* Xw, const double* obs, int n, int e0, int lane,
                                                  const double* K, double delta, double dsqr, double* tw) {
  const int iA = e0 + lane, iB = e0 + 64 + lane;
  double tA[POSE_QW], tB[POSE_QW];
  edge_wave_terms<W>(est, Xw, obs, min(iA, n - 1), K, delta, dsqr, tA);
  if (TWO) edge_wave_terms<W>(est, Xw, obs, min(iB, n - 1), K, delta, dsqr, tB);
  // (every lane stores: rows beyond the last edge hold +0.0 - the sums run over whole groups of sixteen rows -, and so do the two
  // padding columns of every row, which the matrix-core sum's idle columns read)
#pragma unroll
  for (int q = 0; q < POSE_TW; ++q) tw[lane * POSE_TW + q] = (q < POSE_QW && iA < n) ? tA[q < POSE_QW ? q : 0] : 0.0;
  if (TWO) {
#pragma unroll
    for (int q = 0; q < POSE_TW; ++q) tw[(64 + lane) * POSE_TW + q] = (q < POSE_QW && iB < n) ? tB[q < POSE_QW ? q : 0] : 0.0;
  }
}

// The normal equations at `est`: red[0..20] = upper triangle of H, red[21..26] = b, red[27] = chi2 - every entry the IEEE sum of
// its edge terms in edge order, starting from +0.  256 threads: wave w owns the quantities 7 w .. 7 w + 6; it evaluates them for
// up to 128 edges at a time (one or two edges per lane: error, Huber weight, Jacobian - recomputed by each of the four waves, which
// is cheaper than handing them over), stores them to its LDS slab, and adds them to its running sums four edges per MFMA.
__device__ __forceinline__ void build_system_ordered(const Se3& est, const double* Xw, const double* obs, int n, const double* K,
                                                     double delta, double dsqr, double* tw_all /*[4][POSE_CHUNK * POSE_TW]*/, double* red,
                                                     bool on_mfma) {
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  double* tw = tw_all + wv * POSE_CHUNK * POSE_TW;
  double acc = 0.0;
  for (int e0 = 0; e0 < n; e0 += POSE_CHUNK) {
    const int cnt = min(POSE_CHUNK, n - e0);
    if (cnt > 64) {
      switch (wv) {                       // (uniform per wave: scalar branches)
        case 0: wave_terms_to_lds<0, true>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
        case 1: wave_terms_to_lds<1, true>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
        case 2: wave_terms_to_lds<2, true>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
        default: wave_terms_to_lds<3, true>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
      }
    } else {
      switch (wv) {
        case 0: wave_terms_to_lds<0, false>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
        case 1: wave_terms_to_lds<1, false>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
        case 2: wave_terms_to_lds<2, false>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
        default: wave_terms_to_lds<3, false>(est, Xw, obs, n, e0, lane, K, delta, dsqr, tw); break;
      }
    }
    POSE_WSYNC();
    if (on_mfma) acc = pose_chain_sum(tw + min(c, POSE_TW - 1), POSE_TW, cnt, g, acc);   // columns 0..6: quantities; 7, 8 (and the lanes clamped onto 8): zeros
    else acc = pose_lane_sum(tw + min(lane, POSE_QW - 1), POSE_TW, cnt, acc);      // lanes 0..6: one quantity each
    POSE_WSYNC();                         // (the slab is rewritten by the next chunk)
  }
  if (lane < POSE_QW) red[wv * POSE_QW + lane] = acc;     // (b's terms were negated: red[21..26] is b itself)
  __syncthreads();
}

__device__ __forceinline__ void unpack_system(const double* red, double H[36], double b[6]) {
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) { H[6 * r + c] = red[k]; H[6 * c + r] = red[k]; ++k; }
#pragma unroll
  for (int r = 0; r < 6; ++r) b[r] = red[21 + r];
}

struct LmShared {
  Se3 est;       // the estimate the next system is built at (all threads read)
  int go;        // 1: another trial ; 0: iteration finished
  int done;
};

#define PNP_HYP 100        // cv::solvePnPRansac(..., iterationsCount = 100, ...) (reference src/pnpmatch.cc:227)
#define PNP_MAXN 512       // correspondences per problem (one per keypoint at most)

// LDS workspace of the pose-only LM
struct PoseLds {
  double tw[4 * POSE_CHUNK * POSE_TW];   // per wave: the terms of a chunk of edges
  double red[POSE_NQ + 4];
  LmShared sh;
  double K[4];
};

// the normal equations at `est` into red[] by ONE lane, edge by edge as the CPU loop stands (the checker of build_system_ordered)
__device__ __forceinline__ void build_system_one_lane(const Se3& est, const double* Xw, const double* obs, int n, const double* K,
                                                      double delta, double dsqr, double* red) {
  if (threadIdx.x == 0) {
    double acc[POSE_NQ];
    for (int k = 0; k < POSE_NQ; ++k) acc[k] = 0.0;
    for (int i = 0; i < n; ++i) {
      double e[2], pc[3], J[12], rz;
      edge_error(est, Xw + 3 * i, obs + 2 * i, K, e, pc, &rz);
      double rho0 = e[0] * e[0] + e[1] * e[1], rho1 = 1.;
      huber(rho0, delta, dsqr, rho0, rho1);
      edge_jacobian(pc, K, J, rz);
      int k = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) acc[k++] += rho1 * (J[r] * J[c] + J[6 + r] * J[6 + c]);
      for (int r = 0; r < 6; ++r) acc[21 + r] -= rho1 * (J[r] * e[0] + J[6 + r] * e[1]);
      acc[27] += rho0;
    }
    for (int k = 0; k < POSE_NQ; ++k) red[k] = acc[k];
  }
  __syncthreads();
}

// Pose-only LM, called by all NT = 256 threads of the workgroup.  T: row-major 4x4 in/out (global or LDS); stats may be null.
// use_ordered_mfma (svo_set_option "pose_mfma"): 1 (default) the sums over the edges on the matrix core (v_mfma_f64_4x4x4, A = 1:
// four edges per dependent step); 2 the same sums by one lane per quantity, plain additions in order (a little slower); 0 everything
// by ONE lane, edge by edge as the CPU loop stands (the checker).  Identical bits in all three.
//
// Schedule.  g2o evaluates the robust chi2 of a trial with a pass of its own (sparse_optimizer.cpp:100-114) and rebuilds the
// normal equations at the top of the next iteration (buildSystem) - at the SAME estimate whenever the LM goes on, because it
// only goes on after an accepted trial.  Both passes visit the same edges with the same operations, and chi2 is one of the 28
// ordered sums of the build; so here every trial builds the whole system at its estimate: its chi2 entry IS the trial's chi2
// (same values, same order: same bits), and an accepted trial's system IS the next iteration's - one pass over the edges per
// trial instead of two, on all four waves.  The scalar LM state (lambda, the pose, H, b) lives in thread 0's registers.
template <int NT = 256>
__device__ __forceinline__ void pose_opt_block(PoseLds& L, const double* __restrict__ Xw, const double* __restrict__ obs, int n,
                                               const double* __restrict__ Kp, double* T, svo_lm_stats* stats,
                                               int round_in_f32, int use_ordered_mfma) {
  static_assert(NT == 256, "pose_opt_block: four waves (wave w owns the quantities 7 w .. 7 w + 6 of the normal equations)");
  double* red = L.red; LmShared& sh = L.sh; double* K = L.K;
  const int tid = threadIdx.x;
  const double delta = (double)(float)sqrt(5.991);
  const double dsqr = delta * delta;
  __syncthreads();
  if (tid < 4) K[tid] = Kp[tid];
  Se3 est0{};                 // thread 0: the current estimate
  if (tid == 0) {
    if (round_in_f32) {  // the reference stores the PnP pose as CV_32F before optimising it
      double Tf[16];
      for (int j = 0; j < 16; ++j) Tf[j] = (double)(float)T[j];
      se3_from_T(Tf, est0);
    } else {
      se3_from_T(T, est0);
    }
    sh.est = est0;
    sh.done = 0;
  }
  __syncthreads();
  double lambda = -1., ni = 2., currentChi = 0, chi_init = 0;
  int nBad = 0, iters = 0, trials_total = 0, terminated = 0;
  if (n <= 0) {
    if (tid == 0 && stats) {
      stats->n_edges = 0; stats->iterations = 0; stats->trials_total = 0; stats->terminated = 0;
      stats->chi2_initial = 0; stats->chi2_final = 0; stats->lambda_final = 0;
    }
    return;
  }
#ifdef POSE_PROF
  long long pf_build = 0, pf_serial = 0, pf_dec = 0, pf_t = 0;
#define PF(acc) do { const long long _n = clock64(); acc += _n - pf_t; pf_t = _n; } while (0)
  pf_t = clock64();
#else
#define PF(acc) do { } while (0)
#endif
  auto build = [&]() {        // the system at sh.est into red[] (ends with a barrier)
    const Se3 est = sh.est;
    if (use_ordered_mfma) build_system_ordered(est, Xw, obs, n, K, delta, dsqr, L.tw, red, use_ordered_mfma == 1);
    else build_system_one_lane(est, Xw, obs, n, K, delta, dsqr, red);
  };
  build();
  PF(pf_build);
  double H[36], b[6], x[6] = {0, 0, 0, 0, 0, 0};
  if (tid == 0) {
    unpack_system(red, H, b);
    currentChi = red[27];
    chi_init = currentChi;
    double maxDiag = 0;
    for (int j = 0; j < 6; ++j) maxDiag = fmax(fabs(H[7 * j]), maxDiag);
    lambda = 1e-5 * maxDiag;
  }
  for (int it = 0; it < 10; ++it) {
    double iniChi = currentChi, rho = 0;
    int qmax = 0;
    for (int trial = 0; trial < 10; ++trial) {
      int ok2 = 0;
      if (tid == 0) {
        double Hl[36];
#pragma unroll
        for (int j = 0; j < 36; ++j) Hl[j] = H[j];
#pragma unroll
        for (int j = 0; j < 6; ++j) Hl[7 * j] += lambda;
        ok2 = ldlt6_solve(Hl, b, x);          // (x keeps its old value when a pivot is not positive)
        Se3 e2 = est0;
        se3_oplus(x, e2);
        sh.est = e2;                          // the trial estimate
      }
      __syncthreads();
      PF(pf_serial);
      build();                                // chi2 of the trial = red[27]; its system = the next iteration's if it is accepted
      PF(pf_build);
      if (tid == 0) {
        double tempChi = red[27];
        if (!ok2) tempChi = 1.7976931348623157e308;
        rho = currentChi - tempChi;
        double scale = 0;
        for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(tempChi)) {
          const double v = 2 * rho - 1;
          double alpha = 1. - v * v * v;
          alpha = fmin(alpha, 2. / 3.);
          const double scaleFactor = fmax(1. / 3., alpha);
          lambda *= scaleFactor;
          ni = 2;
          currentChi = tempChi;
          est0 = sh.est;                      // accepted: the trial estimate and its system take over
          unpack_system(red, H, b);
        } else {
          lambda *= ni;
          ni *= 2;
        }
        ++qmax;
        ++trials_total;
        sh.go = (rho < 0 && qmax < 10) ? 1 : 0;
      }
      __syncthreads();
      PF(pf_dec);
      if (!sh.go) break;
    }
    if (tid == 0) {
      ++iters;
      int stop = 0;
      if (qmax == 10 || rho == 0) stop = 1;
      else {
        if ((iniChi - currentChi) * 1e3 < iniChi) ++nBad; else nBad = 0;
        if (nBad >= 3) stop = 1;
      }
      if (stop) terminated = 1;
      sh.done = stop;
    }
    __syncthreads();
    if (sh.done) break;
  }
  if (tid == 0) {
    se3_to_T(est0, T);
    if (stats) {
      stats->n_edges = n; stats->iterations = iters; stats->trials_total = trials_total;
      stats->terminated = terminated; stats->chi2_initial = chi_init;
      stats->chi2_final = currentChi; stats->lambda_final = lambda;
#ifdef POSE_PROF
      stats->chi2_initial = (double)pf_build; stats->chi2_final = (double)pf_serial; stats->lambda_final = (double)pf_dec;
#endif
    }
  }
}

// ---- cv::solvePnPRansac: hypotheses + the sequential acceptance rule ---------------------------------------------
// (OpenCV 3.2 modules/calib3d/src/solvepnp.cpp + ptsetreg.cpp)
#include "svo_epnp_dev.h"
#include "svo_epnp_exact_dev.h"
#include "svo_epnp_ord_dev.h"

struct PnpHyp {          // one RANSAC sample: EPnP pose of its five points and its consensus
  double R[9], t[3];
  int32_t cnt, ok;
};

// PnPRansacCallback::computeError for one point: projectPoints in double, rounded once to float, squared distance to
// the float image point in float.  Inlier iff <= (float)(8 * 8).
__device__ __forceinline__ bool pnp_inlier(const double* R, const double* t, const double* X, const double* uv, const double* K) {
  double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  z = z != 0.0 ? 1. / z : 1.;
  x *= z; y *= z;
  const float px = (float)(x * K[0] + K[2]), py = (float)(y * K[1] + K[3]);
  const float dx = (float)uv[0] - px, dy = (float)uv[1] - py;
  const float err = dx * dx + dy * dy;
  return err <= 64.0f;
}

// Hypotheses hyp_base .. hyp_base + (waves of the block) - 1, one wave each.  Xw / uv: the n correspondences in LDS (or
// anywhere), subset: the 100 x 5 sample indices cv::RNG((uint64)-1) yields for this n (svo_pnp_subsets).
__device__ __forceinline__ void pnp_hyp_block(EpnpWaveLds* ws, const double* Xw, const double* uv, int n, const double* K,
                                              const uint16_t* subset, PnpHyp* out, int hyp_base) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = hyp_base + wv;
  if (k >= PNP_HYP) return;
  if (lane < 5) {   // the sample's five correspondences into the wave's workspace
    const int e = min((int)subset[5 * k + lane], n - 1);
    ws[wv].x5[3 * lane] = Xw[3 * e]; ws[wv].x5[3 * lane + 1] = Xw[3 * e + 1]; ws[wv].x5[3 * lane + 2] = Xw[3 * e + 2];
    ws[wv].u5[2 * lane] = uv[2 * e]; ws[wv].u5[2 * lane + 1] = uv[2 * e + 1];
  }
  double R[9], t[3];
  const bool ok = epnp5_wave(ws[wv], K, R, t);
  int cnt = 0;
  if (ok)
    for (int e = lane; e < n; e += 64) cnt += pnp_inlier(R, t, Xw + 3 * e, uv + 2 * e, K) ? 1 : 0;
  cnt = wave_sum_i32_dpp(cnt);
  if (lane == 0) {
    PnpHyp h;
#pragma unroll
    for (int i = 0; i < 9; ++i) h.R[i] = R[i];
    h.t[0] = t[0]; h.t[1] = t[1]; h.t[2] = t[2];
    h.cnt = cnt; h.ok = ok ? 1 : 0;
    out[k] = h;
  }
}

// The same samples in the parity mode (svo_set_option "epnp_exact"): ONE sample per wave - lane 0 walks the sequential
// restatement of OpenCV's loops (svo_epnp_exact_dev.h) with every array in the LDS workspace `W`, then the whole wave
// counts the sample's consensus.  Called by all 64 lanes of the wave that owns sample k.
struct PnpExactLds { epnp_exact::Work W; double R[9], t[3]; int ok; };
__device__ __forceinline__ void pnp_hyp_exact_wave(PnpExactLds& S, const double* Xw, const double* uv, int n, const double* K,
                                                   const uint16_t* subset, PnpHyp* out, int k) {
  const int lane = threadIdx.x & 63;
  if (k >= PNP_HYP) return;
  if (lane == 0) {
    double* x5 = S.W.PW0;        // staging only: solve5 copies the sample into its problem record first
    double* u5 = S.W.gA;
    for (int i = 0; i < 5; ++i) {
      const int e = min((int)subset[5 * k + i], n - 1);
      x5[3 * i] = Xw[3 * e]; x5[3 * i + 1] = Xw[3 * e + 1]; x5[3 * i + 2] = Xw[3 * e + 2];
      u5[2 * i] = uv[2 * e]; u5[2 * i + 1] = uv[2 * e + 1];
    }
    S.ok = epnp_exact::solve5(S.W, x5, u5, K, S.R, S.t, nullptr) ? 1 : 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = S.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = S.t[i];
  const bool ok = S.ok != 0;
  int cnt = 0;
  if (ok)
    for (int e = lane; e < n; e += 64) cnt += pnp_inlier(R, t, Xw + 3 * e, uv + 2 * e, K) ? 1 : 0;
  cnt = wave_sum_i32_dpp(cnt);
  if (lane == 0) {
    PnpHyp h;
#pragma unroll
    for (int i = 0; i < 9; ++i) h.R[i] = R[i];
    h.t[0] = t[0]; h.t[1] = t[1]; h.t[2] = t[2];
    h.cnt = cnt; h.ok = ok ? 1 : 0;
    out[k] = h;
  }
}

// The same samples in the order-preserving wave mode (svo_set_option "epnp_exact" = 2, the default): ONE sample per wave,
// OpenCV's operations spread over the wavefront with their rounding kept (svo_epnp_ord_dev.h), bit-identical to the
// sequential restatement; then the wave counts the sample's consensus.  Called by all 64 lanes of the wave that owns sample k.
struct PnpOrdLds { epnp_ord::Lds S; epnp_exact::Work W; };
// AGENT_OUT: the result is read by ANOTHER workgroup of the same launch - stored with agent-scope (sc1) stores, which reach the
// device-coherent level by themselves (no release fence: see tp_wait_work in svo_track.hip).
template <bool AGENT_OUT = false>
__device__ __forceinline__ void pnp_hyp_ord_wave(PnpOrdLds& L, const double* Xw, const double* uv, int n, const double* K,
                                                 const uint16_t* subset, PnpHyp* out, int k, bool force_seq = false) {
  const int lane = threadIdx.x & 63;
  if (k >= PNP_HYP) return;
  if (lane < 5) {
    const int e = min((int)subset[5 * k + lane], n - 1);
    L.S.pws[3 * lane] = Xw[3 * e]; L.S.pws[3 * lane + 1] = Xw[3 * e + 1]; L.S.pws[3 * lane + 2] = Xw[3 * e + 2];
    L.S.us[2 * lane] = uv[2 * e]; L.S.us[2 * lane + 1] = uv[2 * e + 1];
  }
  double R[9], t[3];
  const bool ok = epnp_ord::solve5_wave(L.S, L.W, K, R, t, nullptr, force_seq);
  int cnt = 0;
  if (ok)
    for (int e = lane; e < n; e += 64) cnt += pnp_inlier(R, t, Xw + 3 * e, uv + 2 * e, K) ? 1 : 0;
  cnt = wave_sum_i32_dpp(cnt);
  if (lane == 0) {
    if (AGENT_OUT) {
      PnpHyp* o = out + k;
#pragma unroll
      for (int i = 0; i < 9; ++i) __hip_atomic_store(reinterpret_cast<long long*>(&o->R[i]), __builtin_bit_cast(long long, R[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 3; ++i) __hip_atomic_store(reinterpret_cast<long long*>(&o->t[i]), __builtin_bit_cast(long long, t[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&o->cnt, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&o->ok, ok ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      PnpHyp h;
#pragma unroll
      for (int i = 0; i < 9; ++i) h.R[i] = R[i];
      h.t[0] = t[0]; h.t[1] = t[1]; h.t[2] = t[2];
      h.cnt = cnt; h.ok = ok ? 1 : 0;
      out[k] = h;
    }
  }
}

// RANSACUpdateNumIters(p = 0.99, ep, modelPoints = 5, maxIters)
__device__ __forceinline__ int pnp_update_iters(double ep, int maxIters) {
  const double p = 0.99;
  ep = fmax(ep, 0.); ep = fmin(ep, 1.);
  double num = fmax(1. - p, 2.2250738585072014e-308);
  double denom = 1. - pow(1. - ep, 5.0);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

// The iteration bound of RANSACPointSetRegistrator::run after its first m samples: samples at or beyond it can never be
// visited (the bound only shrinks), so they need not be solved.
__device__ __forceinline__ int pnp_bound_after(const int* cnt, const int* ok, int n, int m) {
  int niters = PNP_HYP, maxGood = 0;
  if (n == 5) return 1;
  for (int iter = 0; iter < m && iter < niters; ++iter) {
    if (!ok[iter]) continue;
    const int g = cnt[iter];
    if (g > max(maxGood, 4)) { maxGood = g; niters = pnp_update_iters((double)(n - g) / n, niters); }
  }
  return niters;
}

// pnp_update_iters's transcendental part for a sample with consensus g of n, computed by the sample's own thread ahead of the
// sequential rule below: *ld = log(1 - (1 - ep)^5) (or +1: "denominator below DBL_MIN", the function returns 0 then),
// *r = rint(log(0.01) / *ld).
__device__ __forceinline__ void pnp_update_terms(int g, int n, double* ld, int* r) {
  double ep = (double)(n - g) / n;
  ep = fmax(ep, 0.); ep = fmin(ep, 1.);
  const double denom = 1. - pow(1. - ep, 5.0);
  if (denom < 2.2250738585072014e-308) { *ld = 1.0; *r = 0; return; }
  const double num = log(fmax(1. - 0.99, 2.2250738585072014e-308));
  *ld = log(denom);
  *r = *ld < 0 ? (int)rint(num / *ld) : 0;
}
// the sequential rule with those terms at hand (same decisions as pnp_select)
__device__ __forceinline__ int pnp_select_pre(const int* cnt, const int* ok, const double* ld, const int* r, int n, int* good, int* iters) {
  int niters = PNP_HYP, maxGood = 0, best = -1, run = 0;
  if (n == 5) {
    *good = ok[0] ? 5 : 0; *iters = 1;
    return ok[0] ? 0 : -1;
  }
  const double num = log(fmax(1. - 0.99, 2.2250738585072014e-308));
  for (int iter = 0; iter < niters; ++iter) {
    ++run;
    if (!ok[iter]) continue;
    const int g = cnt[iter];
    if (g > max(maxGood, 4)) {
      maxGood = g; best = iter;
      const double d = ld[iter];
      if (d == 1.0) niters = 0;
      else niters = d >= 0 || -num >= niters * (-d) ? niters : r[iter];
    }
  }
  *good = maxGood; *iters = run;
  return best;
}

// The same loop over the first m samples only: the number of samples the rule visits if it ends within them (<= m), or m + 1 if
// it would go on - then nothing is decided yet.
__device__ __forceinline__ int pnp_select_pre_bound(const int* cnt, const int* ok, const double* ld, const int* r, int n, int m) {
  int niters = PNP_HYP, maxGood = 0;
  if (n == 5) return 1;
  const double num = log(fmax(1. - 0.99, 2.2250738585072014e-308));
  int iter = 0;
  for (; iter < niters; ++iter) {
    if (iter >= m) return m + 1;
    if (!ok[iter]) continue;
    const int g = cnt[iter];
    if (g > max(maxGood, 4)) {
      maxGood = g;
      const double d = ld[iter];
      if (d == 1.0) niters = 0;
      else niters = d >= 0 || -num >= niters * (-d) ? niters : r[iter];
    }
  }
  return iter;   // = samples visited (niters = 0 leaves after the sample that set it)
}

// RANSACPointSetRegistrator::run over the precomputed samples (one thread): sample `iter` replaces the best one iff its
// consensus is larger (and > 4), and the iteration bound shrinks with the inlier ratio.  Returns the winning sample
// (-1: none), *good its consensus, *iters the samples visited.
__device__ __forceinline__ int pnp_select(const int* cnt, const int* ok, int n, int* good, int* iters) {
  int niters = PNP_HYP, maxGood = 0, best = -1, run = 0;
  if (n == 5) {   // count == modelPoints: the one model, every point an inlier
    *good = ok[0] ? 5 : 0; *iters = 1;
    return ok[0] ? 0 : -1;
  }
  for (int iter = 0; iter < niters; ++iter) {
    ++run;
    if (!ok[iter]) continue;
    const int g = cnt[iter];
    if (g > max(maxGood, 4)) {
      maxGood = g; best = iter;
      niters = pnp_update_iters((double)(n - g) / n, niters);
    }
  }
  *good = maxGood; *iters = run;
  return best;
}
