// svo_epnp_exact_dev.h - EPnP on a five-point minimal set, ONE LANE per RANSAC sample, in OpenCV's own operation order.
//
// The parity mode behind svo_set_option("epnp_exact", 1).  The default solver (svo_epnp_dev.h) spreads one sample over a
// wavefront: parallel-order two-sided Jacobi, normal equations instead of SVD / QR least squares, FMA contraction and
// refined hardware reciprocals - the same estimator, but its rounding differs from a CPU run of OpenCV's code, and
// EPnP's N = 1 candidate starts from whatever basis of the two-dimensional null space the eigen-solver's rounding leaves.
// Here every sample is solved by one lane that walks the loops of OpenCV 3.2 one after the other
//   modules/calib3d/src/epnp.cpp   epnp::compute_pose: choose_control_points, compute_barycentric_coordinates, fill_M,
//                                  compute_L_6x10, compute_rho, find_betas_approx_{1,2,3}, gauss_newton + qr_solve,
//                                  compute_R_and_t, estimate_R_and_t, reprojection_error
//   modules/core/src/lapack.cpp    JacobiSVDImpl_<double> (cyclic one-sided Jacobi), SVBkSbImpl_ (cvSolve / cvInvert, CV_SVD)
// with IEEE division and square root and no FMA contraction (this file is compiled with -ffp-contract=off like the rest
// of the library; nothing in it opts back in) - IEEE operations only, in the same order, so a CPU run of the same loops gives
// the same bits - the discrete outcome of RANSAC (winner, visited samples, consensus) then matches the CPU restatement
// sample for sample, which is what tests/test_full_length.py demands over all 4,541 frames of the headline run.
// Every array lives in a caller-provided workspace (`Work`, in LDS: one sample per workgroup, lane 0 walks the loops) -
// nothing is a private array, so no scratch memory and no stack frames are involved; all functions are inlined.  An order
// of magnitude slower than the wave solver, by design only a checker for it - the fast mode is validated against this one.
#pragma once
#ifndef HIP_INCLUDE_HIP_HIP_RUNTIME_H
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#ifndef EPNP_X_FN
#define EPNP_X_FN __device__ __forceinline__
#endif

namespace epnp_exact {

struct Problem {
  double uc, vc, fu, fv;
  double pws[15], us[10], alphas[20], pcs[15];
  double cws[4][3], ccs[4][3];
};

// all arrays of one solve
struct Work {
  Problem e;
  double ut[144], vt[144], d[12], M[120], L[60], rho[6];
  double jW[12];                               // JacobiSVDImpl_'s W
  double sA[30], sUt[36], sVt[25], sw[6];      // the small systems: A (6 x nc / 3 x 3), their SVD
  double bx[5], gA[24], gb[6], gx[4], qA1[4], qA2[4];
  double m33[9], inv33[9], PW0[15];
  double dv[4][6][3];
  double Rs[3][9], ts[3][3], errs[3], betas[4], pc0[3], pw0[3];
};

// cv::RNG of JacobiSVDImpl_'s zero-singular-value branch (seed 0x12345678)
struct Rng {
  uint64_t state;
  EPNP_X_FN unsigned next() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
};

// cv::hypot (lapack.cpp, the template JacobiSVDImpl_ calls - not libm's): scaled, IEEE operations only
EPNP_X_FN double cv_hypot(double a, double b) {
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

// The part of JacobiSVDImpl_ behind its sweeps: singular values W[i] = |At[i]|, the selection sort into descending order (rows of
// At - and of Vt, when given - swapped along), rows of zero singular values replaced by orthogonalised vectors drawn from the
// function's own cv::RNG, every row scaled by 1 / W[i].  Separate so that a caller whose sweeps ran elsewhere (the wave engine of
// svo_epnp_ord_dev.h, bit-identical row for row) can finish a decomposition the way OpenCV does.
// (PW / PA: `double*`, or an LDS-qualified pointer - a caller that is NOT inlined into its kernel would otherwise walk the arrays
// through flat loads and stores, at twice the latency per dependent access)
template <typename PW, typename PA>
EPNP_X_FN void jacobi_svd_finish(PW W, PA At, int astep, double* _W, PA Vt, int vstep, int m, int n, int n1) {
  const double minval = 2.2250738585072014e-308, eps = 2.220446049250313e-16 * 10;
  double s, sd;
  for (int i = 0; i < n; i++) {
    sd = 0;
    for (int k = 0; k < m; k++) { const double t = At[i * astep + k]; sd += t * t; }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
      for (int k = 0; k < m; k++) { t = At[i * astep + k]; At[i * astep + k] = At[j * astep + k]; At[j * astep + k] = t; }
      if (Vt) for (int k = 0; k < n; k++) { t = Vt[i * vstep + k]; Vt[i * vstep + k] = Vt[j * vstep + k]; Vt[j * vstep + k] = t; }
    }
  }
  for (int i = 0; i < n; i++) _W[i] = W[i];
  Rng rng{0x12345678};
  for (int i = 0; i < n1; i++) {
    sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      // a zero singular value: a random vector, orthogonalised against the rows found so far
      const double val0 = 1. / m;
      for (int k = 0; k < m; k++) At[i * astep + k] = (rng.next() & 256) != 0 ? val0 : -val0;
      for (int iter = 0; iter < 2; iter++) {
        for (int j = 0; j < i; j++) {
          sd = 0;
          for (int k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
          double asum = 0;
          for (int k = 0; k < m; k++) {
            const double t = At[i * astep + k] - sd * At[j * astep + k];
            At[i * astep + k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < m; k++) At[i * astep + k] *= asum;
        }
        sd = 0;
        for (int k = 0; k < m; k++) { const double t = At[i * astep + k]; sd += t * t; }
        sd = sqrt(sd);
      }
    }
    s = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < m; k++) At[i * astep + k] *= s;
  }
}


// JacobiSVDImpl_<double>: one-sided Jacobi on the n rows (length m) of At; Vt n x n; singular values descending
template <typename PW, typename PA>
EPNP_X_FN void jacobi_svd(PW W /*[12] workspace*/, PA At, int astep, double* _W, PA Vt, int vstep, int m, int n, int n1) {
  const double minval = 2.2250738585072014e-308, eps = 2.220446049250313e-16 * 10;
  const int max_iter = m > 30 ? m : 30;
  double c, s, sd;
  for (int i = 0; i < n; i++) {
    sd = 0;
    for (int k = 0; k < m; k++) { const double t = At[i * astep + k]; sd += t * t; }
    W[i] = sd;
    for (int k = 0; k < n; k++) Vt[i * vstep + k] = 0;
    Vt[i * vstep + i] = 1;
  }
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        PA Ai = At + i * astep;
        PA Aj = At + j * astep;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = cv_hypot(p, beta);
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const double t0 = c * Ai[k] + s * Aj[k];
          const double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        PA Vi = Vt + i * vstep;
        PA Vj = Vt + j * vstep;
        for (int k = 0; k < n; k++) {
          const double t0 = c * Vi[k] + s * Vj[k];
          const double t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0; Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  jacobi_svd_finish(W, At, astep, _W, Vt, vstep, m, n, n1);
}

// cv::SVD::compute of a row-major m x n matrix (m >= n): w[n], Ut rows = left vectors (n x m), Vt rows = right vectors
EPNP_X_FN void svd_compute(double* jW, const double* A, int m, int n, double* w, double* Ut, double* Vt) {
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < m; ++k) Ut[i * m + k] = A[k * n + i];
  jacobi_svd(jW, Ut, m, w, Vt, n, m, n, n);
}
// SVBkSbImpl_: x = V diag(1/w) U^T b, singular values <= 2 eps sum(w) dropped; b == nullptr: the identity (m x m)
EPNP_X_FN void svd_backsubst(int m, int n, const double* w, const double* Ut, const double* Vt, const double* b, int nb,
                                     double* x) {
  double threshold = 0;
  if (!b) nb = m;
  for (int i = 0; i < n * nb; ++i) x[i] = 0;
  for (int i = 0; i < n; ++i) threshold += w[i];
  threshold *= 2.220446049250313e-16 * 2;
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
// cvSolve(A, b, x, CV_SVD), 6 x nc
EPNP_X_FN void solve_svd6(Work& W, const double* A, int nc, const double* b, double* x) {
  svd_compute(W.jW, A, 6, nc, W.sw, W.sUt, W.sVt);
  svd_backsubst(6, nc, W.sw, W.sUt, W.sVt, b, 1, x);
}

EPNP_X_FN double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
EPNP_X_FN double dist2(const double* p1, const double* p2) {
  return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

// epnp::qr_solve, literally (its `eta` scan looks at the diagonal element twice and never at the last row)
EPNP_X_FN void qr_solve_6x4(double* A1, double* A2, double* pA, double* pb, double* pX) {
  const int nr = 6, nc = 4;
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
  double* ppAjj = pA;
  for (int j = 0; j < nc; j++) {
    double* ppAij = ppAjj;
    double tau = 0;
    for (int i = j; i < nr; i++) { tau += *ppAij * pb[i]; ppAij += nc; }
    tau /= A1[j];
    ppAij = ppAjj;
    for (int i = j; i < nr; i++) { pb[i] -= tau * *ppAij; ppAij += nc; }
    ppAjj += nc + 1;
  }
  pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
  for (int i = nc - 2; i >= 0; i--) {
    const double* ppAij = pA + i * nc + (i + 1);
    double sum = 0;
    for (int j = i + 1; j < nc; j++) { sum += *ppAij * pX[j]; ppAij++; }
    pX[i] = (pb[i] - sum) / A2[i];
  }
}

EPNP_X_FN void gauss_newton(Work& W, const double* L, const double* rho, double* betas) {
  double* A = W.gA; double* b = W.gb; double* x = W.gx;
  for (int it = 0; it < 5; it++) {
    for (int i = 0; i < 6; i++) {
      const double* rl = L + i * 10;
      double* ra = A + i * 4;
      ra[0] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
      ra[1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
      ra[2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
      ra[3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
      b[i] = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] +
                       rl[3] * betas[0] * betas[2] + rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] +
                       rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] + rl[8] * betas[2] * betas[3] +
                       rl[9] * betas[3] * betas[3]);
    }
    x[0] = x[1] = x[2] = x[3] = 0;
    qr_solve_6x4(W.qA1, W.qA2, A, b, x);
    for (int i = 0; i < 4; i++) betas[i] += x[i];
  }
}

// compute_ccs, compute_pcs, solve_for_sign, estimate_R_and_t, reprojection_error
EPNP_X_FN double compute_R_and_t(Work& W, const double* ut, const double* betas, double* R, double* t) {
  Problem& e = W.e;
  for (int i = 0; i < 4; i++) e.ccs[i][0] = e.ccs[i][1] = e.ccs[i][2] = 0.0;
  for (int i = 0; i < 4; i++) {
    const double* v = ut + 12 * (11 - i);
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 3; k++) e.ccs[j][k] += betas[i] * v[3 * j + k];
  }
  for (int i = 0; i < 5; i++) {
    const double* a = e.alphas + 4 * i;
    double* pc = e.pcs + 3 * i;
    for (int j = 0; j < 3; j++) pc[j] = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] + a[3] * e.ccs[3][j];
  }
  if (e.pcs[2] < 0.0) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 3; j++) e.ccs[i][j] = -e.ccs[i][j];
    for (int i = 0; i < 15; i++) e.pcs[i] = -e.pcs[i];
  }
  double* pc0 = W.pc0; double* pw0 = W.pw0;
  for (int j = 0; j < 3; j++) { pc0[j] = 0; pw0[j] = 0; }
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 3; j++) { pc0[j] += e.pcs[3 * i + j]; pw0[j] += e.pws[3 * i + j]; }
  for (int j = 0; j < 3; j++) { pc0[j] /= 5; pw0[j] /= 5; }
  double* abt = W.m33; double* d = W.sw; double* Ut = W.sUt; double* Vt = W.sVt;
  for (int i = 0; i < 9; i++) abt[i] = 0;
  for (int i = 0; i < 5; i++) {
    const double* pc = e.pcs + 3 * i;
    const double* pw = e.pws + 3 * i;
    for (int j = 0; j < 3; j++) {
      abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
      abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
      abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
    }
  }
  svd_compute(W.jW, abt, 3, 3, d, Ut, Vt);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = Ut[0 * 3 + i] * Vt[0 * 3 + j] + Ut[1 * 3 + i] * Vt[1 * 3 + j] + Ut[2 * 3 + i] * Vt[2 * 3 + j];
  const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
  if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
  t[0] = pc0[0] - dot3(R, pw0); t[1] = pc0[1] - dot3(R + 3, pw0); t[2] = pc0[2] - dot3(R + 6, pw0);
  double sum2 = 0.0;
  for (int i = 0; i < 5; i++) {
    const double* pw = e.pws + 3 * i;
    const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
    const double ue = e.uc + e.fu * Xc * inv_Zc, ve = e.vc + e.fv * Yc * inv_Zc;
    const double u = e.us[2 * i], v = e.us[2 * i + 1];
    sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  return sum2 / 5;
}

// epnp::compute_pose on five correspondences.  rep (nullable): the three candidates' mean reprojection errors.
EPNP_X_FN bool solve5(Work& W, const double* Xw5, const double* uv5, const double* K, double* R_out, double* t_out, double* rep) {
  Problem& e = W.e;
  e.fu = K[0]; e.fv = K[1]; e.uc = K[2]; e.vc = K[3];
  for (int i = 0; i < 15; ++i) e.pws[i] = Xw5[i];
  for (int i = 0; i < 10; ++i) e.us[i] = uv5[i];
  // choose_control_points
  e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 3; j++) e.cws[0][j] += e.pws[3 * i + j];
  for (int j = 0; j < 3; j++) e.cws[0][j] /= 5;
  {
    double* PW0 = W.PW0; double* pw0tpw0 = W.m33; double* dc = W.sw; double* uct = W.sUt; double* vt = W.sVt;
    for (int i = 0; i < 5; i++)
      for (int j = 0; j < 3; j++) PW0[3 * i + j] = e.pws[3 * i + j] - e.cws[0][j];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double s = 0;
        for (int i = 0; i < 5; ++i) s += PW0[3 * i + a] * PW0[3 * i + b];
        pw0tpw0[3 * a + b] = s;
      }
    svd_compute(W.jW, pw0tpw0, 3, 3, dc, uct, vt);
    for (int i = 1; i < 4; i++) {
      const double k = sqrt(dc[i - 1] / 5);
      for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * uct[3 * (i - 1) + j];
    }
  }
  // compute_barycentric_coordinates
  {
    double* cc = W.m33; double* cc_inv = W.inv33; double* w = W.sw; double* Ut = W.sUt; double* Vt = W.sVt;
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
    svd_compute(W.jW, cc, 3, 3, w, Ut, Vt);
    svd_backsubst(3, 3, w, Ut, Vt, nullptr, 3, cc_inv);
    for (int i = 0; i < 5; i++) {
      const double* pi = e.pws + 3 * i;
      double* a = e.alphas + 4 * i;
      for (int j = 0; j < 3; j++)
        a[1 + j] = cc_inv[3 * j] * (pi[0] - e.cws[0][0]) + cc_inv[3 * j + 1] * (pi[1] - e.cws[0][1]) +
                   cc_inv[3 * j + 2] * (pi[2] - e.cws[0][2]);
      a[0] = 1.0f - a[1] - a[2] - a[3];
    }
  }
  // M, M^T M, its SVD (cvSVD(&MtM, &D, &Ut, 0, CV_SVD_MODIFY_A | CV_SVD_U_T))
  double* ut = W.ut; double* vt = W.vt; double* d = W.d;
  {
    double* M = W.M;
    for (int i = 0; i < 5; i++) {
      double* M1 = M + 2 * i * 12;
      double* M2 = M1 + 12;
      const double* as = e.alphas + 4 * i;
      const double u = e.us[2 * i], v = e.us[2 * i + 1];
      for (int q = 0; q < 4; q++) {
        M1[3 * q] = as[q] * e.fu; M1[3 * q + 1] = 0.0; M1[3 * q + 2] = as[q] * (e.uc - u);
        M2[3 * q] = 0.0; M2[3 * q + 1] = as[q] * e.fv; M2[3 * q + 2] = as[q] * (e.vc - v);
      }
    }
    // temp_a = (M^T M)^T, built straight into the SVD's working rows
    for (int a = 0; a < 12; ++a)
      for (int b = 0; b < 12; ++b) {
        double s = 0;
        for (int r = 0; r < 10; ++r) s += M[12 * r + a] * M[12 * r + b];
        ut[12 * b + a] = s;
      }
    jacobi_svd(W.jW, ut, 12, d, vt, 12, 12, 12, 12);
  }
  // compute_L_6x10, compute_rho
  double* L = W.L; double* rho = W.rho;
  {
    double (*dv)[6][3] = W.dv;
    for (int i = 0; i < 4; i++) {
      const double* v = ut + 12 * (11 - i);
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = v[3 * a] - v[3 * b];
        dv[i][j][1] = v[3 * a + 1] - v[3 * b + 1];
        dv[i][j][2] = v[3 * a + 2] - v[3 * b + 2];
        b++;
        if (b > 3) { a++; b = a + 1; }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = L + 10 * i;
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
    rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]); rho[2] = dist2(e.cws[0], e.cws[3]);
    rho[3] = dist2(e.cws[1], e.cws[2]); rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);
  }
  double (*Rs)[9] = W.Rs; double (*ts)[3] = W.ts; double* errs = W.errs;
  for (int cand = 0; cand < 3; ++cand) {
    double* betas = W.betas; double* l = W.sA; double* bx = W.bx;
    if (cand == 0) {          // find_betas_approx_1: [B11 B12 B13 B14]
      for (int i = 0; i < 6; i++) { l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6]; }
      solve_svd6(W, l, 4, rho, bx);
      if (bx[0] < 0) { betas[0] = sqrt(-bx[0]); betas[1] = -bx[1] / betas[0]; betas[2] = -bx[2] / betas[0]; betas[3] = -bx[3] / betas[0]; }
      else { betas[0] = sqrt(bx[0]); betas[1] = bx[1] / betas[0]; betas[2] = bx[2] / betas[0]; betas[3] = bx[3] / betas[0]; }
    } else if (cand == 1) {   // find_betas_approx_2: [B11 B12 B22]
      for (int i = 0; i < 6; i++) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
      solve_svd6(W, l, 3, rho, bx);
      if (bx[0] < 0) { betas[0] = sqrt(-bx[0]); betas[1] = (bx[2] < 0) ? sqrt(-bx[2]) : 0.0; }
      else { betas[0] = sqrt(bx[0]); betas[1] = (bx[2] > 0) ? sqrt(bx[2]) : 0.0; }
      if (bx[1] < 0) betas[0] = -betas[0];
      betas[2] = 0.0; betas[3] = 0.0;
    } else {                  // find_betas_approx_3: [B11 B12 B22 B13 B23]
      for (int i = 0; i < 6; i++)
        for (int c = 0; c < 5; ++c) l[5 * i + c] = L[10 * i + c];
      solve_svd6(W, l, 5, rho, bx);
      if (bx[0] < 0) { betas[0] = sqrt(-bx[0]); betas[1] = (bx[2] < 0) ? sqrt(-bx[2]) : 0.0; }
      else { betas[0] = sqrt(bx[0]); betas[1] = (bx[2] > 0) ? sqrt(bx[2]) : 0.0; }
      if (bx[1] < 0) betas[0] = -betas[0];
      betas[2] = bx[3] / betas[0];
      betas[3] = 0.0;
    }
    gauss_newton(W, L, rho, betas);
    errs[cand] = compute_R_and_t(W, ut, betas, Rs[cand], ts[cand]);
  }
  int N = 0;
  if (errs[1] < errs[0]) N = 1;
  if (errs[2] < errs[N]) N = 2;
  bool fin = true;
  for (int k = 0; k < 9; ++k) { R_out[k] = Rs[N][k]; fin = fin && isfinite(Rs[N][k]); }
  for (int k = 0; k < 3; ++k) { t_out[k] = ts[N][k]; fin = fin && isfinite(ts[N][k]); }
  if (rep) { rep[0] = errs[0]; rep[1] = errs[1]; rep[2] = errs[2]; }
  return fin;
}

}  // namespace epnp_exact
