// svo_epnp_ord_dev.h - EPnP on a five-point minimal set, ONE WAVE per RANSAC sample, OpenCV's rounding kept operation by
// operation ("order-preserving"): the solver behind svo_set_option("epnp_exact", 2), the tracker's default.
//
// What has to hold for a RANSAC sample's pose to come out with the bits a CPU run of OpenCV 3.2 gives
// (cv::solvePnPRansac of reference src/pnpmatch.cc:227 -> modules/calib3d/src/epnp.cpp, modules/core/src/lapack.cpp):
// every floating-point operation sees the same operands and is rounded once - IEEE add / mul / div / sqrt, no FMA
// contraction, every `s += x[k] * y[k]` loop summed in its own k order.  What does NOT have to hold is the order in which
// INDEPENDENT operations run.  svo_epnp_exact_dev.h walks the loops on one lane (the checker, an order of magnitude slower
// than the statistical wave solver of svo_epnp_dev.h); this file spreads exactly those operations over the wavefront:
//
//   * JacobiSVDImpl_ (cyclic one-sided Jacobi): a pair (i, j) reads and writes rows i and j only.  The row update
//     t0 = c Ai[k] + s Aj[k], t1 = -s Ai[k] + c Aj[k] is elementwise in k: one lane per column.  The three k-ordered sums
//     of a pair (p = sum Ai[k] Aj[k], a = sum t0^2, b = sum t1^2) stay sequential chains, but every lane of the pair's
//     16-lane DPP row runs the chain on operands broadcast from lane k (v_mov_b64_dpp row_newbcast:k), so the rotation
//     (c, s) is known to all lanes of the row without a further exchange.  Pairs on disjoint rows commute exactly: the
//     12 x 12 eigen-problem of M^T M runs FOUR pairs at a time - one per DPP row of the wave - from a static list
//     schedule of the cyclic order (tools/gen_jacobi_schedule.py: 16.5 steps per sweep instead of 66, sweeps overlapped,
//     verified bit-identical to the sequential loop).  V is not accumulated there: epnp only asks for U.
//   * the small decompositions (3 x 3: control points, barycentric inverse, absolute orientation; 6 x 3 / 6 x 4 / 6 x 5:
//     the three beta initialisations) run one problem per DPP row, the three EPnP candidates side by side.
//   * everything else (M, M^T M, L_6x10, rho, cvInvert / cvSolve back-substitution, compute_ccs / pcs, estimate_R_and_t,
//     reprojection_error) is one lane per OUTPUT element, each lane summing its element in the loop's own order; the
//     five Gauss-Newton steps with epnp::qr_solve are scalar code, one candidate per DPP row, arrays in registers.
//   * the branches OpenCV takes once in a blue moon - a zero singular value (JacobiSVDImpl_ then draws a random vector),
//     two equal singular values (the selection sort's swap order would matter) - are not reproduced here: the wave
//     notices them and lane 0 re-solves the sample with the sequential restatement (epnp_exact::solve5).
//
// The file is compiled with -ffp-contract=off like the rest of the library; `/` and sqrt() are the compiler's IEEE
// (correctly rounded) sequences.
#pragma once
#ifndef HIP_INCLUDE_HIP_HIP_RUNTIME_H
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "svo_epnp_exact_dev.h"

namespace epnp_ord {

#define EO_FN __device__ __forceinline__

#include "svo_epnp_ord_tab.h"

// One wave, its own LDS workspace: LDS instructions of a wave execute in issue order, so a write is visible to the reads
// issued after it; the wavefront-scope fence only keeps the COMPILER from moving a read above the write it depends on.
#define EO_SYNC()                                         \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
  } while (0)

struct Lds {
  double jr[16 * 16];       // Jacobi rows [A (M columns) | V (n columns)], row stride 16; 12 x 12: rows 0..11, small ones: rows 5 p .. 5 p + n - 1 of problem p
  double jw[16];            // their W (squared row norms while rotating, singular values afterwards)
  double pws[16], us[10], alphas[20], cws[12], ccinv[9];
  double M[120];
  double ut4[4][12];        // ut + 12 * (11 - q): vectors of the four smallest singular values, q = 0 the smallest
  double L[60], rho[6];
  double sw[3][6];          // per small problem: singular values, descending
  double bx[3][6];          // per candidate: solution of the beta initialisation
  double ccs[3][12], pcs[3][16], abt[3][9], Rc[3][9], rep[3][8];
  double out[3][16];        // per candidate: R (9), t (3), mean reprojection error
  int srow[3][6];           // per small problem: Jacobi row (index into jr rows) at sorted position p
  int urow[4];              // 12 x 12: row of the (11 - q)-th singular value
  int flag;                 // a branch this file does not reproduce was met: re-solve sequentially
  int sweeps;               // diagnostics: sweeps of the 12 x 12 decomposition
  long long stamp[8];
};

// ---- row broadcast and in-order sums -------------------------------------------------------------------------------
// v_mov_b64_dpp row_newbcast:K - every lane of a 16-lane row receives lane K's value.  Inline assembly (the DPP builtin
// of this compiler is 32-bit only); FIRST additionally waits out the VALU-write -> DPP-read and EXEC -> DPP hazards the
// compiler's hazard recogniser does not see inside an asm blob - the later moves of a chain read the same, long-written
// source register.
template <int K, bool FIRST>
EO_FN double rbc(double x) {
  double y;
  if (FIRST)
    asm volatile("s_nop 4\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x), "n"(K));
  else
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x), "n"(K));
  return y;
}
template <int K, int M>
struct RowSum {
  static EO_FN double run(double acc, double x) { return RowSum<K + 1, M>::run(acc + rbc<K, K == 0>(x), x); }
  static EO_FN void run2(double& a, double& b, double x, double y) {
    a = a + rbc<K, K == 0>(x);
    b = b + rbc<K, false>(y);
    RowSum<K + 1, M>::run2(a, b, x, y);
  }
};
template <int M>
struct RowSum<M, M> {
  static EO_FN double run(double acc, double) { return acc; }
  static EO_FN void run2(double&, double&, double, double) {}
};
// s = 0; for (k = 0; k < M; k++) s += x[lane k of the row]; - in every lane of the row.  All 64 lanes must be active.
template <int M>
EO_FN double row_sum(double x) { return RowSum<0, M>::run(0.0, x); }
template <int M>
EO_FN void row_sum2(double x, double y, double& a, double& b) {
  a = 0.0; b = 0.0;
  RowSum<0, M>::run2(a, b, x, y);
}

EO_FN double xdiv(double a, double b) { return a / b; }
EO_FN double xsqrt(double x) { return sqrt(x); }

// ---- one (i, j) visit of JacobiSVDImpl_<double> for the pair held by this lane's DPP row ---------------------------------
// ai / aj: this lane's column of rows i and j (columns >= M: the V part, rotated along), wi / wj: W[i], W[j].  Returns whether
// THIS row's pair rotated (ai, aj, wi, wj updated then); `any` tells whether any row of the wave did.
template <int M>
EO_FN bool jpair(double& ai, double& aj, double& wi, double& wj, bool valid, bool& any) {
  const double eps = 2.220446049250313e-16 * 10;
  const double p0 = row_sum<M>(ai * aj);
  const bool rot = valid && !(fabs(p0) <= eps * xsqrt(wi * wj));
  any = __any(rot);
  if (!any) return false;
  const double p = p0 * 2;
  const double beta = wi - wj;
  // cv::hypot(p, beta)
  const double pa = fabs(p), pb = fabs(beta);
  const bool agb = pa > pb;
  const double hi = agb ? pa : pb, lo = agb ? pb : pa;
  const double q = xdiv(lo, hi);
  const double g = hi * xsqrt(1 + q * q);
  const double gamma = (agb || pb > 0) ? g : 0.0;
  const bool neg = beta < 0;
  //   beta < 0:  delta = (gamma - beta) * 0.5; s = sqrt(delta / gamma); c = p / (gamma * s * 2)
  //   else:      c = sqrt((gamma + beta) / (gamma * 2));                s = p / (gamma * c * 2)
  const double num = neg ? (gamma - beta) * 0.5 : (gamma + beta);
  const double den = neg ? gamma : gamma * 2;
  const double r1 = xsqrt(xdiv(num, den));
  const double r2 = xdiv(p, gamma * r1 * 2);
  const double s = neg ? r1 : r2, c = neg ? r2 : r1;
  const double t0 = c * ai + s * aj;
  const double t1 = -s * ai + c * aj;
  double a, b;
  row_sum2<M>(t0 * t0, t1 * t1, a, b);
  if (rot) { ai = t0; aj = t1; wi = a; wj = b; }
  return rot;
}

// ---- up to four independent small problems, one per DPP row, each walked in the cyclic order ------------------------------
// Problem of this lane's row: n Jacobi rows (n = 0: the row idles) starting at row `base` of S.jr, M columns of A followed by
// n columns of V.  JacobiSVDImpl_ up to the end of its sweeps: rows rotated in place, S.jw = squared row norms.
template <int M>
EO_FN void jacobi_seq(Lds& S, int n, int base, int lane) {
  const int col = lane & 15;
  for (int r = 0; r < 5; ++r) {
    const bool act = r < n;
    const double x = S.jr[(base + (act ? r : 0)) * 16 + col];
    const double w = row_sum<M>(x * x);
    if (act && col == 0) S.jw[base + r] = w;
  }
  EO_SYNC();
  int i = 0, j = 1, iter = 0;
  bool changed = false, active = n >= 2;
  while (__any(active)) {
    const int ri = (base + i) * 16 + col, rj = (base + j) * 16 + col;
    double ai = S.jr[ri], aj = S.jr[rj], wi = S.jw[base + i], wj = S.jw[base + j];
    bool any;
    const bool rot = jpair<M>(ai, aj, wi, wj, active, any);
    if (rot) {
      S.jr[ri] = ai; S.jr[rj] = aj;
      if (col == 0) { S.jw[base + i] = wi; S.jw[base + j] = wj; }
    }
    EO_SYNC();
    changed = changed || rot;
    if (active) {
      ++j;
      if (j == n) {
        ++i; j = i + 1;
        if (i == n - 1) {         // end of a sweep: for (iter < max(m, 30)) { ...; if (!changed) break; }
          if (!changed || iter + 1 == 30) active = false;
          ++iter; i = 0; j = 1; changed = false;
        }
      }
    }
  }
}

// After the sweeps: W[i] = sqrt(sum At[i][k]^2), descending order, rows scaled by 1 / W (cv::SVD's U^T); the V part stays.
// Per problem p (this lane's row, p < 3): S.sw[p][pos] the singular values, S.srow[p][pos] the Jacobi row at sorted position pos.
// A singular value <= DBL_MIN or two equal ones: S.flag.
template <int M>
EO_FN void svd_finish(Lds& S, int n, int base, int p, int lane) {
  const int col = lane & 15;
  for (int r = 0; r < 5; ++r) {
    const bool act = r < n;
    const double x = S.jr[(base + (act ? r : 0)) * 16 + col];
    const double w = xsqrt(row_sum<M>(x * x));
    if (act && col == 0) S.jw[base + r] = w;
  }
  EO_SYNC();
  if (col < n) {
    const double my = S.jw[base + col];
    int rank = 0;
    bool bad = !(my > 2.2250738585072014e-308);
    for (int q = 0; q < 5; ++q)
      if (q < n && q != col) {
        const double o = S.jw[base + q];
        rank += o > my ? 1 : 0;
        bad = bad || o == my;
      }
    if (bad) S.flag = 1;
    S.sw[p][rank] = my;
    S.srow[p][rank] = base + col;
  }
  for (int r = 0; r < 5; ++r) {
    const bool act = r < n;
    const int a = (base + (act ? r : 0)) * 16 + col;
    const double s = 1 / S.jw[base + (act ? r : 0)];
    const double x = S.jr[a];
    if (act && col < M) S.jr[a] = x * s;
  }
  EO_SYNC();
}

// ---- the 12 x 12 decomposition of M^T M: four pairs per step from the static schedule --------------------------------
EO_FN void jacobi12(Lds& S, int lane) {
  const int slot = lane >> 4, col = lane & 15;
  for (int r0 = 0; r0 < 12; r0 += 4) {
    const double x = S.jr[(r0 + slot) * 16 + col];
    const double w = row_sum<12>(x * x);
    if (col == 0) S.jw[r0 + slot] = w;
  }
  EO_SYNC();
  unsigned chg = 0;   // bit s: a pair of sweep s handled by this row rotated
  // the step's four entries are one uniform 64-bit (scalar) load, requested one step ahead
  const uint64_t* tab64 = reinterpret_cast<const uint64_t*>(&c_j12_tab[0][0]);
  int tt = 0, sbase = 0, last = 0;
  uint64_t e4 = tab64[0];
  int cl = c_j12_close[0];
  for (;;) {
    int tn = tt + 1, sbn = sbase;
    if (tn == EO_J12_STEPS) { tn = EO_J12_PROLOGUE; sbn += 2; }
    const uint64_t e4n = tab64[tn];
    const int cln = c_j12_close[tn];
    const unsigned e = (unsigned)(e4 >> (16 * slot)) & 0xffffu;
    const int i = e & 15, j = (e >> 4) & 15, sw = sbase + (int)((e >> 8) & 3);
    const bool valid = e != 0xffffu && sw < 30;
    const int ri = (valid ? i : 0) * 16 + col, rj = (valid ? j : 1) * 16 + col;
    double ai = S.jr[ri], aj = S.jr[rj], wi = S.jw[valid ? i : 0], wj = S.jw[valid ? j : 1];
    bool any;
    const bool rot = jpair<12>(ai, aj, wi, wj, valid, any);
    if (rot) {
      S.jr[ri] = ai; S.jr[rj] = aj;
      if (col == 0) { S.jw[i] = wi; S.jw[j] = wj; }
      chg |= 1u << sw;
    }
    EO_SYNC();
    if (cl >= 0) {    // the (10, 11) pair of sweep sbase + cl ran in this step: that sweep is complete
      const int sc = sbase + cl;
      last = sc;
      if (sc >= 29 || !__any((chg >> sc) & 1u)) break;
    }
    tt = tn; sbase = sbn; e4 = e4n; cl = cln;
  }
  if (lane == 0) S.sweeps = last + 1;
  // singular values; the rows of the four smallest, scaled: S.ut4[q] = ut + 12 * (11 - q)
  for (int r0 = 0; r0 < 12; r0 += 4) {
    const double x = S.jr[(r0 + slot) * 16 + col];
    const double w = xsqrt(row_sum<12>(x * x));
    if (col == 0) S.jw[r0 + slot] = w;
  }
  EO_SYNC();
  if (lane < 12) {
    const double my = S.jw[lane];
    int rank = 0;
    bool bad = !(my > 2.2250738585072014e-308);
    for (int q = 0; q < 12; ++q)
      if (q != lane) {
        const double o = S.jw[q];
        rank += o > my ? 1 : 0;
        bad = bad || o == my;
      }
    if (bad) S.flag = 1;
    if (rank >= 8) S.urow[11 - rank] = lane;
  }
  EO_SYNC();
  if (lane < 48) {
    const int q = lane / 12, k = lane % 12, r = S.urow[q] & 15;
    const double s = 1 / S.jw[r];
    S.ut4[q][k] = S.jr[r * 16 + k] * s;
  }
  EO_SYNC();
}

// ---- epnp::gauss_newton + qr_solve for one candidate, scalar code (arrays in registers) ---------------------------------
EO_FN void gauss_newton(const double* L, const double* rho, double* betas) {
  for (int it = 0; it < 5; ++it) {
    double A[6][4], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double* rl = L + 10 * i;
      A[i][0] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
      A[i][1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
      A[i][2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
      A[i][3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
      b[i] = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] +
                       rl[3] * betas[0] * betas[2] + rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] +
                       rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] + rl[8] * betas[2] * betas[3] +
                       rl[9] * betas[3] * betas[3]);
    }
    // epnp::qr_solve, literally (its `eta` scan looks at the diagonal element twice and never at the last row); the early
    // `return` on eta == 0 (x stays 0) becomes the `alive` predicate
    double A1[4], A2[4], x[4] = {0, 0, 0, 0};
    bool alive = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double eta = fabs(A[k][k]);
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        const double elt = fabs(A[i - 1][k]);
        if (eta < elt) eta = elt;
      }
      if (eta == 0) alive = false;
      const double inv_eta = 1. / eta;
      double sum2 = 0.0;
#pragma unroll
      for (int i = k; i < 6; ++i) {
        A[i][k] *= inv_eta;
        sum2 += A[i][k] * A[i][k];
      }
      double sigma = xsqrt(sum2);
      if (A[k][k] < 0) sigma = -sigma;
      A[k][k] += sigma;
      A1[k] = sigma * A[k][k];
      A2[k] = -eta * sigma;
#pragma unroll
      for (int j = k + 1; j < 4; ++j) {
        double sum = 0;
#pragma unroll
        for (int i = k; i < 6; ++i) sum += A[i][k] * A[i][j];
        const double tau = sum / A1[k];
#pragma unroll
        for (int i = k; i < 6; ++i) A[i][j] -= tau * A[i][k];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double tau = 0;
#pragma unroll
      for (int i = j; i < 6; ++i) tau += A[i][j] * b[i];
      tau /= A1[j];
#pragma unroll
      for (int i = j; i < 6; ++i) b[i] -= tau * A[i][j];
    }
    x[3] = b[3] / A2[3];
#pragma unroll
    for (int i = 2; i >= 0; --i) {
      double sum = 0;
#pragma unroll
      for (int j = i + 1; j < 4; ++j) sum += A[i][j] * x[j];
      x[i] = (b[i] - sum) / A2[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) betas[i] += alive ? x[i] : 0.0;
  }
}

EO_FN double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// epnp::compute_pose on the five correspondences in S.pws (5 x 3) / S.us (5 x 2) (doubles holding float values), K = {fu, fv,
// uc, vc}.  Called by all 64 lanes of a wave; returns (in every lane) whether the pose is finite, R_out / t_out in every lane.
// `xw` is the LDS workspace of the sequential fallback (force_seq: take it regardless - tests).
EO_FN bool solve5_wave(Lds& S, epnp_exact::Work& xw, const double* K, double* R_out, double* t_out, double* rep, bool force_seq = false) {
  const int lane = threadIdx.x & 63, slot = lane >> 4, col = lane & 15;
  const double fu = K[0], fv = K[1], uc = K[2], vc = K[3];
  if (lane == 0) { S.flag = force_seq ? 1 : 0; S.stamp[0] = clock64(); }   // force_seq: tests exercise the sequential fallback
  EO_SYNC();
  // ---- choose_control_points -----------------------------------------------------------------------------------------
  if (lane < 3) {
    double c = 0;
    for (int i = 0; i < 5; i++) c += S.pws[3 * i + lane];
    S.cws[lane] = c / 5;
  }
  EO_SYNC();
  if (lane < 9) {            // PW0^T PW0 (cvMulTransposed), loaded transposed into the Jacobi rows (temp_a = A^T)
    const int a = lane / 3, b = lane % 3;
    const double ca = S.cws[a], cb = S.cws[b];
    double s = 0;
    for (int i = 0; i < 5; ++i) s += (S.pws[3 * i + a] - ca) * (S.pws[3 * i + b] - cb);
    S.jr[b * 16 + a] = s;
  } else if (lane < 18) {
    const int r = (lane - 9) / 3, k = (lane - 9) % 3;
    S.jr[r * 16 + 3 + k] = r == k ? 1.0 : 0.0;
  }
  EO_SYNC();
  jacobi_seq<3>(S, slot == 0 ? 3 : 0, 0, lane);
  svd_finish<3>(S, slot == 0 ? 3 : 0, 0, 0, lane);
  if (lane < 9) {
    const int i = 1 + lane / 3, j = lane % 3;
    const double k = xsqrt(S.sw[0][i - 1] / 5);
    S.cws[3 * i + j] = S.cws[j] + k * S.jr[S.srow[0][i - 1] * 16 + j];
  }
  EO_SYNC();
  // ---- compute_barycentric_coordinates: cvInvert(CC, CV_SVD) ---------------------------------------------------------
  if (lane < 9) {            // cc[3 i + j - 1] = cws[j][i] - cws[0][i]; Jacobi row r, column k = cc[3 k + r]
    const int r = lane / 3, k = lane % 3;
    S.jr[r * 16 + k] = S.cws[3 * (r + 1) + k] - S.cws[k];
  } else if (lane < 18) {
    const int r = (lane - 9) / 3, k = (lane - 9) % 3;
    S.jr[r * 16 + 3 + k] = r == k ? 1.0 : 0.0;
  }
  EO_SYNC();
  jacobi_seq<3>(S, slot == 0 ? 3 : 0, 0, lane);
  svd_finish<3>(S, slot == 0 ? 3 : 0, 0, 0, lane);
  if (lane < 9) {            // SVBkSbImpl_ with the identity as right-hand side: x[3 j + c] += (Ut[i][c] / w[i]) * Vt[i][j]
    const int j = lane / 3, c = lane % 3;
    double threshold = 0;
    for (int i = 0; i < 3; ++i) threshold += S.sw[0][i];
    threshold *= 2.220446049250313e-16 * 2;
    double x = 0;
    for (int i = 0; i < 3; ++i) {
      double wi = S.sw[0][i];
      if (fabs(wi) <= threshold) continue;
      wi = 1 / wi;
      const int r = S.srow[0][i];
      double s = S.jr[r * 16 + c];
      s *= wi;
      x = x + s * S.jr[r * 16 + 3 + j];
    }
    S.ccinv[lane] = x;
  }
  EO_SYNC();
  if (lane < 15) {
    const int i = lane / 3, j = lane % 3;
    const double* ci = S.ccinv;
    const double* pi = S.pws + 3 * i;
    S.alphas[4 * i + 1 + j] = ci[3 * j] * (pi[0] - S.cws[0]) + ci[3 * j + 1] * (pi[1] - S.cws[1]) + ci[3 * j + 2] * (pi[2] - S.cws[2]);
  }
  EO_SYNC();
  if (lane < 5) {
    const double* a = S.alphas + 4 * lane;
    S.alphas[4 * lane] = 1.0f - a[1] - a[2] - a[3];
  }
  EO_SYNC();
  if (lane == 0) S.stamp[1] = clock64();
  // ---- fill_M, M^T M (cvMulTransposed), loaded as temp_a = (M^T M)^T ---------------------------------------------------
  for (int e = lane; e < 120; e += 64) {
    const int r = e / 12, cc = e % 12, i = r >> 1, q = cc / 3, t = cc % 3;
    const double as = S.alphas[4 * i + q], u = S.us[2 * i], v = S.us[2 * i + 1];
    double m;
    if (r & 1) m = t == 0 ? 0.0 : (t == 1 ? as * fv : as * (vc - v));
    else m = t == 0 ? as * fu : (t == 1 ? 0.0 : as * (uc - u));
    S.M[e] = m;
  }
  EO_SYNC();
  for (int e = lane; e < 144; e += 64) {
    const int a = e / 12, b = e % 12;
    double s = 0;
    for (int r = 0; r < 10; ++r) s += S.M[12 * r + a] * S.M[12 * r + b];
    S.jr[b * 16 + a] = s;
  }
  EO_SYNC();
  if (lane == 0) S.stamp[2] = clock64();
  jacobi12(S, lane);
  if (lane == 0) S.stamp[3] = clock64();
  // ---- compute_L_6x10, compute_rho ---------------------------------------------------------------------------------------
  if (lane < 60) {
    const int i = lane / 10, c = lane % 10;
    const int pa = i < 3 ? 0 : (i < 5 ? 1 : 2), pb = i < 3 ? i + 1 : (i < 5 ? i - 1 : 3);
    const int y = c >= 6 ? 3 : (c >= 3 ? 2 : (c >= 1 ? 1 : 0)), x = c - y * (y + 1) / 2;
    double dx[3], dy[3];
    for (int k = 0; k < 3; ++k) {
      dx[k] = S.ut4[x][3 * pa + k] - S.ut4[x][3 * pb + k];
      dy[k] = S.ut4[y][3 * pa + k] - S.ut4[y][3 * pb + k];
    }
    const double d = dot3(dx, dy);
    S.L[lane] = x == y ? d : 2.0f * d;
  }
  if (lane < 6) {
    const int pa = lane < 3 ? 0 : (lane < 5 ? 1 : 2), pb = lane < 3 ? lane + 1 : (lane < 5 ? lane - 1 : 3);
    const double* p1 = S.cws + 3 * pa;
    const double* p2 = S.cws + 3 * pb;
    S.rho[lane] = (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
  }
  EO_SYNC();
  // ---- the three candidates, one per DPP row: find_betas_approx_{1,2,3} = cvSolve(L_6xnc, rho, CV_SVD) -----------------------
  const int cand = slot < 3 ? slot : 0;
  const int nc = slot == 0 ? 4 : (slot == 1 ? 3 : (slot == 2 ? 5 : 0));
  const int base = 5 * cand;
  if (slot < 3) {
    for (int i = 0; i < nc; ++i) {
      // l[k][i] = L[10 k + column i of the candidate]: approx_1 [B11 B12 B13 B14] = 0 1 3 6, approx_2 = 0 1 2, approx_3 = 0 1 2 3 4
      const int lc = slot == 0 ? (i == 2 ? 3 : (i == 3 ? 6 : i)) : i;
      double v = 0.0;
      if (col < 6) v = S.L[10 * col + lc];
      else if (col - 6 == i) v = 1.0;
      S.jr[(base + i) * 16 + col] = v;
    }
  }
  EO_SYNC();
  jacobi_seq<6>(S, nc, base, lane);
  svd_finish<6>(S, nc, base, cand, lane);
  if (slot < 3 && col < nc) {   // SVBkSbImpl_: x[j] += (sum_k Ut[i][k] b[k] / w[i]) * Vt[i][j]
    double threshold = 0;
    for (int i = 0; i < nc; ++i) threshold += S.sw[cand][i];
    threshold *= 2.220446049250313e-16 * 2;
    double x = 0;
    for (int i = 0; i < nc; ++i) {
      double wi = S.sw[cand][i];
      if (fabs(wi) <= threshold) continue;
      wi = 1 / wi;
      const int r = S.srow[cand][i];
      double s = 0;
      for (int k = 0; k < 6; ++k) s += S.jr[r * 16 + k] * S.rho[k];
      s *= wi;
      x = x + s * S.jr[r * 16 + 6 + col];
    }
    S.bx[cand][col] = x;
  }
  EO_SYNC();
  if (lane == 0) S.stamp[4] = clock64();
  double betas[4];
  {
    const double* bx = S.bx[cand];
    const double b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3];
    if (slot == 0) {
      if (b0 < 0) { betas[0] = xsqrt(-b0); betas[1] = -b1 / betas[0]; betas[2] = -b2 / betas[0]; betas[3] = -b3 / betas[0]; }
      else { betas[0] = xsqrt(b0); betas[1] = b1 / betas[0]; betas[2] = b2 / betas[0]; betas[3] = b3 / betas[0]; }
    } else {
      if (b0 < 0) { betas[0] = xsqrt(-b0); betas[1] = (b2 < 0) ? xsqrt(-b2) : 0.0; }
      else { betas[0] = xsqrt(b0); betas[1] = (b2 > 0) ? xsqrt(b2) : 0.0; }
      if (b1 < 0) betas[0] = -betas[0];
      betas[2] = slot == 2 ? b3 / betas[0] : 0.0;
      betas[3] = 0.0;
    }
  }
  {
    double rho[6];
    for (int k = 0; k < 6; ++k) rho[k] = S.rho[k];
    gauss_newton(S.L, rho, betas);
  }
  if (lane == 0) S.stamp[5] = clock64();
  // ---- compute_R_and_t: compute_ccs, compute_pcs, solve_for_sign, estimate_R_and_t, reprojection_error ---------------------
  if (slot < 3 && col < 12) {
    double c = 0.0;
    for (int i = 0; i < 4; ++i) c += betas[i] * S.ut4[i][col];
    S.ccs[cand][col] = c;
  }
  EO_SYNC();
  if (slot < 3 && col < 15) {
    const int i = col / 3, j = col % 3;
    const double* a = S.alphas + 4 * i;
    const double* cc = S.ccs[cand];
    S.pcs[cand][col] = a[0] * cc[j] + a[1] * cc[3 + j] + a[2] * cc[6 + j] + a[3] * cc[9 + j];
  }
  EO_SYNC();
  double pc0[3], pw0[3];
  {
    const double* pc = S.pcs[cand];
    const bool flip = pc[2] < 0.0;
    for (int j = 0; j < 3; ++j) { pc0[j] = 0; pw0[j] = 0; }
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 3; ++j) { pc0[j] += flip ? -pc[3 * i + j] : pc[3 * i + j]; pw0[j] += S.pws[3 * i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= 5; pw0[j] /= 5; }
    if (slot < 3 && col < 9) {        // ABt, loaded transposed into the Jacobi rows
      const int j = col / 3, k = col % 3;
      double s = 0;
      for (int i = 0; i < 5; ++i) s += ((flip ? -pc[3 * i + j] : pc[3 * i + j]) - pc0[j]) * (S.pws[3 * i + k] - pw0[k]);
      S.jr[(base + k) * 16 + j] = s;
    } else if (slot < 3 && col < 12) {
      const int r = col - 9;
      for (int k = 0; k < 3; ++k) S.jr[(base + r) * 16 + 3 + k] = r == k ? 1.0 : 0.0;
    }
  }
  EO_SYNC();
  jacobi_seq<3>(S, slot < 3 ? 3 : 0, base, lane);
  svd_finish<3>(S, slot < 3 ? 3 : 0, base, cand, lane);
  if (slot < 3 && col < 9) {
    const int i = col / 3, j = col % 3;
    const int r0 = S.srow[cand][0], r1 = S.srow[cand][1], r2 = S.srow[cand][2];
    S.Rc[cand][col] = S.jr[r0 * 16 + i] * S.jr[r0 * 16 + 3 + j] + S.jr[r1 * 16 + i] * S.jr[r1 * 16 + 3 + j] + S.jr[r2 * 16 + i] * S.jr[r2 * 16 + 3 + j];
  }
  EO_SYNC();
  double R[9], t[3];
  for (int k = 0; k < 9; ++k) R[k] = S.Rc[cand][k];
  {
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
  }
  t[0] = pc0[0] - dot3(R, pw0); t[1] = pc0[1] - dot3(R + 3, pw0); t[2] = pc0[2] - dot3(R + 6, pw0);
  if (slot < 3 && col < 5) {
    const double* pw = S.pws + 3 * col;
    const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
    const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
    const double u = S.us[2 * col], v = S.us[2 * col + 1];
    S.rep[cand][col] = xsqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  EO_SYNC();
  if (slot < 3 && col == 0) {
    double sum2 = 0.0;
    for (int i = 0; i < 5; ++i) sum2 += S.rep[cand][i];
    double* o = S.out[cand];
    for (int k = 0; k < 9; ++k) o[k] = R[k];
    o[9] = t[0]; o[10] = t[1]; o[11] = t[2];
    o[12] = sum2 / 5;
  }
  EO_SYNC();
  if (lane == 0) S.stamp[6] = clock64();
  bool fin = true;
  if (S.flag) {                 // a branch not reproduced here: the sequential restatement decides (lane 0)
    if (lane == 0) {
      double* x5 = xw.PW0;      // staging only: solve5 copies the sample first
      double* u5 = xw.gA;
      for (int i = 0; i < 15; ++i) x5[i] = S.pws[i];
      for (int i = 0; i < 10; ++i) u5[i] = S.us[i];
      double Rx[9], tx[3], rx[3];
      const bool ok = epnp_exact::solve5(xw, x5, u5, K, Rx, tx, rx);
      for (int k = 0; k < 9; ++k) S.out[0][k] = Rx[k];
      for (int k = 0; k < 3; ++k) S.out[0][9 + k] = tx[k];
      S.out[0][12] = rx[0]; S.out[1][12] = rx[1]; S.out[2][12] = rx[2];
      S.out[0][13] = ok ? 1.0 : 0.0;
    }
    EO_SYNC();
    for (int k = 0; k < 9; ++k) R_out[k] = S.out[0][k];
    for (int k = 0; k < 3; ++k) t_out[k] = S.out[0][9 + k];
    fin = S.out[0][13] != 0.0;
  } else {
    // N = 1; if (rep_errors[2] < rep_errors[1]) N = 2; if (rep_errors[3] < rep_errors[N]) N = 3;
    int N = 0;
    if (S.out[1][12] < S.out[0][12]) N = 1;
    if (S.out[2][12] < S.out[N][12]) N = 2;
    for (int k = 0; k < 9; ++k) { R_out[k] = S.out[N][k]; fin = fin && isfinite(S.out[N][k]); }
    for (int k = 0; k < 3; ++k) { t_out[k] = S.out[N][9 + k]; fin = fin && isfinite(S.out[N][9 + k]); }
  }
  if (rep) { rep[0] = S.out[0][12]; rep[1] = S.out[1][12]; rep[2] = S.out[2][12]; }
  if (lane == 0) S.stamp[7] = clock64();
  EO_SYNC();
  return fin;
}

}  // namespace epnp_ord
