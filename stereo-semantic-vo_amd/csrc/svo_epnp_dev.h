// svo_epnp_dev.h - EPnP on a five-point minimal set, ONE WAVE per hypothesis (gfx950, float64).
//
// What cv::solvePnPRansac runs for every RANSAC sample (reference src/pnpmatch.cc:227 -> OpenCV 3.2
// modules/calib3d/src/solvepnp.cpp: model_points = 5, SOLVEPNP_EPNP; modules/calib3d/src/epnp.cpp = Lepetit,
// Moreno-Noguer, Fua, "EPnP: an accurate O(n) solution to the PnP problem", IJCV 2009): control points from the PCA
// of the object points, barycentric coordinates, the 12x12 Gram matrix M^T M, its four smallest eigenvectors, three
// closed-form initialisations of the betas (N = 1, 2, 3), five Gauss-Newton steps each on the six control-point
// distance constraints, R / t by absolute orientation, the candidate with the smallest reprojection error wins.
// No initial pose enters anywhere.
//
// Mapping to a wavefront:
//   * M^T M: 144 entries over 64 lanes; its eigen-decomposition is a PARALLEL-ORDER two-sided Jacobi - in each of the
//     11 steps of a sweep six disjoint index pairs are rotated at once (round-robin tournament), six lanes computing
//     the rotations, all lanes applying J^T A J and V J to their entries from LDS.  OpenCV's JacobiSVDImpl_ visits the
//     66 pairs one after the other; both converge to the same eigenvectors (up to sign and up to the basis of the
//     two-dimensional null space a five-point M has - EPnP's result does not depend on either).
//   * the three beta branches run side by side on lanes 0, 1, 2 (scalar float64 code per lane);
//   * the small linear least-squares problems (cvSolve(..., CV_SVD) on 6x3 / 6x4 / 6x5, epnp::qr_solve on 6x4) are
//     solved through their normal equations with a pivoted Gauss-Jordan - same solutions for the full-rank systems
//     EPnP produces, far fewer dependent operations than a Jacobi SVD.
// Multiply-adds are contracted to FMAs in this file (the rest of the library is built with -ffp-contract=off): half the
// dependent instructions of the scalar stages, and nothing downstream depends on the rounding of a RANSAC sample's pose.
// A CPU restatement that follows OpenCV's own loops is kept with the tests; the two are compared to a stated tolerance.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct EpnpWaveLds {
  alignas(16) double A[12 * 12];      // M^T M, diagonalised in place
  alignas(16) double V[12 * 12];      // accumulated rotations: columns = eigenvectors
  alignas(16) double cs[8][2];  // per pair of the running step: cos, sin of its rotation; [6] = (1, 0): the identity, for the lanes that update V
  double v4[4][12];       // eigenvectors of the four smallest eigenvalues, v4[0] = smallest (OpenCV's ut + 12 * 11)
  double L[6 * 10], rho[6];
  double alphas[5 * 4];
  double cws[4][3];
  double x5[15], u5[10];  // the sample's object / image points
  double out[3][16];      // per branch: R (9), t (3), reprojection error
  int order[12];
  long long stamp[8];     // s_memtime after each stage (diagnostics)
  int sweeps;
};

#define EPNP_WAVE_SYNC()                                  \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); \
    __builtin_amdgcn_wave_barrier();                      \
  } while (0)

// Inside the eigen-solver's loop the wave is alone with its LDS workspace: LDS instructions of one wave execute in issue
// order, so a write is visible to the reads issued after it without waiting for its completion - only the compiler must
// not reorder them.  (The full fence costs the write's latency twice per Jacobi step.)
#define EPNP_WAVE_ORDER() __builtin_amdgcn_wave_barrier()

// float64 reciprocal square root / reciprocal: the hardware's own f64 approximation (v_rsq_f64 / v_rcp_f64) plus one
// Newton step - a handful of dependent instructions instead of the IEEE sqrt / division sequences.  The Jacobi
// rotations only need c^2 + s^2 = 1 to rounding.
__device__ __forceinline__ double epnp_rsqrt(double x) {
#pragma clang fp contract(fast)
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - (0.5 * x) * y * y);
  return y;
}
__device__ __forceinline__ double epnp_rcp(double x) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  return r;
}
// the same with a second Newton step: full double precision (the pivots of the small normal-equation solves), still a
// third of the dependent instructions of an IEEE division
__device__ __forceinline__ double epnp_rcp2(double x) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}
// square root as x * rsqrt(x) with two Newton steps (the scalar stages' IEEE sqrt / division sequences are ~100 dependent
// cycles each; a sample's pose is compared statistically, an ulp does not matter)
__device__ __forceinline__ double epnp_sqrt(double x) {
#pragma clang fp contract(fast)
  if (!(x > 0.0)) return 0.0;
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - (0.5 * x) * y * y);
  y = y * (1.5 - (0.5 * x) * y * y);
  return x * y;
}
// where position x moves after a step of the systolic tournament: 0 stays; 2 -> 4 -> 6 -> 8 -> 10 -> 11 -> 9 -> 7 -> 5 -> 3 -> 1 -> 2
__device__ __forceinline__ int epnp_pi(int x) {
  return x == 0 ? 0 : (x == 1 ? 2 : (x == 10 ? 11 : ((x & 1) ? x - 2 : x + 2)));
}
// partner of index x in step r of the tournament below
__device__ __forceinline__ int epnp_partner(int x, int r) {
  int y = 2 * r - x + 11;            // 1 .. 31
  y = y >= 22 ? y - 22 : (y >= 11 ? y - 11 : y);
  return x == 11 ? r : (x == r ? 11 : y);
}

// round-robin tournament on 12 indices (circle method, index 11 fixed): step r in 0..10, pair g in 0..5; over the 11
// steps every pair of indices meets exactly once, and the six pairs of a step are disjoint
__device__ __forceinline__ void epnp_pair(int r, int g, int& p, int& q) {
  const int a = g == 0 ? 11 : (r + g) % 11, b = g == 0 ? r : (r - g + 11) % 11;
  p = a < b ? a : b; q = a < b ? b : a;
}

// cyclic one-sided Jacobi SVD of a 3x3 (rows of At = columns of A), as OpenCV's SVD::compute on a 3x3: w descending,
// Ut rows = left, Vt rows = right singular vectors.  Scalar code (one lane).
__device__ inline void epnp_svd3(const double* A, double* w, double* Ut, double* Vt) {
#pragma clang fp contract(fast)
  double At[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, W[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) At[3 * i + k] = A[3 * k + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) W[i] = At[3 * i] * At[3 * i] + At[3 * i + 1] * At[3 * i + 1] + At[3 * i + 2] * At[3 * i + 2];
  const double eps = 2.220446049250313e-15;
  for (int iter = 0; iter < 30; ++iter) {
    bool changed = false;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
      const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2;
      double* Ai = At + 3 * i; double* Aj = At + 3 * j;
      double a = W[i], b = W[j];
      double p = Ai[0] * Aj[0] + Ai[1] * Aj[1] + Ai[2] * Aj[2];
      if (p * p <= eps * eps * a * b) continue;
      p *= 2;
      const double beta = a - b, g2 = p * p + beta * beta, ig = epnp_rsqrt(g2), gamma = g2 * ig;
      double c, s;
      if (beta < 0) { const double delta = (gamma - beta) * 0.5, dg = delta * ig; s = dg * epnp_rsqrt(dg); c = p * ig * 0.5 * epnp_rcp(s); }
      else { const double cg = (gamma + beta) * ig * 0.5; c = cg * epnp_rsqrt(cg); s = p * ig * 0.5 * epnp_rcp(c); }
      a = b = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
        Ai[k] = t0; Aj[k] = t1; a += t0 * t0; b += t1 * t1;
        const double v0 = c * V[3 * i + k] + s * V[3 * j + k], v1 = -s * V[3 * i + k] + c * V[3 * j + k];
        V[3 * i + k] = v0; V[3 * j + k] = v1;
      }
      W[i] = a; W[j] = b;
      changed = true;
    }
    if (!changed) break;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) W[i] = epnp_sqrt(At[3 * i] * At[3 * i] + At[3 * i + 1] * At[3 * i + 1] + At[3 * i + 2] * At[3 * i + 2]);
  // selection sort, descending
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int j = i;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k > i && W[j] < W[k]) j = k;
    if (j != i) {
      double t = W[i]; W[i] = W[j]; W[j] = t;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        t = At[3 * i + k]; At[3 * i + k] = At[3 * j + k]; At[3 * j + k] = t;
        t = V[3 * i + k]; V[3 * i + k] = V[3 * j + k]; V[3 * j + k] = t;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    w[i] = W[i];
    const double s = W[i] > 2.2250738585072014e-308 ? epnp_rcp2(W[i]) : 0.;
#pragma unroll
    for (int k = 0; k < 3; ++k) { Ut[3 * i + k] = At[3 * i + k] * s; Vt[3 * i + k] = V[3 * i + k]; }
  }
}

// least squares min |A x - b| for a 6 x NC system through the normal equations, solved by an unpivoted LDL^T
// (the Gram matrix of a full-rank A is positive definite; a rank-deficient candidate produces non-finite betas and
// loses the comparison of reprojection errors)
// `active` < NC: only the first `active` columns are unknowns (the others must be zero columns): their diagonal is set to
// 1, which leaves the leading block's factorisation - every operation of it - as it is and makes the padding solve to 0.
template <int NC>
__device__ inline void epnp_lsq6(const double* A /*6 x NC row-major*/, const double* b, double* x, int active = NC) {
#pragma clang fp contract(fast)
  double N[NC][NC], y[NC], Lm[NC][NC], D[NC], Dinv[NC];
#pragma unroll
  for (int r = 0; r < NC; ++r) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < r) continue;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += A[NC * k + r] * A[NC * k + c];
      N[r][c] = s;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += A[NC * k + r] * b[k];
    y[r] = s;
    if (r >= active) N[r][r] = 1.0;
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    double d = N[j][j];
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < j) d -= Lm[j][k] * Lm[j][k] * D[k];
    D[j] = d;
    const double inv = epnp_rcp2(d);
    Dinv[j] = inv;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if (i <= j) continue;
      double sacc = N[j][i];
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k < j) sacc -= Lm[i][k] * Lm[j][k] * D[k];
      Lm[i][j] = sacc * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < i) y[i] -= Lm[i][k] * y[k];
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) y[i] *= Dinv[i];
#pragma unroll
  for (int i = NC - 1; i >= 0; --i) {
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k > i) y[i] -= Lm[k][i] * y[k];
  }
#pragma unroll
  for (int r = 0; r < NC; ++r) x[r] = y[r];
}

__device__ __forceinline__ double epnp_dot3(const double* a, const double* b) { 
#pragma clang fp contract(fast)
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Eigen-decomposition of the symmetric 12 x 12 matrix in S.A (V must hold the identity), by the whole wave: on return the
// diagonal of S.A holds the eigenvalues (in no particular order) and column j of S.V the eigenvector of S.A[13 j].
// Also used by the 8-point fundamental matrix (svo_track.hip: its 9 x 9 normal matrix padded to 12 x 12).
__device__ inline void epnp_eig12_wave(EpnpWaveLds& S) {
#pragma clang fp contract(fast)
  const int lane = threadIdx.x & 63;
  // ---- eigen-decomposition: parallel-order two-sided Jacobi in its SYSTOLIC form (Brent & Luk) -----------------------
  // The six pairs of a step always sit at the adjacent POSITIONS (0,1) (2,3) ... (10,11); after the rotations the rows
  // and columns of A and the columns of V are moved by the fixed permutation `epnp_pi` (the circle method's rotation of
  // the players around position 0), which brings the next step's pairs side by side.  Every LDS address a lane touches
  // is therefore the same in every step - no index arithmetic inside the loop.
  double trace = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) trace += S.A[13 * i];
  const double tau = 1e-15 * trace;
  // work split of a step: A' = J^T A J and V' = V J are the SAME instructions on different 2 x 2 blocks.  Lanes 0..20 own the
  // upper-triangular blocks (gi <= gj) of the symmetric A - B = blk J_j, blk' = J_i^T B - and store every entry at its new
  // position and at the mirrored one; lanes 21..56 own the blocks (rows 2 ri, 2 ri + 1; pair gj) of V with J_i = the
  // identity (cs[6]).  One instruction stream instead of one for A and one for V; every LDS address is fixed per lane.
  typedef double epnp_d2 __attribute__((ext_vector_type(2)));
  double* const AV = S.A;                      // A at [0, 144), V behind it at [144, 288)
  static_assert(offsetof(EpnpWaveLds, V) == offsetof(EpnpWaveLds, A) + 144 * sizeof(double), "V must follow A");
  int bi = 0, gj = 0, ci_idx = 6, mbase = 144;
  bool isA = false;
  {
    const int t = min(lane, 56);
    if (t < 21) {
      int k = t, r = 0;
      while (k >= 6 - r) { k -= 6 - r; ++r; }   // row r of the upper triangle has 6 - r blocks
      bi = r; gj = r + k; ci_idx = r; mbase = 0; isA = true;
    } else {
      bi = (t - 21) / 6; gj = (t - 21) % 6;
    }
  }
  const int rd0 = mbase + 12 * (2 * bi) + 2 * gj, rd1 = rd0 + 12;
  const bool diag = isA && bi == gj;
  int wr[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int c = epnp_pi(2 * gj + b);
      if (isA) {
        const int r = epnp_pi(2 * bi + a);
        wr[a][b][0] = 12 * r + c; wr[a][b][1] = 12 * c + r;
      } else {
        wr[a][b][0] = 144 + 12 * (2 * bi + a) + c; wr[a][b][1] = wr[a][b][0];
      }
    }
  if (lane == 0) { S.cs[6][0] = 1.0; S.cs[6][1] = 0.0; }
  for (int sweep = 0; sweep < 14; ++sweep) {
    double maxoff = 0;
    for (int r = 0; r < 11; ++r) {
      if (lane < 6) {
        const int p = 2 * lane, q = p + 1;
        const epnp_d2 row = *reinterpret_cast<const epnp_d2*>(&S.A[12 * p + p]);   // app, apq
        const double apq = row.y, app = row.x, aqq = S.A[13 * q];
        double c = 1.0, s = 0.0;
        maxoff = fmax(maxoff, fabs(apq));
        if (fabs(apq) > tau) {
          // classical Jacobi: t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), c = 1 / sqrt(t^2 + 1),
          // s = t c.  With d = aqq - app, h = sqrt(d^2 + 4 apq^2), w = |d| + h:  t = sgn(d) 2 apq / w,  w^2 + 4 apq^2 = 2 h w,
          // so  c = w / sqrt(2 h w),  s = sgn(d) 2 apq / sqrt(2 h w)  - two reciprocal square roots, no division.
          const double d = aqq - app, two = apq + apq;
          const double x = d * d + two * two;
          const double h = x * epnp_rsqrt(x);
          const double w = fabs(d) + h;
          const double q2 = epnp_rsqrt((h + h) * w);
          c = w * q2;
          s = (d >= 0 ? two : -two) * q2;
        }
        // column p' = c col_p - s col_q ; column q' = s col_p + c col_q
        epnp_d2 o; o.x = c; o.y = s;
        *reinterpret_cast<epnp_d2*>(S.cs[lane]) = o;
      }
      EPNP_WAVE_ORDER();
      // every operand address is fixed: all loads of the step in one round trip
      const epnp_d2 ci = *reinterpret_cast<const epnp_d2*>(S.cs[ci_idx]), cj = *reinterpret_cast<const epnp_d2*>(S.cs[gj]);
      const epnp_d2 a0 = *reinterpret_cast<const epnp_d2*>(&AV[rd0]);
      const epnp_d2 a1 = *reinterpret_cast<const epnp_d2*>(&AV[rd1]);
      // B = blk J_j, blk' = J_i^T B
      const double b00 = a0.x * cj.x - a0.y * cj.y, b01 = a0.x * cj.y + a0.y * cj.x;
      const double b10 = a1.x * cj.x - a1.y * cj.y, b11 = a1.x * cj.y + a1.y * cj.x;
      double n00 = ci.x * b00 - ci.y * b10, n01 = ci.x * b01 - ci.y * b11;
      double n10 = ci.y * b00 + ci.x * b10, n11 = ci.y * b01 + ci.x * b11;
      if (diag) { n01 = 0.0; n10 = 0.0; }      // the rotated pair's own off-diagonal entry
      __builtin_amdgcn_wave_barrier();   // all reads above are issued before any write below (one wave, in-order LDS)
      AV[wr[0][0][0]] = n00; AV[wr[0][1][0]] = n01; AV[wr[1][0][0]] = n10; AV[wr[1][1][0]] = n11;
      AV[wr[0][0][1]] = n00; AV[wr[0][1][1]] = n01; AV[wr[1][0][1]] = n10; AV[wr[1][1][1]] = n11;
      EPNP_WAVE_ORDER();
    }
    if (lane == 0) S.sweeps = sweep + 1;
    // quadratic convergence: once a sweep met no off-diagonal entry above 1e-7 * trace, what it leaves behind is of the
    // order of 1e-14 * trace - below what the eigenvectors of the well separated small eigenvalues can resolve; no
    // confirming sweep needed
    const uint64_t big = __ballot(lane < 6 && maxoff > 1e-7 * trace);
    if (big == 0) break;
  }
}

// One hypothesis.  Called by all 64 lanes of a wave; the result (R row-major, t) is returned in every lane.  The sample
// is expected in S.x5 (5 x 3) / S.u5 (5 x 2) (doubles holding float values); K = {fu, fv, uc, vc}.
__device__ inline bool epnp5_wave(EpnpWaveLds& S, const double* K, double* R_out, double* t_out) {
#pragma clang fp contract(fast)
  const int lane = threadIdx.x & 63;
  // the sample lives in LDS, S.x5 / S.u5, filled by the caller (uniform reads broadcast): keeps ~50 registers free
  // across the eigen-solver
  EPNP_WAVE_SYNC();
  const double* Xw = S.x5;
  const double* uv = S.u5;
  const double fu = K[0], fv = K[1], uc = K[2], vc = K[3];
  if (lane == 0) S.stamp[0] = clock64();
  // ---- choose_control_points + compute_barycentric_coordinates (every lane, same scalar code) -------------------
  double cw0[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 5; ++i) { cw0[0] += Xw[3 * i]; cw0[1] += Xw[3 * i + 1]; cw0[2] += Xw[3 * i + 2]; }
  cw0[0] *= 0.2; cw0[1] *= 0.2; cw0[2] *= 0.2;
  double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double d[3] = {Xw[3 * i] - cw0[0], Xw[3 * i + 1] - cw0[1], Xw[3 * i + 2] - cw0[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) cov[3 * a + b] += d[a] * d[b];
  }
  double dc[3], uct[9], vt3[9];
  epnp_svd3(cov, dc, uct, vt3);
  double kk[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) kk[i] = epnp_sqrt(dc[i] * 0.2);
  // control points cws[0] = centroid, cws[i] = centroid + k_i u_i; CC = [k1 u1 | k2 u2 | k3 u3] has orthogonal columns: its
  // inverse (OpenCV: cvInvert through an SVD) is diag(1/k) U^T.  Both go to LDS: later stages index them dynamically.
  if (lane < 12) {
    const int i = lane / 3, j = lane % 3;
    const double ki = i == 1 ? kk[0] : (i == 2 ? kk[1] : kk[2]);
    const double uij = i == 0 ? 0.0 : (i == 1 ? (j == 0 ? uct[0] : (j == 1 ? uct[1] : uct[2]))
                                               : (i == 2 ? (j == 0 ? uct[3] : (j == 1 ? uct[4] : uct[5]))
                                                         : (j == 0 ? uct[6] : (j == 1 ? uct[7] : uct[8]))));
    const double c0 = j == 0 ? cw0[0] : (j == 1 ? cw0[1] : cw0[2]);
    S.cws[i][j] = i == 0 ? c0 : c0 + ki * uij;
  }
  if (lane < 5) {
    const double d[3] = {Xw[3 * lane] - cw0[0], Xw[3 * lane + 1] - cw0[1], Xw[3 * lane + 2] - cw0[2]};
    double a[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) a[1 + j] = kk[j] > 0 ? epnp_dot3(&uct[3 * j], d) * epnp_rcp2(kk[j]) : 0.0;
    a[0] = 1.0 - a[1] - a[2] - a[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) S.alphas[4 * lane + j] = a[j];
  }
  EPNP_WAVE_SYNC();
  const double* alphas = S.alphas;
  // ---- M^T M: entry (a, b) = sum over the 10 rows of M; M row pair of point i = alpha (x) (fu, 0, uc - u), (0, fv, vc - v)
  for (int e = lane; e < 144; e += 64) {
    const int a = e / 12, b = e % 12, ia = a / 3, ca = a % 3, ib = b / 3, cb = b % 3;
    double s = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const double du = uc - uv[2 * i], dv = vc - uv[2 * i + 1];
      const double r1a = ca == 0 ? fu : (ca == 1 ? 0.0 : du), r2a = ca == 0 ? 0.0 : (ca == 1 ? fv : dv);
      const double r1b = cb == 0 ? fu : (cb == 1 ? 0.0 : du), r2b = cb == 0 ? 0.0 : (cb == 1 ? fv : dv);
      s += alphas[4 * i + ia] * alphas[4 * i + ib] * (r1a * r1b + r2a * r2b);
    }
    S.A[e] = s;
    S.V[e] = a == b ? 1.0 : 0.0;
  }
  EPNP_WAVE_SYNC();
  if (lane == 0) S.stamp[1] = clock64();
  epnp_eig12_wave(S);
  if (lane == 0) S.stamp[2] = clock64();
  // ---- the four smallest eigenvalues' vectors: v4[0] = smallest (ut + 12 * 11 of OpenCV's descending order) --------
  if (lane < 12) {
    const double mine = S.A[13 * lane];
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const double o = S.A[13 * j];
      rank += (o < mine || (o == mine && j < lane)) ? 1 : 0;
    }
    S.order[lane] = rank;   // ascending rank of eigenvalue `lane`
  }
  EPNP_WAVE_SYNC();
  if (lane < 48) {
    const int col = lane % 12, which = lane / 12;   // which-th smallest eigenvalue, component `col` of its vector
    int src = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) src = S.order[j] == which ? j : src;
    S.v4[which][col] = S.V[12 * col + src];
  }
  EPNP_WAVE_SYNC();
  // ---- compute_L_6x10 (lanes 0..59) and compute_rho -------------------------------------------------------------
  if (lane < 60) {
    const int i = lane / 10, c = lane % 10;
    // pair i of the control points: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    const int pa = i < 3 ? 0 : (i < 5 ? 1 : 2), pb = i < 3 ? i + 1 : (i < 5 ? i - 1 : 3);
    // column c <-> (x, y) of betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]
    const int y = c >= 6 ? 3 : (c >= 3 ? 2 : (c >= 1 ? 1 : 0)), x = c - y * (y + 1) / 2;
    double dx[3], dy[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dx[k] = S.v4[x][3 * pa + k] - S.v4[x][3 * pb + k];
      dy[k] = S.v4[y][3 * pa + k] - S.v4[y][3 * pb + k];
    }
    S.L[lane] = (x == y ? 1.0 : 2.0) * epnp_dot3(dx, dy);
  }
  if (lane < 6) {
    const int pa = lane < 3 ? 0 : (lane < 5 ? 1 : 2), pb = lane < 3 ? lane + 1 : (lane < 5 ? lane - 1 : 3);
    const double d[3] = {S.cws[pa][0] - S.cws[pb][0], S.cws[pa][1] - S.cws[pb][1], S.cws[pa][2] - S.cws[pb][2]};
    S.rho[lane] = epnp_dot3(d, d);
  }
  EPNP_WAVE_SYNC();
  if (lane == 0) S.stamp[3] = clock64();
  // ---- the three beta branches, one lane each -------------------------------------------------------------------
  if (lane < 3) {
    const double* L = S.L;
    double rho[6], betas[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) rho[k] = S.rho[k];
    // The three initialisations solve 6 x 4, 6 x 3 and 6 x 5 systems over different columns of L: one padded 6 x 5
    // solve for all three lanes (same operations on the active block as the small solves), so that the lanes do not
    // run three different code paths one after the other.
    {
      const int nc = lane == 0 ? 4 : (lane == 1 ? 3 : 5);
      double l[30], b5[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int col = lane == 0 ? (k == 2 ? 3 : (k == 3 ? 6 : k)) : k;   // approx_1: [B11 B12 B13 B14] = columns 0 1 3 6
#pragma unroll
        for (int i = 0; i < 6; ++i) l[5 * i + k] = k < nc ? L[10 * i + col] : 0.0;
      }
      epnp_lsq6<5>(l, rho, b5, nc);
      if (lane == 0) {          // find_betas_approx_1
        const double sg = b5[0] < 0 ? -1.0 : 1.0;
        betas[0] = epnp_sqrt(sg * b5[0]);
        const double ib0 = sg * epnp_rcp2(betas[0]);
        betas[1] = b5[1] * ib0; betas[2] = b5[2] * ib0; betas[3] = b5[3] * ib0;
      } else {                  // find_betas_approx_2: [B11 B12 B22], approx_3: [B11 B12 B22 B13 B23]
        if (b5[0] < 0) { betas[0] = epnp_sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? epnp_sqrt(-b5[2]) : 0.0; }
        else { betas[0] = epnp_sqrt(b5[0]); betas[1] = (b5[2] > 0) ? epnp_sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) betas[0] = -betas[0];
        betas[2] = lane == 2 ? b5[3] * epnp_rcp2(betas[0]) : 0.0;
        betas[3] = 0.0;
      }
    }
    if (lane == 0) S.stamp[5] = clock64();
    // gauss_newton: five steps on the six distance constraints
    for (int it = 0; it < 5; ++it) {
      double A[24], b[6], x[4];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double* rl = L + 10 * i;
        A[4 * i] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
        A[4 * i + 1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
        A[4 * i + 2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
        A[4 * i + 3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
        b[i] = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] +
                         rl[3] * betas[0] * betas[2] + rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] +
                         rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] + rl[8] * betas[2] * betas[3] +
                         rl[9] * betas[3] * betas[3]);
      }
      epnp_lsq6<4>(A, b, x);
#pragma unroll
      for (int i = 0; i < 4; ++i) betas[i] += x[i];
    }
    if (lane == 0) S.stamp[6] = clock64();
    // compute_R_and_t: control points in the camera frame, the five points, sign, absolute orientation
    double ccs[4][3], pcs[15];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        ccs[j][k] = betas[0] * S.v4[0][3 * j + k] + betas[1] * S.v4[1][3 * j + k] + betas[2] * S.v4[2][3 * j + k] + betas[3] * S.v4[3][3 * j + k];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        pcs[3 * i + j] = S.alphas[4 * i] * ccs[0][j] + S.alphas[4 * i + 1] * ccs[1][j] + S.alphas[4 * i + 2] * ccs[2][j] + S.alphas[4 * i + 3] * ccs[3][j];
    if (pcs[2] < 0.0) {
#pragma unroll
      for (int i = 0; i < 15; ++i) pcs[i] = -pcs[i];
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) { pc0[j] += pcs[3 * i + j]; pw0[j] += Xw[3 * i + j]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) { pc0[j] *= 0.2; pw0[j] *= 0.2; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) abt[3 * j + k] += (pcs[3 * i + j] - pc0[j]) * (Xw[3 * i + k] - pw0[k]);
    double d3[3], Ut[9], Vt[9], R[9];
    if (lane == 0) S.stamp[7] = clock64();
    epnp_svd3(abt, d3, Ut, Vt);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = Ut[i] * Vt[j] + Ut[3 + i] * Vt[3 + j] + Ut[6 + i] * Vt[6 + j];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    double t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] = pc0[j] - epnp_dot3(&R[3 * j], pw0);
    double sum2 = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const double Xc = epnp_dot3(&R[0], &Xw[3 * i]) + t[0], Yc = epnp_dot3(&R[3], &Xw[3 * i]) + t[1],
                   inv_Zc = epnp_rcp2(epnp_dot3(&R[6], &Xw[3 * i]) + t[2]);
      const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
      const double du = uv[2 * i] - ue, dv = uv[2 * i + 1] - ve;
      sum2 += epnp_sqrt(du * du + dv * dv);
    }
    double* o = S.out[lane];
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = R[k];
    o[9] = t[0]; o[10] = t[1]; o[11] = t[2];
    o[12] = sum2 * 0.2;
  }
  EPNP_WAVE_SYNC();
  if (lane == 0) S.stamp[4] = clock64();
  // N = 1; if (rep[2] < rep[1]) N = 2; if (rep[3] < rep[N]) N = 3   (a NaN error never wins a `<`)
  int N = 0;
  if (S.out[1][12] < S.out[0][12]) N = 1;
  if (S.out[2][12] < S.out[N][12]) N = 2;
  bool fin = true;
#pragma unroll
  for (int k = 0; k < 12; ++k) fin = fin && isfinite(S.out[N][k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) R_out[k] = S.out[N][k];
  t_out[0] = S.out[N][9]; t_out[1] = S.out[N][10]; t_out[2] = S.out[N][11];
  EPNP_WAVE_SYNC();
  return fin;
}
