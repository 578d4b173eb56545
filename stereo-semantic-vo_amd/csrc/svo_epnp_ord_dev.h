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
//   * JacobiSVDImpl_ (cyclic one-sided Jacobi): a pair (i, j) reads and writes rows i and j only, so pairs on disjoint rows
//     commute exactly.  Lane 16 g + r of the wave holds columns g, 4 + g, 8 + g of matrix row r (r < 16 rows, 12 columns: A
//     followed by V); a pair is worked on by the lanes of its two rows, each fetching the partner row through LDS.  The row
//     update t0 = c Ai[k] + s Aj[k], t1 = -s Ai[k] + c Aj[k] is elementwise.  The three k-ordered sums of a pair
//     (p = sum Ai[k] Aj[k], a = sum t0^2, b = sum t1^2) are sequential chains over the columns - which is exactly what
//     v_mfma_f64_4x4x4_4b_f64 computes over the four 16-lane groups: D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C))))
//     per (row, column), rounded after every step (measured: tools/microbench/mfma_f64_4x4.hip, 128,000 of 128,000 entries
//     bit-identical).  With A = 1.0 the products are exact, so three chained MFMAs ARE the loop `s = 0; for k < 12: s += x[k]`
//     in IEEE arithmetic, for all 16 rows at once, the result landing in every lane of the row.  The rotation (c, s) is then
//     computed redundantly by all lanes of the pair from identical operands.  Up to floor(n / 2) pairs run per step from a
//     static list schedule of the cyclic order (tools/gen_jacobi_schedule.py: 12 steps per sweep instead of 66 for the
//     12 x 12 problem, sweeps overlapped, verified bit-identical to the sequential loop); problems on disjoint rows - the
//     three EPnP candidates' decompositions - share the steps.  V is not accumulated for M^T M: epnp only asks for U.
//   * the small decompositions (3 x 3: control points, barycentric inverse, absolute orientation; 6 x 3 / 6 x 4 / 6 x 5:
//     the three beta initialisations) go through the same engine, the three EPnP candidates side by side.
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
typedef double d2 __attribute__((ext_vector_type(2)));

#include "svo_epnp_ord_tab.h"
#include "svo_epnp_ord_asm.h"

// One wave, its own LDS workspace: LDS instructions of a wave execute in issue order, so a write is visible to the reads
// issued after it; the wavefront-scope fence only keeps the COMPILER from moving a read above the write it depends on.
#define EO_SYNC()                                         \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
  } while (0)

struct Lds {
  alignas(16) double jr[16 * 16];   // Jacobi rows [A (M columns) | V (n columns) | .. | W in column 15], row stride 16; 12 x 12: rows 0..11, small ones: rows 5 p .. 5 p + n - 1 of problem p
  alignas(16) double xch[16][4][4];   // row exchange of the Jacobi engine: [row][lane group] = its three columns (+ pad)
  unsigned tab[EO_TAB_TOTAL];         // the schedules (svo_epnp_ord_tab.h), bit 4 (second row of the pair) moved to bit 31
  double pws[16], us[10], alphas[20], cws[12], ccinv[9];
  double M[120];
  double ut4[4][12];        // ut + 12 * (11 - q): vectors of the four smallest singular values, q = 0 the smallest
  double L[60], rho[6];
  double sw[3][6];          // per small problem: singular values, descending
  double bx[3][6];          // per candidate: solution of the beta initialisation
  double ccs[3][12], pcs[3][16], abt[3][9], Rc[3][9], rep[3][8];
  alignas(16) double gn[3][6][6];    // gauss_newton: the rows of A (4) and b of a candidate, built one per lane
  double out[3][16];        // per candidate: R (9), t (3), mean reprojection error
  int srow[3][6];           // per small problem: Jacobi row (index into jr rows) at sorted position p
  int urow[4];              // 12 x 12: row of the (11 - q)-th singular value
  double xw[16];            // jacobi12_safe: W of every row (the assembly loop recomputes it instead)
  int flag;                 // bit 0: a branch this file does not reproduce was met in the 12 x 12 decomposition (or the step loop ran out of
                            // sweeps there): re-solve the sample sequentially; bit 1: a small decomposition was finished sequentially
  int sweeps;               // diagnostics: sweeps of the 12 x 12 decomposition
  int why;                  // diagnostics: 1 the 12 x 12 step loop ran out of sweeps, 2 a non-zero singular value of it out of range,
                            // 4 its finish by lane 0 (zero / equal singular values), 8 a small decomposition finished by lane 0, 16 forced
  long long stamp[8];
#ifdef EO_PROFILE
  long long prof[8];
#endif
};

EO_FN double xdiv(double a, double b) { return a / b; }
EO_FN double xsqrt(double x) { return sqrt(x); }

// The compiler's IEEE division and square root WITHOUT their range scaling (v_div_scale / v_ldexp) and special-case fix-up
// (v_div_fixup / v_cmp_class): the same reciprocal (square root) estimate, Newton steps and final correction, instruction
// for instruction - bit-identical whenever no scaling would have happened, i.e. for operands whose exponents stay a few
// hundred binades away from the ends of the range (no infinities, NaNs, denormals; the dividend may be zero).  Only used
// inside the Jacobi rotations, whose operands are bounded by the singular values: jacobi12 / svd_finish check that these
// lie in [2^-100, 2^100] and hand the sample to the sequential solver otherwise.
EO_FN double ndiv(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = a * r;
  const double rem = __builtin_fma(-b, q, a);
  return __builtin_fma(rem, r, q);
}
EO_FN double nsqrt(double x) {   // x > 0
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
EO_FN bool w_in_range(double w) { return w >= 7.888609052210118e-31 && w <= 1.2676506002282294e30; }   // [2^-100, 2^100]

// ---- in-order sums over the columns on the matrix core ------------------------------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 with A = 1.0: D(lane 16 i + c) = fma(1, B(48 + c), fma(1, B(32 + c), fma(1, B(16 + c), fma(1, B(c), C)))) for
// every group i - the IEEE sum ((((C + b0) + b1) + b2) + b3) of the four lane groups' values, in group order, in all four lanes.
EO_FN double gsum(double term, double acc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, term, acc, 0, 0, 0); }
// s = 0; for (c = 0; c < M; c++) s += t[c]; with column c = 4 q + g held as t_q by lane group g; columns >= M (the V part) are
// replaced by +0.0 (adding it leaves every partial sum as it is)
template <int M>
EO_FN double colsum(double t0, double t1, double t2, int g) {
  double acc = gsum(M >= 4 || g < M ? t0 : 0.0, 0.0);
  if (M > 4) acc = gsum(M >= 8 || 4 + g < M ? t1 : 0.0, acc);
  if (M > 8) acc = gsum(M >= 12 || 8 + g < M ? t2 : 0.0, acc);
  return acc;
}

#define EO_TAB3_OFF 0
#define EO_TAB4_OFF (EO_TAB3_OFF + EO_TAB3_STEPS * 3)
#define EO_TAB5_OFF (EO_TAB4_OFF + EO_TAB4_STEPS * 4)
#define EO_TAB12_OFF (EO_TAB5_OFF + EO_TAB5_STEPS * 5)
EO_FN unsigned tab_entry(unsigned e) { return (e & ~16u) | ((e & 16u) << 27); }
EO_FN void load_tables(Lds& S, int lane) {
  for (int e = lane; e < EO_TAB3_STEPS * 3; e += 64) S.tab[EO_TAB3_OFF + e] = tab_entry(c_tab3[0][e]);
  for (int e = lane; e < EO_TAB4_STEPS * 4; e += 64) S.tab[EO_TAB4_OFF + e] = tab_entry(c_tab4[0][e]);
  for (int e = lane; e < EO_TAB5_STEPS * 5; e += 64) S.tab[EO_TAB5_OFF + e] = tab_entry(c_tab5[0][e]);
  for (int e = lane; e < EO_TAB12_STEPS * 12; e += 64) S.tab[EO_TAB12_OFF + e] = tab_entry(c_tab12[0][e]);
}

// The pair test of JacobiSVDImpl_ as one hand-scheduled instruction stream: p = sum_k Ai[k] Aj[k] (three chained MFMAs) and
// eps * sqrt(W[i] W[j]) (the compiler's IEEE square root without its range scaling, see nsqrt) - the square root's eleven
// instructions sit in the shadows of the MFMAs (4 wait states between dependent DMFMAs, 6 before a VALU read of the result,
// 2 after the VALU write of an operand, 1 after v_rsq_f64: the distances the compiler itself keeps).
// mk: 1.0 / 0.0 per lane for the term of the partly filled column group (M = 6: group 1, M = 3: group 0).
template <int M>
EO_FN void pair_test(double x0, double x1, double x2, double q0, double q1, double q2, double W, double wP, double mk,
                     double& p, double& thr) {
  const double one = 1.0, eps = 2.220446049250313e-16 * 10;
  double t0, t1, t2, ab, y, g, h, r, d;
  if (M == 12) {
    asm volatile(
        "v_mul_f64 %[t0], %[x0], %[q0]\n\t"
        "v_mul_f64 %[ab], %[W], %[wP]\n\t"
        "v_mul_f64 %[t1], %[x1], %[q1]\n\t"
        "v_mul_f64 %[t2], %[x2], %[q2]\n\t"
        "v_rsq_f64 %[y], %[ab]\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[p], %[one], %[t0], 0\n\t"
        "v_mul_f64 %[g], %[ab], %[y]\n\t"
        "v_mul_f64 %[h], %[y], 0.5\n\t"
        "v_fma_f64 %[r], -%[h], %[g], 0.5\n\t"
        "v_fma_f64 %[g], %[g], %[r], %[g]\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[p], %[one], %[t1], %[p]\n\t"
        "v_fma_f64 %[h], %[h], %[r], %[h]\n\t"
        "v_fma_f64 %[d], -%[g], %[g], %[ab]\n\t"
        "v_fma_f64 %[g], %[d], %[h], %[g]\n\t"
        "v_fma_f64 %[d], -%[g], %[g], %[ab]\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[p], %[one], %[t2], %[p]\n\t"
        "v_fma_f64 %[g], %[d], %[h], %[g]\n\t"
        "v_mul_f64 %[thr], %[g], %[eps]\n\t"
        "s_nop 4"
        : [p] "=&v"(p), [thr] "=&v"(thr), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [ab] "=&v"(ab), [y] "=&v"(y),
          [g] "=&v"(g), [h] "=&v"(h), [r] "=&v"(r), [d] "=&v"(d)
        : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [W] "v"(W), [wP] "v"(wP),
          [one] "v"(one), [eps] "v"(eps));
  } else if (M == 6) {
    asm volatile(
        "v_mul_f64 %[t0], %[x0], %[q0]\n\t"
        "v_mul_f64 %[t1], %[x1], %[q1]\n\t"
        "v_mul_f64 %[ab], %[W], %[wP]\n\t"
        "v_mul_f64 %[t1], %[t1], %[mk]\n\t"
        "v_rsq_f64 %[y], %[ab]\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[p], %[one], %[t0], 0\n\t"
        "v_mul_f64 %[g], %[ab], %[y]\n\t"
        "v_mul_f64 %[h], %[y], 0.5\n\t"
        "v_fma_f64 %[r], -%[h], %[g], 0.5\n\t"
        "v_fma_f64 %[g], %[g], %[r], %[g]\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[p], %[one], %[t1], %[p]\n\t"
        "v_fma_f64 %[h], %[h], %[r], %[h]\n\t"
        "v_fma_f64 %[d], -%[g], %[g], %[ab]\n\t"
        "v_fma_f64 %[g], %[d], %[h], %[g]\n\t"
        "v_fma_f64 %[d], -%[g], %[g], %[ab]\n\t"
        "v_fma_f64 %[g], %[d], %[h], %[g]\n\t"
        "v_mul_f64 %[thr], %[g], %[eps]\n\t"
        "s_nop 0"
        : [p] "=&v"(p), [thr] "=&v"(thr), [t0] "=&v"(t0), [t1] "=&v"(t1), [ab] "=&v"(ab), [y] "=&v"(y), [g] "=&v"(g),
          [h] "=&v"(h), [r] "=&v"(r), [d] "=&v"(d)
        : [x0] "v"(x0), [x1] "v"(x1), [q0] "v"(q0), [q1] "v"(q1), [W] "v"(W), [wP] "v"(wP), [mk] "v"(mk), [one] "v"(one),
          [eps] "v"(eps));
  } else {
    asm volatile(
        "v_mul_f64 %[t0], %[x0], %[q0]\n\t"
        "v_mul_f64 %[ab], %[W], %[wP]\n\t"
        "v_mul_f64 %[t0], %[t0], %[mk]\n\t"
        "v_rsq_f64 %[y], %[ab]\n\t"
        "s_nop 0\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[p], %[one], %[t0], 0\n\t"
        "v_mul_f64 %[g], %[ab], %[y]\n\t"
        "v_mul_f64 %[h], %[y], 0.5\n\t"
        "v_fma_f64 %[r], -%[h], %[g], 0.5\n\t"
        "v_fma_f64 %[g], %[g], %[r], %[g]\n\t"
        "v_fma_f64 %[h], %[h], %[r], %[h]\n\t"
        "v_fma_f64 %[d], -%[g], %[g], %[ab]\n\t"
        "v_fma_f64 %[g], %[d], %[h], %[g]\n\t"
        "v_fma_f64 %[d], -%[g], %[g], %[ab]\n\t"
        "v_fma_f64 %[g], %[d], %[h], %[g]\n\t"
        "v_mul_f64 %[thr], %[g], %[eps]"
        : [p] "=&v"(p), [thr] "=&v"(thr), [t0] "=&v"(t0), [ab] "=&v"(ab), [y] "=&v"(y), [g] "=&v"(g), [h] "=&v"(h), [r] "=&v"(r),
          [d] "=&v"(d)
        : [x0] "v"(x0), [q0] "v"(q0), [W] "v"(W), [wP] "v"(wP), [mk] "v"(mk), [one] "v"(one), [eps] "v"(eps));
  }
}

// a = sum t0^2 over the A columns of the updated row (n0, n1, n2: its three columns): squares, then the chained MFMAs; the trailing
// wait states make the result readable by whatever the compiler schedules next
template <int M>
EO_FN double row_norm2(double n0, double n1, double n2, double mk) {
  const double one = 1.0;
  double s0, s1, s2, a;
  if (M == 12) {
    asm volatile(
        "v_mul_f64 %[s0], %[n0], %[n0]\n\t"
        "v_mul_f64 %[s1], %[n1], %[n1]\n\t"
        "v_mul_f64 %[s2], %[n2], %[n2]\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[a], %[one], %[s0], 0\n\t"
        "s_nop 3\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[a], %[one], %[s1], %[a]\n\t"
        "s_nop 3\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[a], %[one], %[s2], %[a]\n\t"
        "s_nop 5"
        : [a] "=&v"(a), [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2)
        : [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2), [one] "v"(one));
  } else if (M == 6) {
    asm volatile(
        "v_mul_f64 %[s0], %[n0], %[n0]\n\t"
        "v_mul_f64 %[s1], %[n1], %[n1]\n\t"
        "v_mul_f64 %[s1], %[s1], %[mk]\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[a], %[one], %[s0], 0\n\t"
        "s_nop 3\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[a], %[one], %[s1], %[a]\n\t"
        "s_nop 5"
        : [a] "=&v"(a), [s0] "=&v"(s0), [s1] "=&v"(s1)
        : [n0] "v"(n0), [n1] "v"(n1), [mk] "v"(mk), [one] "v"(one));
  } else {
    asm volatile(
        "v_mul_f64 %[s0], %[n0], %[n0]\n\t"
        "v_mul_f64 %[s0], %[s0], %[mk]\n\t"
        "s_nop 1\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %[a], %[one], %[s0], 0\n\t"
        "s_nop 5"
        : [a] "=&v"(a), [s0] "=&v"(s0)
        : [n0] "v"(n0), [mk] "v"(mk), [one] "v"(one));
  }
  return a;
}

// ---- JacobiSVDImpl_<double> on the rows of S.jr, all problems of the wave together ------------------------------------------
// Lane 16 g + r works on matrix row r (a row of S.jr: M columns of A, then V, 12 columns in all - the columns behind V must hold
// finite values, they are rotated along) and holds its columns g, 4 + g, 8 + g.  (n, base): the problem row r belongs to -
// n rows starting at row `base`, n in {3, 4, 5, 12} - or n = 0: the row idles.
// On return the rows are what JacobiSVDImpl_ leaves before its sort: rotated, the A part scaled by 1 / W[i], and
// W[i] = sqrt(sum At[i][k]^2) in column 15 of the row.
// Returns (uniform) the lanes whose step loop ran out of sweeps.
template <int M>
EO_FN unsigned long long jacobi_rows(Lds& S, int n, int base, int lane) {
  const int r = lane & 15, g = lane >> 4, q = r - base;
  // 1.0 where this lane's column of the partly filled group is an A column, 0.0 where it belongs to V: x * 1.0 is x, and a
  // (finite) x * 0.0 adds nothing to a sum that started at +0.0
  const double mk = M == 6 ? (g < 2 ? 1.0 : 0.0) : (M == 3 ? (g < 3 ? 1.0 : 0.0) : 1.0);
  const double eps = 2.220446049250313e-16 * 10;
  double x0 = S.jr[r * 16 + g], x1 = S.jr[r * 16 + 4 + g], x2 = S.jr[r * 16 + 8 + g];
  double* const mine = S.xch[r][g];
  { d2 lo; lo.x = x0; lo.y = x1; *reinterpret_cast<d2*>(mine) = lo; mine[2] = x2; }
  const int tb = n == 12 ? EO_TAB12_OFF : (n == 5 ? EO_TAB5_OFF : (n == 4 ? EO_TAB4_OFF : EO_TAB3_OFF));
  const int steps = n == 12 ? EO_TAB12_STEPS : (n == 5 ? EO_TAB5_STEPS : (n == 4 ? EO_TAB4_STEPS : EO_TAB3_STEPS));
  const int pro = n == 12 ? EO_TAB12_PROLOGUE : (n == 5 ? EO_TAB5_PROLOGUE : (n == 4 ? EO_TAB4_PROLOGUE : EO_TAB3_PROLOGUE));
  static_assert(EO_TAB3_SWEEPS == 1 && EO_TAB4_SWEEPS == 1 && EO_TAB5_SWEEPS == 1 && EO_TAB12_SWEEPS == 1, "one sweep per period");
  // LDS byte addresses (the low half of a generic pointer into the LDS aperture)
  auto lds = [](const void* p) { return (unsigned)(unsigned long long)p; };
  const unsigned a_tab = lds(&S.tab[0]);
  const unsigned twrap = a_tab + 4u * (unsigned)(n > 0 ? tb + pro * n + q : 0);
  const unsigned cfg = (unsigned)steps | (unsigned)pro << 8 | (unsigned)(4 * n) << 16;
  const unsigned pm = n > 0 ? ((1u << n) - 1u) << base : 0u;   // the rows of my problem, as lanes of group 0
  EO_SYNC();
  // The step loop (svo_epnp_ord_asm.h, generated by tools/gen_jacobi_asm.py) is software-pipelined over the schedule: the entry
  // of step t + 2 and the partner row of step t + 1 are requested from LDS while step t computes.  Per step and row:
  //   partner row of the pair (i, j) this row belongs to (none: the row idles), fetched from S.xch;
  //   p = sum_k Ai[k] Aj[k], W[i] = sum_k Ai[k]^2, W[j] = sum_k Aj[k]^2 - three interleaved chains of DMFMAs.  (JacobiSVDImpl_
  //   carries W along; it is the k-ordered sum of the squares of the row as last written, which is what is recomputed here -
  //   same operands, same order, same bits.)
  //   rotate iff |p| > eps sqrt(W[i] W[j]): decided on the squares with a 2^-40 margin, by the expression itself inside it;
  //   (c, s) by OpenCV's formulas, IEEE division / square root; row i: c Ai + s Aj, row j: -s Ai + c Aj, back to S.xch;
  //   when the pair (n - 2, n - 1) of a sweep has run: the problem stops if none of its rows rotated in that sweep.
  const double eps2 = eps * eps;                             // exact: eps = 1.25 * 2^-49
  const double eps2hi = eps2 * (1.0 + 9.094947017729282e-13), eps2lo = eps2 * (1.0 - 9.094947017729282e-13);   // 2^-40
  unsigned chg = 0, nst = 0, flag = 0;
  int tt = 0, tp = n > 0 ? tb + q : 0, sb = 0;
  const unsigned e0 = S.tab[tp];                             // step 0
  ++tt; tp += n;
  if (tt == steps) { tt = pro; tp = n > 0 ? tb + pro * n + q : 0; ++sb; }
  const unsigned e1 = S.tab[tp];                             // step 1
  const unsigned tpa = a_tab + 4u * (unsigned)tp;
  const unsigned act = n >= 2 ? 1u : 0u;
  const unsigned amine = lds(mine), axch = lds(&S.xch[0][g][0]);
#define EO_JACOBI_OPERANDS                                                                                                       \
  : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [chg] "+v"(chg), [nst] "+v"(nst), [flag] "+v"(flag)                              \
  : [e0] "v"(e0), [e1] "v"(e1), [tt] "v"(tt), [tp] "v"(tpa), [sb] "v"(sb), [cfg] "v"(cfg), [twrap] "v"(twrap), [lane] "v"(lane), \
    [base] "v"(base), [act] "v"(act), [pm] "v"(pm), [mk] "v"(mk), [eps] "v"(eps), [eps2hi] "v"(eps2hi), [eps2lo] "v"(eps2lo),   \
    [amine] "v"(amine), [axch] "v"(axch)                                                                                          \
  : EO_JACOBI_ASM_CLOBBERS
  if (M == 12) asm volatile(EO_JACOBI_ASM_12 EO_JACOBI_OPERANDS);
  else if (M == 6) asm volatile(EO_JACOBI_ASM_6 EO_JACOBI_OPERANDS);
  else asm volatile(EO_JACOBI_ASM_3 EO_JACOBI_OPERANDS);
#undef EO_JACOBI_OPERANDS
  // 25 sweeps without convergence: the sequential code decides - the whole sample for the 12 x 12 problem, this decomposition
  // alone for a small one (the caller, svd_small_seq)
  const unsigned long long exhausted = __ballot(flag != 0 && n > 0);
  if (M == 12 && flag) S.why |= 1;
  if (M == 12 && lane == 0) S.sweeps = (int)nst;             // diagnostics: steps of the 12 x 12 decomposition
  // W[i] = sqrt(sum At[i][k]^2); At[i] *= 1 / W[i]  (the sort is the caller's: svd_rank)
  const double w = xsqrt(row_norm2<M>(x0, x1, x2, mk));
  const double sc1 = 1 / w;
  if (n > 0) {
    S.jr[r * 16 + g] = M >= 4 || g < M ? x0 * sc1 : x0;
    S.jr[r * 16 + 4 + g] = M >= 8 || 4 + g < M ? x1 * sc1 : x1;
    S.jr[r * 16 + 8 + g] = M >= 12 || 8 + g < M ? x2 * sc1 : x2;
    if (g == 0) S.jr[r * 16 + 15] = w;
  }
  EO_SYNC();
  return exhausted;
}

// cv::SVD's descending order for the problem of this lane's DPP row (p < 3; n rows from `base`): S.sw[p][pos] the singular values,
// S.srow[p][pos] the Jacobi row at sorted position pos.  A singular value outside [2^-100, 2^100] (zero: JacobiSVDImpl_ would
// draw a random vector) or two equal ones (the selection sort's swaps would matter): the lane reports it (the returned ballot) -
// the caller reloads the problem's matrix and has lane 0 walk JacobiSVDImpl_ itself on it (svd_small_seq).
EO_FN unsigned long long svd_rank(Lds& S, int n, int base, int p, int lane) {
  const int col = lane & 15;
  bool bad = false;
  if (col < n) {
    const double my = S.jr[(base + col) * 16 + 15];
    int rank = 0;
    bad = !w_in_range(my);
    for (int q = 0; q < 5; ++q)
      if (q < n && q != col) {
        const double o = S.jr[(base + q) * 16 + 15];
        rank += o > my ? 1 : 0;
        bad = bad || o == my;
      }
    S.sw[p][rank] = my;
    S.srow[p][rank] = base + col;
  }
  EO_SYNC();
  return __ballot(bad);      // (uniform) lanes of the DPP rows whose problem needs the sequential finish: bits 16 p .. 16 p + 15
}
// which of the (up to three) side-by-side problems need the sequential finish: bit p.  `ex`: jacobi_rows' lanes (row r of any
// lane group belongs to problem r / 5), `bad`: svd_rank's (DPP row p)
EO_FN unsigned need_seq(unsigned long long ex, unsigned long long bad) {
  unsigned m = 0;
  for (int p = 0; p < 3; ++p) {
    const unsigned long long rows = 0x001F001F001F001FULL << (5 * p);
    if ((ex & rows) || (bad & (0xFFFFULL << (16 * p)))) m |= 1u << p;
  }
  return m;
}

// the 12 x 12 decomposition of M^T M: the rows of the four smallest singular values, S.ut4[q] = ut + 12 * (11 - q)
// Returns (uniform) 0: the four vectors are in S.ut4; 1: a singular value is exactly zero or two are equal - the caller finishes
// the decomposition sequentially (svd12_finish_seq) instead of using what is stored here; 2: a value outside [2^-100, 2^100]
// (NaN included) - the assembly loop's unscaled divisions were not IEEE's there: the caller redoes the sweeps (jacobi12_safe).
EO_FN int svd12_smallest(Lds& S, int lane) {
  bool local = false, full = false;
  if (lane < 12) {
    const double my = S.jr[lane * 16 + 15];
    int rank = 0;
    local = my == 0.0;
    full = !w_in_range(my);     // (0 included: rows of exact zeros next to rows of 1e-300 are the same degenerate case)
    for (int q = 0; q < 12; ++q)
      if (q != lane) {
        const double o = S.jr[q * 16 + 15];
        rank += o > my ? 1 : 0;
        local = local || o == my;
      }
    if (full) S.why |= 2;
    if (rank >= 8) S.urow[11 - rank] = lane;
  }
  const bool any_local = __ballot(local) != 0, any_full = __ballot(full) != 0;
  EO_SYNC();
  if (any_full) return 2;
  if (any_local) return 1;
  if (lane < 48) {
    const int q = lane / 12, k = lane % 12, r = S.urow[q] & 15;
    S.ut4[q][k] = S.jr[r * 16 + k];
  }
  EO_SYNC();
  return 0;
}

// The 12 x 12 step loop in plain C++ with the compiler's full IEEE division and square root (range scaling, special cases) and
// JacobiSVDImpl_'s expressions as they stand - the assembly loop's predecessor (round 4), ~1.5x its time.  The assembly loop's
// unscaled divisions are IEEE's only for operands well inside the range, which a DEGENERATE sample's M^T M is not (an exactly
// coplanar or duplicated set of world points: zero rows, rows of 1e-300): it then reports trouble (out of sweeps, a singular value
// out of range) and the decomposition is redone here from the reloaded matrix - still all pairs of a step side by side, instead of
// the 66 pairs of a sweep one after the other on one lane (~2 ms per sample).  Same schedule tables, same k-ordered sums.
__device__ __attribute__((noinline)) void jacobi12_safe(Lds& S, int lane) {
  const double eps = 2.220446049250313e-16 * 10;
  const int r = lane & 15, g = lane >> 4, n = r < 12 ? 12 : 0, q = r;
  double x0 = S.jr[r * 16 + g], x1 = S.jr[r * 16 + 4 + g], x2 = S.jr[r * 16 + 8 + g];
  double W = colsum<12>(x0 * x0, x1 * x1, x2 * x2, g);
  double* const mine = S.xch[r][g];
  mine[0] = x0; mine[1] = x1; mine[2] = x2;
  S.xw[r] = W;
  const int tb = EO_TAB12_OFF, steps = EO_TAB12_STEPS, pro = EO_TAB12_PROLOGUE;
  const unsigned pm = n > 0 ? (1u << 12) - 1u : 0u;   // the rows of the problem, as lanes of group 0
  EO_SYNC();
  int tt = 0, tp = n > 0 ? tb + q : 0, sbase = 0;
  unsigned chg = 0;      // bit s: this row rotated in sweep s
  bool active = n >= 2;
  unsigned e = S.tab[tp];
  int nsteps = 0;
  while (__any(active)) {
    int tn = tt + 1, tpn = tp + n, sbn = sbase;
    if (tn == steps) { tn = pro; tpn = tb + pro * n + q; sbn = sbase + 1; }
    const unsigned en = S.tab[n > 0 ? tpn : 0];           // next step's entry
    const int pq = e & 15, sw = sbase + (int)((e >> 5) & 3);
    const bool is_j = (e >> 31) != 0;                     // (bit 4 of the table entry, moved up by load_tables)
    const bool valid = active && pq != q && sw < 30;
    const int pr = valid ? pq : r;
    const double* const theirs = S.xch[pr][g];
    const double y0 = theirs[0], y1 = theirs[1], y2 = theirs[2];
    const double p0 = colsum<12>(x0 * y0, x1 * y1, x2 * y2, g);
    const double wP = S.xw[pr];
    const bool rot = valid && !(fabs(p0) <= eps * xsqrt(W * wP));
    if (__any(rot)) {
      const double p = p0 * 2;
      double beta = W - wP;                    // W[i] - W[j]: the j row sees the operands swapped
      if (is_j) beta = -beta;
      double gamma;                            // cv::hypot(p, beta)
      {
        double a = fabs(p), b = fabs(beta);
        if (a > b) { b = xdiv(b, a); gamma = a * xsqrt(1 + b * b); }
        else if (b > 0) { a = xdiv(a, b); gamma = b * xsqrt(1 + a * a); }
        else gamma = 0;
      }
      double c, sn;
      if (beta < 0) {
        const double delta = (gamma - beta) * 0.5;
        sn = xsqrt(xdiv(delta, gamma));
        c = xdiv(p, gamma * sn * 2);
      } else {
        c = xsqrt(xdiv(gamma + beta, gamma * 2));
        sn = xdiv(p, gamma * c * 2);
      }
      // row i: t0 = c Ai + s Aj; row j: t1 = -s Ai + c Aj = c (mine) + (-s) (theirs)
      const double se = is_j ? -sn : sn;
      const double n0 = c * x0 + se * y0, n1 = c * x1 + se * y1, n2 = c * x2 + se * y2;
      const double Wn = colsum<12>(n0 * n0, n1 * n1, n2 * n2, g);
      EO_SYNC();                               // (every lane has fetched its partner's row before any row is rewritten)
      if (rot) {
        x0 = n0; x1 = n1; x2 = n2; W = Wn;
        chg |= 1u << sw;
        mine[0] = x0; mine[1] = x1; mine[2] = x2;
        S.xw[r] = W;
      }
    }
    EO_SYNC();
    const bool closes = active && (e & 0x80u) != 0;        // the pair (n - 2, n - 1) of sweep sc ran in this step: the sweep is complete
    if (__any(closes)) {
      const int sc = sbase + (int)((e >> 8) & 3);
      const unsigned long long bal = __ballot(closes && ((chg >> (sc & 31)) & 1u));
      if (closes && (sc >= 29 || ((unsigned)bal & pm) == 0u)) active = false;   // for (iter < 30) { ...; if (!changed) break; }
    }
    tt = tn; tp = tpn; sbase = sbn; e = en;
    ++nsteps;
  }
  if (lane == 0) S.sweeps = 1000000 + nsteps;      // diagnostics (1e6 + steps: the safe loop ran)
  EO_SYNC();
}

// A small decomposition the wave engine cannot finish the way OpenCV does (a zero singular value: JacobiSVDImpl_ draws a random
// vector from its own RNG and orthogonalises it; equal singular values: its selection sort's swaps decide the order; a step loop
// that ran out of sweeps): lane 0 walks JacobiSVDImpl_ itself (svo_epnp_exact_dev.h, the restatement the sequential solver uses)
// over the problem's rows IN PLACE - the rows are At (n rows of M columns, pitch 16) with Vt behind them (columns M ...), which is
// the layout that function works on - and leaves what svd_rank leaves: S.sw[p] descending, S.srow[p][pos] = base + pos (the rows
// come out sorted).  The caller has reloaded the problem's ORIGINAL matrix into the rows.  Costs microseconds for a 3 x 3 or
// 6 x 5 problem, where re-solving the whole sample on one lane cost ~840 us (the 12 x 12 Jacobi on one lane): an exactly
// coplanar set of world points - a zero singular value in choose_control_points' 3 x 3 problem - no longer is a latency cliff.
// One copy of the code (not inlined): this path is rare.
// (`jW`: n doubles of workspace of the calling lane's own - the three candidates' problems are finished by three lanes side by side)
typedef __attribute__((address_space(3))) double lds_double;    // (the function is not inlined: without the qualifier its accesses are flat ones)
__device__ __attribute__((noinline)) void svd_small_seq(Lds& S, double* jW, int M, int n, int base, int p) {
  double w[6];
  epnp_exact::jacobi_svd((lds_double*)jW, (lds_double*)&S.jr[base * 16], 16, w, (lds_double*)&S.jr[base * 16 + M], 16, M, n, n);
  for (int i = 0; i < n; ++i) {
    S.sw[p][i] = w[i];
    S.srow[p][i] = base + i;
    S.jr[(base + i) * 16 + 15] = w[i];
  }
  S.flag |= 2; S.why |= 8;
}

// The 12 x 12 decomposition with a ZERO singular value or two EQUAL ones (an exactly coplanar / duplicated set of world points
// leaves structural zeros in M): the wave engine's sweeps are JacobiSVDImpl_'s sweeps row for row, only what follows them - the
// selection sort's order among equal values, the random vectors it draws for zero ones - is not reproduced there.  Lane 0 runs
// exactly that part (epnp_exact::jacobi_svd_finish) on the rows as the sweeps left them (unscaled, in the engine's exchange
// buffer) and hands the four smallest singular values' vectors on.  A few thousand instructions, where re-solving the whole
// sample on one lane costs ~2 ms - the 12 x 12 sweeps on one lane.
__device__ __attribute__((noinline)) void svd12_finish_seq(Lds& S, epnp_exact::Work& xw) {
  for (int i = 0; i < 12; ++i)
    for (int k = 0; k < 12; ++k) xw.ut[i * 12 + k] = S.xch[i][k & 3][k >> 2];     // lane group g = k & 3 holds columns g, 4 + g, 8 + g
  epnp_exact::jacobi_svd_finish((lds_double*)xw.jW, (lds_double*)xw.ut, 12, xw.d, (lds_double*)nullptr, 0, 12, 12, 12);
  for (int q = 0; q < 4; ++q)
    for (int k = 0; k < 12; ++k) S.ut4[q][k] = xw.ut[(11 - q) * 12 + k];
  S.flag |= 2; S.why |= 4;
}

// ---- epnp::gauss_newton + qr_solve for one candidate, scalar code (arrays in registers) ---------------------------------
// `cand`: this lane's candidate (its DPP row), `col`: lane within the row.  The six rows of A and b are built by six lanes of the
// row (compute_A_and_b_gauss_newton is elementwise in the row index), handed round through S.gn, and every lane then walks
// qr_solve on its own copy.
EO_FN void gauss_newton(Lds& S, int cand, int col, bool owner, const double* L, const double* rho, double* betas) {
  for (int it = 0; it < 5; ++it) {
    double A[6][4], b[6];
    {
      const int i = col < 6 ? col : 0;
      const double* rl = L + 10 * i;
      double* g = S.gn[cand][i];
      const double a0 = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
      const double a1 = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
      const double a2 = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
      const double a3 = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
      const double bi = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] +
                                  rl[3] * betas[0] * betas[2] + rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] +
                                  rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] + rl[8] * betas[2] * betas[3] +
                                  rl[9] * betas[3] * betas[3]);
      if (owner && col < 6) { g[0] = a0; g[1] = a1; g[2] = a2; g[3] = a3; g[4] = bi; }   // (the idle fourth DPP row shadows candidate 1: it must not write)
    }
    EO_SYNC();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double* g = S.gn[cand][i];
      A[i][0] = g[0]; A[i][1] = g[1]; A[i][2] = g[2]; A[i][3] = g[3]; b[i] = g[4];
    }
    EO_SYNC();
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
  const int lane = threadIdx.x & 63, slot = lane >> 4, col = lane & 15, r16 = lane & 15;
  const double fu = K[0], fv = K[1], uc = K[2], vc = K[3];
  if (lane == 0) { S.flag = force_seq ? 1 : 0; S.why = force_seq ? 16 : 0; S.stamp[0] = clock64(); }
  load_tables(S, lane);   // force_seq: tests exercise the sequential fallback
  EO_SYNC();
  // ---- choose_control_points -----------------------------------------------------------------------------------------
  if (lane < 3) {
    double c = 0;
    for (int i = 0; i < 5; i++) c += S.pws[3 * i + lane];
    S.cws[lane] = c / 5;
  }
  EO_SYNC();
  auto load_pw0 = [&]() {
    if (lane < 9) {            // PW0^T PW0 (cvMulTransposed), loaded transposed into the Jacobi rows (temp_a = A^T)
      const int a = lane / 3, b = lane % 3;
      const double ca = S.cws[a], cb = S.cws[b];
      double s = 0;
      for (int i = 0; i < 5; ++i) s += (S.pws[3 * i + a] - ca) * (S.pws[3 * i + b] - cb);
      S.jr[b * 16 + a] = s;
    } else if (lane < 18) {
      const int r = (lane - 9) / 3, k = (lane - 9) % 3;
      S.jr[r * 16 + 3 + k] = r == k ? 1.0 : 0.0;
    } else if (lane < 36) {
      S.jr[((lane - 18) / 6) * 16 + 6 + (lane - 18) % 6] = 0.0;   // the columns behind V: finite
    }
    EO_SYNC();
  };
  load_pw0();
  unsigned long long ex = jacobi_rows<3>(S, r16 < 3 ? 3 : 0, r16 < 3 ? 0 : r16, lane);
  unsigned long long bd = svd_rank(S, slot == 0 ? 3 : 0, 0, 0, lane);
  if (ex | bd) {             // (uniform) e.g. exactly coplanar world points: a zero singular value
    load_pw0();
    if (lane == 0) svd_small_seq(S, xw.jW, 3, 3, 0, 0);
    EO_SYNC();
  }
  if (lane < 9) {
    const int i = 1 + lane / 3, j = lane % 3;
    const double k = xsqrt(S.sw[0][i - 1] / 5);
    S.cws[3 * i + j] = S.cws[j] + k * S.jr[S.srow[0][i - 1] * 16 + j];
  }
  EO_SYNC();
  // ---- compute_barycentric_coordinates: cvInvert(CC, CV_SVD) ---------------------------------------------------------
  auto load_cc = [&]() {
    if (lane < 9) {            // cc[3 i + j - 1] = cws[j][i] - cws[0][i]; Jacobi row r, column k = cc[3 k + r]
      const int r = lane / 3, k = lane % 3;
      S.jr[r * 16 + k] = S.cws[3 * (r + 1) + k] - S.cws[k];
    } else if (lane < 18) {
      const int r = (lane - 9) / 3, k = (lane - 9) % 3;
      S.jr[r * 16 + 3 + k] = r == k ? 1.0 : 0.0;
    } else if (lane < 36) {
      S.jr[((lane - 18) / 6) * 16 + 6 + (lane - 18) % 6] = 0.0;   // the columns behind V: finite
    }
    EO_SYNC();
  };
  load_cc();
  ex = jacobi_rows<3>(S, r16 < 3 ? 3 : 0, r16 < 3 ? 0 : r16, lane);
  bd = svd_rank(S, slot == 0 ? 3 : 0, 0, 0, lane);
  if (ex | bd) {
    load_cc();
    if (lane == 0) svd_small_seq(S, xw.jW, 3, 3, 0, 0);
    EO_SYNC();
  }
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
  auto load_mtm = [&]() {
    for (int e = lane; e < 144; e += 64) {
      const int a = e / 12, b = e % 12;
      double s = 0;
      for (int r = 0; r < 10; ++r) s += S.M[12 * r + a] * S.M[12 * r + b];
      S.jr[b * 16 + a] = s;
    }
    EO_SYNC();
  };
  load_mtm();
  if (lane == 0) S.stamp[2] = clock64();
  // (a sample whose control points already needed the sequential finish - exactly coplanar world points - is degenerate here too:
  // straight to the full-IEEE loop instead of running the assembly loop into its sweep bound first)
  const bool degenerate = (S.why & 8) != 0;      // (uniform: LDS, written before the last EO_SYNC)
  unsigned long long ex12 = 0;
  int fin12 = 2;
  if (!degenerate) {
    ex12 = jacobi_rows<12>(S, r16 < 12 ? 12 : 0, r16 < 12 ? 0 : r16, lane);
    fin12 = svd12_smallest(S, lane);             // (uniform) 0: done; 1: zero / equal singular values; 2: one out of range
  }
  if (ex12 || fin12) {
    if (ex12 || fin12 == 2) {       // the assembly loop's arithmetic left its range: the sweeps in full IEEE, from the reloaded matrix
      if (!degenerate) load_mtm();
      jacobi12_safe(S, lane);
    }
    if (lane == 0) svd12_finish_seq(S, xw);      // OpenCV's sort and its random vectors for zero singular values, by lane 0
    EO_SYNC();
  }
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
  auto load_l = [&](int only) {      // the candidates' systems into their rows (only >= 0: that candidate's alone)
    if (slot < 3 && (only < 0 || slot == only)) {
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
  };
  load_l(-1);
  {   // rows 0..3: candidate 1 (6 x 4), rows 5..7: candidate 2 (6 x 3), rows 10..14: candidate 3 (6 x 5)
    const int pc = r16 / 5, pn = pc == 0 ? 4 : (pc == 1 ? 3 : 5);
    const bool mine = r16 < 15 && r16 % 5 < pn;
    ex = jacobi_rows<6>(S, mine ? pn : 0, mine ? 5 * pc : r16, lane);
  }
  bd = svd_rank(S, nc, base, cand, lane);
  if (ex | bd) {             // (uniform)
    const unsigned need = need_seq(ex, bd);
    for (int pp = 0; pp < 3; ++pp)
      if (need >> pp & 1) load_l(pp);
    // one lane per candidate, side by side (each with a workspace of its own)
    if (slot < 3 && col == 0 && (need >> slot & 1)) svd_small_seq(S, slot == 0 ? xw.jW : (slot == 1 ? xw.sA : xw.sUt), 6, nc, base, cand);
    EO_SYNC();
  }
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
    gauss_newton(S, cand, col, slot < 3, S.L, rho, betas);
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
  bool flip;
  {
    const double* pc = S.pcs[cand];
    flip = pc[2] < 0.0;
    for (int j = 0; j < 3; ++j) { pc0[j] = 0; pw0[j] = 0; }
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 3; ++j) { pc0[j] += flip ? -pc[3 * i + j] : pc[3 * i + j]; pw0[j] += S.pws[3 * i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= 5; pw0[j] /= 5; }
  }
  auto load_abt = [&](int only) {    // the candidates' ABt into their rows (only >= 0: that candidate's alone)
    const double* pc = S.pcs[cand];
    if (slot < 3 && (only < 0 || slot == only) && col < 9) {        // ABt, loaded transposed into the Jacobi rows
      const int j = col / 3, k = col % 3;
      double s = 0;
      for (int i = 0; i < 5; ++i) s += ((flip ? -pc[3 * i + j] : pc[3 * i + j]) - pc0[j]) * (S.pws[3 * i + k] - pw0[k]);
      S.jr[(base + k) * 16 + j] = s;
    } else if (slot < 3 && (only < 0 || slot == only) && col < 12) {
      const int r = col - 9;
      for (int k = 0; k < 3; ++k) S.jr[(base + r) * 16 + 3 + k] = r == k ? 1.0 : 0.0;
      for (int k = 6; k < 12; ++k) S.jr[(base + r) * 16 + k] = 0.0;   // the columns behind V: finite
    }
    EO_SYNC();
  };
  load_abt(-1);
  {
    const bool mine = r16 < 15 && r16 % 5 < 3;
    ex = jacobi_rows<3>(S, mine ? 3 : 0, mine ? 5 * (r16 / 5) : r16, lane);
  }
  bd = svd_rank(S, slot < 3 ? 3 : 0, base, cand, lane);
  if (ex | bd) {             // (uniform)
    const unsigned need = need_seq(ex, bd);
    for (int pp = 0; pp < 3; ++pp)
      if (need >> pp & 1) load_abt(pp);
    if (slot < 3 && col == 0 && (need >> slot & 1)) svd_small_seq(S, slot == 0 ? xw.jW : (slot == 1 ? xw.sA : xw.sUt), 3, 3, base, cand);
    EO_SYNC();
  }
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
  if (S.flag & 1) {             // the 12 x 12 decomposition met a branch not reproduced here: the sequential restatement decides (lane 0)
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
