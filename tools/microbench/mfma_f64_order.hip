// mfma_f64_order.hip - is v_mfma_f64_16x16x4_f64 a SEQUENTIAL chain of fused multiply-adds over k = 0..3, rounded after each
// step, starting from C?  With A = 1.0 the products are exact, so D[.][col] would be the in-order sum ((((C + b0) + b1) + b2) + b3)
// of IEEE additions - usable for the k-ordered sums of OpenCV's loops.  Compares against several candidate orders, bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(const double* A, const double* B, const double* C, double* D, long long* cyc) {
  const int lane = threadIdx.x;
  const double a = A[lane], b = B[lane];
  d4 c;
  // C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
  for (int r = 0; r < 4; ++r) c[r] = C[((lane >> 4) + 4 * r) * 16 + (lane & 15)];
  d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = d[r];
  // latency of a dependent chain (C = previous D)
  __builtin_amdgcn_sched_barrier(0);
  long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  d4 e = d;
#pragma unroll
  for (int i = 0; i < 64; ++i) e = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::"v"(e));
  long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  // chain where a VALU op sits between (MFMA -> VALU add -> MFMA): the hand-over latency
  d4 f = d;
#pragma unroll
  for (int i = 0; i < 64; ++i) { f = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, f, 0, 0, 0); f[0] = f[0] + a; f[1] = f[0]; f[2] = f[0]; f[3] = f[0]; }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::"v"(f));
  long long t2 = clock64();
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
  D[256 + lane] = e[0] + e[1] + e[2] + e[3] + f[0];
}
static double rnd_wide() {
  const double m = 1.0 + rand() / (double)RAND_MAX;
  const int e = rand() % 80 - 40;
  return (rand() & 1 ? -1 : 1) * ldexp(m, e) * (1.0 + rand() / (double)RAND_MAX * 1e-9);
}
int main() {
  srand(7);
  int same_seq = 0, same_end = 0, same_rev = 0, same_pair = 0, total = 0, ones_seq = 0, ones_total = 0;
  double *dA, *dB, *dC, *dD; long long* dc;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 512 * 8); hipMalloc(&dc, 16);
  long long cyc[2] = {0, 0};
  for (int trial = 0; trial < 400; ++trial) {
    double A[64], B[64], C[256], D[256];
    const bool ones = trial & 1;            // A = 1.0: exact products (the use case); otherwise general products (fma semantics)
    for (int i = 0; i < 64; ++i) { A[i] = ones ? 1.0 : rnd_wide(); B[i] = rnd_wide(); }
    for (int i = 0; i < 256; ++i) C[i] = (trial % 4 < 2) ? rnd_wide() : 0.0;
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice); hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, dc);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    // A[row][k]: lane = k * 16 + row; B[k][col]: lane = k * 16 + col
    for (int row = 0; row < 16; ++row)
      for (int col = 0; col < 16; ++col) {
        const double c = C[row * 16 + col];
        double s1 = c;
        for (int kk = 0; kk < 4; ++kk) s1 = fma(A[kk * 16 + row], B[kk * 16 + col], s1);          // sequential from C
        double s2 = 0;
        for (int kk = 0; kk < 4; ++kk) s2 = fma(A[kk * 16 + row], B[kk * 16 + col], s2);
        s2 = s2 + c;                                                                              // products first, C last
        double s3 = c;
        for (int kk = 3; kk >= 0; --kk) s3 = fma(A[kk * 16 + row], B[kk * 16 + col], s3);         // reverse
        const double p01 = fma(A[0 * 16 + row], B[0 * 16 + col], A[16 + row] * B[16 + col]);
        const double p23 = fma(A[32 + row], B[32 + col], A[48 + row] * B[48 + col]);
        const double s4 = (p01 + p23) + c;                                                        // pairwise
        const double g = D[row * 16 + col];
        ++total;
        same_seq += !memcmp(&g, &s1, 8); same_end += !memcmp(&g, &s2, 8); same_rev += !memcmp(&g, &s3, 8); same_pair += !memcmp(&g, &s4, 8);
        if (ones) { ++ones_total; ones_seq += !memcmp(&g, &s1, 8); }
      }
  }
  printf("entries %d: identical to sequential-from-C %d, products-then-C %d, reverse %d, pairwise %d\n", total, same_seq, same_end, same_rev, same_pair);
  printf("A = 1.0 entries %d: identical to the in-order IEEE sum %d\n", ones_total, ones_seq);
  printf("dependent MFMA chain: %.1f ticks per MFMA; MFMA -> v_add -> MFMA: %.1f ticks per round\n", cyc[0] / 64.0, cyc[1] / 64.0);
  return 0;
}
