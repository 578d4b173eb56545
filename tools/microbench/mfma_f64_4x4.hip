// mfma_f64_4x4.hip - v_mfma_f64_4x4x4_4b_f64: operand layout, accumulation order and latency.
// Layout probe: A one-hot / B one-hot patterns tell which lane feeds which (block, i, k) / (block, k, j) and where D[i][j] lands.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__global__ __launch_bounds__(64) void k(const double* A, const double* B, const double* C, double* D, long long* cyc) {
  const int lane = threadIdx.x;
  const double a = A[lane], b = B[lane], c = C[lane];
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  D[lane] = d;
  __builtin_amdgcn_sched_barrier(0);
  long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  double e = d;
#pragma unroll
  for (int i = 0; i < 64; ++i) e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::"v"(e));
  long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  double f = d;
#pragma unroll
  for (int i = 0; i < 64; ++i) { f = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f, 0, 0, 0); f = f + a; }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::"v"(f));
  long long t2 = clock64();
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
  D[64 + lane] = e + f;
}
static double rnd_wide() {
  const double m = 1.0 + rand() / (double)RAND_MAX;
  const int e = rand() % 80 - 40;
  return (rand() & 1 ? -1 : 1) * ldexp(m, e);
}
int main() {
  double *dA, *dB, *dC, *dD; long long* dc;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 64 * 8); hipMalloc(&dD, 128 * 8); hipMalloc(&dc, 16);
  double A[64], B[64], C[64], D[64];
  long long cyc[2];
  // layout hypothesis H: block = lane >> 4; A[i][k]: i = lane & 3, k = (lane >> 2) & 3; B[k][j]: j = lane & 3, k = (lane >> 2) & 3;
  // D[i][j]: j = lane & 3, i = (lane >> 2) & 3.  Checked with random data below (and its transposed variants).
  srand(3);
  int ok[4] = {0, 0, 0, 0}, total = 0, ones_ok = 0, ones_total = 0;
  for (int trial = 0; trial < 2000; ++trial) {
    const bool ones = trial & 1;
    for (int i = 0; i < 64; ++i) { A[i] = ones ? 1.0 : rnd_wide(); B[i] = rnd_wide(); C[i] = (trial % 4 < 2) ? rnd_wide() : 0.0; }
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice); hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, dc);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; ++lane) {
      // measured layout: D[b][i][j] at lane 16 i + 4 b + j, A[b][i][k] at lane 16 k + 4 b + i, B[b][k][j] at lane 16 k + 4 b + j
      const int i = lane >> 4, bb = (lane & 15) >> 2, j = lane & 3;
      for (int v = 0; v < 4; ++v) {
        double s = C[lane];
        if (v == 0) for (int kk = 0; kk < 4; ++kk) s = fma(A[16 * kk + 4 * bb + i], B[16 * kk + 4 * bb + j], s);
        if (v == 1) for (int kk = 3; kk >= 0; --kk) s = fma(A[16 * kk + 4 * bb + i], B[16 * kk + 4 * bb + j], s);
        if (v == 2) { double t = 0; for (int kk = 0; kk < 4; ++kk) t = fma(A[16 * kk + 4 * bb + i], B[16 * kk + 4 * bb + j], t); s = t + C[lane]; }
        if (v == 3) { const double p01 = fma(A[4 * bb + i], B[4 * bb + j], A[16 + 4 * bb + i] * B[16 + 4 * bb + j]);
                      const double p23 = fma(A[32 + 4 * bb + i], B[32 + 4 * bb + j], A[48 + 4 * bb + i] * B[48 + 4 * bb + j]); s = (p01 + p23) + C[lane]; }
        ok[v] += !memcmp(&s, &D[lane], 8);
        if (ones && v == 0) { ++ones_total; ones_ok += !memcmp(&s, &D[lane], 8); }
      }
      ++total;
    }
  }
  printf("entries %d; sequential-from-C %d, reverse %d, products-then-C %d, pairwise %d\n", total, ok[0], ok[1], ok[2], ok[3]);
  printf("A = 1.0: %d of %d identical to the in-order sum over k = (lane >> 2) & 3 for chain j = lane & 3 (variant 0)\n", ones_ok, ones_total);
  printf("dependent 4x4x4 MFMA chain: %.1f ticks per MFMA; MFMA -> v_add -> MFMA: %.1f ticks per round\n", cyc[0] / 64.0, cyc[1] / 64.0);
  return 0;
}
