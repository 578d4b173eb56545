// f64_latency.hip - dependent-chain latency / issue cost of the float64 operations the order-preserving EPnP is built from,
// ONE wave (optionally four waves of one workgroup, one per SIMD), cycles from s_memtime around N chained operations.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define N 256
template <int K> __device__ __forceinline__ double bc(double x) {
  double y;
  if (K == 0) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  if (K == 1) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  if (K == 2) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  if (K == 3) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  return y;
}
__device__ __forceinline__ double shr1(double x) {   // two 32-bit DPP moves: row_shr:1
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double x, int addr) {
  int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
  int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdlane(double x, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
// division without scaling / fix-up (operands known to be normal, quotient in range): rcp + 2 Newton + residual correction
__device__ __forceinline__ double div_noscale(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
  double q = a * r;
  double rem = __builtin_fma(-b, q, a);
  return __builtin_fma(rem, r, q);
}
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, const double* in, int addr_in) {
  __shared__ double sh[256 * 4];
  const int lane = threadIdx.x;
  double x = in[lane & 63], c = in[64 + (lane & 63)];
  double r[16];
  long long t0, t1;
  int slot = 0;
#define BEGIN() __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#define END() __builtin_amdgcn_sched_barrier(0); asm volatile("" :: "v"(x)); t1 = clock64(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[slot] = t1 - t0; ++slot;
  // 0 dependent add
  BEGIN(); for (int i = 0; i < N; ++i) x = x + c; END();
  // 1 dependent mul
  BEGIN(); for (int i = 0; i < N; ++i) x = x * c; END();
  // 2 dependent fma
  BEGIN(); for (int i = 0; i < N; ++i) x = __builtin_fma(x, c, c); END();
  // 3 dependent IEEE div
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N; ++i) x = c / x; END();
  // 4 dependent IEEE sqrt (+ add)
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N; ++i) x = sqrt(x) + c; END();
  // 5 eight independent add chains (issue rate): 8 * N adds
  for (int j = 0; j < 8; ++j) r[j] = x + j;
  BEGIN(); for (int i = 0; i < N; ++i) { for (int j = 0; j < 8; ++j) r[j] = r[j] + c; } 
  for (int j = 1; j < 8; ++j) r[0] += r[j]; x = r[0]; END();
  // 6 eight independent fma chains
  for (int j = 0; j < 8; ++j) r[j] = x + j;
  BEGIN(); for (int i = 0; i < N; ++i) { for (int j = 0; j < 8; ++j) r[j] = __builtin_fma(r[j], c, c); }
  for (int j = 1; j < 8; ++j) r[0] += r[j]; x = r[0]; END();
  // 7 chain: v_mov_b64_dpp row_newbcast -> add (the in-order sum over the lanes of a row)
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N / 4; ++i) { x = x + bc<0>(x); x = x + bc<1>(x); x = x + bc<2>(x); x = x + bc<3>(x); } END();
  // 8 adds whose DPP operands are all ready beforehand (products known up front): acc += bcast_k(p)
  { double p = in[lane & 63] * c, acc = 0;
    BEGIN(); for (int i = 0; i < N / 4; ++i) { acc = acc + bc<0>(p); acc = acc + bc<1>(p); acc = acc + bc<2>(p); acc = acc + bc<3>(p); } x = acc; END(); }
  // 9 chain: 2 x v_mov_b32_dpp row_shr:1 -> add
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N; ++i) x = shr1(x) + c; END();
  // 10 chain: ds_bpermute (2 x b32) -> add
  x = in[lane & 63];
  { const int addr = ((lane + addr_in) & 63) * 4;
    BEGIN(); for (int i = 0; i < N; ++i) x = bperm(x, addr) + c; END(); }
  // 11 chain: LDS write b64 -> read b64 (other lane's slot) -> add
  x = in[lane & 63];
  { const int wi = lane, ri = (lane & ~63) | ((lane + addr_in) & 63);
    BEGIN(); for (int i = 0; i < N; ++i) { sh[wi] = x; __builtin_amdgcn_wave_barrier(); x = sh[ri] + c; __builtin_amdgcn_wave_barrier(); } END(); }
  // 12 dependent v_rcp_f64
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x); END();
  // 13 dependent v_rsq_f64
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x); END();
  // 14 chain: readlane (2 x b32) -> add
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N; ++i) x = rdlane(x, 3) + c; END();
  // 15 dependent division without scaling
  x = in[lane & 63];
  BEGIN(); for (int i = 0; i < N; ++i) x = div_noscale(c, x); END();
  // 16 three interleaved chains of (bcast -> add): p, a, b sums of one Jacobi pair
  { double p = in[lane & 63] * c, q = p + 1, s = p + 2, a0 = 0, a1 = 0, a2 = 0;
    BEGIN(); for (int i = 0; i < N / 4; ++i) {
      a0 += bc<0>(p); a1 += bc<0>(q); a2 += bc<0>(s); a0 += bc<1>(p); a1 += bc<1>(q); a2 += bc<1>(s);
      a0 += bc<2>(p); a1 += bc<2>(q); a2 += bc<2>(s); a0 += bc<3>(p); a1 += bc<3>(q); a2 += bc<3>(s); }
    x = a0 + a1 + a2; END(); }
  // 17 LDS: write b64, then one lane-group reads 12 values as 6 x b128 and sums them in order (transposed sum)
  x = in[lane & 63];
  { typedef double d2 __attribute__((ext_vector_type(2)));
    BEGIN(); for (int i = 0; i < N / 8; ++i) {
      sh[lane] = x; __builtin_amdgcn_wave_barrier();
      const d2* row = reinterpret_cast<const d2*>(&sh[(lane & ~15)]);
      double acc = 0;
      for (int k2 = 0; k2 < 6; ++k2) { d2 v = row[k2]; acc = acc + v.x; acc = acc + v.y; }
      x = acc; __builtin_amdgcn_wave_barrier(); }
    END(); }
  out[threadIdx.x] = x;
}
int main() {
  double h[128];
  for (int i = 0; i < 64; ++i) { h[i] = 1.0 + 0.01 * i; h[64 + i] = 1.0000001 + 1e-9 * i; }
  double *din, *dout; long long* dc;
  hipMalloc(&din, sizeof h); hipMalloc(&dout, 256 * 8); hipMalloc(&dc, 64 * 8);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  const char* names[] = {"dep add", "dep mul", "dep fma", "dep IEEE div", "dep IEEE sqrt + add", "8 indep add chains (per add)",
    "8 indep fma chains (per fma)", "dep (bcast64 -> add)", "add chain, bcast operands ready", "dep (2 x dpp32 row_shr -> add)",
    "dep (bpermute x2 -> add)", "dep (LDS write -> read -> add)", "dep v_rcp_f64", "dep v_rsq_f64", "dep (readlane x2 -> add)",
    "dep div, no scale/fixup", "3 interleaved bcast-add chains (per k)", "LDS transposed 12-sum (per sum)"};
  const double per[] = {N, N, N, N, N, 8.0 * N, 8.0 * N, N, N, N, N, N, N, N, N, N, N, N / 8};
  for (int threads = 64; threads <= 256; threads *= 4) {
    hipMemset(dc, 0, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, dout, dc, din, 1);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, dout, dc, din, 1);
    hipDeviceSynchronize();
    long long c[64]; hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
    printf("---- %d threads (wave 0's clock) ----\n", threads);
    for (int i = 0; i < 18; ++i) printf("%-44s %8lld ticks  %7.2f per op\n", names[i], c[i], c[i] / per[i]);
  }
  // tick rate: s_memtime against a timed kernel
  return 0;
}
