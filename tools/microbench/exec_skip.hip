// Does a wave with only its first 16 (or 32) lanes active issue VALU instructions faster? (pass skipping)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 512
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, const double* in, int nact) {
  const int lane = threadIdx.x;
  double x = in[lane], c = in[64 + lane];
  float xf = (float)x, cf = (float)c;
  long long t0 = 0, t1 = 0, t2 = 0;
  if (lane < nact) {
    __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll 32
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, c, c);
    __builtin_amdgcn_sched_barrier(0); asm volatile("" ::"v"(x)); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll 32
    for (int i = 0; i < N; ++i) xf = __builtin_fmaf(xf, cf, cf);
    __builtin_amdgcn_sched_barrier(0); asm volatile("" ::"v"(xf)); t2 = clock64(); __builtin_amdgcn_sched_barrier(0);
  }
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
  out[lane] = x + xf;
}
int main() {
  double h[128];
  for (int i = 0; i < 128; ++i) h[i] = 1.0 + 1e-9 * i;
  double *din, *dout; long long* dc;
  hipMalloc(&din, sizeof h); hipMalloc(&dout, 64 * 8); hipMalloc(&dc, 16);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  for (int nact : {64, 32, 16, 8, 1}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, dc, din, nact);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, dc, din, nact);
    hipDeviceSynchronize();
    long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    printf("%2d active lanes: dep fma f64 %.2f ticks/op, dep fma f32 %.2f ticks/op\n", nact, c[0] / (double)N, c[1] / (double)N);
  }
  return 0;
}
