#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(const double* A, const double* B, double* D) {
  const int lane = threadIdx.x;
  D[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[lane], B[lane], 0.0, 0, 0, 0);
}
int main() {
  double *dA, *dB, *dD;
  hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 64 * 8);
  double A[64], B[64], D[64];
  printf("B one-hot at lane x (A = 1): output lanes that see it\n");
  for (int x = 0; x < 64; ++x) {
    for (int i = 0; i < 64; ++i) { A[i] = 1.0; B[i] = i == x ? 1.0 : 0.0; }
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    printf("B %2d ->", x);
    for (int i = 0; i < 64; ++i) if (D[i] != 0) printf(" %d", i);
    printf("\n");
  }
  printf("A one-hot at lane x (B = 1): output lanes that see it\n");
  for (int x = 0; x < 64; ++x) {
    for (int i = 0; i < 64; ++i) { B[i] = 1.0; A[i] = i == x ? 1.0 : 0.0; }
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    printf("A %2d ->", x);
    for (int i = 0; i < 64; ++i) if (D[i] != 0) printf(" %d", i);
    printf("\n");
  }
  return 0;
}
