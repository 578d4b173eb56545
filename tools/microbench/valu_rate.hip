// Integer-VALU issue-rate microbenchmark for gfx950: how many cycles does one wave64
// instruction of each kind occupy a SIMD?  (Sets the VALU roofline DESIGN.md quotes.)
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v2s __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u;
  uint32_t b = blockIdx.x * 97u + 13u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t o = a[(i + 3) & 7];   // data-dependent second operand: nothing folds
        if (KIND == 0) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2s, a[i]), __builtin_bit_cast(v2s, o)) + __builtin_bit_cast(v2s, b));  // pk_min + pk_add
        if (KIND == 1) a[i] = (uint32_t)min((int)a[i], (int)o) ^ b;     // v_min_i32 + v_xor
        if (KIND == 2) a[i] = __builtin_amdgcn_perm(a[i], o, 0x07020500u + i);
        if (KIND == 3) a[i] = (a[i] + o) ^ b;                            // v_add_u32 + v_xor (or v_xad)
        if (KIND == 4) { float f = __builtin_bit_cast(float, a[i]); f = f * __builtin_bit_cast(float, o) + 0.5f; a[i] = __builtin_bit_cast(uint32_t, f); }
      }
    b += 7;
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
double run(const char* name, int ops_per_inner) {
  const int blocks = 256 * 8, iters = 4096;
  uint32_t* d;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 64 * ops_per_inner;  // wave-instructions
  const double rate = winstr / (ms * 1e-3);
  printf("%-22s %8.3f ms  %7.1f G wave-instr/s  -> %.2f cycles per wave-instr per SIMD @2.4GHz (1024 SIMDs)\n",
         name, ms, rate / 1e9, 1024 * 2.4e9 / rate);
  hipFree(d);
  return rate;
}
int main() {
  run<0>("v_pk_min_i16+pk_add", 2);
  run<1>("v_min_i32+v_xor", 2);
  run<2>("v_perm_b32", 1);
  run<3>("v_add_u32+v_xor", 2);
  run<4>("v_fma_f32", 1);
  return 0;
}
