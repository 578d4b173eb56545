// Second integer-VALU issue-rate table for gfx950 (see valu_rate.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v2s __attribute__((ext_vector_type(2)));
#define BC(T, x) __builtin_bit_cast(T, x)
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t o = a[(i + 3) & 7], p = a[(i + 5) & 7];
        if (KIND == 0) a[i] = (a[i] & o) | p;                               // v_and_or_b32
        if (KIND == 1) a[i] = (a[i] << 3) | o;                              // v_lshl_or_b32
        if (KIND == 2) a[i] = a[i] - o;                                     // v_sub_u32
        if (KIND == 3) a[i] = __builtin_amdgcn_ubfe(a[i] ^ o, 8, 8) + p;    // v_xor + v_bfe_u32 + v_add
        if (KIND == 4) a[i] = (a[i] > o) ? p : a[i];                        // v_cmp + v_cndmask
        if (KIND == 5) a[i] = max(a[i], o);                                 // v_max_u32
        if (KIND == 6) a[i] = __builtin_amdgcn_sad_u8(a[i], o, p);          // v_sad_u8
        if (KIND == 7) a[i] = __builtin_amdgcn_alignbyte(a[i], o, 1);       // v_alignbyte_b32
        if (KIND == 8) a[i] = (a[i] & 0xffffff) * (o & 0xffffff) + p;       // v_mad_u32_u24
        if (KIND == 9) a[i] = a[i] + o + p;                                 // v_add3_u32
        if (KIND == 10) a[i] = a[i] >> (o & 7);                             // v_lshrrev_b32
        if (KIND == 11) a[i] = (uint32_t)min(min((int)a[i], (int)o), (int)p);  // v_min3_i32
        if (KIND == 12) a[i] = BC(uint32_t, BC(v2s, a[i]) - BC(v2s, o));    // v_pk_sub_i16
        if (KIND == 13) a[i] = (a[i] ^ o) & p;                              // v_xor + v_and (or v_bfi)
        if (KIND == 14) a[i] = __builtin_popcount(a[i] ^ o) + p;            // v_xor + v_bcnt (bcnt has add)
        if (KIND == 15) a[i] = a[i] * o;                                    // v_mul_lo_u32
      }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, int ops) {
  const int blocks = 256 * 8, iters = 2048;
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
  const double winstr = (double)blocks * 4 * iters * 64 * ops;
  printf("%-28s %8.3f ms  -> %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, ms,
         1024 * 2.4e9 / (winstr / (ms * 1e-3)));
  hipFree(d);
}
int main() {
  run<0>("v_and_or_b32", 1); run<1>("v_lshl_or_b32", 1); run<2>("v_sub_u32", 1);
  run<3>("xor+bfe_u32+add (3)", 3); run<4>("v_cmp+v_cndmask (2)", 2); run<5>("v_max_u32", 1);
  run<6>("v_sad_u8", 1); run<7>("v_alignbyte_b32", 1); run<8>("and,and,v_mad_u32_u24 (3)", 3);
  run<9>("v_add3_u32", 1); run<10>("and+v_lshrrev (2)", 2); run<11>("v_min3_i32", 1);
  run<12>("v_pk_sub_i16", 1); run<13>("xor+and (2)", 2); run<14>("xor+bcnt (2)", 2);
  run<15>("v_mul_lo_u32", 1);
  return 0;
}
