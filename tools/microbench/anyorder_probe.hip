// Does hipExtAnyOrderLaunch clear the barrier bit of a kernel's AQL packet on gfx950, i.e. may a kernel start while the
// previous kernel of the SAME stream still runs?  (hip_ext.h says "not supported on GFX9xx".)  Kernel A spins (bounded) on a
// flag; kernel B, enqueued behind it on the same stream with the flag, sets it.  If B overtakes, A leaves within microseconds.
// Second part: the gap between two dependent empty kernels of one stream, launched the ordinary way and with the flag.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void k_wait(int* f, int limit) {
  int spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < limit) { __builtin_amdgcn_s_sleep(8); ++spins; }
  f[1] = spins;
}
__global__ void k_set(int* f) { __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_nop(int* f) { if (f == nullptr) __builtin_trap(); }
int main() {
  int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
    hipMemsetAsync(d, 0, 64, s); hipStreamSynchronize(s);
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, d, 200000);
    hipExtLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s, nullptr, nullptr, flags, d);
    hipStreamSynchronize(s);
    int h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("{\"test\": \"overtake\", \"flags\": %d, \"waiter_spins\": %d, \"limit\": 200000, \"overtook\": %s}\n", flags, h[1], h[1] < 200000 ? "true" : "false");
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, s);
      for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, d);
      hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("{\"test\": \"empty_kernel_chain\", \"flags\": %d, \"us_per_kernel\": %.3f}\n", flags, 1e3 * ms / N);
    }
  }
  return 0;
}
