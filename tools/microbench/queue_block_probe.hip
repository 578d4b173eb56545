// Does a long multi-workgroup dispatch on one hardware queue hold up short kernels on ANOTHER queue - and for which pairs?
// (The tracker's tail was seen to slow down in some process states with its index-chain kernels 5x and the front end's 1.5x
// longer while every pair of its streams passed the one-wave stream probe: docs/NEXT_ROUNDS.md.  One explanation: two queues
// behind one dispatch pipe of the command processor - while the pipe is busy placing the workgroups of a large grid, the
// other queue's packets wait, however small.)
//
//   queue_block_probe <masked|pooled|mixed> [N = 12] [pad = 0]
//     masked: N streams from hipExtStreamCreateWithCUMask (every CU allowed): one dedicated hardware queue each
//     pooled: N high-priority streams (the runtime's pool: 4 hardware queues per priority)
//     mixed : N masked streams, `pad` pooled normal-priority streams created and used in between each
//     destroy: N masked streams; the table, then stream 1 is destroyed and the table is taken again (do the others move up?)
// For every ordered pair (a, b): stream a runs ONE launch of 4096 workgroups that each hold 64 KB of LDS (two per CU: the grid
// stays in dispatch for its whole ~400 us) while stream b runs a chain of 8 dependent one-wave kernels.  Output: one JSON line
// per blocker stream with the victim chains' times relative to the chain alone (1.0 = not held up).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
__global__ void k_spin(long long cycles, int* sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
  if (cycles < 0) *sink = 1;
}
__global__ void k_block(long long cycles, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = (int)cycles;
  __syncthreads();
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
  if (cycles < 0) *sink = lds[(threadIdx.x + 1) & 63];
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void victim(hipStream_t s, int* d) { for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 2000LL, d); }
int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "masked";
  const int N = argc > 2 ? atoi(argv[2]) : 12, pad = argc > 3 ? atoi(argv[3]) : 0;
  int least = 0, greatest = 0;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int* d; hipMalloc(&d, 64);
  hipFuncSetAttribute((const void*)k_block, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  std::vector<hipStream_t> s(N), pads;
  std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
  for (int i = 0; i < N; ++i) {
    if (!strcmp(mode, "pooled")) hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, greatest);
    else hipExtStreamCreateWithCUMask(&s[i], (uint32_t)mask.size(), mask.data());
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], 100LL, d);
    hipStreamSynchronize(s[i]);
    for (int p = 0; p < (strcmp(mode, "mixed") ? 0 : pad); ++p) {
      hipStream_t q; hipStreamCreateWithPriority(&q, hipStreamNonBlocking, (least + greatest) / 2);
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, q, 100LL, d); hipStreamSynchronize(q); pads.push_back(q);
    }
  }
  hipDeviceSynchronize();
  std::vector<double> alone(N);
  for (int i = 0; i < N; ++i)
    for (int rep = 0; rep < 3; ++rep) { const double t0 = now_us(); victim(s[i], d); hipStreamSynchronize(s[i]); alone[i] = now_us() - t0; }
  double block_us = 0;
  { const double t0 = now_us(); hipLaunchKernelGGL(k_block, dim3(4096), dim3(64), 65536, s[0], 100000LL, d); hipStreamSynchronize(s[0]); block_us = now_us() - t0; }
  const bool destroy = !strcmp(mode, "destroy");   // masked queues; after the first table stream 1 is destroyed and the table taken again
  for (int pass = 0; pass < (destroy ? 2 : 1); ++pass) {
  if (pass == 1) { hipStreamDestroy(s[1]); s[1] = nullptr; hipDeviceSynchronize(); printf("{\"note\": \"stream 1 destroyed\"}\n"); }
  for (int a = 0; a < N; ++a) {
    if (!s[a]) continue;
    printf("{\"mode\": \"%s\", \"pad\": %d, \"blocker\": %d, \"blocker_alone_us\": %.0f, \"victim_alone_us\": %.0f, \"victims\": [", mode, pad, a, block_us, alone[a]);
    for (int b = 0; b < N; ++b) {
      double best = 1e30;
      if (a != b && s[b])
        for (int rep = 0; rep < 2; ++rep) {
          hipLaunchKernelGGL(k_block, dim3(4096), dim3(64), 65536, s[a], 100000LL, d);
          const double t0 = now_us();
          victim(s[b], d); hipStreamSynchronize(s[b]);
          const double t = now_us() - t0;
          hipStreamSynchronize(s[a]);
          if (t < best) best = t;
        }
      printf("%s%.1f", b ? ", " : "", a == b || !s[b] ? 0.0 : best / alone[b]);
    }
    printf("]}\n");
    fflush(stdout);
  }
  }
  return 0;
}
