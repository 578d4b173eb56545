// Does a high-priority stream keep its priority whatever the process did before?
// The tracker's pose chain (a high-priority stream) was seen to wait behind the batched front end's grids - which fill every wave slot
// of the eighth of the CUs their stream is confined to - in processes that had used other streams before, and not in fresh ones.
//
//   queue_prio_probe [idle_high = 0] [idle_normal = 0] [N = 8] [blocker_first = 0]
// creates `idle_high` high- and `idle_normal` normal-priority streams (used once, kept), then N high-priority streams and one
// CU-masked stream (first mask word: four CUs of every XCD) - before the N streams if blocker_first.  For every one of the N: a grid
// of 3072 one-wave workgroups alone, and the same grid while four launches of 4096 x 256 threads (8 workgroups = all wave slots
// per CU, ~45 us each round) run on the masked stream.  One JSON line per stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
__global__ void k_spin(long long cycles) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(2);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double grid_us(hipStream_t q) {
  hipStreamSynchronize(q);
  const double t0 = now_us();
  hipLaunchKernelGGL(k_spin, dim3(3072), dim3(64), 0, q, 4000LL);
  hipStreamSynchronize(q);
  return now_us() - t0;
}
int main(int argc, char** argv) {
  const int ih = argc > 1 ? atoi(argv[1]) : 0, in = argc > 2 ? atoi(argv[2]) : 0, N = argc > 3 ? atoi(argv[3]) : 8, bf = argc > 4 ? atoi(argv[4]) : 0;
  int least = 0, greatest = 0;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, 100LL); hipDeviceSynchronize();
  std::vector<hipStream_t> idle;
  for (int i = 0; i < ih + in; ++i) {
    hipStream_t s; hipStreamCreateWithPriority(&s, hipStreamNonBlocking, i < ih ? greatest : (least + greatest) / 2);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 100LL); hipStreamSynchronize(s); idle.push_back(s);
  }
  std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u); mask[0] = 0xffffffffu;
  hipStream_t blk = nullptr;
  if (bf) { hipExtStreamCreateWithCUMask(&blk, (uint32_t)mask.size(), mask.data()); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, blk, 100LL); hipStreamSynchronize(blk); }
  std::vector<hipStream_t> s(N);
  for (int i = 0; i < N; ++i) { hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, greatest); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], 100LL); hipStreamSynchronize(s[i]); }
  if (!bf) { hipExtStreamCreateWithCUMask(&blk, (uint32_t)mask.size(), mask.data()); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, blk, 100LL); hipStreamSynchronize(blk); }
  hipDeviceSynchronize();
  for (int i = 0; i < N; ++i) {
    (void)grid_us(s[i]);
    const double alone = grid_us(s[i]);
    double beside = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
      for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, blk, 100000LL);
      const double t = grid_us(s[i]);
      if (t < beside) beside = t;
      hipStreamSynchronize(blk);
    }
    printf("{\"idle_high\": %d, \"idle_normal\": %d, \"blocker_first\": %d, \"stream\": %d, \"alone_us\": %.0f, \"beside_front_end_grids_us\": %.0f}\n", ih, in, bf, i, alone, beside);
    fflush(stdout);
  }
  return 0;
}
