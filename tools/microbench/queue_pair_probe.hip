// Which pairs of high-priority streams run two dependent kernel chains side by side at full speed?
// The tracker's tail is two chains on two streams (svo_track.hip); its rate was seen to depend on which hardware queues the
// runtime hands the two streams (profiles/README.md "Hardware queues").  This program makes six high-priority streams one after
// the other (optionally after `argv[1]` idle normal-priority + `argv[2]` idle high-priority streams: a process's history) and
// times, for every pair, chain A (16 x one launch of 101 workgroups spinning ~35 us) beside chain B (16 x [768 short
// workgroups, then one 1024-thread workgroup spinning ~25 us]) - the shapes of the pose and the index chain.
// Output: one JSON line per pair: microseconds for both chains together, and each chain alone on its stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
__global__ void k_spin(long long cycles, int* sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
  if (cycles < 0) *sink = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void chainA(hipStream_t s, int* d) { for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_spin, dim3(101), dim3(256), 0, s, 70000LL, d); }
static void chainB(hipStream_t s, int* d) {
  for (int i = 0; i < 16; ++i) {
    hipLaunchKernelGGL(k_spin, dim3(768), dim3(256), 0, s, 4000LL, d);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, s, 50000LL, d);
  }
}
int main(int argc, char** argv) {
  const int n_norm = argc > 1 ? atoi(argv[1]) : 0, n_high = argc > 2 ? atoi(argv[2]) : 0;
  int least = 0, greatest = 0;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  int* d; hipMalloc(&d, 64);
  std::vector<hipStream_t> idle;
  for (int i = 0; i < n_norm; ++i) { hipStream_t s; hipStreamCreateWithPriority(&s, hipStreamNonBlocking, (least + greatest) / 2); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 100LL, d); idle.push_back(s); }
  for (int i = 0; i < n_high; ++i) { hipStream_t s; hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 100LL, d); idle.push_back(s); }
  hipDeviceSynchronize();
  const int NS = 6;
  hipStream_t h[NS];
  for (int i = 0; i < NS; ++i) { hipStreamCreateWithPriority(&h[i], hipStreamNonBlocking, greatest); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, h[i], 100LL, d); }
  hipDeviceSynchronize();
  double aloneA[NS], aloneB[NS];
  for (int i = 0; i < NS; ++i) {
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now_us(); chainA(h[i], d); hipStreamSynchronize(h[i]); aloneA[i] = now_us() - t0;
      t0 = now_us(); chainB(h[i], d); hipStreamSynchronize(h[i]); aloneB[i] = now_us() - t0;
    }
  }
  for (int i = 0; i < NS; ++i)
    for (int j = 0; j < NS; ++j) {
      if (i == j) continue;
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now_us();
        chainB(h[j], d); chainA(h[i], d);
        hipStreamSynchronize(h[i]); hipStreamSynchronize(h[j]);
        const double t = now_us() - t0;
        if (t < best) best = t;
      }
      printf("{\"idle_normal\": %d, \"idle_high\": %d, \"pose_stream\": %d, \"index_stream\": %d, \"both_us\": %.0f, \"pose_alone_us\": %.0f, \"index_alone_us\": %.0f}\n",
             n_norm, n_high, i, j, best, aloneA[i], aloneB[j]);
    }
  return 0;
}
