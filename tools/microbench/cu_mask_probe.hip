// Where do the workgroups of a CU-masked stream land?  hipExtStreamCreateWithCUMask with the first K 32-bit words set (what
// svo_stream_create_masked does): histogram of HW_REG_XCC_ID and of the (SE, CU) fields of HW_REG_HW_ID over many workgroups.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/cu_mask_probe tools/microbench/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ void k_where(unsigned* out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // keep the workgroup alive for a while so that the grid spreads over everything the mask allows
  long long t0 = clock64();
  while (clock64() - t0 < 20000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  printf("%d CUs, %d mask words\n", cus, words);
  const int NB = 8192;
  unsigned* d;
  hipMalloc(&d, NB * 8);
  std::vector<unsigned> h(2 * NB);
  for (int keep = 1; keep <= words; keep = keep == 1 ? 2 : keep * 2) {
    std::vector<uint32_t> mask(words, 0u);
    for (int w = 0; w < keep; ++w) mask[w] = 0xffffffffu;
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, words, mask.data()) != hipSuccess) { printf("mask stream failed\n"); return 1; }
    hipMemsetAsync(d, 0xff, NB * 8, st);
    hipLaunchKernelGGL(k_where, dim3(NB), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_xcc;
    std::map<unsigned, int> per_cu;
    for (int b = 0; b < NB; ++b) {
      const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      per_xcc[xcc]++;
      per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("first %d word(s) set: workgroups per XCC:", keep);
    for (auto& p : per_xcc) printf(" xcc%u=%d", p.first, p.second);
    printf("; distinct (xcc, se, sh, cu) = %zu\n", per_cu.size());
    hipStreamDestroy(st);
  }
  // one XCD's worth by interleave: bits i with i % 8 == 0
  {
    std::vector<uint32_t> mask(words, 0x01010101u);
    hipStream_t st;
    hipExtStreamCreateWithCUMask(&st, words, mask.data());
    hipLaunchKernelGGL(k_where, dim3(NB), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_xcc;
    for (int b = 0; b < NB; ++b) per_xcc[h[2 * b] & 0xf]++;
    printf("bits i %% 8 == 0 set: workgroups per XCC:");
    for (auto& p : per_xcc) printf(" xcc%u=%d", p.first, p.second);
    printf("\n");
    hipStreamDestroy(st);
  }
  // (A mask with a single bit set is not honoured as such - the workgroups then land on nearly every CU -, so the
  // bit -> CU mapping cannot be read off one bit at a time.)
  return 0;
}
