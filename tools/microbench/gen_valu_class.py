#!/usr/bin/env python3
"""Generate valu_class.hip: the issue cost of one wave64 instruction on a gfx950 SIMD, per instruction, at 1 / 2 / 4 / 8
waves per SIMD.  Every kernel is one inline-assembly block of 64 independent instructions (eight accumulators, each
instruction reads its own accumulator and two others written >= 3 instructions earlier) inside a counted loop, so the
instruction that is timed is the one that is named - nothing is left to instruction selection.

Output of the binary: one JSON object per line
  {"instr": ..., "waves_per_simd": W, "cyc_per_instr_simd": elapsed shader cycles x 1024 SIMDs / wave-instructions,
   "cyc_per_instr_wave": median over waves of (s_memtime delta / instructions), "clock_ghz": s_memtime ticks / wall}
build: python3 gen_valu_class.py > valu_class.hip && hipcc -O2 --offload-arch=gfx950 valu_class.hip -o valu_class
"""
import sys

# name, template ({d} own accumulator, {a} = own, {b}, {c} others), kind: v32 | v64 (accumulators are register pairs)
INSTR = [
    ("v_add_u32", "v_add_u32 {d}, {a}, {b}", "v32"),
    ("v_sub_u32", "v_sub_u32 {d}, {a}, {b}", "v32"),
    ("v_and_b32", "v_and_b32 {d}, {a}, {b}", "v32"),
    ("v_and_b32_literal", "v_and_b32 {d}, 0x3f3f3f3f, {b}", "v32"),
    ("v_or_b32", "v_or_b32 {d}, {a}, {b}", "v32"),
    ("v_xor_b32", "v_xor_b32 {d}, {a}, {b}", "v32"),
    ("v_not_b32", "v_not_b32 {d}, {b}", "v32"),
    ("v_mov_b32", "v_mov_b32 {d}, {b}", "v32"),
    ("v_mov_b32_dpp_row_shr1", "v_mov_b32_dpp {d}, {b} row_shr:1 row_mask:0xf bank_mask:0xf", "v32"),
    ("v_add_u32_dpp_row_shr1", "v_add_u32_dpp {d}, {b}, {a} row_shr:1 row_mask:0xf bank_mask:0xf", "v32"),
    ("v_lshlrev_b32_imm", "v_lshlrev_b32 {d}, 3, {b}", "v32"),
    ("v_lshrrev_b32_imm", "v_lshrrev_b32 {d}, 3, {b}", "v32"),
    ("v_lshrrev_b32_reg", "v_lshrrev_b32 {d}, {a}, {b}", "v32"),
    ("v_ashrrev_i32_imm", "v_ashrrev_i32 {d}, 3, {b}", "v32"),
    ("v_bfe_u32", "v_bfe_u32 {d}, {b}, 8, 8", "v32"),
    ("v_bfi_b32", "v_bfi_b32 {d}, {a}, {b}, {c}", "v32"),
    ("v_and_or_b32", "v_and_or_b32 {d}, {a}, {b}, {c}", "v32"),
    ("v_or3_b32", "v_or3_b32 {d}, {a}, {b}, {c}", "v32"),
    ("v_add3_u32", "v_add3_u32 {d}, {a}, {b}, {c}", "v32"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {a}, 8, {b}", "v32"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {a}, 2, {b}", "v32"),
    ("v_add_lshl_u32", "v_add_lshl_u32 {d}, {a}, {b}, 2", "v32"),
    ("v_xad_u32", "v_xad_u32 {d}, {a}, {b}, {c}", "v32"),
    ("v_perm_b32", "v_perm_b32 {d}, {a}, {b}, {c}", "v32"),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {a}, {b}, 16", "v32"),
    ("v_alignbyte_b32", "v_alignbyte_b32 {d}, {a}, {b}, 1", "v32"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {a}, {b}", "v32"),
    ("v_pk_sub_i16", "v_pk_sub_i16 {d}, {a}, {b}", "v32"),
    ("v_pk_min_i16", "v_pk_min_i16 {d}, {a}, {b}", "v32"),
    ("v_pk_max_i16", "v_pk_max_i16 {d}, {a}, {b}", "v32"),
    ("v_pk_min_u16", "v_pk_min_u16 {d}, {a}, {b}", "v32"),
    ("v_pk_mul_lo_u16", "v_pk_mul_lo_u16 {d}, {a}, {b}", "v32"),
    ("v_pk_mad_u16", "v_pk_mad_u16 {d}, {a}, {b}, {c}", "v32"),
    ("v_pk_lshrrev_b16", "v_pk_lshrrev_b16 {d}, 2, {b}", "v32"),
    ("v_add_u16", "v_add_u16 {d}, {a}, {b}", "v32"),
    ("v_min_u16", "v_min_u16 {d}, {a}, {b}", "v32"),
    ("v_min_i32", "v_min_i32 {d}, {a}, {b}", "v32"),
    ("v_max_u32", "v_max_u32 {d}, {a}, {b}", "v32"),
    ("v_min3_u32", "v_min3_u32 {d}, {a}, {b}, {c}", "v32"),
    ("v_max3_i32", "v_max3_i32 {d}, {a}, {b}, {c}", "v32"),
    ("v_med3_i32", "v_med3_i32 {d}, {a}, {b}, {c}", "v32"),
    ("v_cmp_lt_u32_vcc", "v_cmp_lt_u32 vcc, {a}, {b}", "v32"),
    ("v_cmp_lt_u32_sgpr", "v_cmp_lt_u32 s[20:21], {a}, {b}", "v32"),
    ("v_cndmask_b32_vcc", "v_cndmask_b32 {d}, {a}, {b}, vcc", "v32"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {a}, {b}", "v32"),
    ("v_mul_hi_u32", "v_mul_hi_u32 {d}, {a}, {b}", "v32"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {d}, {a}, {b}", "v32"),
    ("v_mul_hi_u32_u24", "v_mul_hi_u32_u24 {d}, {a}, {b}", "v32"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {a}, {b}, {c}", "v32"),
    ("v_mad_i32_i24", "v_mad_i32_i24 {d}, {a}, {b}, {c}", "v32"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 {d}, {b}, {c}, {a}", "v32"),
    ("v_dot4_i32_i8", "v_dot4_i32_i8 {d}, {b}, {c}, {a}", "v32"),
    ("v_dot2_u32_u16", "v_dot2_u32_u16 {d}, {b}, {c}, {a}", "v32"),
    ("v_dot8_u32_u4", "v_dot8_u32_u4 {d}, {b}, {c}, {a}", "v32"),
    ("v_sad_u8", "v_sad_u8 {d}, {b}, {c}, {a}", "v32"),
    ("v_sad_u16", "v_sad_u16 {d}, {b}, {c}, {a}", "v32"),
    ("v_sad_u32", "v_sad_u32 {d}, {b}, {c}, {a}", "v32"),
    ("v_msad_u8", "v_msad_u8 {d}, {b}, {c}, {a}", "v32"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {d}, {b}, {a}", "v32"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 {d}, {b}, {a}", "v32"),
    ("v_ffbh_u32", "v_ffbh_u32 {d}, {b}", "v32"),
    ("v_sub_u32_sdwa_bytes", "v_sub_u32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", "v32"),
    ("v_add_u32_sdwa_dst_byte", "v_add_u32_sdwa {d}, {a}, {b} dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD", "v32"),
    ("v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 {d}, {b}", "v32"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {d}, {b}", "v32"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 {d}, {b}", "v32"),
    ("v_add_f32", "v_add_f32 {d}, {a}, {b}", "v32"),
    ("v_mul_f32", "v_mul_f32 {d}, {a}, {b}", "v32"),
    ("v_fma_f32", "v_fma_f32 {d}, {a}, {b}, {c}", "v32"),
    ("v_rcp_f32", "v_rcp_f32 {d}, {b}", "v32"),
    ("v_readlane_b32", "v_readlane_b32 s20, {b}, 5", "v32"),
    ("v_cndmask_b32_sgpr", "v_cndmask_b32 {d}, {a}, {b}, s[22:23]", "v32"),
    ("v_max_u16", "v_max_u16 {d}, {a}, {b}", "v32"),
    ("v_min_i16", "v_min_i16 {d}, {a}, {b}", "v32"),
    ("v_max_i16", "v_max_i16 {d}, {a}, {b}", "v32"),
    ("v_sub_u16", "v_sub_u16 {d}, {a}, {b}", "v32"),
    ("v_mul_lo_u16", "v_mul_lo_u16 {d}, {a}, {b}", "v32"),
    ("v_lshlrev_b16", "v_lshlrev_b16 {d}, 3, {b}", "v32"),
    ("v_lshrrev_b16", "v_lshrrev_b16 {d}, 3, {b}", "v32"),
    ("v_mad_u16", "v_mad_u16 {d}, {a}, {b}, {c}", "v32"),
    ("v_min_f32", "v_min_f32 {d}, {a}, {b}", "v32"),
    ("v_max_f32", "v_max_f32 {d}, {a}, {b}", "v32"),
    ("v_sub_f32", "v_sub_f32 {d}, {a}, {b}", "v32"),
    ("v_fmac_f32", "v_fmac_f32 {d}, {b}, {c}", "v32"),
    ("v_add_f16", "v_add_f16 {d}, {a}, {b}", "v32"),
    ("v_min_f16", "v_min_f16 {d}, {a}, {b}", "v32"),
    ("v_max_f16", "v_max_f16 {d}, {a}, {b}", "v32"),
    ("v_pk_min_f16", "v_pk_min_f16 {d}, {a}, {b}", "v32"),
    ("v_pk_add_f16", "v_pk_add_f16 {d}, {a}, {b}", "v32"),
    ("v_add_co_u32", "v_add_co_u32 {d}, vcc, {a}, {b}", "v32"),
    ("v_addc_co_u32", "v_addc_co_u32 {d}, vcc, {a}, {b}, vcc", "v32"),
    ("v_subrev_u32", "v_subrev_u32 {d}, {a}, {b}", "v32"),
    ("v_xnor_b32", "v_xnor_b32 {d}, {a}, {b}", "v32"),
    ("v_and_b32_sgpr", "v_and_b32 {d}, s22, {b}", "v32"),
    ("v_add_u32_inline", "v_add_u32 {d}, 17, {b}", "v32"),
    ("v_cmp_lt_f32_vcc", "v_cmp_lt_f32 vcc, {a}, {b}", "v32"),
    ("v_cmp_lt_u16_vcc", "v_cmp_lt_u16 vcc, {a}, {b}", "v32"),
    ("v_rndne_f32", "v_rndne_f32 {d}, {b}", "v32"),
    ("v_cvt_i32_f32", "v_cvt_i32_f32 {d}, {b}", "v32"),
    ("v_lshl_add_u64", "v_lshl_add_u64 {d}, {a}, 2, {b}", "v64"),
    ("v_mov_b64", "v_mov_b64 {d}, {b}", "v64"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 s20, {b}", "v32"),
    ("v_pk_fma_f32", "v_pk_fma_f32 {d}, {a}, {b}, {c}", "v64"),
    ("v_pk_add_f32", "v_pk_add_f32 {d}, {a}, {b}", "v64"),
    ("v_fma_f64", "v_fma_f64 {d}, {a}, {b}, {c}", "v64"),
    ("v_add_f64", "v_add_f64 {d}, {a}, {b}", "v64"),
    ("v_mul_f64", "v_mul_f64 {d}, {a}, {b}", "v64"),
    ("v_lshlrev_b64", "v_lshlrev_b64 {d}, 3, {b}", "v64"),
    ("v_mad_u64_u32", "v_mad_u64_u32 {d}, vcc, {a_lo}, {b_lo}, {c}", "v64x"),
    ("s_nop_0", "s_nop 0", "v32"),
]

HEAD = r'''// GENERATED by gen_valu_class.py - do not edit.  See the generator for what is measured.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#define REPS 64
__device__ __forceinline__ uint64_t ticks() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
'''

TAIL = r'''
struct Entry { const char* name; void (*fn)(uint64_t*, uint32_t*, int); };
static Entry entries[] = {
%s
};
int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : nullptr;
  const int iters = 2048;
  uint64_t* d_t; uint32_t* d_v;
  const int maxw = 256 * 8 * 4;
  hipMalloc(&d_t, sizeof(uint64_t) * maxw); hipMalloc(&d_v, sizeof(uint32_t) * maxw * 64);
  std::vector<uint64_t> h(maxw);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Entry& e : entries) {
    if (only && !strstr(e.name, only)) continue;
    for (int W : {1, 2, 4, 8}) {
      const int blocks = 256 * W;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_t, d_v, 8);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_t, d_v, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const int nw = blocks * 4;
      hipMemcpy(h.data(), d_t, sizeof(uint64_t) * nw, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.begin() + nw);
      const double ninstr = (double)iters * REPS;
      const double med = (double)h[nw / 2], mx = (double)h[nw - 1];
      const double clock_ghz = mx / (ms * 1e6);   // the slowest wave spans (nearly) the whole launch
      const double cyc_simd = mx * 1024.0 / (ninstr * nw);
      printf("{\"instr\": \"%%s\", \"waves_per_simd\": %%d, \"cyc_per_instr_simd\": %%.3f, \"cyc_per_instr_wave\": %%.3f, "
             "\"launch_ms\": %%.4f, \"clock_ghz_est\": %%.3f}\n", e.name, W, cyc_simd, med / ninstr, ms, clock_ghz);
      fflush(stdout);
    }
  }
  return 0;
}
'''


def body(tmpl, kind):
    lines = []
    for r in range(8):
        for i in range(8):
            a, b, c = i, (i + 3) & 7, (i + 5) & 7
            if kind == "v64x":
                s = tmpl.format(d=f"%{i}", a=f"%{a}", b=f"%{b}", c=f"%{c}", a_lo=f"%{8 + a}", b_lo=f"%{8 + b}")
            else:
                s = tmpl.format(d=f"%{i}", a=f"%{a}", b=f"%{b}", c=f"%{c}")
            lines.append(s)
    return "\\n\\t".join(lines)


def kernel(name, tmpl, kind):
    wide = kind in ("v64", "v64x")
    ty = "uint64_t" if wide else "uint32_t"
    out = [f"__global__ __launch_bounds__(256) void k_{name}(uint64_t* tout, uint32_t* vout, int iters) {{"]
    out.append(f"  {ty} a[8];")
    out.append("  for (int i = 0; i < 8; ++i) a[i] = (" + ty + ")(threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) | 0x3f80000000010001ull;" if wide else
               "  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;")
    if kind == "v64x":
        out.append("  uint32_t lo[8]; for (int i = 0; i < 8; ++i) lo[i] = threadIdx.x + i;")
    out.append("  asm volatile(\"v_cmp_lt_u32 vcc, %0, %1\\n\\ts_mov_b64 s[22:23], vcc\" :: \"v\"(a[0]), \"v\"(a[1]) : \"vcc\", \"s22\", \"s23\");" if not wide else "")
    out.append("  const uint64_t t0 = ticks();")
    out.append("  for (int it = 0; it < iters; ++it) {")
    ops = ", ".join(f"\"+v\"(a[{i}])" for i in range(8))
    ins = ""
    if kind == "v64x":
        ins = ", ".join(f"\"v\"(lo[{i}])" for i in range(8))
    out.append(f"    asm volatile(\"{body(tmpl, kind)}\" : {ops} : {ins} : \"vcc\", \"s20\", \"s21\");")
    out.append("  }")
    out.append("  const uint64_t t1 = ticks();")
    out.append("  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= (uint32_t)a[i] ^ (uint32_t)((uint64_t)a[i] >> 32);")
    out.append("  vout[blockIdx.x * 256 + threadIdx.x] = s;")
    out.append("  if ((threadIdx.x & 63) == 0) tout[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;")
    out.append("}")
    return "\n".join(out)


def main():
    print(HEAD)
    for name, tmpl, kind in INSTR:
        print(kernel(name, tmpl, kind))
    print(TAIL % ",\n".join(f'  {{"{n}", k_{n}}}' for n, _, _ in INSTR))


if __name__ == "__main__":
    main()
