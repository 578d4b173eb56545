// svo_wave.h - wave64 reductions on DPP (no LDS crossbar).
// A __shfl_xor butterfly lowers to six dependent ds_bpermute_b32 (~500 cycles per reduction on
// gfx950); the DPP form below is four row-local v_mov_dpp + op steps and four v_readlane, ~100
// cycles, which is what the serial chains (greedy matching, LM, RANSAC refit) are made of.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SVO_DPP(v, ctrl) __builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), 0xf, 0xf, false)
#define SVO_DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define SVO_DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define SVO_DPP_HALF_MIRROR 0x141 // reverse within 8 lanes
#define SVO_DPP_MIRROR 0x140      // reverse within the 16-lane row

// min over the wave, result in every lane
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
  v = min(v, (uint32_t)SVO_DPP(v, SVO_DPP_XOR1));
  v = min(v, (uint32_t)SVO_DPP(v, SVO_DPP_XOR2));
  v = min(v, (uint32_t)SVO_DPP(v, SVO_DPP_HALF_MIRROR));
  v = min(v, (uint32_t)SVO_DPP(v, SVO_DPP_MIRROR));
  const uint32_t a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16),
                 c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

__device__ __forceinline__ int wave_sum_i32_dpp(int v) {
  v += SVO_DPP(v, SVO_DPP_XOR1);
  v += SVO_DPP(v, SVO_DPP_XOR2);
  v += SVO_DPP(v, SVO_DPP_HALF_MIRROR);
  v += SVO_DPP(v, SVO_DPP_MIRROR);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
         (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

template <int CTRL>
__device__ __forceinline__ double svo_dpp_f64(double v) {
  const int lo = SVO_DPP(__double2loint(v), CTRL), hi = SVO_DPP(__double2hiint(v), CTRL);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double svo_readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}
// sum over the wave, result in every lane (fixed association: rows first, then (r0+r1)+(r2+r3))
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
  v += svo_dpp_f64<SVO_DPP_XOR1>(v);
  v += svo_dpp_f64<SVO_DPP_XOR2>(v);
  v += svo_dpp_f64<SVO_DPP_HALF_MIRROR>(v);
  v += svo_dpp_f64<SVO_DPP_MIRROR>(v);
  return (svo_readlane_f64(v, 0) + svo_readlane_f64(v, 16)) + (svo_readlane_f64(v, 32) + svo_readlane_f64(v, 48));
}
