// svo_orb.hip - ORB extraction for gfx950: pyramid, FAST-9/16 + NMS, Harris top-k,
// intensity-centroid orientation, on-demand 7x7 Gaussian + rBRIEF.
//
// Replaces frame::featuredetect (reference src/frame.cc:75-79), i.e. the
// cv::ORB::create()->detectAndCompute call with default parameters; stage
// semantics are the ones listed in SURVEY.md section 8 a-notes and DESIGN.md.
//
// Data layout in HBM (per image slot): level 0 is read in place from the caller's
// gray image; levels 1..7 live in one pyramid slot with 64-byte-aligned row
// pitches; FAST corners are appended as packed 32-bit words x | y<<12 | score<<24
// into a per-(image, level) segment sized for the NMS density bound (1/4 of the
// pixels), so no overflow path exists; a 256-bin score histogram per
// (image, level) gives the retainBest threshold without sorting.
#include "svo_internal.h"
#include "svo_wave.h"
#include "../../include/svo_brief_pattern.h"

__constant__ __attribute__((aligned(16))) int8_t c_pattern[SVO_BRIEF_NTESTS][4] = SVO_BRIEF_PATTERN_INIT;
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

struct ImgSrc {
  const uint8_t* L;
  const uint8_t* R;
  int stride;
  int B;
  const uint8_t* pyr;
};

__device__ __forceinline__ const uint8_t* level_ptr(const SvoGeom& g, const ImgSrc& s, int img,
                                                     int l, int* pitch) {
  if (l == 0) {
    *pitch = s.stride;
    return img < s.B ? s.L + (size_t)img * s.stride * g.H
                     : s.R + (size_t)(img - s.B) * s.stride * g.H;
  }
  *pitch = g.pitch[l];
  return s.pyr + (size_t)img * g.pyr_bytes + g.loff[l];
}

// ---------------------------------------------------------------------------------
// Pyramid: level l from level l-1, cv::resize INTER_LINEAR fixed-point semantics
// (11-bit coefficients; vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2).
// One thread = 4 output pixels = one dword store; rows are coalesced.
// ---------------------------------------------------------------------------------
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_pyr_level(SvoGeom g, ImgSrc s, int l, uint8_t* pyr,
                                                   const int32_t* __restrict__ xofs,
                                                   const int32_t* __restrict__ xalpha,
                                                   const int32_t* __restrict__ yofs,
                                                   const int32_t* __restrict__ ybeta) {
  // One thread = a 4 (wide) x 4 (tall) block of output pixels: the x coefficients are fetched once
  // for four rows and all source rows of the block are requested before any is used, so the
  // table -> source dependency is paid once per 16 pixels.
  const int img = blockIdx.z;
  const int dy0 = (blockIdx.y * 4 + threadIdx.y) * 4;
  const int dx4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int dw = g.w[l], dh = g.h[l];
  if (dy0 >= dh || dx4 >= dw) return;
  int sp;
  const uint8_t* src = level_ptr(g, s, img, l - 1, &sp);
  const int sw = g.w[l - 1], sh = g.h[l - 1];
  const int4 so = *reinterpret_cast<const int4*>(xofs + g.xtab_off[l] + dx4);
  const int4 sa = *reinterpret_cast<const int4*>(xalpha + g.xtab_off[l] + dx4);
  int syv[4], bbv[4];
  const int dy0u = __builtin_amdgcn_readfirstlane(dy0);   // a wave is one row of the (64, 4) block: scalar table loads
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int dy = min(dy0u + r, dh - 1);
    syv[r] = yofs[g.ytab_off[l] + dy];
    bbv[r] = ybeta[g.ytab_off[l] + dy];
  }
  const int sx[4] = {so.x, so.y, so.z, so.w};
  const int al[4] = {sa.x, sa.y, sa.z, sa.w};
  const int base = sx[0];
  const bool wide = base + 8 <= sw;
  uint64_t w0[4], w1[4];
  // 4-byte aligned rows (every pyramid level; source images with a stride that is a multiple of 4): three aligned
  // dwords per row and two v_alignbit instead of one byte-aligned 8-byte load - the memory pipeline handles the
  // aligned form several times faster
  struct __attribute__((aligned(4))) u96 { uint32_t a, b, c; };
  const bool rows_aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)sp) & 3) == 0;
  const int base_al = base & ~3;
  const uint32_t shb = (uint32_t)(base & 3) * 8u;
  if (wide && rows_aligned && base_al + 12 <= sw) {
    u96 q0[4], q1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      q0[r] = *reinterpret_cast<const u96*>(src + (size_t)syv[r] * sp + base_al);
      q1[r] = *reinterpret_cast<const u96*>(src + (size_t)min(syv[r] + 1, sh - 1) * sp + base_al);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w0[r] = (uint64_t)__builtin_amdgcn_alignbit(q0[r].b, q0[r].a, shb) | ((uint64_t)__builtin_amdgcn_alignbit(q0[r].c, q0[r].b, shb) << 32);
      w1[r] = (uint64_t)__builtin_amdgcn_alignbit(q1[r].b, q1[r].a, shb) | ((uint64_t)__builtin_amdgcn_alignbit(q1[r].c, q1[r].b, shb) << 32);
    }
  } else if (wide) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w0[r] = *reinterpret_cast<const u64_unaligned*>(src + (size_t)syv[r] * sp + base);
      w1[r] = *reinterpret_cast<const u64_unaligned*>(src + (size_t)min(syv[r] + 1, sh - 1) * sp + base);
    }
  }
  uint8_t* dst = pyr + (size_t)img * g.pyr_bytes + g.loff[l] + (size_t)dy0 * g.pitch[l] + dx4;
  // the two horizontal taps of output column k are adjacent bytes off, off+1 of the 8-byte window:
  // one v_perm with a per-column selector spreads them into u16 lanes (shared by all rows), and
  // v_dot2_u32_u16 applies the packed (a0, a1) pair
  uint32_t sel[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sel[k] = 0x0c010c00u + (uint32_t)(sx[k] - base) * 0x00010001u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (dy0 + r >= dh) break;
    const int b0 = (int)(int16_t)(bbv[r] & 0xffff), b1 = bbv[r] >> 16;
    // (b * (S >> 4)) >> 16 as the high half of a 24 x 24-bit product: (b << 12) * (S & ~15) >> 32 (0 <= b <= 2048, S < 2^19)
    const uint32_t B0 = (uint32_t)b0 << 12, B1 = (uint32_t)b1 << 12;
    auto vterm = [](uint32_t Bs, uint32_t S) { return (uint32_t)(((uint64_t)(Bs & 0xffffffu) * ((S & ~15u) & 0xffffffu)) >> 32); };
    uint32_t out = 0;
    if (wide) {
      const uint32_t lo0 = (uint32_t)w0[r], hi0 = (uint32_t)(w0[r] >> 32);
      const uint32_t lo1 = (uint32_t)w1[r], hi1 = (uint32_t)(w1[r] >> 32);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const v2u16 p0 = __builtin_bit_cast(v2u16, __builtin_amdgcn_perm(hi0, lo0, sel[k]));
        const v2u16 p1 = __builtin_bit_cast(v2u16, __builtin_amdgcn_perm(hi1, lo1, sel[k]));
        const v2u16 al2 = __builtin_bit_cast(v2u16, (uint32_t)al[k]);
        const uint32_t S0 = __builtin_amdgcn_udot2(p0, al2, 0u, false);
        const uint32_t S1 = __builtin_amdgcn_udot2(p1, al2, 0u, false);
        // weights are non-negative and sum to 2048 in both directions: the result cannot leave 0..255
        const uint32_t v = (vterm(B0, S0) + vterm(B1, S1) + 2) >> 2;
        out |= (uint32_t)v << (8 * k);
      }
    } else {
      const uint8_t* r0 = src + (size_t)syv[r] * sp;
      const uint8_t* r1 = src + (size_t)min(syv[r] + 1, sh - 1) * sp;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (dx4 + k < dw) {
          const int sx0 = sx[k], sx1 = min(sx0 + 1, sw - 1);
          const int a0 = (int)(int16_t)(al[k] & 0xffff), a1 = al[k] >> 16;
          const int S0 = r0[sx0] * a0 + r0[sx1] * a1;
          const int S1 = r1[sx0] * a0 + r1[sx1] * a1;
          int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
          v = min(max(v, 0), 255);
          out |= (uint32_t)v << (8 * k);
        }
      }
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)r * g.pitch[l]) = out;
  }
}

// ---------------------------------------------------------------------------------
// Two or three consecutive levels per launch.  One workgroup = (image, strip of rows of the LAST of its levels, full
// width): it computes the rows of the first level that strip depends on from the stored level below (HBM), keeps them in
// LDS - and stores them: they are part of the pyramid - then the next level from LDS, and so on.  Neighbouring strips
// recompute the one or two rows they share (identical values, stored twice).  Against one launch per level: every level
// but the first is read from LDS instead of HBM / L2, and the seven dependent launches of a batch - the last of them a
// few waves per CU - become three.  Arithmetic: k_pyr_level's, bit for bit.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pyr_four(uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1, const int (&sx)[4],
                                             const int (&al)[4], int base, int bb) {
  const int b0 = (int)(int16_t)(bb & 0xffff), b1 = bb >> 16;
  const uint32_t B0 = (uint32_t)b0 << 12, B1 = (uint32_t)b1 << 12;
  auto vterm = [](uint32_t Bs, uint32_t S) { return (uint32_t)(((uint64_t)(Bs & 0xffffffu) * ((S & ~15u) & 0xffffffu)) >> 32); };
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t sel = 0x0c010c00u + (uint32_t)(sx[k] - base) * 0x00010001u;
    const v2u16 p0 = __builtin_bit_cast(v2u16, __builtin_amdgcn_perm(hi0, lo0, sel));
    const v2u16 p1 = __builtin_bit_cast(v2u16, __builtin_amdgcn_perm(hi1, lo1, sel));
    const v2u16 al2 = __builtin_bit_cast(v2u16, (uint32_t)al[k]);
    const uint32_t S0 = __builtin_amdgcn_udot2(p0, al2, 0u, false);
    const uint32_t S1 = __builtin_amdgcn_udot2(p1, al2, 0u, false);
    out |= ((vterm(B0, S0) + vterm(B1, S1) + 2) >> 2) << (8 * k);
  }
  return out;
}

template <int NL>
__global__ __launch_bounds__(256) void k_pyr_fused(SvoGeom g, ImgSrc s, int l0, int RT, int cap0, int cap1, uint8_t* pyr,
                                                   const int32_t* __restrict__ xofs, const int32_t* __restrict__ xalpha,
                                                   const int32_t* __restrict__ yofs, const int32_t* __restrict__ ybeta) {
  extern __shared__ uint32_t pyr_lds[];   // rows of level l0 (cap0 rows of pitch[l0]), then (NL == 3) of level l0 + 1; 16 bytes of slack each
  const int img = blockIdx.y, lt = l0 + NL - 1;
  const int r0 = blockIdx.x * RT, r1 = min(g.h[lt], r0 + RT);
  if (r0 >= g.h[lt]) return;
  int lo[NL], hi[NL];
  lo[NL - 1] = r0; hi[NL - 1] = r1;
#pragma unroll
  for (int k = NL - 2; k >= 0; --k) {   // rows of level l0 + k that rows lo .. hi - 1 of level l0 + k + 1 read
    const int child = l0 + k + 1;
    lo[k] = lo[k + 1] == 0 ? 0 : yofs[g.ytab_off[child] + lo[k + 1]];
    hi[k] = hi[k + 1] == g.h[child] ? g.h[l0 + k] : min(yofs[g.ytab_off[child] + hi[k + 1] - 1] + 2, g.h[l0 + k]);
  }
  uint32_t* buf[2] = {pyr_lds, pyr_lds + (size_t)cap0 * (g.pitch[l0] >> 2) + 4};
  (void)cap1;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int l = l0 + k, dw = g.w[l], sw = g.w[l - 1], sh = g.h[l - 1], ncol = (dw + 3) >> 2;
    const int rows = hi[k] - lo[k], nrb = (rows + 3) >> 2, dpw = g.pitch[l] >> 2;
    uint8_t* dst = pyr + (size_t)img * g.pyr_bytes + g.loff[l];
    int sp = 0;
    const uint8_t* src = k == 0 ? level_ptr(g, s, img, l - 1, &sp) : nullptr;
    const int spw = k > 0 ? g.pitch[l - 1] >> 2 : 0;
    const uint32_t* lsrc = k > 0 ? buf[k - 1] : nullptr;
    const bool rows_aligned = k == 0 && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)sp) & 3) == 0;
    // one item = 4 (wide) x 4 (tall) output pixels, as in k_pyr_level: the x coefficients are fetched once for four rows
    // and all eight source rows are requested before any is used
    for (int idx = threadIdx.x; idx < nrb * ncol; idx += 256) {
      const int rb = idx / ncol, c4 = idx - rb * ncol, dx4 = 4 * c4, ry0 = 4 * rb;
      const int4 so = *reinterpret_cast<const int4*>(xofs + g.xtab_off[l] + dx4);
      const int4 sa = *reinterpret_cast<const int4*>(xalpha + g.xtab_off[l] + dx4);
      int syv[4], bbv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int y = min(lo[k] + ry0 + r, hi[k] - 1);
        syv[r] = yofs[g.ytab_off[l] + y];
        bbv[r] = ybeta[g.ytab_off[l] + y];
      }
      const int sx[4] = {so.x, so.y, so.z, so.w};
      const int al[4] = {sa.x, sa.y, sa.z, sa.w};
      const int base = sx[0], base_al = base & ~3;
      const uint32_t shb = (uint32_t)(base & 3) * 8u;
      uint32_t out[4] = {0, 0, 0, 0};
      if (k > 0) {
        // from LDS: three aligned words per source row (reads past the row's width land in its padding / the next row and
        // are never selected: a tap beyond the last column only occurs with weight 0)
        uint32_t a[4][3], b[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t* q0 = lsrc + (size_t)(syv[r] - lo[k - 1]) * spw + (base_al >> 2);
          const uint32_t* q1 = lsrc + (size_t)(min(syv[r] + 1, sh - 1) - lo[k - 1]) * spw + (base_al >> 2);
          a[r][0] = q0[0]; a[r][1] = q0[1]; a[r][2] = q0[2]; b[r][0] = q1[0]; b[r][1] = q1[1]; b[r][2] = q1[2];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[r] = pyr_four(__builtin_amdgcn_alignbit(a[r][1], a[r][0], shb), __builtin_amdgcn_alignbit(a[r][2], a[r][1], shb),
                            __builtin_amdgcn_alignbit(b[r][1], b[r][0], shb), __builtin_amdgcn_alignbit(b[r][2], b[r][1], shb), sx, al,
                            base, bbv[r]);
      } else {
        struct __attribute__((aligned(4))) u96 { uint32_t a, b, c; };
        if (rows_aligned && base_al + 12 <= sw) {
          u96 q0[4], q1[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            q0[r] = *reinterpret_cast<const u96*>(src + (size_t)syv[r] * sp + base_al);
            q1[r] = *reinterpret_cast<const u96*>(src + (size_t)min(syv[r] + 1, sh - 1) * sp + base_al);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[r] = pyr_four(__builtin_amdgcn_alignbit(q0[r].b, q0[r].a, shb), __builtin_amdgcn_alignbit(q0[r].c, q0[r].b, shb),
                              __builtin_amdgcn_alignbit(q1[r].b, q1[r].a, shb), __builtin_amdgcn_alignbit(q1[r].c, q1[r].b, shb), sx, al,
                              base, bbv[r]);
        } else if (base + 8 <= sw) {
          uint64_t w0[4], w1[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            w0[r] = *reinterpret_cast<const u64_unaligned*>(src + (size_t)syv[r] * sp + base);
            w1[r] = *reinterpret_cast<const u64_unaligned*>(src + (size_t)min(syv[r] + 1, sh - 1) * sp + base);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[r] = pyr_four((uint32_t)w0[r], (uint32_t)(w0[r] >> 32), (uint32_t)w1[r], (uint32_t)(w1[r] >> 32), sx, al, base, bbv[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint8_t* p0 = src + (size_t)syv[r] * sp;
            const uint8_t* p1 = src + (size_t)min(syv[r] + 1, sh - 1) * sp;
            const int b0 = (int)(int16_t)(bbv[r] & 0xffff), b1 = bbv[r] >> 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (dx4 + q < dw) {
                const int sx0 = sx[q], sx1 = min(sx0 + 1, sw - 1);
                const int a0 = (int)(int16_t)(al[q] & 0xffff), a1 = al[q] >> 16;
                const int S0 = p0[sx0] * a0 + p0[sx1] * a1;
                const int S1 = p1[sx0] * a0 + p1[sx1] * a1;
                int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                v = min(max(v, 0), 255);
                out[r] |= (uint32_t)v << (8 * q);
              }
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (ry0 + r >= rows) break;
        *reinterpret_cast<uint32_t*>(dst + (size_t)(lo[k] + ry0 + r) * g.pitch[l] + dx4) = out[r];
        if (k < NL - 1) buf[k][(size_t)(ry0 + r) * dpw + c4] = out[r];
      }
    }
    if (k < NL - 1) __syncthreads();
  }
}

// Corner score of TWO horizontally adjacent pixels at once in packed 16-bit lanes
// (v_pk_sub/min/max_i16): d[i] = centre - ring[i]; score = max over the 16 arcs of 9 of
// min(d) (dark ring) and of min(-d) (bright ring), minus 1; 0 unless that exceeds the
// threshold.  Branch-free - a 64-wide wave never diverges on the rare corner pixels.  The
// ring pixel pairs are cut out of the staged 16-byte row windows with v_perm_b32.
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }
template <int B>
__device__ __forceinline__ v2s pix_pair(const uint32_t* row) {  // bytes B, B+1 -> two i16 lanes
  constexpr int w = B >> 2, k = B & 3;
  constexpr uint32_t sel = 0x0c000c00u | ((uint32_t)(k + 1) << 16) | (uint32_t)k;
  return as_v2s(__builtin_amdgcn_perm(row[w + 1 > 3 ? 3 : w + 1], row[w], sel));
}
#define VMIN(a, b) __builtin_elementwise_min(a, b)
#define VMAX(a, b) __builtin_elementwise_max(a, b)
template <int K>
__device__ __forceinline__ uint32_t fast_pair(const uint32_t (&rw)[7][4]) {
  constexpr int cc = 4 + K;  // byte column of the first centre inside the 16-byte window
  const v2s v = pix_pair<cc>(rw[3]);
  v2s d[16];
  d[0] = v - pix_pair<cc>(rw[6]);      d[1] = v - pix_pair<cc + 1>(rw[6]);  d[2] = v - pix_pair<cc + 2>(rw[5]);
  d[3] = v - pix_pair<cc + 3>(rw[4]);  d[4] = v - pix_pair<cc + 3>(rw[3]);  d[5] = v - pix_pair<cc + 3>(rw[2]);
  d[6] = v - pix_pair<cc + 2>(rw[1]);  d[7] = v - pix_pair<cc + 1>(rw[0]);  d[8] = v - pix_pair<cc>(rw[0]);
  d[9] = v - pix_pair<cc - 1>(rw[0]);  d[10] = v - pix_pair<cc - 2>(rw[1]); d[11] = v - pix_pair<cc - 3>(rw[2]);
  d[12] = v - pix_pair<cc - 3>(rw[3]); d[13] = v - pix_pair<cc - 3>(rw[4]); d[14] = v - pix_pair<cc - 2>(rw[5]);
  d[15] = v - pix_pair<cc - 1>(rw[6]);
  v2s a1[16], b1[16], a2[16], b2[16], a9[16], b9[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a1[i] = VMIN(d[i], d[(i + 1) & 15]); b1[i] = VMAX(d[i], d[(i + 1) & 15]); }
#pragma unroll
  for (int i = 0; i < 16; ++i) { a2[i] = VMIN(a1[i], a1[(i + 2) & 15]); b2[i] = VMAX(b1[i], b1[(i + 2) & 15]); }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a9[i] = VMIN(VMIN(a2[i], a2[(i + 4) & 15]), d[(i + 8) & 15]);
    b9[i] = VMAX(VMAX(b2[i], b2[(i + 4) & 15]), d[(i + 8) & 15]);
  }
#pragma unroll
  for (int s = 8; s > 0; s >>= 1)   // balanced trees: dependent v_pk ops need a wait state
#pragma unroll
    for (int i = 0; i < s; ++i) { a9[i] = VMAX(a9[i], a9[i + s]); b9[i] = VMIN(b9[i], b9[i + s]); }
  const v2s zero = {0, 0}, cT = {SVO_FAST_THR, SVO_FAST_THR}, one = {1, 1},
            cTm1 = {SVO_FAST_THR - 1, SVO_FAST_THR - 1};
  const v2s S = VMAX(a9[0], zero - b9[0]);
  const v2s u = VMAX(S, cT) - cT;             // S - T if S > T else 0
  const v2s r = u + VMIN(u, one) * cTm1;      // S - 1 if S > T else 0
  const uint32_t x = as_u32(r);
  return (x & 0xffu) | ((x >> 8) & 0xff00u);  // two score bytes
}

// Necessary condition for a 9-arc, from the 4 compass ring pixels only: an arc of 9 covers at
// least two ADJACENT compass points (0,4,8,12), so two adjacent ones must both be darker (or both
// brighter) than the centre by more than the threshold.  Returns nonzero if either pixel of the
// pair can still be a corner.
template <int K>
__device__ __forceinline__ uint32_t compass_pair(const uint32_t (&r0)[4], const uint32_t (&r3)[4],
                                                 const uint32_t (&r6)[4]) {
  constexpr int cc = 4 + K;
  const v2s v = pix_pair<cc>(r3);
  const v2s d0 = v - pix_pair<cc>(r6), d8 = v - pix_pair<cc>(r0);
  const v2s d4 = v - pix_pair<cc + 3>(r3), d12 = v - pix_pair<cc - 3>(r3);
  const v2s A = VMAX(VMAX(VMIN(d0, d4), VMIN(d4, d8)), VMAX(VMIN(d8, d12), VMIN(d12, d0)));
  const v2s B = VMIN(VMIN(VMAX(d0, d4), VMAX(d4, d8)), VMIN(VMAX(d8, d12), VMAX(d12, d0)));
  const v2s zero = {0, 0}, cT1 = {SVO_FAST_THR + 1, SVO_FAST_THR + 1};
  const uint32_t x = as_u32(VMAX(A, zero - B) - cT1);   // lane >= 0  <=>  possible corner
  return ~x & 0x80008000u;
}

#define FAST_CAND_CAP 2048
#define PXW 34  // dwords per staged pixel row (136 bytes)
#define SCW 128 // bytes per score row
#define SCR (FAST_TH + 2)   // score rows: the tile + one ring for NMS
#define PXR (FAST_TH + 8)   // staged pixel rows: score rows + 3 above and below

__global__ __launch_bounds__(256) void k_fast(SvoGeom g, ImgSrc s, uint32_t* corners,
                                              int32_t* counters, int32_t* hist, int cand_cap) {
  __shared__ uint32_t px[PXR * PXW];
  __shared__ uint32_t sc[SCR * (SCW / 4)];
  __shared__ int lhist[256];
  // 18.5 KB of LDS per workgroup: 8 workgroups (8 waves per SIMD) fit one CU, which is what hides the staging loads
  // of one tile behind the arithmetic of the others (6 per CU: +12 % on the kernel, 4: +45 %)
  __shared__ uint32_t qbuf[SCR * 32];       // phases A-B: the queue; phase C on: lcorn (the queue is dead)
  uint16_t* queue = reinterpret_cast<uint16_t*>(qbuf);   // surviving pixel pairs: id = r*64 + pc
  uint32_t* lcorn = qbuf;                   // FAST_TW * FAST_TH / 4 entries at most (NMS density bound)
  static_assert(FAST_TW * FAST_TH / 4 + 64 <= SCR * 32, "lcorn must fit the queue buffer");
  __shared__ uint16_t cand[FAST_CAND_CAP];  // pixels with a nonzero score: id = r*128 + col (overflow: phase C scans sc)
  __shared__ int nq, ncand, lcount, lbase;

  const int tid = threadIdx.x, lane = tid & 63;
  // XCD-aware tile order: workgroup b is observed to run on XCD b % 8 and each XCD has a private
  // L2, so the linear block id is remapped (bijectively) to give every XCD one CONTIGUOUS eighth of
  // the (image, tile) sequence - vertically adjacent tiles then share their halo rows in one L2
  // instead of each XCD fetching its own copy.  Placement only affects speed, never results.
  int img, tile;
  {
    const uint32_t total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t xcd = lin & 7u, idx = lin >> 3, q = total >> 3, r = total & 7u;
    const uint32_t mapped = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    img = (int)(mapped / gridDim.x);
    tile = (int)(mapped - (uint32_t)img * gridDim.x);
  }
  int l = 0;
#pragma unroll
  for (int k = 1; k < SVO_NLEVELS; ++k)
    if (tile >= g.tile_base[k]) l = k;
  tile -= g.tile_base[l];
  const int ty = tile / g.tiles_x[l], tx = tile - ty * g.tiles_x[l];
  const int x0 = 28 + tx * FAST_TW, y0 = SVO_EDGE + ty * FAST_TH;
  const int w = g.w[l], h = g.h[l];
  int pitch;
  const uint8_t* img_p = level_ptr(g, s, img, l, &pitch);

  lhist[tid] = 0;
  for (int i = tid; i < SCR * (SCW / 4); i += 256) sc[i] = 0;
  if (tid == 0) { lcount = 0; nq = 0; ncand = 0; }

  // stage pixels: rows y0-4 .. y0+FAST_TH+3, columns x0-8 .. x0+127
  {
    // Every thread issues all of its (unconditional, possibly unaligned) dword loads before the first LDS store: one
    // memory round trip per tile instead of six.  A thread keeps one dword column (7 row groups x 34 columns = 238
    // threads) and walks down the rows, so everything that depends on the column is computed once: the column
    // clamped to the last full dword of the row and the v_perm selector that rebuilds a dword hanging over the right
    // image border from it (replicated last pixel; identity elsewhere).
    typedef uint32_t __attribute__((aligned(1))) u32u;
    constexpr int NROWG = 256 / PXW, NSLOT = (PXR + NROWG - 1) / NROWG;
    const int rg = tid / PXW, c = tid - rg * PXW;   // row group, dword column
    const int over = x0 - 8 + 4 * c - (w - 4);   // > 0: this dword starts `over` bytes right of the loaded one
    uint32_t fix = 0x03020100u;
    if (over > 0) fix = (uint32_t)min(over, 3) | ((uint32_t)min(over + 1, 3) << 8) | ((uint32_t)min(over + 2, 3) << 16) | (3u << 24);
    const uint8_t* col = img_p + min(x0 - 8 + 4 * c, w - 4);
    uint32_t v[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int gy = min(max(y0 - 4 + rg + NROWG * k, 0), h - 1);
      v[k] = *reinterpret_cast<const u32u*>(col + (size_t)gy * pitch);
    }
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int r = rg + NROWG * k;
      if (rg < NROWG && r < PXR) px[r * PXW + c] = __builtin_amdgcn_perm(v[k], v[k], fix);
    }
  }
  __syncthreads();

  // phase A: compass pre-test of 4 pixel pairs per thread; score row r <-> y = y0-1+r, pair
  // column pc = 4c + K/2 <-> x = x0-4+2pc.  Survivors are compacted per wave with a ballot.
  for (int r = tid >> 4; r < SCR; r += 16) {
    const int c = tid & 15;
    uint32_t r0[4], r3[4], r6[4];
    {
      const uint2 a = *reinterpret_cast<const uint2*>(&px[(r + 0) * PXW + 2 * c]);
      const uint2 b = *reinterpret_cast<const uint2*>(&px[(r + 0) * PXW + 2 * c + 2]);
      r0[0] = a.x; r0[1] = a.y; r0[2] = b.x; r0[3] = b.y;
      const uint2 e = *reinterpret_cast<const uint2*>(&px[(r + 3) * PXW + 2 * c]);
      const uint2 f = *reinterpret_cast<const uint2*>(&px[(r + 3) * PXW + 2 * c + 2]);
      r3[0] = e.x; r3[1] = e.y; r3[2] = f.x; r3[3] = f.y;
      const uint2 p = *reinterpret_cast<const uint2*>(&px[(r + 6) * PXW + 2 * c]);
      const uint2 q = *reinterpret_cast<const uint2*>(&px[(r + 6) * PXW + 2 * c + 2]);
      r6[0] = p.x; r6[1] = p.y; r6[2] = q.x; r6[3] = q.y;
    }
    // Compass pre-test on 4 pixels per 32-bit operation: pixels quantised to 6 bits (x >> 2); c - n >= 21 implies
    // cq - nq >= 5, which is bit 7 of (cq + 123 - nq) computed byte-wise (60 <= . <= 186: no borrow between bytes),
    // and the "brighter" test is bit 7 of 246 - that.  A 9-arc needs two adjacent compass points both darker or both
    // brighter: (N|S)&(E|W).  Slightly looser than the exact compass test; phase B scores exactly either way.
    uint32_t pass[4];
    {
      auto q6 = [](uint32_t x) { return (x >> 2) & 0x3f3f3f3fu; };
      const uint32_t qL = q6(r3[0]), qC1 = q6(r3[1]), qC2 = q6(r3[2]), qR = q6(r3[3]);
      const uint32_t qN1 = q6(r0[1]), qN2 = q6(r0[2]), qS1 = q6(r6[1]), qS2 = q6(r6[2]);
      const uint32_t qE1 = __builtin_amdgcn_alignbyte(qC2, qC1, 3), qE2 = __builtin_amdgcn_alignbyte(qR, qC2, 3);
      const uint32_t qW1 = __builtin_amdgcn_alignbyte(qC1, qL, 1), qW2 = __builtin_amdgcn_alignbyte(qC2, qC1, 1);
      auto flags = [](uint32_t qc, uint32_t qn, uint32_t qs, uint32_t qe, uint32_t qw) {
        const uint32_t cg = qc + 0x7b7b7b7bu;
        const uint32_t dn = cg - qn, ds = cg - qs, de = cg - qe, dw = cg - qw;
        const uint32_t bn = 0xf6f6f6f6u - dn, bs = 0xf6f6f6f6u - ds, be = 0xf6f6f6f6u - de, bw = 0xf6f6f6f6u - dw;
        return (((dn | ds) & (de | dw)) | ((bn | bs) & (be | bw))) & 0x80808080u;
      };
      const uint32_t m1 = flags(qC1, qN1, qS1, qE1, qW1), m2 = flags(qC2, qN2, qS2, qE2, qW2);
      pass[0] = m1 & 0x00008080u; pass[1] = m1 & 0x80800000u; pass[2] = m2 & 0x00008080u; pass[3] = m2 & 0x80800000u;
    }
    {
      // one LDS atomic per wave and row pair: the four ballots are counted on the scalar unit
      const uint64_t m0 = __ballot(pass[0] != 0), m1b = __ballot(pass[1] != 0), m2b = __ballot(pass[2] != 0),
                     m3b = __ballot(pass[3] != 0);
      const int n0 = __popcll(m0), n1 = __popcll(m1b), n2 = __popcll(m2b), n3 = __popcll(m3b);
      if (n0 + n1 + n2 + n3) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&nq, n0 + n1 + n2 + n3);
        base = __builtin_amdgcn_readfirstlane(base);
        const uint64_t below = (1ull << lane) - 1ull;
        const int id = r * 64 + 4 * c;
        if (pass[0]) queue[base + __popcll(m0 & below)] = (uint16_t)id;
        if (pass[1]) queue[base + n0 + __popcll(m1b & below)] = (uint16_t)(id + 1);
        if (pass[2]) queue[base + n0 + n1 + __popcll(m2b & below)] = (uint16_t)(id + 2);
        if (pass[3]) queue[base + n0 + n1 + n2 + __popcll(m3b & below)] = (uint16_t)(id + 3);
      }
    }
  }
  __syncthreads();

  // phase B: exact score of the surviving pairs only.  A pair starts at byte 4 (even pair column) or byte 6 (odd) of
  // the window beginning at its own dword; odd ones are shifted down by two bytes (v_alignbit, shift 0 or 16) so that
  // ONE loop serves both - with ~95 survivors of each parity per tile two separate loops ran at 37 % lane occupancy.
  uint16_t* sc16 = reinterpret_cast<uint16_t*>(sc);
  {
    const int n = nq;
    for (int q = tid; q < n; q += 256) {
      const int id = queue[q];
      const int r = id >> 6, pc = id & 63;
      const uint32_t sh = (uint32_t)(pc & 1) * 16u;
      uint32_t rw[7][4];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const uint32_t* p = &px[(r + j) * PXW + (pc >> 1)];
        const uint32_t a = p[0], b = p[1], c = p[2];
        rw[j][0] = __builtin_amdgcn_alignbit(b, a, sh);
        rw[j][1] = __builtin_amdgcn_alignbit(c, b, sh);
        rw[j][2] = c >> sh;
        rw[j][3] = rw[j][2];
      }
      const uint32_t sp = fast_pair<0>(rw);
      if (sp) {
        sc16[r * 64 + pc] = (uint16_t)sp;
        if (sp & 0xffu) { const int k = atomicAdd(&ncand, 1); if (k < cand_cap) cand[k] = (uint16_t)(r * 128 + 2 * pc); }
        if (sp & 0xff00u) { const int k = atomicAdd(&ncand, 1); if (k < cand_cap) cand[k] = (uint16_t)(r * 128 + 2 * pc + 1); }
      }
    }
  }
  __syncthreads();

  // phase C: strict 3x3 NMS + border filter, over the corner candidates only
  const uint8_t* scb = reinterpret_cast<const uint8_t*>(sc);
  const bool listed = ncand <= cand_cap;   // otherwise (dense texture) every score byte is visited
  const int nc = listed ? ncand : SCR * 128;
  for (int i = tid; i < nc; i += 256) {
    const int id = listed ? cand[i] : i;
    const int sr = id >> 7, scol = id & 127;
    if (!listed && scb[id] == 0) continue;
    const int oy = sr - 1, ox = scol - 4;
    if (oy < 0 || oy >= FAST_TH || ox < 0 || ox >= FAST_TW) continue;
    const int x = x0 + ox, y = y0 + oy;
    if (x < SVO_EDGE || x >= w - SVO_EDGE || y >= h - SVO_EDGE) continue;
    const int sv = scb[sr * SCW + scol];
    bool keep = true;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        if ((dx | dy) != 0) keep = keep && (scb[(sr + dy) * SCW + scol + dx] < sv);
    if (keep) {
      const int p = atomicAdd(&lcount, 1);
      lcorn[p] = (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)sv << 24);
      atomicAdd(&lhist[sv], 1);
    }
  }
  __syncthreads();
  const int cnt = lcount;
  if (cnt == 0) return;
  if (tid == 0) lbase = atomicAdd(&counters[img * SVO_NLEVELS + l], cnt);
  __syncthreads();
  uint32_t* dst = corners + (size_t)img * g.corner_entries + g.coff[l] + lbase;
  for (int i = tid; i < cnt; i += 256) dst[i] = lcorn[i];
  const int hv = lhist[tid];
  if (hv) atomicAdd(&hist[(img * SVO_NLEVELS + l) * 256 + tid], hv);
}

// ---------------------------------------------------------------------------------
// Per (image, level): retainBest(2*quota) by FAST score with ties kept (threshold
// from the histogram) -> exact Harris measure -> rank by (R desc, raster asc) ->
// keep `quota`.  One workgroup; candidates live in LDS.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int64_t harris_at(const uint8_t* img, int pitch, int x, int y) {
  typedef uint32_t __attribute__((aligned(1))) u32u;
  // 9 x 9 neighbourhood (bytes x-4 .. x+4): all 27 unaligned dword loads are issued up front (one
  // memory round trip), then the gradients of the inner 7 x 7 are accumulated row by row
  uint32_t w[9][3];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const uint8_t* p = img + (size_t)(y - 4 + r) * pitch + x - 4;
    w[r][0] = *reinterpret_cast<const u32u*>(p);
    w[r][1] = *reinterpret_cast<const u32u*>(p + 4);
    w[r][2] = *reinterpret_cast<const u32u*>(p + 8);
  }
#define HB(r, i) (int)((w[r][(i) >> 2] >> (8 * ((i) & 3))) & 0xffu)
  int64_t a = 0, b = 0, c = 0;
#pragma unroll
  for (int r = 1; r <= 7; ++r) {
    int sa = 0, sb = 0, scv = 0;
#pragma unroll
    for (int i = 1; i <= 7; ++i) {
      const int Ix = (HB(r, i + 1) - HB(r, i - 1)) * 2 + (HB(r - 1, i + 1) - HB(r - 1, i - 1)) +
                     (HB(r + 1, i + 1) - HB(r + 1, i - 1));
      const int Iy = (HB(r + 1, i) - HB(r - 1, i)) * 2 + (HB(r + 1, i - 1) - HB(r - 1, i - 1)) +
                     (HB(r + 1, i + 1) - HB(r - 1, i + 1));
      sa += Ix * Ix; sb += Iy * Iy; scv += Ix * Iy;
    }
    a += sa; b += sb; c += scv;
  }
#undef HB
  return 25 * (a * b - c * c) - (a + b) * (a + b);
}

__global__ __launch_bounds__(256) void k_select(SvoGeom g, ImgSrc s, const uint32_t* corners,
                                                const int32_t* counters, int32_t* hist,
                                                SvoSel* sel, int32_t* selcnt) {
  __shared__ int64_t cR[SVO_CAP1];
  __shared__ uint32_t cxy[SVO_CAP1];   // x | y<<12 | score<<24; the raster key y*w+x is derived from it
  __shared__ int cum[257];
  __shared__ int sT, nc;
  const int tid = threadIdx.x;
  const int l = blockIdx.x, img = blockIdx.y;
  const int n = counters[img * SVO_NLEVELS + l];
  const int quota = g.quota[l];
  // the histogram is consumed here and handed back zeroed: k_fast of the next call accumulates into it without a
  // 2 MB memset in between (svo_create zeroes it once)
  int32_t* hbin = &hist[(img * SVO_NLEVELS + l) * 256 + tid];
  const int hv = n ? *hbin : 0;
  if (hv) *hbin = 0;
  if (n == 0 || quota == 0) {
    if (tid == 0) selcnt[img * SVO_NLEVELS + l] = 0;
    return;
  }
  cum[tid] = hv;
  if (tid == 0) { cum[256] = 0; nc = 0; sT = 0; }
  __syncthreads();
  // suffix counts cum[s] = #corners with score >= s (parallel scan over the 256 bins)
  for (int o = 1; o < 256; o <<= 1) {
    const int t = tid + o < 256 ? cum[tid + o] : 0;
    __syncthreads();
    cum[tid] += t;
    __syncthreads();
  }
  {
    // retainBest(2*quota) with ties: the largest s whose suffix count reaches the target;
    // then the CAP1 rule: the smallest s whose suffix count fits.  cum[] is non-increasing in s,
    // so each condition holds at exactly one bin.
    const int target = 2 * quota;
    int cand = 0;
    if (n > target && cum[tid] >= target && cum[tid + 1] < target) cand = tid;
    if (cum[tid] <= SVO_CAP1 && (tid == 0 || cum[tid - 1] > SVO_CAP1)) cand = max(cand, tid);
    if (cand) atomicMax(&sT, cand);
  }
  __syncthreads();
  const int T = sT;
  const uint32_t* src = corners + (size_t)img * g.corner_entries + g.coff[l];
  for (int i0 = tid; i0 < n; i0 += 1024) {   // four independent loads in flight per lane
    uint32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = i0 + 256 * k < n ? src[i0 + 256 * k] : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + 256 * k < n && (int)(e[k] >> 24) >= T) {
        const int p = atomicAdd(&nc, 1);
        cxy[p] = e[k];
      }
  }
  __syncthreads();
  const int ncand = nc;
  int pitch;
  const uint8_t* img_p = level_ptr(g, s, img, l, &pitch);
  for (int c = tid; c < ncand; c += 256) {
    const uint32_t e = cxy[c];
    const int x = e & 0xfff, y = (e >> 12) & 0xfff;
    cR[c] = harris_at(img_p, pitch, x, y);
  }
  __syncthreads();
  // Order the candidates by (Harris measure descending, raster index ascending) with a bitonic
  // sort in LDS - O(n log^2 n) instead of the O(n^2) rank count, which dominated when many FAST
  // scores tie at the retainBest threshold.  Raster order y*w+x == order of (y<<12 | x), x < 4096.
  int npow = 64;
  while (npow < ncand) npow <<= 1;
  for (int c = ncand + tid; c < npow; c += 256) { cR[c] = INT64_MIN; cxy[c] = 0x00ffffffu; }  // padding sorts last
  __syncthreads();
  for (int k = 2; k <= npow; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow; i += 256) {
        const int p2 = i ^ j;
        if (p2 > i) {
          const int64_t Ra = cR[i], Rb = cR[p2];
          const uint32_t ea = cxy[i], eb = cxy[p2];
          const uint32_t ka = ea & 0x00ffffffu, kb = eb & 0x00ffffffu;   // y<<12 | x
          const bool a_first = (Ra > Rb) || (Ra == Rb && ka < kb);      // a belongs before b
          const bool up = (i & k) == 0;                                  // this block sorts "best first"
          if (a_first != up) { cR[i] = Rb; cR[p2] = Ra; cxy[i] = eb; cxy[p2] = ea; }
        }
      }
      __syncthreads();
    }
  }
  SvoSel* out = sel + (size_t)(img * SVO_NLEVELS + l) * SVO_QMAX;
  for (int c = tid; c < min(ncand, quota); c += 256) {
    const uint32_t e = cxy[c];
    SvoSel v;
    v.R = cR[c]; v.x = (int16_t)(e & 0xfff); v.y = (int16_t)((e >> 12) & 0xfff); v.pad = 0;
    out[c] = v;
  }
  if (tid == 0) selcnt[img * SVO_NLEVELS + l] = min(ncand, quota);
}

// ---------------------------------------------------------------------------------
// Per keypoint (one wave): orientation + descriptor.  The 37 x 37 source window is
// staged in LDS once; the 7 x 7 Gaussian (fixed-point {18,34,49,55,49,34,18},
// (sum+2^15)>>16) is evaluated only on the 31 x 31 patch the rotated pattern can
// reach, so the blurred pyramid is never written to HBM.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + 2.220446e-16f);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + 2.220446e-16f);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ void det_sincos(float angle_rad, float* s, float* c) {
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double PIO2_HI = 1.57079632679489655800e+00;
  const double PIO2_LO = 6.12323399573676603587e-17;
  const double x = (double)angle_rad;
  const double kq = floor(x * TWO_OVER_PI + 0.5);
  const double r = (x - kq * PIO2_HI) - kq * PIO2_LO;
  const double r2 = r * r;
  double ps = -1.0 / 1307674368000.0;
  ps = ps * r2 + 1.0 / 6227020800.0;
  ps = ps * r2 - 1.0 / 39916800.0;
  ps = ps * r2 + 1.0 / 362880.0;
  ps = ps * r2 - 1.0 / 5040.0;
  ps = ps * r2 + 1.0 / 120.0;
  ps = ps * r2 - 1.0 / 6.0;
  const double sr = r + r * (r2 * ps);
  double pc = 1.0 / 20922789888000.0;
  pc = pc * r2 - 1.0 / 87178291200.0;
  pc = pc * r2 + 1.0 / 479001600.0;
  pc = pc * r2 - 1.0 / 3628800.0;
  pc = pc * r2 + 1.0 / 40320.0;
  pc = pc * r2 - 1.0 / 720.0;
  pc = pc * r2 + 1.0 / 24.0;
  pc = pc * r2 - 0.5;
  const double cr = 1.0 + r2 * pc;
  const int q = (int)((long long)kq & 3);
  double sv, cv;
  switch (q) {
    case 0: sv = sr; cv = cr; break;
    case 1: sv = cr; cv = -sr; break;
    case 2: sv = -sr; cv = -cr; break;
    default: sv = -cr; cv = sr; break;
  }
  *s = (float)sv;
  *c = (float)cv;
}

__device__ __forceinline__ int wave_sum(int v) { return wave_sum_i32_dpp(v); }

#define PW 37      // staged window side (31 + 2*3)
#define PROW 10    // dwords per staged row: byte 0 = column x-19, pixel column x+u at byte u+19

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

// Per-row byte weights of the intensity-centroid moments over the radius-15 disc (umax table):
// row v (-15..15), byte c (0..31) <-> u = c-15; w = u (as int8) inside the disc, else 0; one = 1/0.
struct MomTables { uint32_t w[31][8]; uint32_t one[31][8]; };
constexpr MomTables make_mom_tables() {
  MomTables t{};
  const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  for (int v = -15; v <= 15; ++v)
    for (int c = 0; c < 32; ++c) {
      const int u = c - 15, av = v < 0 ? -v : v, au = u < 0 ? -u : u;
      const bool inside = c < 31 && au <= umax[av];
      if (inside) {
        t.w[v + 15][c / 4] |= (uint32_t)((uint8_t)(int8_t)u) << (8 * (c % 4));
        t.one[v + 15][c / 4] |= 1u << (8 * (c % 4));
      }
    }
  return t;
}
__constant__ MomTables c_mom = make_mom_tables();

// One wave per keypoint.  Everything between the staged window and the descriptor bits runs on
// the byte/short dot-product instructions: v_dot4 for the moments and the horizontal Gaussian
// pass (bytes x {18,34,49,55 | 49,34,18,0}), v_dot2_u32_u16 for the vertical pass.
__global__ __launch_bounds__(64) void k_describe(SvoGeom g, ImgSrc s, const SvoSel* sel,
                                                 const int32_t* selcnt, svo_kp* kp, uint8_t* desc,
                                                 int32_t* nkp, int max_kp) {
  __shared__ uint32_t patch[PW * PROW];          // 37 rows x 40 bytes
  __shared__ uint32_t hbT[31 * 20];              // horizontal pass, TRANSPOSED: [column][row] u16, 40 rows pitch
  __shared__ uint32_t blT[31 * 8];               // blurred 31 x 31 patch, TRANSPOSED: [column][row] u8, 32 pitch
  const int lane = threadIdx.x;
  // XCD-aware order (as k_fast): workgroup b runs on XCD b % 8 and every XCD has its own L2 - the linear id is remapped
  // (bijectively) so that each XCD gets one contiguous eighth of the (image, keypoint) sequence: the 37 x 40-byte windows
  // of one image's keypoints overlap heavily, and this way they meet in ONE L2 instead of being fetched by eight.
  int slot, img;
  {
    const uint32_t total_wg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t xcd = lin & 7u, idx = lin >> 3, q = total_wg >> 3, r = total_wg & 7u;
    const uint32_t mapped = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    img = (int)(mapped / gridDim.x);
    slot = (int)(mapped - (uint32_t)img * gridDim.x);
  }
  // slot -> (level, rank)
  int l = -1, rank = 0, total = 0;
#pragma unroll
  for (int k = 0; k < SVO_NLEVELS; ++k) {
    const int c = selcnt[img * SVO_NLEVELS + k];
    if (l < 0 && slot < total + c) { l = k; rank = slot - total; }
    total += c;
  }
  total = min(total, max_kp);
  if (slot == 0 && lane == 0) nkp[img] = total;
  if (l < 0 || slot >= total) return;
  const SvoSel sv = sel[(size_t)(img * SVO_NLEVELS + l) * SVO_QMAX + rank];
  // the four test pairs of this lane, requested now so that they arrive under the patch loads
  char4 pat[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) pat[t] = *reinterpret_cast<const char4*>(c_pattern[t * 64 + lane]);
  uint32_t mw[8], mo[8];   // moment weights of this lane's disc row (same reason)
#pragma unroll
  for (int k = 0; k < 8; ++k) { mw[k] = c_mom.w[min(lane, 30)][k]; mo[k] = c_mom.one[min(lane, 30)][k]; }
  const int x = sv.x, y = sv.y;
  int pitch;
  const uint8_t* img_p = level_ptr(g, s, img, l, &pitch);
  const uint8_t* org = img_p + (size_t)(y - 18) * pitch + (x - 19);
  {
    // a lane keeps one dword column (6 row groups x 10 columns = 60 lanes) and walks down the rows: its seven loads
    // are in flight together (one memory round trip) and need no per-load index arithmetic
    constexpr int NROWG = 64 / PROW, NSLOT = (PW + NROWG - 1) / NROWG;
    const int rg = lane / PROW, d = lane - rg * PROW;
    const uint8_t* col = org + 4 * d;
    uint32_t v[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) v[k] = *reinterpret_cast<const u32_unaligned*>(col + (size_t)min(rg + NROWG * k, PW - 1) * pitch);
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
      if (rg < NROWG && rg + NROWG * k < PW) patch[(rg + NROWG * k) * PROW + d] = v[k];
  }
  if (lane < 62) hbT[(lane >> 1) * 20 + 18 + (lane & 1)] = 0;   // rows 37..39 of each column stay zero (36 is written below)
  __syncthreads();
  // intensity centroid: lane = disc row v+15; m10 = sum u*I (via I-128, the row weights sum to 0),
  // m01 = v * sum I
  int m10 = 0, m01 = 0;
  if (lane < 31) {
    const uint32_t* row = &patch[(lane + 3) * PROW + 1];   // bytes 4..35 = u -15..16
    int rs = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t px4 = row[k];
      m10 = __builtin_amdgcn_sdot4((int)(px4 ^ 0x80808080u), (int)mw[k], m10, false);
      rs = (int)__builtin_amdgcn_udot4(px4, mo[k], (uint32_t)rs, false);
    }
    m01 = (lane - 15) * rs;
  }
  m10 = wave_sum(m10);
  m01 = wave_sum(m01);
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // horizontal pass: item = (row r, group of 4 output columns c0 = 4g); taps = bytes c+1 .. c+7
  const uint32_t K0 = 18u | (34u << 8) | (49u << 16) | (55u << 24), K1 = 49u | (34u << 8) | (18u << 16);
  uint16_t* hb16 = reinterpret_cast<uint16_t*>(hbT);
  for (int i = lane; i < PW * 8; i += 64) {
    const int r = i >> 3, gq = i & 7;
    const uint32_t* p = &patch[r * PROW + gq];
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    uint32_t o[4];
    o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), K1,
                                  __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), K0, 0u, false), false);
    o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), K1,
                                  __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), K0, 0u, false), false);
    o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), K1,
                                  __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), K0, 0u, false), false);
    o[3] = __builtin_amdgcn_udot4(d2, K1, __builtin_amdgcn_udot4(d1, K0, 0u, false), false);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * gq + j < 31) hb16[(4 * gq + j) * 40 + r] = (uint16_t)o[j];
  }
  __syncthreads();
  // vertical pass: item = (column c, group of 4 output rows r0 = 4h); rows r..r+6 of hbT[c]
  const uint32_t W01 = 18u | (34u << 16), W23 = 49u | (55u << 16), W45 = 49u | (34u << 16), W6 = 18u;
  for (int i = lane; i < 31 * 8; i += 64) {
    const int c = i >> 3, hq = i & 7;
    const uint32_t* p = &hbT[c * 20 + 2 * hq];
    const uint32_t q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
    auto dot7 = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
      uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, a0), __builtin_bit_cast(v2u16, W01), 32768u, false);
      acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, a1), __builtin_bit_cast(v2u16, W23), acc, false);
      acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, a2), __builtin_bit_cast(v2u16, W45), acc, false);
      acc = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, a3), __builtin_bit_cast(v2u16, W6), acc, false);
      return min(acc >> 16, 255u);
    };
    const uint32_t s01 = __builtin_amdgcn_alignbyte(q1, q0, 2), s12 = __builtin_amdgcn_alignbyte(q2, q1, 2),
                   s23 = __builtin_amdgcn_alignbyte(q3, q2, 2), s34 = __builtin_amdgcn_alignbyte(q4, q3, 2),
                   s45 = __builtin_amdgcn_alignbyte(0u, q4, 2);
    const uint32_t b0 = dot7(q0, q1, q2, q3), b1 = dot7(s01, s12, s23, s34), b2 = dot7(q1, q2, q3, q4),
                   b3 = dot7(s12, s23, s34, s45);
    blT[c * 8 + hq] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
  }
  __syncthreads();
  const uint8_t* bl = reinterpret_cast<const uint8_t*>(blT);
  float sn, cs;
  det_sincos(angle * 0.017453292f, &sn, &cs);
  const size_t oidx = (size_t)img * max_kp + slot;
  uint64_t* dout = reinterpret_cast<uint64_t*>(desc + oidx * SVO_DESC_BYTES);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float px0 = (float)pat[t].x, py0 = (float)pat[t].y;
    const float px1 = (float)pat[t].z, py1 = (float)pat[t].w;
    const float rx0 = px0 * cs - py0 * sn, ry0 = px0 * sn + py0 * cs;
    const float rx1 = px1 * cs - py1 * sn, ry1 = px1 * sn + py1 * cs;
    const int v0 = bl[(15 + __float2int_rn(rx0)) * 32 + 15 + __float2int_rn(ry0)];
    const int v1 = bl[(15 + __float2int_rn(rx1)) * 32 + 15 + __float2int_rn(ry1)];
    const uint64_t m = __ballot(v0 < v1);
    if (lane == 0) dout[t] = m;
  }
  if (lane == 0) {
    svo_kp k;
    const float sc = g.scale[l];
    k.x = (float)x * sc;
    k.y = (float)y * sc;
    k.size = 31.0f * sc;
    k.angle = angle;
    const double kscale = 1.0 / (4.0 * 7.0 * 255.0);
    k.response = (float)((double)sv.R * (kscale * kscale * kscale * kscale / 25.0));
    k.octave = l;
    k.class_id = -1;
    kp[oidx] = k;
  }
}

// ---------------------------------------------------------------------------------
int svo_launch_orb(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR, int stride,
                   int B, int nimg) {
  const SvoGeom& g = ctx->g;
  if (nimg > ctx->max_images || B < 1) return SVO_E_CAPACITY;
  ImgSrc s{d_grayL, d_grayR, stride, B, ctx->d_pyr};
  hipStream_t st = ctx->stream;
  SVO_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(int32_t) * (size_t)nimg * SVO_NLEVELS, st));
  {
    const bool fused = ctx->opt_pyr_fused && !ctx->pyr_plan.empty();
    SvoTimer t(ctx, fused ? "k_pyr_fused" : "k_pyr_level");   // one timer around the pyramid's launches (three fused ones, or seven)
    if (fused) {
      for (const SvoPyrGroup& pg : ctx->pyr_plan) {
        const int lt = pg.l0 + pg.nl - 1;
        const dim3 grid((g.h[lt] + pg.rt - 1) / pg.rt, nimg);
        if (pg.nl == 2)
          hipLaunchKernelGGL(k_pyr_fused<2>, grid, dim3(256), pg.lds, st, g, s, pg.l0, pg.rt, pg.cap0, pg.cap1, ctx->d_pyr, ctx->d_xofs,
                             ctx->d_xalpha, ctx->d_yofs, ctx->d_ybeta);
        else
          hipLaunchKernelGGL(k_pyr_fused<3>, grid, dim3(256), pg.lds, st, g, s, pg.l0, pg.rt, pg.cap0, pg.cap1, ctx->d_pyr, ctx->d_xofs,
                             ctx->d_xalpha, ctx->d_yofs, ctx->d_ybeta);
      }
    } else {
      for (int l = 1; l < SVO_NLEVELS; ++l) {
        dim3 grid((g.w[l] / 4 + 64) / 64, (g.h[l] + 15) / 16, nimg);
        hipLaunchKernelGGL(k_pyr_level, grid, dim3(64, 4, 1), 0, st, g, s, l, ctx->d_pyr, ctx->d_xofs,
                           ctx->d_xalpha, ctx->d_yofs, ctx->d_ybeta);
      }
    }
  }
  {
    SvoTimer t(ctx, "k_fast");
    hipLaunchKernelGGL(k_fast, dim3(g.tile_base[SVO_NLEVELS], nimg), dim3(256), 0, st, g, s,
                       ctx->d_corners, ctx->d_counters, ctx->d_hist, ctx->opt_fast_cand_cap);
  }
  {
    SvoTimer t(ctx, "k_select");
    hipLaunchKernelGGL(k_select, dim3(SVO_NLEVELS, nimg), dim3(256), 0, st, g, s, ctx->d_corners,
                       ctx->d_counters, ctx->d_hist, ctx->d_sel, ctx->d_selcnt);
  }
  {
    SvoTimer t(ctx, "k_describe");
    hipLaunchKernelGGL(k_describe, dim3(ctx->max_kp, nimg), dim3(64), 0, st, g, s, ctx->d_sel,
                       ctx->d_selcnt, ctx->d_kp, ctx->d_desc, ctx->d_nkp, ctx->max_kp);
  }
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}
