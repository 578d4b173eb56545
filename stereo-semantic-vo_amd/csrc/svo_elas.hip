// svo_elas.hip - dense ELAS stereo for gfx950.
//
// Replaces the reference's vendored libelas, `Elas::process` (Thirdparty/libelas/src/elas.cpp:32-150).
// Stage map (reference file:line -> here):
//   Descriptor (descriptor.cpp:27-112, filter.cpp:176-262,372-413)   -> k_elas_desc        (GPU)
//   computeSupportMatches / computeMatchingDisparity (elas.cpp:267-443) -> k_elas_support   (GPU)
//   removeInconsistent/RedundantSupportPoints, addCorners (elas.cpp:152-265) -> host (in-place, order dependent)
//   computeDelaunayTriangulation (elas.cpp:445-503)                  -> svo_delaunay.hip   (host)
//   computeDisparityPlanes (elas.cpp:505-577, matrix.cpp:414-503)    -> host (float64 Gauss-Jordan)
//   createGrid (elas.cpp:579-658)                                    -> host
//   computeDisparity / findMatch (elas.cpp:682-907)                  -> k_elas_raster + k_elas_match (GPU)
//   leftRightConsistencyCheck (elas.cpp:909-979)                     -> k_elas_lr          (GPU)
//   removeSmallSegments (elas.cpp:981-1099)                          -> k_cc_* union-find  (GPU)
//   gapInterpolation (elas.cpp:1101-1285)                            -> k_elas_gap         (GPU)
//   adaptiveMean (elas.cpp:1287-1483)                                -> k_elas_mean_h/_v   (GPU)
//   median (elas.cpp:1485-1560)                                      -> k_elas_median_h/_v (GPU)
// All integer stages are bit-exact against the compiled reference; the float stages repeat its
// operation order (-ffp-contract=off), including libelas' `_mm_set1_ps(0x7FFFFFFF)` "abs mask", which is
// really the bit pattern of 2^31f (0x4F000000).  Memory the reference reads without having written it
// (descriptor rows/columns 2 and N-3; D_tmp of adaptiveMean) is defined as 0 here.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <system_error>
#include <thread>
#include <vector>

#include "svo_internal.h"
#include "svo_wave.h"

extern "C" int svo_elas_delaunay(const int32_t* xy, int32_t n, int32_t* tri, int32_t cap, int32_t* n_tri);

namespace {

struct ElasState {
  int W = 0, H = 0, Wc = 0, Hc = 0, gw = 0, gh = 0, gd = 0;
  int cap_sp = 0, cap_tri = 0;
  uint8_t* d_img[2] = {nullptr, nullptr};
  uint4* d_desc[2] = {nullptr, nullptr};
  int16_t* d_can = nullptr;
  int32_t* d_sp = nullptr;
  int32_t* d_tri[2] = {nullptr, nullptr};
  float* d_plane[2] = {nullptr, nullptr};
  int32_t* d_grid[2] = {nullptr, nullptr};
  int32_t* d_P = nullptr;
  int32_t* d_owner[2] = {nullptr, nullptr};
  float* d_D[2] = {nullptr, nullptr};
  float* d_T[2] = {nullptr, nullptr};
  int32_t* d_lab = nullptr;
  int32_t* d_size = nullptr;
  uint8_t* h_img = nullptr;   // pinned staging of the host API: 2 x W*H bytes in, 2 x W*H floats out
  float* h_D = nullptr;
  int16_t* h_can = nullptr;   // pinned: lattice candidates back from the GPU
  struct ElasTab* d_tab = nullptr;   // 1-entry table of the one-pair path
  std::vector<void*> allocs;
  void release() {
    for (void* p : allocs) hipFree(p);
    allocs.clear();
    if (h_img) hipHostFree(h_img);
    if (h_D) hipHostFree(h_D);
    if (h_can) hipHostFree(h_can);
    h_img = nullptr; h_D = nullptr; h_can = nullptr;
    W = H = 0;
  }
};

// ------------------------------------------------------------------------------------------------
// Descriptor: 3x3 Sobel responses du, dv (8-bit, offset 128, saturated) and the 16-byte descriptor.
// ------------------------------------------------------------------------------------------------
#define DT_X 32
// global -> LDS staging with a thread's loads in flight together: written as a plain loop the compiler waits for every
// load before the LDS store (one memory round trip per iteration); DEPTH loads are issued back to back instead
template <int DEPTH, typename T>
__device__ __forceinline__ void stage_lds(T* dst, const T* src, int n, size_t stride, int tid) {
  for (int i0 = tid; i0 < n; i0 += 256 * DEPTH) {
    T v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) v[k] = src[(size_t)min(i0 + 256 * k, n - 1) * stride];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
      if (i0 + 256 * k < n) dst[i0 + 256 * k] = v[k];
  }
}

#define DT_Y 8
__device__ __forceinline__ void d_elas_desc(int side, const uint8_t* img0, const uint8_t* img1, int pitch, int W, int H,
                                            int half, uint4* desc0, uint4* desc1) {
  __shared__ uint8_t im[DT_Y + 6][DT_X + 8];
  __shared__ uint8_t du[DT_Y + 4][DT_X + 4], dv[DT_Y + 4][DT_X + 4];
  const uint8_t* img = side ? img1 : img0;
  uint4* desc = side ? desc1 : desc0;
  const int x0 = blockIdx.x * DT_X, y0 = blockIdx.y * DT_Y, tid = threadIdx.x;
  {
    constexpr int NPX = (DT_Y + 6) * (DT_X + 6), NSLOT = (NPX + 255) / 256;
    uint8_t pv[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int i = min(tid + 256 * k, NPX - 1);
      const int r = i / (DT_X + 6), c = i - r * (DT_X + 6);
      const int gy = min(max(y0 - 3 + r, 0), H - 1), gx = min(max(x0 - 3 + c, 0), W - 1);
      pv[k] = img[(size_t)gy * pitch + gx];
    }
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int i = tid + 256 * k;
      if (i < NPX) im[i / (DT_X + 6)][i % (DT_X + 6)] = pv[k];
    }
  }
  __syncthreads();
  for (int i = tid; i < (DT_Y + 4) * (DT_X + 4); i += 256) {
    const int r = i / (DT_X + 4), c = i - r * (DT_X + 4);   // pixel (y0-2+r, x0-2+c); im index (r+1, c+1)
    const int sl = im[r][c] + 2 * im[r + 1][c] + im[r + 2][c];
    const int sr = im[r][c + 2] + 2 * im[r + 1][c + 2] + im[r + 2][c + 2];
    const int tl = im[r][c] - im[r + 2][c], tc = im[r][c + 1] - im[r + 2][c + 1], tr = im[r][c + 2] - im[r + 2][c + 2];
    du[r][c] = (uint8_t)min(max(((sl - sr) >> 2) + 128, 0), 255);
    dv[r][c] = (uint8_t)min(max(((tl + 2 * tc + tr) >> 2) + 128, 0), 255);
  }
  __syncthreads();
  const int tx = tid & (DT_X - 1), ty = tid / DT_X;
  const int u = x0 + tx, v = y0 + ty;
  if (u >= W || v >= H) return;
  uint4 o = make_uint4(0, 0, 0, 0);
  // half resolution (descriptor.cpp:44-72): only the even rows 4, 6, ... are built
  if (u >= 3 && u < W - 3 && v >= 3 && v < H - 3 && (!half || (v >= 4 && (v & 1) == 0))) {
    const int j = tx + 2, i = ty + 2;   // (v, u) in du/dv indices
    o.x = du[i - 2][j] | (du[i - 1][j - 2] << 8) | (du[i - 1][j] << 16) | ((uint32_t)du[i - 1][j + 2] << 24);
    o.y = du[i][j - 1] | (du[i][j] << 8) | (du[i][j] << 16) | ((uint32_t)du[i][j + 1] << 24);
    o.z = du[i + 1][j - 2] | (du[i + 1][j] << 8) | (du[i + 1][j + 2] << 16) | ((uint32_t)du[i + 2][j] << 24);
    o.w = dv[i - 1][j] | (dv[i][j - 1] << 8) | (dv[i][j + 1] << 16) | ((uint32_t)dv[i + 1][j] << 24);
  }
  desc[(size_t)v * W + u] = o;
}

__device__ __forceinline__ int sad16(const uint4& a, const uint4& b) {
  uint32_t s = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
  s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
  s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
  return (int)__builtin_amdgcn_sad_u8(a.w, b.w, s);
}
__device__ __forceinline__ int texture16(const uint4& a) {   // sum |byte - 128|
  return sad16(a, make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u));
}

// ------------------------------------------------------------------------------------------------
// Support matches: one wave per lattice candidate, lanes over disparities.
// ------------------------------------------------------------------------------------------------
__device__ int elas_support_match(const uint4* __restrict__ I1d, const uint4* __restrict__ I2d, int u, int v,
                                  bool right, int W, int H, const svo_elas_params& p, int lane) {
  if (!(u >= 5 && u <= W - 6 && v >= 5 && v <= H - 6)) return -1;
  if (texture16(I1d[(size_t)v * W + u]) < p.support_texture) return -1;
  const size_t r0 = (size_t)(v - 2) * W, r1 = (size_t)(v + 2) * W;
  const uint4 b1 = I1d[r0 + u - 2], b2 = I1d[r0 + u + 2], b3 = I1d[r1 + u - 2], b4 = I1d[r1 + u + 2];
  const int dmin = max(p.disp_min, 0);
  const int dmax = right ? min(p.disp_max, W - u - 5) : min(p.disp_max, u - 5);
  if (dmax - dmin < 10) return -1;
  uint32_t k1 = 0xffffffffu, m2 = 0x7fffffu;
  for (int d = dmin + lane; d <= dmax; d += 64) {
    const int uw = right ? u + d : u - d;
    const uint32_t sum = (uint32_t)(sad16(b1, I2d[r0 + uw - 2]) + sad16(b2, I2d[r0 + uw + 2]) +
                                    sad16(b3, I2d[r1 + uw - 2]) + sad16(b4, I2d[r1 + uw + 2]));
    const uint32_t key = (sum << 9) | (uint32_t)d;
    if (key < k1) { m2 = k1 >> 9; k1 = key; }
    else if (sum < m2) m2 = sum;
  }
  const uint32_t K1 = wave_min_u32_dpp(k1);
  const uint32_t second = wave_min_u32_dpp(k1 == K1 ? m2 : (k1 >> 9));
  const uint32_t min1 = K1 >> 9;
  if (second >= 32767u) return -1;   // no second candidate
  return ((float)min1 < p.support_threshold * (float)second) ? (int)(K1 & 511u) : -1;
}

__device__ __forceinline__ void d_elas_support(const uint4* desc1, const uint4* desc2, int W, int H,
                                                      int Wc, int Hc, svo_elas_params p, int16_t* D_can) {
  const int lane = threadIdx.x & 63;
  const int cand = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cand >= (Wc - 1) * (Hc - 1)) return;
  const int v_can = 1 + cand / (Wc - 1), u_can = 1 + cand % (Wc - 1);
  const int u = u_can * p.candidate_stepsize, v = v_can * p.candidate_stepsize;
  int res = -1;
  const int d = elas_support_match(desc1, desc2, u, v, false, W, H, p, lane);
  if (d >= 0) {
    const int d2 = elas_support_match(desc2, desc1, u - d, v, true, W, H, p, lane);
    if (d2 >= 0 && abs(d - d2) <= p.lr_threshold) res = d;
  }
  if (lane == 0) D_can[v_can * Wc + u_can] = (int16_t)res;
}

// ------------------------------------------------------------------------------------------------
// Dense matching.  k_elas_raster repeats the reference's scanline rasterisation (its float edge
// equations and truncations) and leaves, per pixel, the LAST triangle covering it (what the
// reference's sequential overwrite leaves); k_elas_match then runs findMatch once per pixel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_elas_raster(const int32_t* sp, const int32_t* tri0, const int32_t* tri1,
                                                     int n0, int n1, int W, int H, int sub, int32_t* own0, int32_t* own1) {
  const bool right = blockIdx.y != 0;
  const int32_t* tri = right ? tri1 : tri0;
  int32_t* owner = right ? own1 : own0;
  const int n = right ? n1 : n0;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n) return;
  float tu[3], tv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = tri[3 * t + k];
    tu[k] = right ? (float)(sp[3 * c] - sp[3 * c + 2]) : (float)sp[3 * c];
    tv[k] = (float)sp[3 * c + 1];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < j; ++k)
      if (tu[k] > tu[j]) {
        const float a = tu[j]; tu[j] = tu[k]; tu[k] = a;
        const float b = tv[j]; tv[j] = tv[k]; tv[k] = b;
      }
  const float A_u = tu[0], A_v = tv[0], B_u = tu[1], B_v = tv[1], C_u = tu[2], C_v = tv[2];
  float AB_a = 0, AC_a = 0, BC_a = 0;
  if ((int)A_u != (int)B_u) AB_a = (A_v - B_v) / (A_u - B_u);
  if ((int)A_u != (int)C_u) AC_a = (A_v - C_v) / (A_u - C_u);
  if ((int)B_u != (int)C_u) BC_a = (B_v - C_v) / (B_u - C_u);
  const float AB_b = A_v - AB_a * A_u, AC_b = A_v - AC_a * A_u, BC_b = B_v - BC_a * B_u;
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    const float s_u = part ? B_u : A_u, e_u = part ? C_u : B_u;
    const float m_a = part ? BC_a : AB_a, m_b = part ? BC_b : AB_b;
    if ((int)s_u == (int)e_u) continue;
    // the lattice triangles are ~10-25 columns wide: the 64 lanes form lx x ly, lx = the power of two that
    // covers the columns of this part, the other lanes share each column's rows
    const int u_lo = max((int)s_u, 0), u_hi = min((int)e_u, W);
    int lx = 4;
    while (lx < 64 && lx < u_hi - u_lo) lx <<= 1;
    const int ly = 64 / lx, my_x = lane & (lx - 1), my_y = lane / lx;
    for (int u = u_lo + my_x; u < u_hi; u += lx) {
      if (sub && (u & 1)) continue;
      const int v_1 = (int)(AC_a * (float)u + AC_b);
      const int v_2 = (int)(m_a * (float)u + m_b);
      for (int v = max(min(v_1, v_2), 0) + my_y; v < min(max(v_1, v_2), H); v += ly) {
        if (!sub) atomicMax(&owner[v * W + u], t);
        else if (!(v & 1) && (v >> 1) < H / 2 && (u >> 1) < W / 2) atomicMax(&owner[(v >> 1) * (W / 2) + (u >> 1)], t);
      }
    }
  }
}

__device__ __forceinline__ void d_elas_match(bool right, const uint4* desc1, const uint4* desc2, const int32_t* own0,
                                                    const int32_t* own1, const float* pl0, const float* pl1,
                                                    const int32_t* grid0, const int32_t* grid1, const int32_t* P,
                                                    int W, int H, int gw, int gd, int plane_radius,
                                                    svo_elas_params p, float* D0, float* D1) {
  const int sub = p.subsampling, Wd = sub ? W / 2 : W;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  const int u = sub ? 2 * x : x, v = sub ? 2 * y : y;   // subsampling: every second pixel (elas.cpp:846-871)
  const uint4* Is = right ? desc2 : desc1;
  const uint4* Io = right ? desc1 : desc2;
  const size_t line = (size_t)max(min(v, H - 3), 2) * W;
  // the descriptors of the other image this block can reach (its columns -255 .. +0 for the left map,
  // +0 .. +255 for the right one) are staged in LDS once: every pixel then reads its 20-60 candidates there
  __shared__ uint4 win[768];
  const int u_first = sub ? 2 * (int)(blockIdx.x * 256) : (int)(blockIdx.x * 256);
  const int u_last = min(sub ? u_first + 510 : u_first + 255, W - 1);
  const int w0 = max(right ? u_first : u_first - 255, 0), w1 = min(right ? u_last + 255 : u_last, W - 1);
  for (int i = threadIdx.x; i <= w1 - w0; i += 256) win[i] = Io[line + w0 + i];
  __syncthreads();
  if (x >= Wd) return;
  const int addr = y * Wd + x;
  float* D = right ? D1 : D0;
  const int t = (right ? own1 : own0)[addr];
  float out = -10.0f;
  if (t >= 0 && u >= 2 && u < W - 2) {
    const uint4 self = Is[line + u];
    if (texture16(self) >= p.match_texture) {
      const float* pl = (right ? pl1 : pl0) + 6 * t;
      const float pa = right ? pl[3] : pl[0], pb = right ? pl[4] : pl[1], pc = right ? pl[5] : pl[2];
      const float pd = right ? pl[0] : pl[3];
      const bool valid = (double)fabsf(pa) < 0.7 && (double)fabsf(pd) < 0.7;
      const int disp_num = gd - 1;
      const int d_plane = (int)(pa * (float)u + pb * (float)v + pc);
      const int d_plane_min = max(d_plane - plane_radius, 0);
      const int d_plane_max = min(d_plane + plane_radius, disp_num - 1);
      const int gx = (int)floorf((float)u / (float)p.grid_size), gy = (int)floorf((float)v / (float)p.grid_size);
      // the cell's disparity candidates: a 256-bit set, visited in ascending order like the reference's list
      const uint32_t* cell = reinterpret_cast<const uint32_t*>(right ? grid1 : grid0) + (size_t)(gy * gw + gx) * 8;
      int min_val = 10000, min_d = -1;
      // the whole 256-bit set at once (two 16-byte loads in flight) instead of one dependent dword load per word
      uint32_t cw[8];
      {
        const uint4 c0 = *reinterpret_cast<const uint4*>(cell), c1 = *reinterpret_cast<const uint4*>(cell + 4);
        cw[0] = c0.x; cw[1] = c0.y; cw[2] = c0.z; cw[3] = c0.w; cw[4] = c1.x; cw[5] = c1.y; cw[6] = c1.z; cw[7] = c1.w;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        uint32_t m = cw[q];
        while (m) {
          const int d = 32 * q + __ffs(m) - 1;
          m &= m - 1;
          if (d < d_plane_min || d > d_plane_max) {
            const int uw = right ? u + d : u - d;
            if (uw < 2 || uw >= W - 2) continue;
            const int val = sad16(self, win[uw - w0]);
            if (val < min_val) { min_val = val; min_d = d; }
          }
        }
      }
      for (int d = d_plane_min; d <= d_plane_max; ++d) {
        const int uw = right ? u + d : u - d;
        if (uw < 2 || uw >= W - 2) continue;
        const int val = sad16(self, win[uw - w0]) + (valid ? P[abs(d - d_plane)] : 0);
        if (val < min_val) { min_val = val; min_d = d; }
      }
      out = min_d >= 0 ? (float)min_d : -1.0f;
    }
  }
  D[addr] = out;
}

__device__ __forceinline__ void d_elas_lr(const float* D1, const float* D2, int W, int H, int lr_threshold,
                                                 int sub, float* O1, float* O2) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  const int addr = v * W + u;
  const float d1 = D1[addr], d2 = D2[addr];
  const float uw1 = sub ? (float)u - d1 / 2 : (float)u - d1, uw2 = sub ? (float)u + d2 / 2 : (float)u + d2;
  float o1 = d1, o2 = d2;
  if (d1 >= 0 && uw1 >= 0 && uw1 < (float)W) {
    if (fabsf(D2[v * W + (int)uw1] - d1) > (float)lr_threshold) o1 = -10.0f;
  } else o1 = -10.0f;
  if (d2 >= 0 && uw2 >= 0 && uw2 < (float)W) {
    if (fabsf(D1[v * W + (int)uw2] - d2) > (float)lr_threshold) o2 = -10.0f;
  } else o2 = -10.0f;
  O1[addr] = o1; O2[addr] = o2;
}

// ------------------------------------------------------------------------------------------------
// removeSmallSegments: the reference's region growing visits exactly the connected components of the
// graph "4-neighbours, both valid, |d_a - d_b| <= threshold" (invalid pixels are -10 after the L/R
// check and can never join); components smaller than speckle_size are invalidated.  Union-find.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cc_load(const int32_t* L, int i) { return __atomic_load_n(&L[i], __ATOMIC_RELAXED); }
__device__ int cc_find(const int32_t* L, int i) {
  int p = cc_load(L, i);
  while (p != i) { i = p; p = cc_load(L, i); }
  return i;
}
// find with path halving: every visited node is re-pointed to its grandparent.  The plain store can race
// with an atomicMin on the same entry only when that entry is not a root any more, and then either value is
// an ancestor inside the same (already united or about to be re-united) set - cc_union re-reads and retries.
__device__ int cc_find_halving(int32_t* L, int i) {
  int p = cc_load(L, i);
  while (p != i) {
    const int gp = cc_load(L, p);
    if (gp != p) __atomic_store_n(&L[i], gp, __ATOMIC_RELAXED);
    i = p; p = gp;
  }
  return i;
}
__device__ void cc_union(int32_t* L, int a, int b) {
  bool done;
  do {
    a = cc_find_halving(L, a); b = cc_find_halving(L, b);
    if (a < b) { const int old = atomicMin(&L[b], a); done = old == b; b = old; }
    else if (b < a) { const int old = atomicMin(&L[a], b); done = old == a; a = old; }
    else done = true;
  } while (!done);
}
// Rows first: every pixel is labelled with the first pixel of its horizontal run (a per-row scan in
// LDS, no atomics), the run head records the run length.  Only run-to-run contacts are then united
// vertically (the first column of each contact), and sizes are accumulated per run, not per pixel.
#define CC_MAXLEN 4096
__device__ __forceinline__ void d_cc_rows(const float* D, int W, float thr, int32_t* L, int32_t* rlen,
                                                 int32_t* size) {
  __shared__ float val[CC_MAXLEN];
  __shared__ int16_t start[CC_MAXLEN];
  __shared__ int cs[256];
  const int v = blockIdx.x, tid = threadIdx.x;
  const float* row = D + (size_t)v * W;
  stage_lds<8>(val, row, W, 1, tid);
  __syncthreads();
  const int C = (W + 255) / 256, b = tid * C, e = min(W, b + C);
  // brk(i): pixel i starts a run (invalid pixels are runs of their own)
  auto brk = [&](int i) { return i == 0 || !(val[i] >= 0) || !(val[i - 1] >= 0) || !(fabsf(val[i] - val[i - 1]) <= thr); };
  int l = -1;
  for (int i = b; i < e; ++i) if (brk(i)) l = i;
  cs[tid] = l;
  __syncthreads();
  if (tid == 0) {
    int carry = 0;
    for (int t = 0; t < 256; ++t) { const int own = cs[t]; cs[t] = carry; if (own >= 0) carry = own; }
  }
  __syncthreads();
  l = cs[tid];
  for (int i = b; i < e; ++i) { if (brk(i)) l = i; start[i] = (int16_t)l; }
  __syncthreads();
  for (int i = tid; i < W; i += 256) {
    const int st = start[i];
    L[v * W + i] = v * W + st;
    size[v * W + i] = 0;
    rlen[v * W + i] = 0;
  }
  __syncthreads();
  for (int i = tid; i < W; i += 256)
    if (i == W - 1 || start[i + 1] != start[i]) rlen[v * W + start[i]] = i - start[i] + 1;
}
__device__ __forceinline__ void d_cc_merge(const float* D, int W, int H, float thr, int32_t* L) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
  if (u >= W || v + 1 >= H) return;
  const int i = v * W + u;
  const float d = D[i], e = D[i + W];
  if (!(d >= 0 && e >= 0 && fabsf(d - e) <= thr)) return;
  if (u > 0) {   // both runs continue from u-1 and already touch there: that column makes the union
    const float d0 = D[i - 1], e0 = D[i - 1 + W];
    if (d0 >= 0 && e0 >= 0 && fabsf(d - d0) <= thr && fabsf(e - e0) <= thr && fabsf(d0 - e0) <= thr) return;
  }
  cc_union(L, i, i + W);
}
__device__ __forceinline__ void d_cc_count(int n, const int32_t* L, const int32_t* rlen, int32_t* size) {
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  int root = -1, len = 0;
  if (i < n && rlen[i] > 0) { root = cc_find(L, i); len = rlen[i]; }   // run heads only
  // neighbouring runs mostly share one root (the big surfaces): add them up per wave first, one atomic
  // per distinct root, instead of tens of thousands of atomics on the same address
  uint64_t active = __ballot(root >= 0);
  while (active) {
    const int leader = __ffsll((unsigned long long)active) - 1;
    const int lr = __builtin_amdgcn_readlane(root, leader);
    const bool same = root == lr;
    const int sum = wave_sum_i32_dpp(same ? len : 0);
    if (lane == leader) atomicAdd(&size[lr], sum);
    active &= ~__ballot(same);
  }
}
__device__ __forceinline__ void d_cc_apply(float* D, int n, const int32_t* L, const int32_t* size, int speckle) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (size[cc_find(L, i)] < speckle) D[i] = -10.0f;
}

// Strips: the same labelling, but the rows are cut into strips of CCS_ROWS rows and one workgroup unites
// everything INSIDE its strip in LDS (run starts per row, vertical contacts between the strip's rows, LDS
// atomics); only the contacts across strip borders are left for the device-wide union-find - an eighth of
// the global atomics on the hot roots, which is what the per-pixel version spends its time on.  What the
// strip kernel leaves behind has the shape the later kernels expect: every pixel points at the root pixel
// of its strip component, that root carries the component's pixel count in `rlen` (others 0).
#define CCS_ROWS 8
__device__ __forceinline__ int lds_find(const int32_t* lab, int i) {
  int p = lab[i];
  while (p != i) { i = p; p = lab[i]; }
  return i;
}
__device__ __forceinline__ void lds_union(int32_t* lab, int a, int b) {
  bool done;
  do {
    a = lds_find(lab, a); b = lds_find(lab, b);
    if (a < b) { const int old = atomicMin(&lab[b], a); done = old == b; b = old; }
    else if (b < a) { const int old = atomicMin(&lab[a], b); done = old == a; a = old; }
    else done = true;
  } while (!done);
}
__device__ __forceinline__ void d_cc_strip(const float* D, int W, int H, float thr, int32_t* L, int32_t* rlen, int32_t* size) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ccs_smem[];
  float* val = reinterpret_cast<float*>(ccs_smem);                           // [rows][W], later the counters
  int32_t* lab = reinterpret_cast<int32_t*>(ccs_smem) + (size_t)CCS_ROWS * W;   // [rows][W]
  __shared__ int carry[CCS_ROWS][32];
  const int tid = threadIdx.x, v0 = blockIdx.x * CCS_ROWS, rows = min(CCS_ROWS, H - v0), n = rows * W;
  // a strip is 8 * W floats: its byte offset is a multiple of 16, so it moves as float4 (then the tail)
  if ((reinterpret_cast<uintptr_t>(D) & 15) == 0) {
    const int n4 = n / 4;
    stage_lds<5>(reinterpret_cast<float4*>(val), reinterpret_cast<const float4*>(D + (size_t)v0 * W), n4, 1, tid);
    if (tid < n - 4 * n4) val[4 * n4 + tid] = D[(size_t)v0 * W + 4 * n4 + tid];
  } else {
    stage_lds<8>(val, D + (size_t)v0 * W, n, 1, tid);
  }
  __syncthreads();
  // run starts: 32 threads per row, each a chunk; a run start is a break = first pixel, an invalid pixel, or
  // a jump of more than thr against the left neighbour
  const int r = tid >> 5, k = tid & 31, chunk = (W + 31) / 32, cb = k * chunk, ce = min(W, cb + chunk);
  auto brk = [&](int row, int x) {
    const float c = val[row * W + x];
    if (x == 0 || !(c >= 0)) return true;
    const float l = val[row * W + x - 1];
    return !(l >= 0) || !(fabsf(c - l) <= thr);
  };
  int last = -1;
  if (r < rows) for (int x = cb; x < ce; ++x) if (brk(r, x)) last = x;
  carry[r][k] = last;
  __syncthreads();
  if (k == 0 && r < rows) {
    int c = 0;
    for (int q = 0; q < 32; ++q) { const int own = carry[r][q]; carry[r][q] = c; if (own >= 0) c = own; }
  }
  __syncthreads();
  if (r < rows) {
    int st = carry[r][k];
    for (int x = cb; x < ce; ++x) { if (brk(r, x)) st = x; lab[r * W + x] = r * W + st; }
  }
  __syncthreads();
  // vertical contacts inside the strip (first column of each run-to-run contact)
  for (int i = tid; i < n - W; i += 256) {
    const int x = i % W;
    const float d = val[i], e = val[i + W];
    if (!(d >= 0 && e >= 0 && fabsf(d - e) <= thr)) continue;
    if (x > 0) {
      const float d0 = val[i - 1], e0 = val[i - 1 + W];
      if (d0 >= 0 && e0 >= 0 && fabsf(d - d0) <= thr && fabsf(e - e0) <= thr && fabsf(d0 - e0) <= thr) continue;
    }
    lds_union(lab, lab[i] == i ? i : lab[i], lab[i + W] == i + W ? i + W : lab[i + W]);
  }
  __syncthreads();
  int32_t* cnt = reinterpret_cast<int32_t*>(val);   // the disparities are not needed any more
  for (int i = tid; i < n; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) atomicAdd(&cnt[lds_find(lab, i)], 1);
  __syncthreads();
  const int g0 = v0 * W;
  for (int i = tid; i < n; i += 256) {
    const int root = lds_find(lab, i);
    L[g0 + i] = g0 + root;
    size[g0 + i] = 0;
    rlen[g0 + i] = root == i ? cnt[i] : 0;
  }
}
// the contacts the strips could not see: between the last row of a strip and the first row of the next
__device__ __forceinline__ void d_cc_merge_borders(const float* D, int W, int H, float thr, int32_t* L) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = (blockIdx.y + 1) * CCS_ROWS - 1;
  if (u >= W || v + 1 >= H) return;
  const int i = v * W + u;
  const float d = D[i], e = D[i + W];
  if (!(d >= 0 && e >= 0 && fabsf(d - e) <= thr)) return;
  if (u > 0) {
    const float d0 = D[i - 1], e0 = D[i - 1 + W];
    if (d0 >= 0 && e0 >= 0 && fabsf(d - d0) <= thr && fabsf(e - e0) <= thr && fabsf(d0 - e0) <= thr) return;
  }
  cc_union(L, i, i + W);
}

// ------------------------------------------------------------------------------------------------
// gapInterpolation, one pass over lines (rows: estride 1, lstride W; columns: estride W, lstride 1).
// A maximal run of invalid pixels strictly inside a line with 1 <= length <= gap is filled with the
// mean of its two valid neighbours (|d1-d2| < 3) or their minimum; with add_corners the leading and
// trailing runs are filled up to `gap` pixels from the first / last valid pixel.
// ------------------------------------------------------------------------------------------------
#define GAP_MAXLEN 4096
__device__ __forceinline__ void d_elas_gap(float* D, int n, int estride, int lstride, int gap, int add_corners) {
  __shared__ float val[GAP_MAXLEN];
  __shared__ int16_t last[GAP_MAXLEN], nxt[GAP_MAXLEN];
  __shared__ int cl[256], cn[256];
  float* line = D + (size_t)blockIdx.x * lstride;
  const int tid = threadIdx.x;
  stage_lds<8>(val, const_cast<const float*>(line), n, (size_t)estride, tid);
  __syncthreads();
  const int C = (n + 255) / 256, b = tid * C, e = min(n, b + C);
  int l = -1, x = n;
  for (int i = b; i < e; ++i) if (val[i] >= 0) l = i;
  for (int i = e - 1; i >= b; --i) if (val[i] >= 0) x = i;
  cl[tid] = l; cn[tid] = x;
  __syncthreads();
  if (tid == 0) {
    int carry = -1;
    for (int t = 0; t < 256; ++t) { const int own = cl[t]; cl[t] = carry; if (own >= 0) carry = own; }
    carry = n;
    for (int t = 255; t >= 0; --t) { const int own = cn[t]; cn[t] = carry; if (own < n) carry = own; }
  }
  __syncthreads();
  l = cl[tid];
  for (int i = b; i < e; ++i) { last[i] = (int16_t)l; if (val[i] >= 0) l = i; }
  x = cn[tid];
  for (int i = e - 1; i >= b; --i) { nxt[i] = (int16_t)x; if (val[i] >= 0) x = i; }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    if (val[i] >= 0) continue;
    const int lo = last[i], hi = nxt[i];
    if (lo >= 0 && hi < n) {
      if (hi - lo - 1 <= gap) {
        const float d1 = val[lo], d2 = val[hi];
        line[(size_t)i * estride] = fabsf(d1 - d2) < 3.0f ? (d1 + d2) / 2 : fminf(d1, d2);
      }
    } else if (add_corners) {
      if (lo < 0 && hi < n) { if (i >= hi - gap) line[(size_t)i * estride] = val[hi]; }
      else if (hi >= n && lo >= 0) { if (i <= lo + gap) line[(size_t)i * estride] = val[lo]; }
    }
  }
}

// The column pass of gapInterpolation on a tile of C adjacent columns (C a power of two, chosen so that the
// tile fits the LDS): rows of the tile are read and written as C consecutive floats - the line-per-block
// kernel above reads a column with a stride of one image row per element.  256 threads = C columns x
// (256 / C) row chunks; same fill rule.
__device__ __forceinline__ void d_elas_gap_cols(float* D, int W, int H, int C, int gap, int add_corners) {
  extern __shared__ unsigned char gap_smem[];
  float* val = reinterpret_cast<float*>(gap_smem);                    // [H][C]
  int16_t* nxt = reinterpret_cast<int16_t*>(val + (size_t)H * C);     // [H][C]
  __shared__ int cl[256], cn[256];
  const int tid = threadIdx.x, c = tid & (C - 1), t = tid / C, T = 256 / C;
  const int x0 = blockIdx.x * C, x = x0 + c;
  const bool col_ok = x < W;
  for (int v = t; v < H; v += T) val[v * C + c] = col_ok ? D[(size_t)v * W + x] : -10.0f;
  __syncthreads();
  const int R = (H + T - 1) / T, b = t * R, e = min(H, b + R);
  int l = -1, nx = H;
  for (int v = b; v < e; ++v) if (val[v * C + c] >= 0) l = v;
  for (int v = e - 1; v >= b; --v) if (val[v * C + c] >= 0) nx = v;
  cl[t * C + c] = l; cn[t * C + c] = nx;
  __syncthreads();
  if (t == 0) {   // one thread per column turns the chunk summaries into carries
    int carry = -1;
    for (int k = 0; k < T; ++k) { const int own = cl[k * C + c]; cl[k * C + c] = carry; if (own >= 0) carry = own; }
    carry = H;
    for (int k = T - 1; k >= 0; --k) { const int own = cn[k * C + c]; cn[k * C + c] = carry; if (own < H) carry = own; }
  }
  __syncthreads();
  nx = cn[t * C + c];
  for (int v = e - 1; v >= b; --v) { nxt[v * C + c] = (int16_t)nx; if (val[v * C + c] >= 0) nx = v; }
  l = cl[t * C + c];
  for (int v = b; v < e; ++v) {
    const float cur = val[v * C + c];
    if (cur >= 0) { l = v; continue; }
    if (!col_ok) continue;
    const int lo = l, hi = nxt[v * C + c];
    float* dst = D + (size_t)v * W + x;
    if (lo >= 0 && hi < H) {
      if (hi - lo - 1 <= gap) {
        const float d1 = val[lo * C + c], d2 = val[hi * C + c];
        *dst = fabsf(d1 - d2) < 3.0f ? (d1 + d2) / 2 : fminf(d1, d2);
      }
    } else if (add_corners) {
      if (lo < 0 && hi < H) { if (v >= hi - gap) *dst = val[hi * C + c]; }
      else if (hi >= H && lo >= 0) { if (v <= lo + gap) *dst = val[lo * C + c]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// adaptiveMean: 8-tap "bilateral" mean along rows, then along columns, in the reference's SSE lane
// order: window pixel q sits in slot q % 8, lane i adds slots i and i+4, lanes are summed 0..3.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float am_weight(float val, float cur) {
  const float a = __uint_as_float(__float_as_uint(val - cur) & 0x4F000000u);
  return fmaxf(0.0f, 4.0f - a);
}
__device__ __forceinline__ bool am_filter(const float (&slot)[8], float cur, float* out) {
  float w[8], f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { w[i] = am_weight(slot[i], cur); f[i] = slot[i] * w[i]; }
  const float ws = (((w[0] + w[4]) + (w[1] + w[5])) + (w[2] + w[6])) + (w[3] + w[7]);
  const float fs = (((f[0] + f[4]) + (f[1] + f[5])) + (f[2] + f[6])) + (f[3] + f[7]);
  if (ws > 0) {
    const float d = fs / ws;
    if (d >= 0) { *out = d; return true; }
  }
  return false;
}
// subsampled maps use a 4-tap window (elas.cpp:1322-1385): pixel q in slot q % 4, slots summed 0..3,
// window x-2 .. x+1 around the output pixel x
__device__ __forceinline__ bool am_filter4(const float (&slot)[4], float cur, float* out) {
  float w[4], f[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { w[i] = am_weight(slot[i], cur); f[i] = slot[i] * w[i]; }
  const float ws = ((w[0] + w[1]) + w[2]) + w[3];
  const float fs = ((f[0] + f[1]) + f[2]) + f[3];
  if (ws > 0) {
    const float d = fs / ws;
    if (d >= 0) { *out = d; return true; }
  }
  return false;
}
// horizontal: T = filtered copy of D (invalid -> -10, not written -> 0)
__device__ __forceinline__ void d_elas_mean_h(const float* D, int W, int H, int half, float* T) {
  const int x = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
  if (x >= W) return;
  const float* row = D + (size_t)v * W;
  float out = row[x] < 0 ? -10.0f : 0.0f;
  if (half) {
    if (v >= 3 && v < H - 3 && x >= 2 && x <= W - 2) {
      float slot[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = x - 2 + k;
        const float a = row[q];
        slot[q & 3] = a < 0 ? -10.0f : a;
      }
      const float c = row[x];
      am_filter4(slot, c < 0 ? -10.0f : c, &out);
    }
  } else if (v >= 3 && v < H - 3 && x >= 4 && x <= W - 4) {
    float slot[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int q = x - 4 + k;
      const float a = row[q];
      slot[q & 7] = a < 0 ? -10.0f : a;
    }
    const float c = row[x];
    am_filter(slot, c < 0 ? -10.0f : c, &out);
  }
  T[(size_t)v * W + x] = out;
}
__device__ __forceinline__ void d_elas_mean_v(const float* T, int W, int H, int half, float* D) {
  const int u = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (u >= W) return;
  if (half) {
    if (!(u >= 3 && u < W - 3 && y >= 2 && y <= H - 2)) return;
    float slot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = y - 2 + k;
      slot[q & 3] = T[(size_t)q * W + u];
    }
    float out;
    if (am_filter4(slot, T[(size_t)y * W + u], &out)) D[(size_t)y * W + u] = out;
    return;
  }
  if (!(u >= 3 && u < W - 3 && y >= 4 && y <= H - 4)) return;
  float slot[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int q = y - 4 + k;
    slot[q & 7] = T[(size_t)q * W + u];
  }
  float out;
  if (am_filter(slot, T[(size_t)y * W + u], &out)) D[(size_t)y * W + u] = out;
}

__device__ __forceinline__ float median7(float (&a)[7]) {
#pragma unroll
  for (int i = 1; i < 7; ++i)
#pragma unroll
    for (int j = 6; j >= 1; --j)
      if (j >= i) { const float lo = fminf(a[j - 1], a[j]), hi = fmaxf(a[j - 1], a[j]); a[j - 1] = lo; a[j] = hi; }
  return a[3];
}
// horizontal median into T (zero elsewhere), vertical median of T back into D
__device__ __forceinline__ void d_elas_median_h(const float* D, int W, int H, float* T) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  float out = 0.0f;
  if (u >= 3 && u < W - 3 && v >= 3 && v < H - 3) {
    const float c = D[(size_t)v * W + u];
    out = c;
    if (c >= 0) {
      float a[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) a[k] = D[(size_t)v * W + u - 3 + k];
      out = median7(a);
    }
  }
  T[(size_t)v * W + u] = out;
}
__device__ __forceinline__ void d_elas_median_v(const float* T, int W, int H, float* D) {
  const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
  if (u >= W) return;
  if (!(u >= 3 && u < W - 3 && v >= 3 && v < H - 3)) return;
  if (!(D[(size_t)v * W + u] >= 0)) return;
  float a[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) a[k] = T[(size_t)(v - 3 + k) * W + u];
  D[(size_t)v * W + u] = median7(a);
}

// ------------------------------------------------------------------------------------------------
// Plane fit per triangle (elas.cpp:505-577): d = a*u + b*v + c through the three corners, once in left
// image coordinates (t1) and once in right image coordinates (t2), solved as Matrix::solve does
// (matrix.cpp:414-503): Gauss-Jordan with full pivoting in float64, pivot = LAST largest |a| among the
// rows / columns not used yet, singular below 1e-20 -> plane (0, 0, 0).
// ------------------------------------------------------------------------------------------------
__device__ bool solve3(double (&A)[3][3], double (&b)[3]) {
  int ipiv[3] = {0, 0, 0};
  for (int i = 0; i < 3; ++i) {
    double big = 0.0;
    int irow = 0, icol = 0;
    for (int j = 0; j < 3; ++j)
      if (ipiv[j] != 1)
        for (int k = 0; k < 3; ++k)
          if (ipiv[k] == 0 && fabs(A[j][k]) >= big) { big = fabs(A[j][k]); irow = j; icol = k; }
    ++ipiv[icol];
    if (irow != icol) {
      for (int l = 0; l < 3; ++l) { const double t = A[irow][l]; A[irow][l] = A[icol][l]; A[icol][l] = t; }
      const double t = b[irow]; b[irow] = b[icol]; b[icol] = t;
    }
    if (fabs(A[icol][icol]) < 1e-20) return false;
    const double pivinv = 1.0 / A[icol][icol];
    A[icol][icol] = 1.0;
    for (int l = 0; l < 3; ++l) A[icol][l] *= pivinv;
    b[icol] *= pivinv;
    for (int ll = 0; ll < 3; ++ll)
      if (ll != icol) {
        const double dum = A[ll][icol];
        A[ll][icol] = 0.0;
        for (int l = 0; l < 3; ++l) A[ll][l] -= A[icol][l] * dum;
        b[ll] -= b[icol] * dum;
      }
  }
  return true;
}
__device__ __forceinline__ void d_elas_planes(const int32_t* sp, const int32_t* tri0, const int32_t* tri1, int n0,
                                                    int n1, float* pl0, float* pl1) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int32_t* tri = blockIdx.y ? tri1 : tri0;
  float* pl = blockIdx.y ? pl1 : pl0;
  if (t >= (blockIdx.y ? n1 : n0)) return;
  int cu[3], cv[3], cd[3];
  for (int k = 0; k < 3; ++k) { const int c = tri[3 * t + k]; cu[k] = sp[3 * c]; cv[k] = sp[3 * c + 1]; cd[k] = sp[3 * c + 2]; }
  for (int side = 0; side < 2; ++side) {
    double A[3][3], b[3];
    for (int k = 0; k < 3; ++k) {
      A[k][0] = side ? cu[k] - cd[k] : cu[k]; A[k][1] = cv[k]; A[k][2] = 1;
      b[k] = cd[k];
    }
    const bool ok = solve3(A, b);
    for (int k = 0; k < 3; ++k) pl[6 * t + 3 * side + k] = ok ? (float)b[k] : 0.0f;
  }
}

// ------------------------------------------------------------------------------------------------
// Disparity grid (elas.cpp:579-658) as one 256-bit set per cell (disp_max <= 255).  mark: every support
// point sets d-1..d+1 in its cell; diffuse: the reference's nine marching pointers run over the FLAT
// (cell, d) array, whole cells apart, so per disparity they OR the flat CELL indices c + {0,1,2, gw..gw+2,
// 2gw..2gw+2} into cell c + gw + 1 for every c with c + 2gw + 2 < gw*gh (rows wrap); other cells stay empty.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_elas_grid_mark(const int32_t* sp, int nsp, int grid_size, int disp_max, int gw, int gh,
                                 uint32_t* t1_left, uint32_t* t1_right) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nsp) return;
  const int u = sp[3 * i], v = sp[3 * i + 1], d0 = sp[3 * i + 2];
  const int y = (int)floorf((float)v / (float)grid_size);
  const int xl = (int)floorf((float)(u / grid_size));
  const int xr = (int)floorf((float)(u - d0) / (float)grid_size);
  for (int d = max(d0 - 1, 0); d <= min(d0 + 1, disp_max); ++d) {
    if (xl >= 0 && xl < gw && y >= 0 && y < gh) atomicOr(&t1_left[(size_t)(y * gw + xl) * 8 + (d >> 5)], 1u << (d & 31));
    if (xr >= 0 && xr < gw && y >= 0 && y < gh) atomicOr(&t1_right[(size_t)(y * gw + xr) * 8 + (d >> 5)], 1u << (d & 31));
  }
}
__device__ __forceinline__ void d_elas_grid_diffuse(const uint32_t* t1_left, const uint32_t* t1_right, int gw, int gh,
                                    uint32_t* t2_left, uint32_t* t2_right) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // (cell, word)
  const int ncell = gw * gh;
  if (i >= ncell * 8) return;
  const uint32_t* t1 = blockIdx.y ? t1_right : t1_left;
  uint32_t* t2 = blockIdx.y ? t2_right : t2_left;
  const int cell = i >> 3, w = i & 7, c = cell - gw - 1;
  uint32_t r = 0;
  if (c >= 0 && c + 2 * gw + 2 < ncell) {
    const int o[9] = {0, 1, 2, gw, gw + 1, gw + 2, 2 * gw, 2 * gw + 1, 2 * gw + 2};
#pragma unroll
    for (int k = 0; k < 9; ++k) r |= t1[(size_t)(c + o[k]) * 8 + w];
  }
  t2[i] = r;
}

// ------------------------------------------------------------------------------------------------
// Launch side: every kernel serves a whole batch.  `tab` holds, per pair, the pointers of its buffers and
// the sizes of its lists; the pair index rides on the highest grid dimension the stage leaves free.  A
// pair with fewer than 3 support points (`produced` = 0) is skipped by everything after the lattice stage.
// ------------------------------------------------------------------------------------------------
struct ElasTab {
  const uint8_t *imgL, *imgR;
  uint4 *desc0, *desc1;
  int16_t* can;
  int32_t *sp, *tri0, *tri1;
  float *plane0, *plane1;
  uint32_t *bits0, *bits1;      // per side: set t1 (support point cells) then set t2 (after diffusion)
  int32_t *owner0, *owner1;
  float *raw0, *raw1, *out0, *out1;
  int32_t *lab, *size;
  int32_t nsp, nt0, nt1, produced;
};

// before phase B: owner maps to "no triangle", support-point cell sets to empty
__global__ void k_elas_clear(const ElasTab* tab, int nd, int nbits) {
  const ElasTab& E = tab[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (!E.produced || i >= nd) return;
  E.owner0[i] = -1; E.owner1[i] = -1;
  if (i < nbits) { E.bits0[i] = 0; E.bits1[i] = 0; }
}
__global__ __launch_bounds__(256) void k_elas_desc(const ElasTab* tab, int pitch, int W, int H, int half) {
  const ElasTab& E = tab[blockIdx.z >> 1];
  d_elas_desc(blockIdx.z & 1, E.imgL, E.imgR, pitch, W, H, half, E.desc0, E.desc1);
}
__global__ __launch_bounds__(256) void k_elas_support(const ElasTab* tab, int W, int H, int Wc, int Hc, svo_elas_params p) {
  const ElasTab& E = tab[blockIdx.y];
  d_elas_support(E.desc0, E.desc1, W, H, Wc, Hc, p, E.can);
}
__global__ void k_elas_grid_mark(const ElasTab* tab, int grid_size, int disp_max, int gw, int gh) {
  const ElasTab& E = tab[blockIdx.y];
  if (!E.produced) return;
  d_elas_grid_mark(E.sp, E.nsp, grid_size, disp_max, gw, gh, E.bits0, E.bits1);
}
__global__ void k_elas_grid_diffuse(const ElasTab* tab, int gw, int gh) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  const size_t half_set = (size_t)gw * gh * 8;
  d_elas_grid_diffuse(E.bits0, E.bits1, gw, gh, E.bits0 + half_set, E.bits1 + half_set);
}
__global__ __launch_bounds__(64) void k_elas_planes(const ElasTab* tab) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_elas_planes(E.sp, E.tri0, E.tri1, E.nt0, E.nt1, E.plane0, E.plane1);
}
__global__ __launch_bounds__(256) void k_elas_raster(const ElasTab* tab, int W, int H, int sub) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_elas_raster(E.sp, E.tri0, E.tri1, E.nt0, E.nt1, W, H, sub, E.owner0, E.owner1);
}
__global__ __launch_bounds__(256) void k_elas_match(const ElasTab* tab, const int32_t* P, int W, int H, int gw, int gh,
                                                    int gd, int plane_radius, svo_elas_params p) {
  const ElasTab& E = tab[blockIdx.z >> 1];
  if (!E.produced) return;
  const size_t half_set = (size_t)gw * gh * 8;
  d_elas_match((blockIdx.z & 1) != 0, E.desc0, E.desc1, E.owner0, E.owner1, E.plane0, E.plane1,
               reinterpret_cast<const int32_t*>(E.bits0 + half_set), reinterpret_cast<const int32_t*>(E.bits1 + half_set), P,
               W, H, gw, gd, plane_radius, p, E.raw0, E.raw1);
}
__global__ __launch_bounds__(256) void k_elas_lr(const ElasTab* tab, int W, int H, int lr_threshold, int sub) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_elas_lr(E.raw0, E.raw1, W, H, lr_threshold, sub, E.out0, E.out1);
}
// segment removal works on one side at a time; the run lengths borrow owner0 (free once matching is done)
__global__ __launch_bounds__(256) void k_cc_rows(const ElasTab* tab, int side, int W, float thr) {
  const ElasTab& E = tab[blockIdx.y];
  if (!E.produced) return;
  d_cc_rows(side ? E.out1 : E.out0, W, thr, E.lab, E.owner0, E.size);
}
__global__ void k_cc_merge(const ElasTab* tab, int side, int W, int H, float thr) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_cc_merge(side ? E.out1 : E.out0, W, H, thr, E.lab);
}
__global__ __launch_bounds__(256) void k_cc_count(const ElasTab* tab, int n) {
  const ElasTab& E = tab[blockIdx.y];
  if (!E.produced) return;
  d_cc_count(n, E.lab, E.owner0, E.size);
}
__global__ __launch_bounds__(256) void k_cc_strip(const ElasTab* tab, int side, int W, int H, float thr) {
  const ElasTab& E = tab[blockIdx.y];
  if (!E.produced) return;
  d_cc_strip(side ? E.out1 : E.out0, W, H, thr, E.lab, E.owner0, E.size);
}
__global__ void k_cc_merge_borders(const ElasTab* tab, int side, int W, int H, float thr) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_cc_merge_borders(side ? E.out1 : E.out0, W, H, thr, E.lab);
}
__global__ void k_cc_apply(const ElasTab* tab, int side, int n, int speckle) {
  const ElasTab& E = tab[blockIdx.y];
  if (!E.produced) return;
  d_cc_apply(side ? E.out1 : E.out0, n, E.lab, E.size, speckle);
}
__global__ __launch_bounds__(256) void k_elas_gap(const ElasTab* tab, int side, int n, int estride, int lstride, int gap,
                                                  int add_corners) {
  const ElasTab& E = tab[blockIdx.y];
  if (!E.produced) return;
  d_elas_gap(side ? E.out1 : E.out0, n, estride, lstride, gap, add_corners);
}
__global__ __launch_bounds__(256) void k_elas_gap_cols(const ElasTab* tab, int side, int W, int H, int C, int gap,
                                                       int add_corners) {
  const ElasTab& E = tab[blockIdx.y];
  if (!E.produced) return;
  d_elas_gap_cols(side ? E.out1 : E.out0, W, H, C, gap, add_corners);
}
// filters: maps in out*, scratch in raw*
__global__ __launch_bounds__(256) void k_elas_mean_h(const ElasTab* tab, int side, int W, int H, int half) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_elas_mean_h(side ? E.out1 : E.out0, W, H, half, side ? E.raw1 : E.raw0);
}
__global__ __launch_bounds__(256) void k_elas_mean_v(const ElasTab* tab, int side, int W, int H, int half) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_elas_mean_v(side ? E.raw1 : E.raw0, W, H, half, side ? E.out1 : E.out0);
}
__global__ __launch_bounds__(256) void k_elas_median_h(const ElasTab* tab, int side, int W, int H) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_elas_median_h(side ? E.out1 : E.out0, W, H, side ? E.raw1 : E.raw0);
}
__global__ __launch_bounds__(256) void k_elas_median_v(const ElasTab* tab, int side, int W, int H) {
  const ElasTab& E = tab[blockIdx.z];
  if (!E.produced) return;
  d_elas_median_v(side ? E.raw1 : E.raw0, W, H, side ? E.out1 : E.out0);
}

// ------------------------------------------------------------------------------------------------
// Host stages
// ------------------------------------------------------------------------------------------------
struct SupportPt { int32_t u, v, d; };

void remove_inconsistent(std::vector<int16_t>& D, int Wc, int Hc, const svo_elas_params& p) {
  for (int u = 0; u < Wc; ++u)
    for (int v = 0; v < Hc; ++v) {
      const int d = D[v * Wc + u];
      if (d < 0) continue;
      int support = 0;   // only compared with the threshold, so counting stops there
      const int u_lo = std::max(u - p.incon_window_size, 0), u_hi = std::min(u + p.incon_window_size, Wc - 1);
      const int v_lo = std::max(v - p.incon_window_size, 0), v_hi = std::min(v + p.incon_window_size, Hc - 1);
      for (int v2 = v_lo; v2 <= v_hi && support < p.incon_min_support; ++v2)
        for (int u2 = u_lo; u2 <= u_hi; ++u2) {
          const int d2 = D[v2 * Wc + u2];
          if (d2 >= 0 && abs(d - d2) <= p.incon_threshold) ++support;
        }
      if (support < p.incon_min_support) D[v * Wc + u] = -1;
    }
}

void remove_redundant(std::vector<int16_t>& D, int Wc, int Hc, int max_dist, int thr, bool vertical) {
  const int du[2] = {vertical ? 0 : -1, vertical ? 0 : 1}, dv[2] = {vertical ? -1 : 0, vertical ? 1 : 0};
  for (int u = 0; u < Wc; ++u)
    for (int v = 0; v < Hc; ++v) {
      const int d = D[v * Wc + u];
      if (d < 0) continue;
      bool redundant = true;
      for (int i = 0; i < 2 && redundant; ++i) {
        int u2 = u, v2 = v;
        bool support = false;
        for (int j = 0; j < max_dist; ++j) {
          u2 += du[i]; v2 += dv[i];
          if (u2 < 0 || v2 < 0 || u2 >= Wc || v2 >= Hc) break;
          const int d2 = D[v2 * Wc + u2];
          if (d2 >= 0 && abs(d - d2) <= thr) { support = true; break; }
        }
        if (!support) redundant = false;
      }
      if (redundant) D[v * Wc + u] = -1;
    }
}

void add_corner_points(std::vector<SupportPt>& sp, int W, int H) {
  SupportPt b[6] = {{0, 0, 0}, {0, H - 1, 0}, {W - 1, 0, 0}, {W - 1, H - 1, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < 4; ++i) {
    int best = 10000000;
    for (const SupportPt& s : sp) {
      const int du = b[i].u - s.u, dv = b[i].v - s.v, dist = du * du + dv * dv;
      if (dist < best) { best = dist; b[i].d = s.d; }
    }
  }
  b[4] = {b[2].u + b[2].d, b[2].v, b[2].d};
  b[5] = {b[3].u + b[3].d, b[3].v, b[3].d};
  for (int i = 0; i < 6; ++i) sp.push_back(b[i]);
}

template <typename T>
int dev_alloc(svo_ctx* ctx, ElasState* st, T** p, size_t count) {
  void* q = nullptr;
  SVO_HIP(ctx, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  st->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return SVO_OK;
}

// host_io: also the staging buffers of the host-buffer API (images in HBM, pinned in/out copies)
int elas_prepare(svo_ctx* ctx, ElasState* st, int W, int H, const svo_elas_params& p, bool host_io) {
  int step = p.candidate_stepsize;
  if (p.subsampling) step += step % 2;   // elas.cpp:379-381: at half resolution only every second line exists
  const int Wc = (W + step - 1) / step, Hc = (H + step - 1) / step;
  const int gw = (int)ceil((float)W / (float)p.grid_size), gh = (int)ceil((float)H / (float)p.grid_size);
  const int gd = p.disp_max + 2;
  const size_t n = (size_t)W * H;
  if (st->W == W && st->H == H && st->Wc == Wc && st->Hc == Hc && st->gw == gw && st->gh == gh && st->gd == gd) {
    if (host_io && !st->h_img) {
      int rc;
      for (int s = 0; s < 2; ++s)
        if ((rc = dev_alloc(ctx, st, &st->d_img[s], n))) return rc;
      SVO_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&st->h_img), 2 * n, hipHostMallocDefault));
      SVO_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&st->h_D), 2 * n * sizeof(float), hipHostMallocDefault));
    }
    return SVO_OK;
  }
  st->release();
  st->cap_sp = Wc * Hc + 8;
  st->cap_tri = 2 * st->cap_sp + 16;
  int rc;
  for (int s = 0; s < 2; ++s) {
    if ((rc = dev_alloc(ctx, st, &st->d_desc[s], n))) return rc;
    if ((rc = dev_alloc(ctx, st, &st->d_tri[s], (size_t)st->cap_tri * 3))) return rc;
    if ((rc = dev_alloc(ctx, st, &st->d_plane[s], (size_t)st->cap_tri * 6))) return rc;
    if ((rc = dev_alloc(ctx, st, &st->d_grid[s], (size_t)gw * gh * 16))) return rc;   // two 256-bit sets per cell
    if ((rc = dev_alloc(ctx, st, &st->d_owner[s], n))) return rc;
    if ((rc = dev_alloc(ctx, st, &st->d_D[s], n))) return rc;
    if ((rc = dev_alloc(ctx, st, &st->d_T[s], n))) return rc;
  }
  if ((rc = dev_alloc(ctx, st, &st->d_can, (size_t)Wc * Hc))) return rc;
  if ((rc = dev_alloc(ctx, st, &st->d_sp, (size_t)st->cap_sp * 3))) return rc;
  if ((rc = dev_alloc(ctx, st, &st->d_P, 256))) return rc;
  if ((rc = dev_alloc(ctx, st, &st->d_lab, n))) return rc;
  if ((rc = dev_alloc(ctx, st, &st->d_size, n))) return rc;
  {
    void* q = nullptr;
    SVO_HIP(ctx, hipMalloc(&q, 1024));   // >= sizeof(ElasTab)
    st->allocs.push_back(q);
    st->d_tab = reinterpret_cast<struct ElasTab*>(q);
  }
  SVO_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&st->h_can), (size_t)Wc * Hc * sizeof(int16_t), hipHostMallocDefault));
  st->W = W; st->H = H; st->Wc = Wc; st->Hc = Hc; st->gw = gw; st->gh = gh; st->gd = gd;
  return host_io ? elas_prepare(ctx, st, W, H, p, true) : SVO_OK;   // second pass adds the staging buffers
}

int tap(svo_ctx* ctx, hipStream_t s, float* dst, const float* src, size_t n) {
  if (!dst) return SVO_OK;
  SVO_HIP(ctx, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToHost, s));
  SVO_HIP(ctx, hipStreamSynchronize(s));
  return SVO_OK;
}

}  // namespace

extern "C" int svo_elas_default_params(int32_t setting, svo_elas_params* q) {
  if (!q || (setting != 0 && setting != 1)) return SVO_E_INVALID;
  const bool mb = setting == 1;   // elas.h:87-142
  q->disp_min = 0; q->disp_max = 255; q->support_threshold = mb ? 0.95f : 0.85f; q->support_texture = 10;
  q->candidate_stepsize = 5; q->incon_window_size = 5; q->incon_threshold = 5; q->incon_min_support = 5;
  q->add_corners = mb ? 1 : 0; q->grid_size = 20; q->beta = 0.02f; q->gamma = mb ? 5.0f : 3.0f; q->sigma = 1.0f;
  q->sradius = mb ? 3.0f : 2.0f; q->match_texture = mb ? 0 : 1; q->lr_threshold = 2;
  q->speckle_sim_threshold = 1.0f; q->speckle_size = 200; q->ipol_gap_width = mb ? 5000 : 3;
  q->filter_median = mb ? 1 : 0; q->filter_adaptive_mean = mb ? 0 : 1; q->postprocess_only_left = mb ? 0 : 1;
  q->subsampling = 0;
  return SVO_OK;
}

namespace {

int elas_check(svo_ctx* ctx, int W, int H, int pitch, const svo_elas_params& p) {
  if ((p.subsampling != 0 && p.subsampling != 1) || W < 16 || H < 16 || W > GAP_MAXLEN || H > GAP_MAXLEN || pitch < W || p.disp_min < 0 ||
      p.disp_max < p.disp_min || p.disp_max > 255 || p.candidate_stepsize < 1 || p.grid_size < 1 ||
      p.incon_window_size < 0 || !(p.sigma * p.sradius <= 250.0f)) {   // plane band must fit the prior table
    ctx->last_error = "svo_elas_process: unsupported parameters (sizes, disparity range)";
    return SVO_E_INVALID;
  }
  return SVO_OK;
}

// What the host stages of one pair produce (kept alive until the pair's uploads have completed).
struct ElasWork {
  std::vector<SupportPt> sp;
  std::vector<int32_t> spflat, tri[2];
  const char* err = nullptr;
};

// the table entry of a pair whose buffers live in `st`
ElasTab tab_entry(const ElasState* st, const uint8_t* imgL, const uint8_t* imgR, int16_t* can, float* out0, float* out1) {
  ElasTab e;
  memset(&e, 0, sizeof e);
  e.imgL = imgL; e.imgR = imgR;
  e.desc0 = st->d_desc[0]; e.desc1 = st->d_desc[1];
  e.can = can;
  e.sp = st->d_sp; e.tri0 = st->d_tri[0]; e.tri1 = st->d_tri[1];
  e.plane0 = st->d_plane[0]; e.plane1 = st->d_plane[1];
  e.bits0 = reinterpret_cast<uint32_t*>(st->d_grid[0]); e.bits1 = reinterpret_cast<uint32_t*>(st->d_grid[1]);
  e.owner0 = st->d_owner[0]; e.owner1 = st->d_owner[1];
  e.raw0 = st->d_D[0]; e.raw1 = st->d_D[1]; e.out0 = out0; e.out1 = out1;
  e.lab = st->d_lab; e.size = st->d_size;
  e.produced = 1;
  return e;
}

// Phase A for B pairs: descriptors of both images and the lattice candidates.  Enqueues on `s`.
int elas_phase_a(svo_ctx* ctx, hipStream_t s, const ElasTab* d_tab, int B, int pitch, int W, int H, int Wc, int Hc,
                 const svo_elas_params& p) {
  svo_elas_params pk = p;   // what the kernels see: the lattice step already adjusted (elas.cpp:379-381)
  pk.candidate_stepsize = p.candidate_stepsize + (p.subsampling ? p.candidate_stepsize % 2 : 0);
  {
    SvoTimer t(ctx, "k_elas_desc");
    hipLaunchKernelGGL(k_elas_desc, dim3((W + DT_X - 1) / DT_X, (H + DT_Y - 1) / DT_Y, 2 * B), dim3(256), 0, s, d_tab,
                       pitch, W, H, p.subsampling);
  }
  if (Wc > 1 && Hc > 1) {
    SvoTimer t(ctx, "k_elas_support");
    const int ncand = (Wc - 1) * (Hc - 1);
    hipLaunchKernelGGL(k_elas_support, dim3((ncand + 3) / 4, B), dim3(256), 0, s, d_tab, W, H, Wc, Hc, pk);
  }
  return SVO_OK;
}

extern "C" int svo_elas_filter_lattice(int16_t* D, int Wc, int Hc, int incon_window, int incon_threshold, int incon_min_support,
                                       int red_max_dist, int red_threshold);

// Host stage 1: the order-dependent clean-up of the lattice candidates -> support point list.
void elas_filter(const int16_t* h_can, int Wc, int Hc, int W, int H, const svo_elas_params& p, ElasWork& w) {
  const int step = p.candidate_stepsize + (p.subsampling ? p.candidate_stepsize % 2 : 0);
  static thread_local std::vector<int16_t> can;   // (per-thread scratch: see svo_elas_delaunay on why nothing big is allocated per pair)
  can.assign(h_can, h_can + (size_t)Wc * Hc);
  // the three passes, eight lattice cells per instruction (svo_elas_filter.cc); the scalar loops serve unusual parameters
  if (svo_elas_filter_lattice(can.data(), Wc, Hc, p.incon_window_size, p.incon_threshold, p.incon_min_support, 5, 1)) {
    remove_inconsistent(can, Wc, Hc, p);
    remove_redundant(can, Wc, Hc, 5, 1, true);
    remove_redundant(can, Wc, Hc, 5, 1, false);
  }
  for (int u = 1; u < Wc; ++u)
    for (int v = 1; v < Hc; ++v)
      if (can[v * Wc + u] >= 0) w.sp.push_back({u * step, v * step, can[v * Wc + u]});
  if (p.add_corners) add_corner_points(w.sp, W, H);
  w.spflat.resize(3 * w.sp.size());
  for (size_t i = 0; i < w.sp.size(); ++i) { w.spflat[3 * i] = w.sp[i].u; w.spflat[3 * i + 1] = w.sp[i].v; w.spflat[3 * i + 2] = w.sp[i].d; }
}

// Host stage 2: the Delaunay triangulation of one image's support points (or the list a test injects).
void elas_triangulate_side(int cap_tri, const svo_elas_taps* taps, ElasWork& w, int side) {
  {
    const int32_t* tin = taps ? (side ? taps->tri2_in : taps->tri1_in) : nullptr;
    if (tin) {
      const int nt = side ? taps->n_tri2_in : taps->n_tri1_in;
      if (nt > cap_tri) { w.err = "svo_elas_process: too many triangles"; return; }
      for (int i = 0; i < 3 * nt; ++i)
        if (tin[i] < 0 || tin[i] >= (int)w.sp.size()) { w.err = "svo_elas_process: bad triangle index"; return; }
      w.tri[side].assign(tin, tin + 3 * (size_t)nt);
      return;
    }
    static thread_local std::vector<int32_t> xy;
    xy.resize(2 * w.sp.size());
    for (size_t i = 0; i < w.sp.size(); ++i) { xy[2 * i] = side ? w.sp[i].u - w.sp[i].d : w.sp[i].u; xy[2 * i + 1] = w.sp[i].v; }
    w.tri[side].resize((size_t)cap_tri * 3);
    int32_t nt = 0;
    const int r = svo_elas_delaunay(xy.data(), (int32_t)w.sp.size(), w.tri[side].data(), cap_tri, &nt);
    if (r || nt > cap_tri) { w.err = "svo_elas_process: triangulation failed"; w.tri[side].clear(); return; }
    w.tri[side].resize((size_t)nt * 3);
  }
}

// Both of them.
void elas_triangulate(int cap_tri, const svo_elas_taps* taps, ElasWork& w, bool two_threads) {
  auto do_side = [&](int side) { elas_triangulate_side(cap_tri, taps, w, side); };
  if (two_threads) {   // the two images are independent: the right one runs on a second host thread
    std::thread right_side(do_side, 1);
    do_side(0);
    right_side.join();
  } else {
    do_side(0);
    do_side(1);
  }
}

// Phase B, first part (needs the support points only): clear the per-pair grids / owner maps and build the
// disparity grids.  In the one-pair path this runs while the host triangulates.
int elas_phase_b_grids(svo_ctx* ctx, hipStream_t s, const ElasTab* d_tab, int B, int max_nsp, int gw, int gh,
                       const svo_elas_params& p) {
  SvoTimer t(ctx, "k_elas_grid");
  hipLaunchKernelGGL(k_elas_grid_mark, dim3((std::max(max_nsp, 1) + 255) / 256, B), dim3(256), 0, s, d_tab, p.grid_size,
                     p.disp_max, gw, gh);
  hipLaunchKernelGGL(k_elas_grid_diffuse, dim3((gw * gh * 8 + 255) / 256, 2, B), dim3(256), 0, s, d_tab, gw, gh);
  return SVO_OK;
}

// Phase B, second part: plane fits, dense matching and all post-processing for B pairs; the final maps are
// built in place in each pair's out0 / out1.  `tap_st` (B = 1 only): tap every intermediate of that pair,
// synchronising after every stage; otherwise it only enqueues on `s`.
int elas_phase_b(svo_ctx* ctx, hipStream_t s, const ElasTab* d_tab, int B, int max_nt, int W, int H, int gw, int gh, int gd,
                 const int32_t* d_P, const svo_elas_params& p, const ElasState* tap_st, const ElasTab* tap_e,
                 const ElasWork* tap_w, svo_elas_taps* taps) {
  const int sub = p.subsampling;
  const int Wd = sub ? W / 2 : W, Hd = sub ? H / 2 : H;   // disparity map size (elas.h:157-160)
  const size_t n = (size_t)Wd * Hd;
  const int ncell = gw * gh;
  const int plane_radius = (int)std::max((float)ceil(p.sigma * p.sradius), (float)2.0);
  const unsigned ub = (unsigned)B;
  int rc;
  if (max_nt > 0) {
    {
      SvoTimer t(ctx, "k_elas_planes");
      hipLaunchKernelGGL(k_elas_planes, dim3((max_nt + 63) / 64, 2, ub), dim3(64), 0, s, d_tab);
    }
    SvoTimer t(ctx, "k_elas_raster");
    hipLaunchKernelGGL(k_elas_raster, dim3((max_nt + 3) / 4, 2, ub), dim3(256), 0, s, d_tab, W, H, sub);
  }
  const dim3 pix((Wd + 255) / 256, Hd, ub), pix2((Wd + 255) / 256, Hd, 2 * ub);
  {
    SvoTimer t(ctx, "k_elas_match");
    hipLaunchKernelGGL(k_elas_match, pix2, dim3(256), 0, s, d_tab, d_P, W, H, gw, gh, gd, plane_radius, p);
  }
  if (taps) {
    SVO_HIP(ctx, hipStreamSynchronize(s));
    const int nt[2] = {(int)tap_w->tri[0].size() / 3, (int)tap_w->tri[1].size() / 3};
    taps->n_tri1 = nt[0]; taps->n_tri2 = nt[1];
    for (int side = 0; side < 2; ++side) {
      const int k = std::min<int>(nt[side], taps->cap_tri);
      int32_t* ti = side ? taps->tri2 : taps->tri1;
      float* pl = side ? taps->planes2 : taps->planes1;
      int32_t* gr = side ? taps->grid2 : taps->grid1;
      if (ti) memcpy(ti, tap_w->tri[side].data(), (size_t)k * 3 * sizeof(int32_t));
      if (pl && k) SVO_HIP(ctx, svo_memcpy_sync(ctx, pl, tap_st->d_plane[side], (size_t)k * 6 * sizeof(float), hipMemcpyDeviceToHost));
      if (gr) {   // the reference's list layout: per cell [count, d0, d1, ...]
        std::vector<uint32_t> bits((size_t)ncell * 8);
        SVO_HIP(ctx, svo_memcpy_sync(ctx, bits.data(), reinterpret_cast<const uint32_t*>(tap_st->d_grid[side]) + (size_t)ncell * 8,
                               bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        memset(gr, 0, (size_t)ncell * gd * sizeof(int32_t));
        for (int c = 0; c < ncell; ++c) {
          int32_t* cell = gr + (size_t)c * gd;
          int cur = 1;
          for (int q = 0; q < 8; ++q)
            for (uint32_t m = bits[(size_t)c * 8 + q]; m; m &= m - 1) cell[cur++] = 32 * q + __builtin_ctz(m);
          cell[0] = cur - 1;
        }
      }
    }
    if ((rc = tap(ctx, s, taps->D1_raw, tap_e->raw0, n))) return rc;
    if ((rc = tap(ctx, s, taps->D2_raw, tap_e->raw1, n))) return rc;
  }
  {
    SvoTimer t(ctx, "k_elas_lr");
    hipLaunchKernelGGL(k_elas_lr, pix, dim3(256), 0, s, d_tab, Wd, Hd, p.lr_threshold, sub);
  }
  if (taps) { if ((rc = tap(ctx, s, taps->D1_lr, tap_e->out0, n))) return rc; if ((rc = tap(ctx, s, taps->D2_lr, tap_e->out1, n))) return rc; }
  const int nsides = p.postprocess_only_left ? 1 : 2;
  // elas.cpp:986-991, 1107-1111: thresholds of the half-resolution maps
  const int speckle_size = sub ? (int)(sqrtf((float)p.speckle_size) * 2) : p.speckle_size;
  const int gap_width = sub ? p.ipol_gap_width / 2 + 1 : p.ipol_gap_width;
  const int nb = (int)((n + 255) / 256);
  // strips in LDS when two strip-sized arrays fit (more than the default 64 KB of dynamic LDS needs an opt-in)
  const size_t strip_lds = (size_t)2 * CCS_ROWS * Wd * sizeof(int32_t);
  // the opt-in is a per-device function attribute: remembered per context (a context is bound to one device and is not
  // shared between threads), never in a process-wide static
  if (ctx->elas_strip_state == 0) {
    ctx->elas_strip_state = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cc_strip), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                150 * 1024) == hipSuccess ? 1 : -1;
    (void)hipGetLastError();
  }
  const bool strip_ok = ctx->elas_strip_state > 0;
  const char* cc_env = getenv("SVO_ELAS_CC_STRIPS");
  const bool use_strips = strip_ok && strip_lds <= 150 * 1024 && Hd > CCS_ROWS && (!cc_env || atoi(cc_env) != 0);
  for (int side = 0; side < nsides; ++side) {
    SvoTimer t(ctx, "k_cc_segments");
    if (use_strips) {
      const int nstrips = (Hd + CCS_ROWS - 1) / CCS_ROWS;
      hipLaunchKernelGGL(k_cc_strip, dim3(nstrips, ub), dim3(256), strip_lds, s, d_tab, side, Wd, Hd, p.speckle_sim_threshold);
      hipLaunchKernelGGL(k_cc_merge_borders, dim3((Wd + 255) / 256, nstrips - 1, ub), dim3(256), 0, s, d_tab, side, Wd, Hd,
                         p.speckle_sim_threshold);
      hipLaunchKernelGGL(k_cc_count, dim3(nb, ub), dim3(256), 0, s, d_tab, (int)n);
      hipLaunchKernelGGL(k_cc_apply, dim3(nb, ub), dim3(256), 0, s, d_tab, side, (int)n, speckle_size);
      continue;
    }
    hipLaunchKernelGGL(k_cc_rows, dim3(Hd, ub), dim3(256), 0, s, d_tab, side, Wd, p.speckle_sim_threshold);
    hipLaunchKernelGGL(k_cc_merge, pix, dim3(256), 0, s, d_tab, side, Wd, Hd, p.speckle_sim_threshold);
    hipLaunchKernelGGL(k_cc_count, dim3(nb, ub), dim3(256), 0, s, d_tab, (int)n);
    hipLaunchKernelGGL(k_cc_apply, dim3(nb, ub), dim3(256), 0, s, d_tab, side, (int)n, speckle_size);
  }
  if (taps) { if ((rc = tap(ctx, s, taps->D1_seg, tap_e->out0, n))) return rc; if ((rc = tap(ctx, s, taps->D2_seg, tap_e->out1, n))) return rc; }
  for (int side = 0; side < nsides; ++side) {
    SvoTimer t(ctx, "k_elas_gap");
    hipLaunchKernelGGL(k_elas_gap, dim3(Hd, ub), dim3(256), 0, s, d_tab, side, Wd, 1, Wd, gap_width, p.add_corners);
    int C = 32;   // columns per tile: val (4 B) + next-valid (2 B) per pixel within 60 KB of dynamic LDS
    while (C > 1 && (size_t)Hd * C * 6 > 60 * 1024) C >>= 1;
    hipLaunchKernelGGL(k_elas_gap_cols, dim3((Wd + C - 1) / C, ub), dim3(256), (size_t)Hd * C * 6, s, d_tab, side, Wd, Hd, C,
                       gap_width, p.add_corners);
  }
  if (taps) { if ((rc = tap(ctx, s, taps->D1_gap, tap_e->out0, n))) return rc; if ((rc = tap(ctx, s, taps->D2_gap, tap_e->out1, n))) return rc; }
  if (p.filter_adaptive_mean)
    for (int side = 0; side < nsides; ++side) {
      SvoTimer t(ctx, "k_elas_mean");
      hipLaunchKernelGGL(k_elas_mean_h, pix, dim3(256), 0, s, d_tab, side, Wd, Hd, sub);
      hipLaunchKernelGGL(k_elas_mean_v, pix, dim3(256), 0, s, d_tab, side, Wd, Hd, sub);
    }
  if (taps) { if ((rc = tap(ctx, s, taps->D1_mean, tap_e->out0, n))) return rc; if ((rc = tap(ctx, s, taps->D2_mean, tap_e->out1, n))) return rc; }
  if (p.filter_median)
    for (int side = 0; side < nsides; ++side) {
      SvoTimer t(ctx, "k_elas_median");
      hipLaunchKernelGGL(k_elas_median_h, pix, dim3(256), 0, s, d_tab, side, Wd, Hd);
      hipLaunchKernelGGL(k_elas_median_v, pix, dim3(256), 0, s, d_tab, side, Wd, Hd);
    }
  return SVO_OK;
}

void fill_prior(const svo_elas_params& p, int32_t (&P)[256]) {   // elas.cpp:795-799
  const int disp_num = p.disp_max + 1;
  const float two_sigma_squared = 2 * p.sigma * p.sigma;
  for (int dd = 0; dd < 256; ++dd)
    P[dd] = dd < disp_num ? (int32_t)((-logf(p.gamma + expf(-dd * dd / two_sigma_squared)) + logf(p.gamma)) / p.beta) : 0;
}

// One pair, latency first: everything between the images in HBM (dL, dR, `pitch` bytes per row) and the two
// final disparity maps in HBM (st->d_T[0], st->d_T[1]; valid until the next call).  *produced = 0 when there
// are fewer than 3 support points (the reference then leaves its outputs untouched).
int elas_core(svo_ctx* ctx, ElasState* st, const uint8_t* dL, const uint8_t* dR, int pitch, int W, int H,
              const svo_elas_params& p, svo_elas_taps* taps, float** outD1, float** outD2, int* produced) {
  hipStream_t s = ctx->stream;
  const size_t nd = p.subsampling ? (size_t)(W / 2) * (H / 2) : (size_t)W * H;
  const int ncell = st->gw * st->gh;
  int rc;
  *produced = 0;
  ElasTab e = tab_entry(st, dL, dR, st->d_can, st->d_T[0], st->d_T[1]);
  SVO_HIP(ctx, hipMemcpyAsync(st->d_tab, &e, sizeof e, hipMemcpyHostToDevice, s));
  // calloc'ed in the reference: row 0 / column 0 of the lattice stay 0
  SVO_HIP(ctx, hipMemsetAsync(st->d_can, 0, (size_t)st->Wc * st->Hc * sizeof(int16_t), s));
  if ((rc = elas_phase_a(ctx, s, st->d_tab, 1, pitch, W, H, st->Wc, st->Hc, p))) return rc;
  {
    HostTimer ht(ctx, "host_elas_wait_candidates");
    SVO_HIP(ctx, hipMemcpyAsync(st->h_can, st->d_can, (size_t)st->Wc * st->Hc * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    SVO_HIP(ctx, hipStreamSynchronize(s));
  }
  if (taps) {
    if (taps->desc1) SVO_HIP(ctx, svo_memcpy_sync(ctx, taps->desc1, st->d_desc[0], (size_t)W * H * 16, hipMemcpyDeviceToHost));
    if (taps->desc2) SVO_HIP(ctx, svo_memcpy_sync(ctx, taps->desc2, st->d_desc[1], (size_t)W * H * 16, hipMemcpyDeviceToHost));
  }
  ElasWork w;
  {
    HostTimer ht(ctx, "host_elas_support_filter");
    elas_filter(st->h_can, st->Wc, st->Hc, W, H, p, w);
  }
  if (taps) {
    taps->n_support = (int32_t)w.sp.size();
    if (taps->support)
      memcpy(taps->support, w.spflat.data(), sizeof(int32_t) * 3 * std::min<size_t>(w.sp.size(), (size_t)std::max(taps->cap_support, 0)));
  }
  if (w.sp.size() < 3) return SVO_OK;   // *produced stays 0
  // support points -> HBM; the disparity grids need nothing else, so the GPU builds them while the host triangulates
  e.nsp = (int32_t)w.sp.size();
  SVO_HIP(ctx, hipMemcpyAsync(st->d_tab, &e, sizeof e, hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(st->d_sp, w.spflat.data(), w.spflat.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  for (int side = 0; side < 2; ++side) {
    SVO_HIP(ctx, hipMemsetAsync(st->d_grid[side], 0, (size_t)ncell * 8 * sizeof(uint32_t), s));
    SVO_HIP(ctx, hipMemsetAsync(st->d_owner[side], 0xff, nd * sizeof(int32_t), s));
  }
  if ((rc = elas_phase_b_grids(ctx, s, st->d_tab, 1, e.nsp, st->gw, st->gh, p))) return rc;
  {
    HostTimer ht(ctx, "host_elas_delaunay");
    elas_triangulate(st->cap_tri, taps, w, true);
  }
  if (w.err) { hipStreamSynchronize(s); ctx->last_error = w.err; return SVO_E_INVALID; }
  {
    HostTimer ht(ctx, "host_elas_upload2_match_sync");
    int32_t P[256];
    fill_prior(p, P);
    e.nt0 = (int32_t)w.tri[0].size() / 3; e.nt1 = (int32_t)w.tri[1].size() / 3;
    SVO_HIP(ctx, hipMemcpyAsync(st->d_tab, &e, sizeof e, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(st->d_P, P, sizeof P, hipMemcpyHostToDevice, s));
    for (int side = 0; side < 2; ++side)
      if (!w.tri[side].empty())
        SVO_HIP(ctx, hipMemcpyAsync(st->d_tri[side], w.tri[side].data(), w.tri[side].size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    rc = elas_phase_b(ctx, s, st->d_tab, 1, std::max(e.nt0, e.nt1), W, H, st->gw, st->gh, st->gd, st->d_P, p, st, &e, &w, taps);
    hipStreamSynchronize(s);   // `w`, `e`, `P` must outlive their uploads; the callers read the maps next anyway
  }
  if (rc) return rc;
  *outD1 = st->d_T[0]; *outD2 = st->d_T[1];
  *produced = 1;
  return SVO_OK;
}

// ---- many pairs at once -----------------------------------------------------------------------------
// One ElasState ("slot") per pair; every stage is ONE launch for a chunk of pairs (pair index in the grid);
// the two sequential host stages of the pairs run on a pool of host threads; the lists they produce go up
// packed, one copy per chunk.
struct ElasBatch {
  std::vector<ElasState*> slots;
  std::vector<ElasWork> work;    // host results per pair (support points, triangle lists), vectors reused
  int cap = 0, can_elems = 0;
  ElasTab* d_tab = nullptr;
  ElasTab* h_tab = nullptr;      // pinned
  int16_t* d_can = nullptr;      // [cap][Wc*Hc]
  int16_t* h_can = nullptr;      // pinned
  int32_t* d_P = nullptr;
  int32_t* d_lists = nullptr;    // per call: support points and triangle lists of all pairs, packed
  int32_t* h_lists = nullptr;    // pinned
  size_t lists_cap = 0;          // ints
  int grow_lists(svo_ctx* ctx, size_t need) {
    if (need <= lists_cap) return SVO_OK;
    if (d_lists) hipFree(d_lists);
    if (h_lists) hipHostFree(h_lists);
    d_lists = nullptr; h_lists = nullptr; lists_cap = 0;
    SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d_lists), need * sizeof(int32_t)));
    SVO_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&h_lists), need * sizeof(int32_t), hipHostMallocDefault));
    lists_cap = need;
    return SVO_OK;
  }
  void release_tables() {
    if (d_tab) hipFree(d_tab);
    if (h_tab) hipHostFree(h_tab);
    if (d_can) hipFree(d_can);
    if (h_can) hipHostFree(h_can);
    d_tab = nullptr; h_tab = nullptr; d_can = nullptr; h_can = nullptr;
    cap = 0; can_elems = 0;
  }
  ~ElasBatch() {
    for (ElasState* st : slots) { st->release(); delete st; }
    release_tables();
    if (d_P) hipFree(d_P);
    if (d_lists) hipFree(d_lists);
    if (h_lists) hipHostFree(h_lists);
  }
};
#define ELAS_BATCH_CHUNK 32

}  // namespace

extern "C" void svo_elas_release(svo_ctx* ctx) {
  if (!ctx) return;
  if (ctx->elas) {
    ElasState* st = reinterpret_cast<ElasState*>(ctx->elas);
    st->release();
    delete st;
    ctx->elas = nullptr;
  }
  if (ctx->elas_batch) {
    delete reinterpret_cast<ElasBatch*>(ctx->elas_batch);
    ctx->elas_batch = nullptr;
  }
  if (ctx->stream_elas_a) { hipStreamSynchronize(ctx->stream_elas_a); hipStreamDestroy(ctx->stream_elas_a); ctx->stream_elas_a = nullptr; ctx->stream_elas_a_pct = -1; }
  if (ctx->ev_elas_setup) { hipEventDestroy(ctx->ev_elas_setup); ctx->ev_elas_setup = nullptr; }
}

// Device-resident entry used by the tracker (svo_track.hip): images already in HBM, maps stay in HBM.
int svo_elas_run_dev(svo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int pitch, int W, int H,
                     const svo_elas_params* params, float** dD1, float** dD2, int* produced) {
  int rc = elas_check(ctx, W, H, pitch, *params);
  if (rc) return rc;
  if (!ctx->elas) ctx->elas = new ElasState();
  ElasState* st = reinterpret_cast<ElasState*>(ctx->elas);
  if ((rc = elas_prepare(ctx, st, W, H, *params, false))) return rc;
  return elas_core(ctx, st, dL, dR, pitch, W, H, *params, nullptr, dD1, dD2, produced);
}

// svo_elas_batch_dev with a hook: hook(user, f0, b) runs on the calling thread right after the last GPU phase of the pairs
// f0 .. f0 + b - 1 has been ENQUEUED on the ctx stream (their `produced` flags are final) - the batched tracker hangs the
// per-keypoint depth lookups and the ordered tail of those frames on it, so that they run while later chunks are still in the
// dense stage (BASELINE configs[4] as a pipeline).
int svo_elas_batch_dev_hooked(svo_ctx* ctx, const uint8_t* d_L, const uint8_t* d_R, int stride, int W, int H,
                              int B, const svo_elas_params* params, float* d_D1, float* d_D2, int32_t* produced,
                              int (*hook)(void*, int, int), void* user);
extern "C" int svo_elas_batch_dev(svo_ctx* ctx, const uint8_t* d_L, const uint8_t* d_R, int stride, int W, int H,
                                  int B, const svo_elas_params* params, float* d_D1, float* d_D2, int32_t* produced) {
  return svo_elas_batch_dev_hooked(ctx, d_L, d_R, stride, W, H, B, params, d_D1, d_D2, produced, nullptr, nullptr);
}
int svo_elas_batch_dev_hooked(svo_ctx* ctx, const uint8_t* d_L, const uint8_t* d_R, int stride, int W, int H,
                              int B, const svo_elas_params* params, float* d_D1, float* d_D2, int32_t* produced,
                              int (*hook)(void*, int, int), void* user) {
  if (!ctx) return SVO_E_INVALID;
  if (!d_L || !d_R || !d_D1 || !d_D2 || !params || B < 1) { ctx->last_error = "svo_elas_batch_dev: invalid argument"; return SVO_E_INVALID; }
  const svo_elas_params p = *params;
  int rc = elas_check(ctx, W, H, stride, p);
  if (rc) return rc;
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->elas_batch) ctx->elas_batch = new ElasBatch();
  ElasBatch* eb = reinterpret_cast<ElasBatch*>(ctx->elas_batch);
  while ((int)eb->slots.size() < B) eb->slots.push_back(new ElasState());
  for (int b = 0; b < B; ++b)
    if ((rc = elas_prepare(ctx, eb->slots[b], W, H, p, false))) return rc;
  const ElasState* s0 = eb->slots[0];
  const int Wc = s0->Wc, Hc = s0->Hc, gw = s0->gw, gh = s0->gh, wh = Wc * Hc;
  if (eb->cap < B || eb->can_elems != wh) {
    SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    eb->release_tables();
    SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&eb->d_tab), sizeof(ElasTab) * (size_t)B));
    SVO_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&eb->h_tab), sizeof(ElasTab) * (size_t)B, hipHostMallocDefault));
    SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&eb->d_can), sizeof(int16_t) * (size_t)B * wh));
    SVO_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&eb->h_can), sizeof(int16_t) * (size_t)B * wh, hipHostMallocDefault));
    eb->cap = B; eb->can_elems = wh;
  }
  if (!eb->d_P) SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&eb->d_P), 256 * sizeof(int32_t)));
  const size_t nd = p.subsampling ? (size_t)(W / 2) * (H / 2) : (size_t)W * H;
  const size_t img = (size_t)stride * H;
  hipStream_t s = ctx->stream;

  // The batch is cut into chunks that move through  A (GPU) -> host stages -> B (GPU)  as a pipeline: while the
  // host threads work on chunk c, the GPU runs phase A of chunk c+1 and phase B of chunk c-1.
  const char* chunk_env = getenv("SVO_ELAS_CHUNK");   // tuning knob, default ELAS_BATCH_CHUNK
  // (with a hook - the tracker's tail hangs on every chunk - chunks of 16: the tail starts after half as many pairs and the
  // last chunk's tail is half as long; measured with boxes, 256 frames per call: 32 -> 6.46 k, 24 -> 6.57 k, 16 -> 6.70 k, 8 -> 5.97 k
  // frames/s.  Without a hook 32: 8.0 k pairs/s against 7.1 k with 16 on the synthetic frames.)
  const int C = std::min(B, chunk_env && atoi(chunk_env) > 0 ? atoi(chunk_env) : (hook ? ELAS_BATCH_CHUNK / 2 : ELAS_BATCH_CHUNK));
  // With a hook the first chunks are SMALL (4, 4, 8 pairs, then C): whatever hangs on the chunks - the tracker's ordered tail, 120 us
  // per frame - starts after the dense stage of four pairs instead of sixteen, and the pipeline's fill (A -> host -> B of the
  // first chunk, ~4 ms of a 36 ms call of 256 frames) shrinks accordingly.  SVO_ELAS_RAMP=0: equal chunks.
  std::vector<int> coff(1, 0);
  {
    static const bool ramp = []() { const char* e = getenv("SVO_ELAS_RAMP"); return !(e && e[0] == '0'); }();
    const int first[3] = {4, 4, 8};
    for (int k = 0; hook && ramp && k < 3 && coff.back() + first[k] < B && first[k] < C; ++k) coff.push_back(coff.back() + first[k]);
    while (coff.back() < B) coff.push_back(std::min(B, coff.back() + C));
  }
  const int NC = (int)coff.size() - 1;
  auto c_b0 = [&coff](int c) { return coff[c]; };
  auto c_nb = [&coff](int c) { return coff[c + 1] - coff[c]; };
  // every fallible set-up step comes BEFORE the worker pool exists (returning past joinable threads would terminate
  // the process); the events are owned by a guard so that no return path leaks them
  struct EventSet {
    std::vector<hipEvent_t> ev;
    ~EventSet() { for (hipEvent_t e : ev) if (e) hipEventDestroy(e); }
  } evs;
  evs.ev.assign(NC, nullptr);
  std::vector<hipEvent_t>& evA = evs.ev;
  for (int c = 0; c < NC; ++c) SVO_HIP(ctx, hipEventCreateWithFlags(&evA[c], hipEventDisableTiming));
  if ((rc = eb->grow_lists(ctx, (size_t)B * 49152))) return rc;   // ~2x the typical 90 KB of lists per pair
  int32_t P[256];
  fill_prior(p, P);
  for (int b = 0; b < B; ++b)
    eb->h_tab[b] = tab_entry(eb->slots[b], d_L + b * img, d_R + b * img, eb->d_can + (size_t)b * wh, d_D1 + b * nd, d_D2 + b * nd);
  SVO_HIP(ctx, hipMemcpyAsync(eb->d_tab, eb->h_tab, sizeof(ElasTab) * (size_t)B, hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(eb->d_P, P, sizeof P, hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemsetAsync(eb->d_can, 0, sizeof(int16_t) * (size_t)B * wh, s));   // lattice row 0 / column 0 stay 0
  // Phase A of the chunks (descriptors + support matching: arithmetic-bound, 0.6 ms per 32 pairs) runs on a stream of its own
  // beside phase B of the earlier chunks (1.8 ms, half of it latency-bound kernels: rasterisation by atomics, segment
  // labelling in strips, gap interpolation) - on one stream the GPU ran them one after the other (2.4 ms per chunk of the
  // 2.9 ms a chunk takes).  Nothing is shared between the phases of different chunks: slots, candidates and table rows are per
  // pair.  (10.6 k -> 11.2 k pairs/s.)  Not beside the tracker's tail (svo_track_batch_dev with depth_source = 1 runs the dense
  // stage on a CU-confined stream): two streams keep that share of the CUs busy without a gap and the tail's single-wave
  // RANSAC workgroups that land there wait - 6.4 k -> 5.6 k frames/s, the effect svo_track_sharded_dev met with two
  // confined front-end streams.
  hipStream_t sA = s;
  {
    static const bool one = []() { const char* e = getenv("SVO_ELAS_ONE_STREAM"); return e && e[0] == '1'; }();
    if (!one && !(ctx->stream_dense && s == ctx->stream_dense)) {
      const int pct = 0;
      if (ctx->stream_elas_a && ctx->stream_elas_a_pct != pct) {
        hipStreamSynchronize(ctx->stream_elas_a); hipStreamDestroy(ctx->stream_elas_a); ctx->stream_elas_a = nullptr;
      }
      if (!ctx->stream_elas_a) {
        {   // a stream that runs beside the caller's (phase A of chunk c + 1 beside phase B of chunk c) - and beside the tracker's two
            // chains, whose tail hangs on the chunks when the maps feed it (measured: phase A on a hardware queue shared with one of
            // them waited 30 ms per 256-frame call, configs[4] 4.8 k instead of 6.8 k frames/s)
          int attempts = 0, percent = 0;
          const int rcp = svo_pick_stream(ctx, [](hipStream_t* q) { return svo_stream_create(q, 0); },
                                          {s, ctx->stream != s ? ctx->stream : nullptr, ctx->stream_idx}, &ctx->stream_elas_a, &attempts, &percent,
                                          {s, ctx->stream != s ? ctx->stream : nullptr, ctx->stream_idx}, {s});
          if (rcp) return rcp;
        }
        ctx->stream_elas_a_pct = pct;
      }
      if (!ctx->ev_elas_setup) SVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_elas_setup, hipEventDisableTiming));
      sA = ctx->stream_elas_a;
      SVO_HIP(ctx, hipEventRecord(ctx->ev_elas_setup, s));          // the table, the candidates' memset, whatever produced the images
      SVO_HIP(ctx, hipStreamWaitEvent(sA, ctx->ev_elas_setup, 0));
    }
  }

  auto enqueue_a = [&](int c) -> int {   // descriptors + lattice candidates of the chunk, candidates on their way back
    const int b0 = c_b0(c), nb = c_nb(c);
    int r = elas_phase_a(ctx, sA, eb->d_tab + b0, nb, stride, W, H, Wc, Hc, p);
    if (r) return r;
    SVO_HIP(ctx, hipMemcpyAsync(eb->h_can + (size_t)b0 * wh, eb->d_can + (size_t)b0 * wh, sizeof(int16_t) * (size_t)nb * wh,
                                hipMemcpyDeviceToHost, sA));
    SVO_HIP(ctx, hipEventRecord(evA[c], sA));
    return SVO_OK;
  };

  // persistent pool: the two host stages of one pair per task
  // the host results of a pair keep their vectors from call to call (same reason)
  if ((int)eb->work.size() < B) eb->work.resize(B);
  std::vector<ElasWork>& work = eb->work;
  for (int b = 0; b < B; ++b) { work[b].sp.clear(); work[b].spflat.clear(); work[b].tri[0].clear(); work[b].tri[1].clear(); work[b].err = nullptr; }
  std::vector<int> prc(B, SVO_OK);
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  int task_begin = 0, task_end = 0, next_task = 0, running = 0;
  bool quit = false;
  // persistent pool.  The host stage of a chunk runs as two waves of tasks: the filter of every pair, then the two
  // triangulations of every pair as separate tasks - a pair's ~3 ms of host work (0.9 + 2 x 1.0) would otherwise be the
  // latency of the whole chunk, and the chunk's GPU phases take less than that.
  std::atomic<long long> dbg_filter_us{0}, dbg_tri_us{0};
  int task_mode = 0, task_b0 = 0;
  auto do_task = [&](int t) {
    const auto ta = std::chrono::steady_clock::now();
    if (task_mode == 0) {
      const int b = task_b0 + t;
      elas_filter(eb->h_can + (size_t)b * wh, Wc, Hc, W, H, p, work[b]);
      dbg_filter_us += (long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ta).count();
    } else {
      const int b = task_b0 + (t >> 1);
      if (work[b].sp.size() >= 3) elas_triangulate_side(eb->slots[b]->cap_tri, nullptr, work[b], t & 1);
      dbg_tri_us += (long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ta).count();
    }
  };
  auto worker = [&]() {
    hipSetDevice(ctx->device);   // the current device is per-thread state
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_go.wait(lk, [&] { return quit || next_task < task_end; });
      if (quit) return;
      const int t = next_task++;
      ++running;
      lk.unlock();
      do_task(t);
      lk.lock();
      if (--running == 0 && next_task >= task_end) cv_done.notify_all();
    }
  };
  const char* thr_env = getenv("SVO_ELAS_THREADS");   // tuning knob
  // (measured on a 16-CPU quota, 512 pairs: 12 threads 5.8 k pairs/s, 16: 7.8 k, 24: 9.0 k, 32: 8.4 k - part of a task is waiting)
  const int thr_cap = thr_env && atoi(thr_env) > 0 ? atoi(thr_env) : (3 * svo_host_cpus() + 1) / 2;
  const int nthreads = std::max(1, std::min(2 * C, thr_cap));
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
  auto host_stage = [&](int c) {
    const int b0 = c_b0(c), nb = c_nb(c);
    std::unique_lock<std::mutex> lk(mu);
    for (int mode = 0; mode < 2; ++mode) {
      task_mode = mode; task_b0 = b0;
      task_begin = 0; next_task = 0; task_end = mode == 0 ? nb : 2 * nb;
      cv_go.notify_all();
      cv_done.wait(lk, [&] { return next_task >= task_end && running == 0; });
    }
  };
  double t_sync_copy = 0;
  size_t lists_used = 0;
  auto enqueue_b = [&](int c) -> int {   // counts into the table, then the whole phase B of the chunk
    const int b0 = c_b0(c), nb = c_nb(c);
    int max_nsp = 0, max_nt = 0;
    for (int b = b0; b < b0 + nb; ++b) {
      const ElasWork& w = work[b];
      if (w.err) { ctx->last_error = w.err; return SVO_E_INVALID; }
      if (prc[b]) { ctx->last_error = "svo_elas_batch_dev: upload failed"; return prc[b]; }
      ElasTab& e = eb->h_tab[b];
      e.produced = w.sp.size() >= 3;   // fewer than 3 support points: outputs untouched, as the reference leaves them
      e.nsp = (int32_t)w.sp.size(); e.nt0 = (int32_t)w.tri[0].size() / 3; e.nt1 = (int32_t)w.tri[1].size() / 3;
      if (produced) produced[b] = e.produced;
      if (e.produced) { max_nsp = std::max(max_nsp, e.nsp); max_nt = std::max(max_nt, std::max(e.nt0, e.nt1)); }
    }
    if (max_nsp == 0) return SVO_OK;
    // the chunk's lists, packed, in ONE copy (a copy per list costs ~25 us each under load)
    const auto tq0 = std::chrono::steady_clock::now();
    size_t need = 0;
    for (int b = b0; b < b0 + nb; ++b)
      if (eb->h_tab[b].produced) need += work[b].spflat.size() + work[b].tri[0].size() + work[b].tri[1].size();
    if (lists_used + need > eb->lists_cap) {   // earlier chunks may still read the arena: drain, then grow
      hipStreamSynchronize(s);
      int r = eb->grow_lists(ctx, std::max(need * (size_t)(NC - c + 1), (size_t)1 << 20));
      if (r) return r;
      lists_used = 0;
    }
    const size_t chunk_off = lists_used;
    for (int b = b0; b < b0 + nb; ++b) {
      ElasTab& e = eb->h_tab[b];
      if (!e.produced) continue;
      const ElasWork& w = work[b];
      const std::vector<int32_t>* src[3] = {&w.spflat, &w.tri[0], &w.tri[1]};
      int32_t** dst[3] = {&e.sp, &e.tri0, &e.tri1};
      for (int k = 0; k < 3; ++k) {
        memcpy(eb->h_lists + lists_used, src[k]->data(), src[k]->size() * sizeof(int32_t));
        *dst[k] = eb->d_lists + lists_used;
        lists_used += src[k]->size();
      }
    }
    SVO_HIP(ctx, hipMemcpyAsync(eb->d_lists + chunk_off, eb->h_lists + chunk_off, (lists_used - chunk_off) * sizeof(int32_t),
                                hipMemcpyHostToDevice, s));
    t_sync_copy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count();
    if (sA != s) SVO_HIP(ctx, hipStreamWaitEvent(s, evA[c], 0));   // (long signalled: the host stage of the chunk came after it)
    SVO_HIP(ctx, hipMemcpyAsync(eb->d_tab + b0, eb->h_tab + b0, sizeof(ElasTab) * (size_t)nb, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_elas_clear, dim3((unsigned)((nd + 255) / 256), (unsigned)nb), dim3(256), 0, s, eb->d_tab + b0, (int)nd,
                       gw * gh * 8);
    int r = elas_phase_b_grids(ctx, s, eb->d_tab + b0, nb, max_nsp, gw, gh, p);
    if (r == SVO_OK)
      r = elas_phase_b(ctx, s, eb->d_tab + b0, nb, max_nt, W, H, gw, gh, s0->gd, eb->d_P, p, nullptr, nullptr, nullptr, nullptr);
    return r;
  };

  const bool dbg = getenv("SVO_ELAS_BATCH_DEBUG") != nullptr;
  auto tnow = []() { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count(); };
  double t_wait = 0, t_host = 0, t_enq = 0;
  const auto t_start = tnow();
  // The host stage of chunk c (the pool: filter, then the triangulations) runs on a driver thread of its own while THIS thread
  // enqueues: phase B of chunk c - 1 (+ the caller's hook: the tracker's depth lookups and tail), then phase A of chunk c + 2 -
  // the pool used to stand still during those 8-10 ms of a 256-pair call.  Only this thread makes HIP calls.
  rc = enqueue_a(0);
  std::thread driver;
  auto finish = [&](int c) -> int {   // chunk c has left the host stage: its phase B, and whatever the caller hangs on it
    int r = enqueue_b(c);
    if (r == SVO_OK && hook) r = hook(user, c_b0(c), c_nb(c));
    return r;
  };
  for (int c = 0; c < NC && rc == SVO_OK; ++c) {
    auto t0 = tnow();
    if (c + 1 < NC) rc = enqueue_a(c + 1);
    auto t1 = tnow();
    if (rc == SVO_OK) hipEventSynchronize(evA[c]);
    auto t2 = tnow();
    if (driver.joinable()) driver.join();          // the pool is free again: chunk c - 1 is through
    auto t3 = tnow();
    if (rc == SVO_OK) {
      // (std::thread's constructor may throw; an exception must not unwind past the joinable pool threads: run the stage here)
      try { driver = std::thread([&host_stage, c]() { host_stage(c); }); }
      catch (const std::system_error&) { host_stage(c); }
    }
    if (rc == SVO_OK && c > 0) rc = finish(c - 1);
    t_enq += ms(t0, t1) + ms(t3, tnow()); t_wait += ms(t1, t2); t_host += ms(t2, t3);
  }
  {
    auto t2 = tnow();
    if (driver.joinable()) driver.join();
    auto t3 = tnow();
    if (rc == SVO_OK) rc = finish(NC - 1);
    t_host += ms(t2, t3); t_enq += ms(t3, tnow());
  }
  const auto t_loop = tnow();
  {
    std::unique_lock<std::mutex> lk(mu);
    quit = true;
    cv_go.notify_all();
  }
  for (std::thread& t : pool) t.join();
  if (sA != s) hipStreamSynchronize(sA);
  hipStreamSynchronize(s);   // the pinned table and list arena are read by copies until here
  if (dbg) {
    fprintf(stderr, "   delaunay per call: sort pts %.0f us, build %.0f, emit %.0f, sort triangles %.0f; points %.0f\n", (double)svo_delaunay_us[0] / (2 * B), (double)svo_delaunay_us[1] / (2 * B), (double)svo_delaunay_us[2] / (2 * B), (double)svo_delaunay_us[3] / (2 * B), (double)svo_delaunay_pts / (2 * B));
    for (int k = 0; k < 4; ++k) svo_delaunay_us[k] = 0; svo_delaunay_pts = 0; }
  if (dbg) fprintf(stderr, "   host work per pair: filter %.0f us, triangulations %.0f us (thread time)\n", (double)dbg_filter_us / B, (double)dbg_tri_us / B);
  if (dbg) fprintf(stderr, "elas batch B=%d chunk=%d: enqueue %.2f ms (of it packing the lists %.2f), wait for A %.2f, host stages %.2f, tail wait %.2f, total %.2f\n",
                   B, C, t_enq, t_sync_copy, t_wait, t_host, ms(t_loop, tnow()), ms(t_start, tnow()));
  if (rc == SVO_OK) SVO_HIP(ctx, hipGetLastError());
  return rc;
}

extern "C" int svo_elas_process_ex(svo_ctx* ctx, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                                   const int32_t* dims, const svo_elas_params* params, svo_elas_taps* taps) {
  if (!ctx) return SVO_E_INVALID;
  if (!I1 || !I2 || !D1 || !D2 || !dims || !params) { ctx->last_error = "svo_elas_process: null argument"; return SVO_E_INVALID; }
  const svo_elas_params p = *params;
  const int W = dims[0], H = dims[1], pitch = dims[2];
  int rc = elas_check(ctx, W, H, pitch, p);
  if (rc) return rc;
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->elas) ctx->elas = new ElasState();
  ElasState* st = reinterpret_cast<ElasState*>(ctx->elas);
  if ((rc = elas_prepare(ctx, st, W, H, p, true))) return rc;
  hipStream_t s = ctx->stream;
  const size_t n = (size_t)W * H;
  HostTimer total(ctx, "host_elas_total");
  {
    HostTimer ht(ctx, "host_elas_upload");
    for (int v = 0; v < H; ++v) {   // pageable -> pinned staging (a pageable 2-D copy costs 14 ms per image)
      memcpy(st->h_img + (size_t)v * W, I1 + (size_t)v * pitch, W);
      memcpy(st->h_img + n + (size_t)v * W, I2 + (size_t)v * pitch, W);
    }
    SVO_HIP(ctx, hipMemcpyAsync(st->d_img[0], st->h_img, n, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(st->d_img[1], st->h_img + n, n, hipMemcpyHostToDevice, s));
  }
  float *dD1 = nullptr, *dD2 = nullptr;
  int produced = 0;
  if ((rc = elas_core(ctx, st, st->d_img[0], st->d_img[1], W, W, H, p, taps, &dD1, &dD2, &produced))) return rc;
  if (!produced) return SVO_OK;   // the reference prints an error and returns with D1/D2 untouched
  {
    HostTimer ht(ctx, "host_elas_download");
    const size_t nd = p.subsampling ? (size_t)(W / 2) * (H / 2) : n;
    SVO_HIP(ctx, hipMemcpyAsync(st->h_D, dD1, nd * sizeof(float), hipMemcpyDeviceToHost, s));
    SVO_HIP(ctx, hipMemcpyAsync(st->h_D + nd, dD2, nd * sizeof(float), hipMemcpyDeviceToHost, s));
    SVO_HIP(ctx, hipStreamSynchronize(s));
    memcpy(D1, st->h_D, nd * sizeof(float));
    memcpy(D2, st->h_D + nd, nd * sizeof(float));
  }
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

extern "C" int svo_elas_process(svo_ctx* ctx, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                                const int32_t* dims, const svo_elas_params* params) {
  return svo_elas_process_ex(ctx, I1, I2, D1, D2, dims, params, nullptr);
}
