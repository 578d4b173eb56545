// svo_match.hip - 256-bit Hamming matching for gfx950.
//
// Replaces pnpmatch::DescriptorDistance (reference src/pnpmatch.cc:14-30), the two
// brute-force passes of pnpmatch::poseEstimationPnP (:61-156, :159-199) and the
// cv BruteForce-Hamming match() + filter of find_feature_matches (:253-300).
//
// Layout: descriptors are rows of 8 dwords.  The train side (<= 1024 rows) is
// staged in LDS; one wave owns one query row, lanes stride the train rows,
// distances are 8 x (xor, popcount), the argmin is a wave reduction over
// (dist << 16 | j) so ties resolve to the lowest j exactly like the reference's
// strict `<` scan.  The greedy passes are order-dependent (a match claims its
// column): a parallel kernel first writes the full distance matrix, then ONE wave
// walks the rows in order with each lane holding the claim bits of its 8 columns.
#include "svo_internal.h"
#include "svo_wave.h"
#include "svo_gate.h"

#define MAXT 1024

__device__ __forceinline__ uint32_t wmin_u32(uint32_t v) { return wave_min_u32_dpp(v); }

__global__ void k_desc_dist(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                            int count, int32_t* __restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  int d = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) d += __popc(a[(size_t)i * 8 + k] ^ b[(size_t)i * 8 + k]);
  dist[i] = d;
}

// Independent rows: best / idx / quirky runner-up (see include/svo.h).
__global__ __launch_bounds__(256) void k_hamming_argmin(const uint32_t* __restrict__ q, int M,
                                                        const uint32_t* __restrict__ t, int N,
                                                        const uint8_t* __restrict__ mask,
                                                        int32_t* best_idx, int32_t* best,
                                                        int32_t* second) {
  __shared__ uint32_t td[MAXT * 8];
  __shared__ uint8_t tm[MAXT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < N * 8; i += 256) td[i] = t[i];
  for (int i = tid; i < N; i += 256) tm[i] = mask ? mask[i] : 0;
  __syncthreads();
  const int row = blockIdx.x * 4 + wv;
  if (row >= M) return;
  uint32_t qd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qd[k] = q[(size_t)row * 8 + k];
  uint32_t bestp = (256u << 16) | 0xffffu;
  for (int j = lane; j < N; j += 64) {
    if (tm[j]) continue;
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ td[j * 8 + k]);
    bestp = min(bestp, ((uint32_t)d << 16) | (uint32_t)j);
  }
  bestp = wmin_u32(bestp);
  const int bj = (int)(bestp & 0xffffu), bd = (int)(bestp >> 16);
  // runner-up = running best just before the last improvement = min over unmasked j < bj
  uint32_t sec = 256;
  if (bj != 0xffff) {
    for (int j = lane; j < bj; j += 64) {
      if (tm[j]) continue;
      int d = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ td[j * 8 + k]);
      sec = min(sec, (uint32_t)d);
    }
  }
  sec = wmin_u32(sec);
  if (lane == 0) {
    best_idx[row] = bj == 0xffff ? -1 : bj;
    best[row] = bj == 0xffff ? 256 : bd;
    second[row] = (int)sec;
  }
}

// Full distance matrix D[M][Npad] as uint16, Npad = 512 or 1024 (columns >= N hold 0x7fff).
__global__ __launch_bounds__(256) void k_dist_matrix(const uint32_t* __restrict__ q, int M,
                                                     const uint32_t* __restrict__ t, int N,
                                                     int Npad, uint16_t* __restrict__ D) {
  __shared__ uint32_t td[MAXT * 8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < N * 8; i += 256) td[i] = t[i];
  __syncthreads();
  const int row = blockIdx.x * 4 + wv;
  if (row >= M) return;
  uint32_t qd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qd[k] = q[(size_t)row * 8 + k];
  for (int j = lane; j < Npad; j += 64) {
    int d = 0x7fff;
    if (j < N) {
      d = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ td[j * 8 + k]);
    }
    D[(size_t)row * Npad + j] = (uint16_t)d;
  }
}

// Order-dependent greedy pass over the rows of D by ONE wave.  Lane L owns columns
// [L*CPL, (L+1)*CPL), CPL = Npad/64, and keeps their claim bits in a register.
template <int CPL>
__global__ __launch_bounds__(64) void k_greedy_serial(const uint16_t* __restrict__ D, int M, int N,
                                                      const uint8_t* __restrict__ q_skip,
                                                      uint8_t* assigned, int max_dist, float ratio,
                                                      int32_t* best_idx, int32_t* best,
                                                      int32_t* second, uint8_t* accepted,
                                                      const float* q_xy, const float* t_xy,
                                                      const int32_t* boxes, int n_boxes,
                                                      const double* F, uint8_t* vetoed) {
  const int lane = threadIdx.x;
  constexpr int Npad = CPL * 64;
  uint32_t claimed = 0;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int j = lane * CPL + k;
    if (j >= N || assigned[j]) claimed |= 1u << k;
  }
  uint16_t cur[CPL], nxt[CPL];
  auto load_row = [&](int r, uint16_t* dst) {
    const uint16_t* p = D + (size_t)r * Npad + lane * CPL;
#pragma unroll
    for (int k = 0; k < CPL; k += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + k);
      dst[k] = v.x & 0xffff; dst[k + 1] = v.x >> 16; dst[k + 2] = v.y & 0xffff; dst[k + 3] = v.y >> 16;
      dst[k + 4] = v.z & 0xffff; dst[k + 5] = v.z >> 16; dst[k + 6] = v.w & 0xffff; dst[k + 7] = v.w >> 16;
    }
  };
  if (M > 0) load_row(0, cur);
  for (int i = 0; i < M; ++i) {
    if (i + 1 < M) load_row(i + 1, nxt);
    int o_idx = -1, o_best = 256, o_sec = 256, o_acc = 0;
    if (!(q_skip && q_skip[i])) {
      uint32_t lp = (256u << 16) | 0xffffu;
#pragma unroll
      for (int k = 0; k < CPL; ++k)
        if (!((claimed >> k) & 1u)) lp = min(lp, ((uint32_t)cur[k] << 16) | (uint32_t)(lane * CPL + k));
      const uint32_t bp = wmin_u32(lp);
      const int bj = (int)(bp & 0xffffu), bd = (int)(bp >> 16);
      if (bj != 0xffff) {
        uint32_t ls = 256;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int j = lane * CPL + k;
          if (!((claimed >> k) & 1u) && j < bj) ls = min(ls, (uint32_t)cur[k]);
        }
        const int sec = (int)wmin_u32(ls);
        bool ok = bd < max_dist;
        if (ok && ratio > 0.f) ok = (float)sec / (float)bd > ratio;
        int veto = 0;
        if (ok && n_boxes > 0) {   // epipolar veto of pass 1 (reference src/pnpmatch.cc:103-144)
          const float cx = t_xy[2 * bj], cy = t_xy[2 * bj + 1];
          if (svo_in_boxes(cx, cy, boxes, n_boxes, 10) &&
              svo_epipolar_distance(F, q_xy[2 * i], q_xy[2 * i + 1], cx, cy) > 0.1) { veto = 1; ok = false; }
        }
        if (vetoed && lane == 0) vetoed[i] = (uint8_t)veto;
        o_idx = bj; o_best = bd; o_sec = sec; o_acc = ok ? 1 : 0;
        if (ok && (bj / CPL) == lane) claimed |= 1u << (bj % CPL);
      }
    }
    if (lane == 0) { best_idx[i] = o_idx; best[i] = o_best; second[i] = o_sec; accepted[i] = (uint8_t)o_acc; }
    if (vetoed && lane == 0 && o_idx < 0) vetoed[i] = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) cur[k] = nxt[k];
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int j = lane * CPL + k;
    if (j < N) assigned[j] = (uint8_t)((claimed >> k) & 1u);
  }
}

// cv BFMatcher::match + the min-distance filter of find_feature_matches.
__global__ __launch_bounds__(256) void k_bf_nearest(const uint32_t* __restrict__ q, int M,
                                                    const uint32_t* __restrict__ t, int N,
                                                    int32_t* train_idx, int32_t* dist,
                                                    int32_t* gmin, const int* M_ptr,
                                                    const int* N_ptr) {
  __shared__ uint32_t td[MAXT * 8];
  if (M_ptr) M = *M_ptr;
  if (N_ptr) N = min(*N_ptr, MAXT);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < N * 8; i += 256) td[i] = t[i];
  __syncthreads();
  const int row = blockIdx.x * 4 + wv;
  if (row >= M) return;
  uint32_t qd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qd[k] = q[(size_t)row * 8 + k];
  uint32_t bestp = (0x7fffu << 16) | 0xffffu;
  for (int j = lane; j < N; j += 64) {
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ td[j * 8 + k]);
    bestp = min(bestp, ((uint32_t)d << 16) | (uint32_t)j);
  }
  bestp = wmin_u32(bestp);
  if (lane == 0) {
    const int bj = (int)(bestp & 0xffffu);
    train_idx[row] = bj == 0xffff ? -1 : bj;
    dist[row] = bj == 0xffff ? -1 : (int)(bestp >> 16);
    if (bj != 0xffff) atomicMin(gmin, (int)(bestp >> 16));
  }
}
__global__ void k_bf_filter(int M, const int32_t* train_idx, const int32_t* dist,
                            const int32_t* gmin, uint8_t* keep, const int* M_ptr) {
  if (M_ptr) M = *M_ptr;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const double md = (double)min(*gmin, 10000);
  const double thr = 2 * md > 30.0 ? 2 * md : 30.0;
  keep[i] = (train_idx[i] >= 0 && (double)dist[i] <= thr) ? 1 : 0;
}

// ---------------------------------------------------------------------------------
int svo_launch_descriptor_distance(svo_ctx* ctx, const uint8_t* a, const uint8_t* b, int count,
                                   int32_t* dist) {
  if (count <= 0) return SVO_OK;
  hipLaunchKernelGGL(k_desc_dist, dim3((count + 255) / 256), dim3(256), 0, ctx->stream,
                     (const uint32_t*)a, (const uint32_t*)b, count, dist);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_launch_hamming_argmin(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                              const uint8_t* mask, int32_t* idx, int32_t* best, int32_t* second) {
  if (N > MAXT) return SVO_E_CAPACITY;
  if (M <= 0) return SVO_OK;
  SvoTimer tm(ctx, "k_hamming_argmin");
  hipLaunchKernelGGL(k_hamming_argmin, dim3((M + 3) / 4), dim3(256), 0, ctx->stream,
                     (const uint32_t*)q, M, (const uint32_t*)t, N, mask, idx, best, second);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_launch_match_greedy(svo_ctx* ctx, const uint8_t* q, const uint8_t* q_skip, int M,
                            const uint8_t* t, int N, uint8_t* assigned, int max_dist, float ratio,
                            int32_t* idx, int32_t* best, int32_t* second, uint8_t* accepted,
                            const float* q_xy, const float* t_xy, const int32_t* boxes, int n_boxes,
                            const double* F, uint8_t* vetoed) {
  if (N > MAXT) return SVO_E_CAPACITY;
  if (M <= 0) return SVO_OK;
  const int Npad = N <= 512 ? 512 : 1024;
  const size_t need = (size_t)M * Npad * sizeof(uint16_t);
  if (need > ctx->scratch_bytes / 2) return SVO_E_CAPACITY;
  uint16_t* D = reinterpret_cast<uint16_t*>(ctx->d_scratch + ctx->scratch_bytes / 2);
  {
    SvoTimer tm(ctx, "k_dist_matrix");
    hipLaunchKernelGGL(k_dist_matrix, dim3((M + 3) / 4), dim3(256), 0, ctx->stream,
                       (const uint32_t*)q, M, (const uint32_t*)t, N, Npad, D);
  }
  {
    SvoTimer tm(ctx, "k_greedy_serial");
    if (Npad == 512)
      hipLaunchKernelGGL(k_greedy_serial<8>, dim3(1), dim3(64), 0, ctx->stream, D, M, N, q_skip,
                         assigned, max_dist, ratio, idx, best, second, accepted, q_xy, t_xy, boxes,
                         n_boxes, F, vetoed);
    else
      hipLaunchKernelGGL(k_greedy_serial<16>, dim3(1), dim3(64), 0, ctx->stream, D, M, N, q_skip,
                         assigned, max_dist, ratio, idx, best, second, accepted, q_xy, t_xy, boxes,
                         n_boxes, F, vetoed);
  }
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_launch_bf_match(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                        int32_t* train_idx, int32_t* dist, uint8_t* keep) {
  if (N > MAXT) return SVO_E_CAPACITY;
  if (M <= 0) return SVO_OK;
  int32_t* gmin = reinterpret_cast<int32_t*>(ctx->d_scratch + ctx->scratch_bytes / 2);
  const int32_t big = 0x7fffffff;
  SVO_HIP(ctx, hipMemcpyAsync(gmin, &big, sizeof big, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_bf_nearest, dim3((M + 3) / 4), dim3(256), 0, ctx->stream, (const uint32_t*)q,
                     M, (const uint32_t*)t, N, train_idx, dist, gmin, (const int*)nullptr,
                     (const int*)nullptr);
  hipLaunchKernelGGL(k_bf_filter, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, M, train_idx,
                     dist, gmin, keep, (const int*)nullptr);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

// device-driven variant for the tracker: row counts live in HBM (M_ptr / N_ptr), grid sized for Mmax
int svo_launch_bf_match_dev(svo_ctx* ctx, const uint8_t* q, const int* M_ptr, const uint8_t* t,
                            const int* N_ptr, int Mmax, int32_t* train_idx, int32_t* dist,
                            uint8_t* keep, int32_t* gmin) {
  const int32_t big = 0x7fffffff;
  SVO_HIP(ctx, hipMemcpyAsync(gmin, &big, sizeof big, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_bf_nearest, dim3((Mmax + 3) / 4), dim3(256), 0, ctx->stream, (const uint32_t*)q,
                     0, (const uint32_t*)t, 0, train_idx, dist, gmin, M_ptr, N_ptr);
  hipLaunchKernelGGL(k_bf_filter, dim3((Mmax + 255) / 256), dim3(256), 0, ctx->stream, 0, train_idx,
                     dist, gmin, keep, M_ptr);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}
