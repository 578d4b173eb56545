// svo_track.hip - Tracking::Track (reference src/Tracking.cc:180-252) with all state in HBM.
//
// The ordered tail of a frame - Tracking::init / pnpmatch::poseEstimationPnP passes 1 and 2
// (src/pnpmatch.cc:61-199), the PnP initial pose (:212-247), Optimizer::PoseOptimization
// (src/Optimizer.cc:15-86), frame::createmappoint (src/frame.cc:182-238) and the local-map
// cull (src/Tracking.cc:239-250) - runs as nine small launches per frame with no host
// round trip: sizes, the frame counter and the map-point pool all live in one TrackState
// block.  The map-point pool is a structure of arrays kept in creation order (the
// deterministic stand-in for the reference's std::set<mappoint*> address order); after each
// frame it is stably compacted into the other half of a ping-pong buffer, so "local map
// point r" is simply row r and the pass-2 distance matrix needs no gather.
// The reference also stores match_score[i] = second/best for every pass-1 row
// (src/pnpmatch.cc:99); nothing ever reads it (its only use is commented out at
// src/Optimizer.cc:54), so the tracker does not materialise it - svo_match_greedy does.
// Offline detection boxes (main.cpp:82-95) gate the path exactly where the reference uses them:
// the +-5 px creation gates (src/Tracking.cc:61-66 with its never-reset flag, src/frame.cc:198-203)
// and the +-10 px epipolar veto of pass 1 (src/pnpmatch.cc:101-144) with F from the 8-point
// algorithm over brute-force matches (src/pnpmatch.cc:302-337; svo_fmat.hip, host side).
#include <cstddef>

#include "svo_internal.h"
#include "svo_wave.h"
#include "svo_gate.h"

#define TRK_MAXKP 512
#define TRK_CAP 4096

struct TrackPool {
  float pos[TRK_CAP * 3];
  uint32_t desc[TRK_CAP * 8];
  int32_t create_id[TRK_CAP];
  int32_t obs_frame[TRK_CAP];
  uint8_t bad[TRK_CAP];
  uint8_t in_local[TRK_CAP];
};

struct TrackState {
  int32_t frame_num, npool, lastN, cur;      // cur: active half of the pool ping-pong
  int32_t nkp, m1, m2, n_edges, skip_match;
  int32_t n_pass1, n_pass2, n_new, n_stereo;
  int32_t n_rows1, n_rows2, n_slow2;         // rows the serial passes visited / re-scanned (diagnostics)
  float lastTcw[16];
  int32_t last_mp[TRK_MAXKP];
  int32_t cur_mp[TRK_MAXKP];
  int32_t dbg_cur_mp[TRK_MAXKP];
  uint8_t assigned[TRK_MAXKP];
  double Xw[TRK_MAXKP * 3], obs[TRK_MAXKP * 2], K[4], Tprior[16], T[16];
  svo_lm_stats lm;
  svo_pnp_stats pnp;
  svo_camera cam;
  // semantic gating state
  int32_t n_boxes, n_vetoed;
  int32_t boxes[SVO_MAX_BOXES * 4];    // {left, right, top, bottom} of the current frame
  double F[9];                         // fundamental matrix cur <- last (row-major)
  float last_xy[TRK_MAXKP * 2];        // LastFrame.keypoints_l[i].pt
  uint32_t last_desc[TRK_MAXKP * 8];   // LastFrame.f_descriptor
  int32_t bf_idx[TRK_MAXKP], bf_dist[TRK_MAXKP], bf_min;
  uint8_t bf_keep[TRK_MAXKP];
  TrackPool pool[2];
  uint16_t rowmin[TRK_CAP];    // min over ALL current keypoints of the row's distances
  uint8_t active[TRK_CAP];     // row takes part in the greedy pass (valid && rowmin < threshold)
  uint2 pre[TRK_CAP];          // speculative row result under the claims at pass start:
                               //   x = best<<16 | idx, y = second<<16 | idx_of_second (0xffff: none)
  uint16_t D[(size_t)TRK_CAP * 512];
};

__device__ __forceinline__ uint32_t tk_wmin(uint32_t v) { return wave_min_u32_dpp(v); }

// exclusive scan of one int per thread over a 512-thread block; returns the total in *total
__device__ __forceinline__ int block_excl_scan512(int v, int* sm /*[512]*/, int* total) {
  const int tid = threadIdx.x;
  sm[tid] = v;
  __syncthreads();
  for (int o = 1; o < 512; o <<= 1) {
    const int t = tid >= o ? sm[tid - o] : 0;
    __syncthreads();
    sm[tid] += t;
    __syncthreads();
  }
  const int incl = sm[tid];
  *total = sm[511];
  __syncthreads();
  return incl - v;
}

__device__ __forceinline__ void tk_unproject(const svo_camera& cam, float u, float v, float z,
                                             const float* Rwc, const float* twc, float* xyz) {
  const float x = (u - cam.cx) * z * (1 / cam.fx);
  const float y = (v - cam.cy) * z * (1 / cam.fy);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double acc = (double)Rwc[3 * r] * (double)x + (double)Rwc[3 * r + 1] * (double)y +
                       (double)Rwc[3 * r + 2] * (double)z;
    xyz[r] = (float)(acc + (double)twc[r]);
  }
}

// ---- 1. frame begin: reset per-frame state; frame 0 runs Tracking::init -------------------
// Every tail kernel serves one sequence per blockIdx.y: TrackState number blockIdx.y and frame slot
// blockIdx.y of the front-end buffers (`kstride` keypoints per slot).  A single sequence is grid.y = 1.
#define TK_SEQ_SELECT(kstride)                                   \
  st += blockIdx.y;                                              \
  kp += (size_t)blockIdx.y * (kstride);

__global__ __launch_bounds__(512) void k_tk_begin(TrackState* st, const svo_kp* kp,
                                                  const uint32_t* desc, const int32_t* nkp_p,
                                                  const float* depth, int kstride) {
  __shared__ int sm[512];
  TK_SEQ_SELECT(kstride)
  desc += (size_t)blockIdx.y * kstride * 8; nkp_p += blockIdx.y; depth += (size_t)blockIdx.y * kstride;
  const int tid = threadIdx.x;
  const int nkp = min(*nkp_p, TRK_MAXKP);
  TrackPool& P = st->pool[st->cur];
  st->cur_mp[tid] = -1;
  st->assigned[tid] = 0;
  const bool has_depth = tid < nkp && depth[tid] > 0.f;
  int n_stereo;
  block_excl_scan512(has_depth ? 1 : 0, sm, &n_stereo);
  // Tracking::init's `dynamic` flag is declared outside the keypoint loop and never reset
  // (src/Tracking.cc:44): once a keypoint falls into a padded box, every later one is skipped.
  bool create0 = has_depth;
  if (st->frame_num == 0 && st->n_boxes > 0) {
    const svo_kp k = kp[min(tid, max(nkp - 1, 0))];
    const bool inb = tid < nkp && svo_in_boxes(k.x, k.y, st->boxes, st->n_boxes, 5);
    int tot_in;
    const int before = block_excl_scan512(inb ? 1 : 0, sm, &tot_in);
    if (before + (inb ? 1 : 0) > 0) create0 = false;
  }
  int total;
  const int rank = block_excl_scan512(create0 ? 1 : 0, sm, &total);
  if (tid == 0) {
    st->nkp = nkp;
    st->n_stereo = n_stereo;
    st->n_pass1 = 0; st->n_pass2 = 0; st->n_new = 0; st->n_vetoed = 0;
    for (int i = 0; i < 4; ++i) st->K[i] = (double)((const float*)&st->cam)[i];
    for (int i = 0; i < 16; ++i) st->Tprior[i] = (double)st->lastTcw[i];
  }
  if (st->frame_num == 0) {
    // Tracking::init (src/Tracking.cc:42-97): pose I, one map point per keypoint with depth
    if (create0) {
      const int m = st->npool + rank;
      const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
      const svo_kp k = kp[tid];
      tk_unproject(st->cam, k.x, k.y, depth[tid], I3, z3, &P.pos[3 * m]);
#pragma unroll
      for (int w = 0; w < 8; ++w) P.desc[8 * m + w] = desc[8 * tid + w];
      P.bad[m] = 0; P.in_local[m] = 1; P.create_id[m] = 0; P.obs_frame[m] = -1;
      st->cur_mp[tid] = m;
    }
    __syncthreads();
    if (tid == 0) {
      st->npool += total;
      st->n_new = total;
      st->skip_match = 1; st->m1 = 0; st->m2 = 0;
      for (int i = 0; i < 16; ++i) st->Tprior[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
  } else if (tid == 0) {
    st->skip_match = 0;
    st->m1 = st->lastN;
    st->m2 = st->npool;
  }
}

// ---- 2/4. distance matrix rows of one pass ------------------------------------------------
// pass 1: row i <-> last frame's keypoint i (map point last_mp[i]); pass 2: row r <-> pool row r.
// Also decides, in parallel, which rows the serial pass must visit: a row can only be accepted
// if its best distance over the unclaimed columns is < max_dist; the minimum over ALL columns
// bounds that from below, so rows failing it never claim a column and are dropped.
// TKD_ROWS map-point rows per workgroup: 4 (one per wave) for a single sequence, where the chain's latency counts and
// every row should get its own wave at once; 16 when many sequences are advanced together and the staging of the
// keypoint descriptors is worth sharing
template <int TKD_ROWS>
__global__ __launch_bounds__(256) void k_tk_dist(TrackState* st, const uint32_t* desc, int pass, int kstride) {
  // descriptors of the frame's keypoints, TRANSPOSED: word k of keypoint j at td[k * TRK_MAXKP + j], so that the 64 lanes
  // of a wave (64 consecutive keypoints) read 64 consecutive LDS words
  __shared__ uint32_t td[8 * TRK_MAXKP];
  st += blockIdx.y; desc += (size_t)blockIdx.y * kstride * 8;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int M = st->skip_match ? 0 : (pass == 1 ? st->m1 : st->m2);
  const int row0 = blockIdx.x * TKD_ROWS;
  if (row0 >= M) {
    if (tid < TKD_ROWS && row0 + tid < TRK_CAP) st->active[row0 + tid] = 0;
    return;
  }
  const int nkp = st->nkp, id = st->frame_num;
  const int max_dist = pass == 1 ? 15 : 30;
  const TrackPool& P = st->pool[st->cur];
  {
    // 16-byte loads, all four of a thread in flight together
    const uint4* d4 = reinterpret_cast<const uint4*>(desc);
    uint4 v[TRK_MAXKP * 2 / 256];
#pragma unroll
    for (int k = 0; k < TRK_MAXKP * 2 / 256; ++k) v[k] = d4[min(tid + 256 * k, max(2 * nkp - 1, 0))];
#pragma unroll
    for (int k = 0; k < TRK_MAXKP * 2 / 256; ++k) {
      const int i = tid + 256 * k;   // 16-byte piece i: keypoint i / 2, words 4 * (i & 1) ..
      if (i < 2 * nkp) {
        const int j = i >> 1, w = 4 * (i & 1);
        td[(w + 0) * TRK_MAXKP + j] = v[k].x; td[(w + 1) * TRK_MAXKP + j] = v[k].y;
        td[(w + 2) * TRK_MAXKP + j] = v[k].z; td[(w + 3) * TRK_MAXKP + j] = v[k].w;
      }
    }
  }
  __syncthreads();
  for (int rr = 0; rr < TKD_ROWS / 4; ++rr) {
    const int row = row0 + wv * (TKD_ROWS / 4) + rr;
    if (row >= TRK_CAP) break;
    bool valid = row < M;
    int m = row;
    if (valid) {
      if (pass == 1) {
        m = st->last_mp[row];
        valid = m >= 0 && !P.bad[m];
      } else {
        valid = P.in_local[m] && !P.bad[m] && P.obs_frame[m] != id;
      }
    }
    if (!valid) {
      if (lane == 0) st->active[row] = 0;
      continue;
    }
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = P.desc[8 * m + k];
    uint32_t mn = 0x7fff;
    int dcol[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      int d = 0x7fff;
      if (j < nkp) {
        d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ td[k * TRK_MAXKP + j]);
      }
      dcol[t] = d;
      mn = min(mn, (uint32_t)d);
      st->D[(size_t)row * 512 + j] = (uint16_t)d;
    }
    mn = tk_wmin(mn);
    if (lane == 0) {
      st->rowmin[row] = (uint16_t)mn;
      st->active[row] = (int)mn < max_dist ? 1 : 0;
    }
    if ((int)mn >= max_dist) continue;
    // Speculative result of this row under the claims at the START of the pass (pass 1: none,
    // pass 2: what pass 1 claimed).  Keys (dist<<16 | j) order by distance, then by column, so the
    // wave minimum is the reference's strict-`<` scan result (first minimum) and the minimum over
    // the columns before it is the "runner-up" (src/pnpmatch.cc:89-94) together with its column.
    uint32_t kb = 0xffffffffu;
    uint32_t keys[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      uint32_t key = 0xffffffffu;
      if (j < nkp && !(pass == 2 && st->assigned[j])) key = ((uint32_t)dcol[t] << 16) | (uint32_t)j;
      keys[t] = key;
      kb = min(kb, key);
    }
    kb = tk_wmin(kb);
    uint32_t ks = 0xffffffffu;
    const uint32_t bj = kb & 0xffffu;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if ((keys[t] & 0xffffu) < bj) ks = min(ks, keys[t]);
    ks = tk_wmin(ks);
    if (lane == 0) {
      const uint32_t y = ks == 0xffffffffu ? ((256u << 16) | 0xffffu) : ks;
      // bit 31 of x: the row would be accepted with this (speculative) result
      const int bd = (int)(kb >> 16), sec = (int)(y >> 16);
      bool ok = (kb & 0xffffu) != 0xffffu && bd < max_dist;
      if (ok && pass == 2) ok = (float)sec / (float)bd > 2.f;
      st->pre[row] = make_uint2(kb | (ok ? 0x80000000u : 0u), y);
    }
  }
}

// ---- 3/5. the order-dependent greedy assignment, one wave -----------------------------------
// The row flags written by k_tk_dist are loaded in ONE round trip (64 bytes per lane), compacted
// in order into LDS with a wave prefix sum, and the surviving rows are walked with an 8-deep
// register prefetch of their distance rows, so the serial chain pays ALU time, not an L2 round
// trip, per row.  Lane L owns columns 8L..8L+7 and keeps their claim bits in a register.
__device__ __forceinline__ int tk_wave_incl_scan(int v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if ((int)(threadIdx.x & 63) >= o) v += t;
  }
  return v;
}

__global__ __launch_bounds__(64) void k_tk_greedy(TrackState* st, int pass, const svo_kp* kp, int kstride) {
  TK_SEQ_SELECT(kstride)
  __shared__ int16_t rows[TRK_CAP];
  __shared__ int16_t rowmp[TRK_CAP];
  __shared__ uint4 drow[32 * 64];   // distance rows of the current 32-row chunk (32 KB)
  __shared__ uint8_t claimedB[512]; // columns claimed during THIS pass
  const int lane = threadIdx.x;
  const int M = pass == 1 ? st->m1 : st->m2;
  if (M <= 0 || st->skip_match) return;
  TrackPool& P = st->pool[st->cur];
  const int nkp = st->nkp, id = st->frame_num;
  const int max_dist = pass == 1 ? 15 : 30;
  const float ratio = pass == 1 ? 0.f : 2.f;
  // flags of rows 64*lane .. 64*lane+63
  uint4 fl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    fl[k] = make_uint4(0, 0, 0, 0);
    if (64 * lane + 16 * k < M) fl[k] = *reinterpret_cast<const uint4*>(&st->active[64 * lane + 16 * k]);
  }
  const uint32_t fw[16] = {fl[0].x, fl[0].y, fl[0].z, fl[0].w, fl[1].x, fl[1].y, fl[1].z, fl[1].w,
                           fl[2].x, fl[2].y, fl[2].z, fl[2].w, fl[3].x, fl[3].y, fl[3].z, fl[3].w};
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) cnt += __popc(fw[k] & 0x01010101u);
  const int incl = tk_wave_incl_scan(cnt);
  const int n = __shfl(incl, 63, 64);
  int pos = incl - cnt;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int r = 64 * lane + 4 * k + b;
      if (((fw[k] >> (8 * b)) & 1u) && r < M) rows[pos++] = (int16_t)r;
    }
  }
  __syncthreads();
  for (int a = lane; a < n; a += 64) rowmp[a] = (int16_t)(pass == 1 ? st->last_mp[rows[a]] : rows[a]);
  __syncthreads();
  uint32_t claimed = 0;       // all claims (initial + this pass), columns 8*lane .. 8*lane+7
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int j = lane * 8 + k;
    if (j >= nkp || st->assigned[j]) claimed |= 1u << k;
  }
  const uint16_t* Dl = st->D + lane * 8;
  int accepted_total = 0, n_slow = 0, n_veto = 0;
  const int n_boxes = st->n_boxes;
  // A row's speculative result (st->pre) stays valid unless one of the two columns it depends on -
  // its best column or the column of its runner-up - was claimed during this pass: removing any
  // other column cannot change a minimum that is still present.  So the only rows that can change
  // any state are EVENTS: rows that would be accepted, and rows whose result went stale.  Each
  // 32-row chunk is evaluated by 32 lanes in parallel against the claim bytes in LDS; the wave
  // then jumps from event to event (ballot + ffs) instead of walking every row.
  for (int j = lane; j < 512; j += 64) claimedB[j] = 0;
  for (int a0 = 0; a0 < n; a0 += 32) {
    const int cntb = min(32, n - a0);
    const int mine = a0 + lane;
    uint2 pv = make_uint2(0x0000ffffu, 0xffffffffu);
    if (lane < cntb) pv = st->pre[rows[mine]];
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < cntb; ++k) drow[k * 64 + lane] = *reinterpret_cast<const uint4*>(Dl + (size_t)rows[a0 + k] * 512);
    __syncthreads();
    const int my_bj = (int)(pv.x & 0xffffu), my_js = (int)(pv.y & 0xffffu);
    const bool my_ok = (pv.x >> 31) != 0;
    int start = 0;
    for (;;) {
      bool my_stale = false;
      if (lane >= start && lane < cntb && my_bj != 0xffff)
        my_stale = claimedB[my_bj] != 0 || (my_js != 0xffff && claimedB[my_js] != 0);
      const bool my_event = lane >= start && lane < cntb && (my_stale || my_ok);
      const uint64_t em = __ballot(my_event);
      if (em == 0) break;
      const int k = __ffsll((long long)em) - 1;              // first event row of the chunk
      const bool stale = (__ballot(my_stale) >> k) & 1ull;
      const uint32_t kb = __builtin_amdgcn_readlane((int)pv.x, k);
      const uint32_t ks = __builtin_amdgcn_readlane((int)pv.y, k);
      int bj = (int)(kb & 0xffffu), bd = (int)((kb >> 16) & 0x7fffu), sec = (int)(ks >> 16);
      bool ok = (kb >> 31) != 0;
      if (stale) {
        ++n_slow;
        const uint4 v = drow[k * 64 + lane];
        const uint32_t cur[8] = {v.x & 0xffff, v.x >> 16, v.y & 0xffff, v.y >> 16,
                                 v.z & 0xffff, v.z >> 16, v.w & 0xffff, v.w >> 16};
        uint32_t lp = (256u << 16) | 0xffffu;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (!((claimed >> c) & 1u)) lp = min(lp, (cur[c] << 16) | (uint32_t)(lane * 8 + c));
        const uint32_t bp = tk_wmin(lp);
        bj = (int)(bp & 0xffffu); bd = (int)(bp >> 16);
        uint32_t ls = 256;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (!((claimed >> c) & 1u) && lane * 8 + c < bj) ls = min(ls, cur[c]);
        sec = (int)tk_wmin(ls);
        ok = bj != 0xffff && bd < max_dist;
        if (ok && ratio > 0.f) ok = (float)sec / (float)bd > ratio;
      }
      if (ok && pass == 1 && n_boxes > 0) {
        // epipolar veto (src/pnpmatch.cc:103-144): the match lands in a padded box and is off
        // the epipolar line -> the map point is marked bad and claims nothing
        const svo_kp kc = kp[bj];
        const int i_last = rows[a0 + k];
        if (svo_in_boxes(kc.x, kc.y, st->boxes, n_boxes, 10) &&
            svo_epipolar_distance(st->F, st->last_xy[2 * i_last], st->last_xy[2 * i_last + 1], kc.x, kc.y) > 0.1) {
          if (lane == 0) P.bad[rowmp[a0 + k]] = 1;
          ++n_veto;
          ok = false;
        }
      }
      if (ok) {
        if ((bj >> 3) == lane) claimed |= 1u << (bj & 7);
        if (lane == 0) {
          claimedB[bj] = 1;
          const int m = rowmp[a0 + k];
          st->cur_mp[bj] = m;
          P.obs_frame[m] = id;
        }
        ++accepted_total;
        __syncthreads();   // claimedB visible to the re-evaluation below
      }
      start = k + 1;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int j = lane * 8 + k;
    if (j < nkp) st->assigned[j] = (uint8_t)((claimed >> k) & 1u);
  }
  if (lane == 0) {
    if (pass == 1) { st->n_pass1 = accepted_total; st->n_rows1 = n; st->n_vetoed = n_veto; }
    else { st->n_pass2 = accepted_total; st->n_rows2 = n; st->n_slow2 = n_slow; }
  }
}

// ---- 6. gather the 3D-2D correspondences (ordered by keypoint index) -----------------------
__global__ __launch_bounds__(512) void k_tk_gather(TrackState* st, const svo_kp* kp, int kstride) {
  TK_SEQ_SELECT(kstride)
  __shared__ int sm[512];
  const int tid = threadIdx.x;
  const TrackPool& P = st->pool[st->cur];
  const int m = tid < st->nkp ? st->cur_mp[tid] : -1;
  st->dbg_cur_mp[tid] = m;
  int total;
  const int pos = block_excl_scan512(m >= 0 ? 1 : 0, sm, &total);
  if (m >= 0) {
    st->Xw[3 * pos] = (double)P.pos[3 * m];
    st->Xw[3 * pos + 1] = (double)P.pos[3 * m + 1];
    st->Xw[3 * pos + 2] = (double)P.pos[3 * m + 2];
    const svo_kp k = kp[tid];
    st->obs[2 * pos] = (double)k.x;
    st->obs[2 * pos + 1] = (double)k.y;
  }
  if (tid == 0) st->n_edges = total;
}

// ---- 9. frame end: SetPose, result record, createmappoint, cull, compaction ----------------
__global__ __launch_bounds__(512) void k_tk_end(TrackState* st, const svo_kp* kp,
                                                const uint32_t* desc, const float* depth,
                                                svo_track_result* res_out, int kstride) {
  TK_SEQ_SELECT(kstride)
  desc += (size_t)blockIdx.y * kstride * 8; depth += (size_t)blockIdx.y * kstride; res_out += blockIdx.y;
  __shared__ int sm[512];
  __shared__ float sT[16], sRwc[9], stwc[3];
  const int tid = threadIdx.x;
  const int nkp = st->nkp, id = st->frame_num;
  TrackPool& P = st->pool[st->cur];
  TrackPool& Q = st->pool[st->cur ^ 1];
  if (tid < 16) sT[tid] = (float)st->T[tid];   // SetPose(pose) stores CV_32F (src/Optimizer.cc:82-83)
  __syncthreads();
  if (tid < 9) sRwc[tid] = sT[4 * (tid % 3) + tid / 3];
  __syncthreads();
  if (tid < 3) {
    const double acc = (double)sRwc[3 * tid] * (double)sT[3] + (double)sRwc[3 * tid + 1] * (double)sT[7] +
                       (double)sRwc[3 * tid + 2] * (double)sT[11];
    stwc[tid] = (float)(-acc);
  }
  __syncthreads();
  // frame::createmappoint for keypoints without a map point and with depth
  int m_cur = tid < nkp ? st->cur_mp[tid] : -1;
  bool create = tid < nkp && m_cur < 0 && depth[tid] > 0.f;
  if (tid < nkp) {
    const svo_kp k = kp[tid];
    if (create && st->n_boxes > 0 && svo_in_boxes(k.x, k.y, st->boxes, st->n_boxes, 5)) create = false;
    st->last_xy[2 * tid] = k.x; st->last_xy[2 * tid + 1] = k.y;
#pragma unroll
    for (int w = 0; w < 8; ++w) st->last_desc[8 * tid + w] = desc[8 * tid + w];
  }
  int n_new;
  const int rank = block_excl_scan512(create ? 1 : 0, sm, &n_new);
  const int np0 = st->npool;
  if (create && np0 + rank < TRK_CAP) {
    const int m = np0 + rank;
    const svo_kp k = kp[tid];
    tk_unproject(st->cam, k.x, k.y, depth[tid], sRwc, stwc, &P.pos[3 * m]);
#pragma unroll
    for (int w = 0; w < 8; ++w) P.desc[8 * m + w] = desc[8 * tid + w];
    P.bad[m] = 0; P.in_local[m] = 1; P.create_id[m] = id; P.obs_frame[m] = -1;
    m_cur = m;
  }
  __syncthreads();
  const int np1 = min(np0 + n_new, TRK_CAP);
  // cull + liveness: thread t owns pool rows 8t .. 8t+7
  if (tid < TRK_MAXKP) st->last_mp[tid] = m_cur;   // pre-compaction indices
  __syncthreads();
  // mark rows referenced by the (new) last frame: reuse obs_frame sign? use a flag pass in Q.bad
  for (int r = tid; r < TRK_CAP; r += 512) Q.bad[r] = 0;
  __syncthreads();
  if (tid < nkp && m_cur >= 0) Q.bad[m_cur] = 1;   // Q.bad is scratch here: "referenced" flag
  __syncthreads();
  int live_cnt = 0;
  uint32_t live_bits = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = tid * 8 + k;
    if (r < np1) {
      bool loc = P.in_local[r] != 0;
      if (id >= 4 && P.create_id[r] <= id - 4) loc = false;   // cull (src/Tracking.cc:239-250)
      P.in_local[r] = loc ? 1 : 0;
      if (loc || Q.bad[r]) { live_bits |= 1u << k; ++live_cnt; }
    }
  }
  int total_live;
  int base = block_excl_scan512(live_cnt, sm, &total_live);
  __syncthreads();
  for (int r = tid; r < TRK_CAP; r += 512) Q.bad[r] = 0;
  __syncthreads();
  // remap table lives in Q.obs_frame temporarily (rows of P -> rows of Q)
  int nl = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = tid * 8 + k;
    if ((live_bits >> k) & 1u) {
      const int q = base++;
      Q.pos[3 * q] = P.pos[3 * r]; Q.pos[3 * q + 1] = P.pos[3 * r + 1]; Q.pos[3 * q + 2] = P.pos[3 * r + 2];
#pragma unroll
      for (int w = 0; w < 8; ++w) Q.desc[8 * q + w] = P.desc[8 * r + w];
      Q.create_id[q] = P.create_id[r];
      Q.bad[q] = P.bad[r];
      Q.in_local[q] = P.in_local[r];
      nl += P.in_local[r];
      P.obs_frame[r] = q;            // remap (P is dead after this kernel)
    } else if (r < np1) {
      P.obs_frame[r] = -1;
    }
  }
  int nl_total;
  block_excl_scan512(nl, sm, &nl_total);
  __syncthreads();
  if (tid < TRK_MAXKP) {
    const int m = st->last_mp[tid];
    st->last_mp[tid] = (tid < nkp && m >= 0) ? P.obs_frame[m] : -1;
  }
  for (int q = tid; q < total_live; q += 512) Q.obs_frame[q] = -1;
  if (tid == 0) {
    svo_track_result r;
    for (int i = 0; i < 16; ++i) { r.Tcw[i] = sT[i]; st->lastTcw[i] = sT[i]; }
    r.frame_id = id; r.n_kp = nkp; r.n_stereo = st->n_stereo;
    r.n_match_pass1 = st->n_pass1; r.n_match_pass2 = st->n_pass2;
    r.n_pnp_inliers = st->skip_match ? 0 : st->pnp.n_inliers;
    r.n_lm_edges = st->n_edges;
    r.n_new_mappoints = (id == 0 ? st->n_new : 0) + n_new;
    r.n_local_map = nl_total;
    r.lm_iterations = st->lm.iterations;
    r.reserved[0] = st->skip_match ? 0 : st->n_rows1;   // diagnostics: rows visited by pass 1; [1]: pass 2 | rescanned<<16
    r.reserved[1] = st->skip_match ? 0 : (st->n_rows2 | (st->n_slow2 << 16));
    *res_out = r;
    st->lastN = nkp;
    st->npool = total_live;
    st->cur ^= 1;
    st->frame_num = id + 1;
  }
}

// --------------------------------------------------------------------------------------------
// One frame of `nseq` sequences at once: sequence q uses TrackState q, frame slot `slot + q` and result
// record d_res[q].  nseq = 1 is the ordinary single chain.
static int tail_launch(svo_ctx* ctx, int slot, svo_track_result* d_res, int nseq = 1) {
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  hipStream_t s = ctx->stream;
  const size_t K = ctx->max_kp;
  const int ks = (int)K;
  const unsigned ny = (unsigned)nseq;
  const svo_kp* kp = ctx->d_kp + slot * K;
  const uint32_t* desc = reinterpret_cast<const uint32_t*>(ctx->d_desc + slot * K * 32);
  const float* depth = ctx->d_depth + slot * K;
  {
    SvoTimer t(ctx, "k_tk_begin");
    hipLaunchKernelGGL(k_tk_begin, dim3(1, ny), dim3(512), 0, s, st, kp, desc, ctx->d_nkp + slot, depth, ks);
  }
  {
    SvoTimer t(ctx, "k_tk_match");
    if (ny >= 8) hipLaunchKernelGGL(k_tk_dist<16>, dim3(TRK_MAXKP / 16, ny), dim3(256), 0, s, st, desc, 1, ks);
    else hipLaunchKernelGGL(k_tk_dist<4>, dim3(TRK_MAXKP / 4, ny), dim3(256), 0, s, st, desc, 1, ks);
    hipLaunchKernelGGL(k_tk_greedy, dim3(1, ny), dim3(64), 0, s, st, 1, kp, ks);
    if (ny >= 8) hipLaunchKernelGGL(k_tk_dist<16>, dim3(TRK_CAP / 16, ny), dim3(256), 0, s, st, desc, 2, ks);
    else hipLaunchKernelGGL(k_tk_dist<4>, dim3(TRK_CAP / 4, ny), dim3(256), 0, s, st, desc, 2, ks);
    hipLaunchKernelGGL(k_tk_greedy, dim3(1, ny), dim3(64), 0, s, st, 2, kp, ks);
  }
  {
    SvoTimer t(ctx, "k_tk_gather");
    hipLaunchKernelGGL(k_tk_gather, dim3(1, ny), dim3(512), 0, s, st, kp, ks);
  }
  int rc;
  if ((rc = svo_launch_pnp_dev(ctx, st->Xw, st->obs, &st->n_edges, st->K, st->Tprior, st->T, &st->pnp,
                               &st->skip_match, &st->frame_num, nseq, sizeof(TrackState))))
    return rc;
  if ((rc = svo_launch_pose_opt_dev(ctx, st->Xw, st->obs, &st->n_edges, st->K, st->T, &st->lm, 1, nseq, sizeof(TrackState))))
    return rc;
  {
    SvoTimer t(ctx, "k_tk_end");
    hipLaunchKernelGGL(k_tk_end, dim3(1, ny), dim3(512), 0, s, st, kp, desc, depth, d_res, ks);
  }
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

// (Re)allocate `nseq` tracker states and reset them: pose I, empty map, frame counter 0.
static int track_reset_n(svo_ctx* ctx, const svo_camera* cam, int nseq) {
  if (!ctx || !cam || nseq < 1) return SVO_E_INVALID;
  if (ctx->max_kp > TRK_MAXKP) return SVO_E_CAPACITY;
  hipSetDevice(ctx->device);
  if (!ctx->d_track || ctx->n_seq != nseq) {
    if (ctx->d_track) { hipStreamSynchronize(ctx->stream); hipFree(ctx->d_track); ctx->d_track = nullptr; }
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(TrackState) * (size_t)nseq) != hipSuccess) return SVO_E_NOMEM;
    ctx->d_track = p;
    ctx->n_seq = nseq;
  }
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int q = 0; q < nseq; ++q) {
    TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track) + q;
    // zero the scalar header + index arrays (everything before the pools), then set identity pose
    SVO_HIP(ctx, hipMemsetAsync(st, 0, offsetof(TrackState, pool), ctx->stream));
    SVO_HIP(ctx, hipMemcpyAsync(st->lastTcw, I, sizeof I, hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(ctx, hipMemcpyAsync(&st->cam, cam, sizeof *cam, hipMemcpyHostToDevice, ctx->stream));
  }
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->cam = *cam;
  ctx->track_frame = 0;
  return SVO_OK;
}

extern "C" int svo_track_reset(svo_ctx* ctx, const svo_camera* cam) { return track_reset_n(ctx, cam, 1); }

extern "C" int svo_track_multi_reset(svo_ctx* ctx, int n_seq, const svo_camera* cam) {
  if (!ctx) return SVO_E_INVALID;
  if (n_seq < 1 || n_seq > ctx->max_batch) return SVO_E_CAPACITY;
  return track_reset_n(ctx, cam, n_seq);
}

extern "C" int svo_fundamental_8point(const double* pts1, const double* pts2, int n, double F[9]);

// pnpmatch::poseEstimation2D_2D (src/pnpmatch.cc:302-337) for the frame in slot 0: brute-force
// matches cur -> last on the device, then the (tiny) 8-point solve on the host.
static int estimate_F(svo_ctx* ctx, const int32_t* boxes, int n_boxes) {
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  const int K = ctx->max_kp;
  int rc = svo_launch_bf_match_dev(ctx, ctx->d_desc, ctx->d_nkp, reinterpret_cast<const uint8_t*>(st->last_desc),
                                   &st->lastN, K, st->bf_idx, st->bf_dist, st->bf_keep, &st->bf_min);
  if (rc) return rc;
  std::vector<int32_t> idx(K);
  std::vector<uint8_t> keep(K);
  std::vector<svo_kp> kp(K);
  std::vector<float> lxy(2 * (size_t)TRK_MAXKP);
  int32_t nkp = 0;
  SVO_HIP(ctx, hipMemcpyAsync(idx.data(), st->bf_idx, 4 * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(keep.data(), st->bf_keep, (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(kp.data(), ctx->d_kp, sizeof(svo_kp) * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(lxy.data(), st->last_xy, sizeof(float) * lxy.size(), hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(&nkp, ctx->d_nkp, 4, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<double> p1, p2;
  for (int i = 0; i < nkp && i < K; ++i) {
    if (!keep[i] || idx[i] < 0) continue;
    if (svo_in_boxes(kp[i].x, kp[i].y, boxes, n_boxes, 10)) continue;   // :318-328
    p1.push_back(kp[i].x); p1.push_back(kp[i].y);
    p2.push_back(lxy[2 * idx[i]]); p2.push_back(lxy[2 * idx[i] + 1]);
  }
  double F[9];
  svo_fundamental_8point(p1.data(), p2.data(), (int)p1.size() / 2, F);
  SVO_HIP(ctx, hipMemcpyAsync(st->F, F, sizeof F, hipMemcpyHostToDevice, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));   // F lives on this stack frame
  return SVO_OK;
}

// Depth source 1 (svo_set_option "depth_source"): the reference's live data flow - a dense disparity map
// (src/Tracking.cc:226 `MB`, here the ELAS map D1) -> frame::disp2Depth (src/frame.cc:140-164: depth =
// bf / disp wherever disp != 0, else -1) -> `depthimg.at<float>(y, x)` at the truncated keypoint position;
// keypoints_r = x - disp unless disp == -1 (src/frame.cc:122-138).  D == nullptr: no map, no depth.
// blockIdx.y = frame slot of a batch: map `map_stride` floats further on (0 floats: one frame), `produced` (may
// be null) says whether that frame has a map at all.
__global__ void k_tk_dense_depth(const svo_kp* kp, const int32_t* nkp, const float* D, int W, float bf,
                                 float* uR, float* depth, int K, size_t map_stride, const int32_t* produced) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  kp += (size_t)blockIdx.y * K; nkp += blockIdx.y; uR += (size_t)blockIdx.y * K; depth += (size_t)blockIdx.y * K;
  if (D) D += (size_t)blockIdx.y * map_stride;
  if (produced && !produced[blockIdx.y]) D = nullptr;
  float u = -1.0f, z = -1.0f;
  if (i < *nkp && D) {
    const float disp = D[(size_t)(int)kp[i].y * W + (int)kp[i].x];
    if (disp != -1.0f) u = kp[i].x - disp;
    if (disp != 0.0f) z = bf / disp;
  }
  uR[i] = u; depth[i] = z;
}

// room for B frames' dense maps (two float maps and one flag per frame)
static int dense_reserve(svo_ctx* ctx, int B) {
  if (ctx->dense_cap >= B) return SVO_OK;
  const size_t n = (size_t)ctx->g.W * ctx->g.H;
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d_dense) hipFree(ctx->d_dense);
  ctx->d_dense = nullptr; ctx->dense_cap = 0;
  SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_dense), (2 * n * sizeof(float) + sizeof(int32_t)) * (size_t)B));
  ctx->dense_cap = B;
  return SVO_OK;
}

extern "C" int svo_track_frame(svo_ctx* ctx, const uint8_t* grayL, int strideL,
                               const uint8_t* grayR, int strideR, double timestamp,
                               const int32_t* boxes, int n_boxes, svo_track_result* res) {
  (void)timestamp;
  if (!ctx || !grayL || !grayR || !res || strideL < ctx->g.W || strideR < ctx->g.W || n_boxes < 0 ||
      n_boxes > SVO_MAX_BOXES || (n_boxes > 0 && !boxes))
    return SVO_E_INVALID;
  if (!ctx->d_track || ctx->n_seq != 1) return SVO_E_INVALID;   // svo_track_reset first
  hipSetDevice(ctx->device);
  const SvoGeom& g = ctx->g;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  uint8_t* dL = ctx->d_stage;
  uint8_t* dR = ctx->d_stage + (size_t)g.H * ctx->stage_pitch;
  int rc = svo_upload_image(ctx, grayL, strideL, 0);
  if (rc) return rc;
  if ((rc = svo_upload_image(ctx, grayR, strideR, 1))) return rc;
  const int32_t nb = n_boxes;
  SVO_HIP(ctx, hipMemcpyAsync(&st->n_boxes, &nb, 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_boxes > 0)
    SVO_HIP(ctx, hipMemcpyAsync(st->boxes, boxes, 16 * (size_t)n_boxes, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->opt_depth_source == 1) {
    if ((rc = svo_launch_orb(ctx, dL, dR, ctx->stage_pitch, 1, 1))) return rc;   // left image only
    svo_elas_params ep;
    svo_elas_default_params(0, &ep);
    float *dD1 = nullptr, *dD2 = nullptr;
    int produced = 0;
    if ((rc = svo_elas_run_dev(ctx, dL, dR, ctx->stage_pitch, g.W, g.H, &ep, &dD1, &dD2, &produced))) return rc;
    SvoTimer t(ctx, "k_tk_dense_depth");
    hipLaunchKernelGGL(k_tk_dense_depth, dim3((ctx->max_kp + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_kp,
                       ctx->d_nkp, produced ? dD1 : nullptr, g.W, ctx->cam.bf, ctx->d_uR, ctx->d_depth, ctx->max_kp, (size_t)0,
                       (const int32_t*)nullptr);
  } else if (ctx->opt_depth_source == 2) {
    // the reference's live configuration: frame::MB = MSA::solve(left, right, 48, 1) (src/Tracking.cc:225-228)
    if ((rc = svo_launch_orb(ctx, dL, dR, ctx->stage_pitch, 1, 1))) return rc;
    if ((rc = dense_reserve(ctx, 1))) return rc;
    if ((rc = svo_msa_run_dev(ctx, dL, dR, ctx->stage_pitch, g.W, g.H, 48, ctx->d_dense))) return rc;
    SvoTimer t(ctx, "k_tk_dense_depth");
    hipLaunchKernelGGL(k_tk_dense_depth, dim3((ctx->max_kp + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_kp,
                       ctx->d_nkp, ctx->d_dense, g.W, ctx->cam.bf, ctx->d_uR, ctx->d_depth, ctx->max_kp, (size_t)0,
                       (const int32_t*)nullptr);
  } else {
    if ((rc = svo_launch_orb(ctx, dL, dR, ctx->stage_pitch, 1, 2))) return rc;
    if ((rc = svo_launch_stereo(ctx, dL, dR, ctx->stage_pitch, 1, &ctx->cam))) return rc;
  }
  if (n_boxes > 0 && ctx->track_frame > 0) {
    rc = estimate_F(ctx, boxes, n_boxes);
    if (rc) return rc;
  }
  svo_track_result* d_res = reinterpret_cast<svo_track_result*>(ctx->d_scratch);
  rc = tail_launch(ctx, 0, d_res);
  if (rc) return rc;
  SVO_HIP(ctx, hipMemcpyAsync(res, d_res, sizeof *res, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->track_frame++;
  return SVO_OK;
}

// One time step of n_seq independent sequences: pair q is the next frame of sequence q.
extern "C" int svo_track_multi_step_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR,
                                        int stride, int n_seq, svo_track_result* d_results) {
  if (!ctx || !d_grayL || !d_grayR || !d_results || stride < ctx->g.W) return SVO_E_INVALID;
  if (!ctx->d_track || n_seq != ctx->n_seq) return SVO_E_INVALID;   // svo_track_multi_reset(n_seq) first
  if (ctx->opt_depth_source != 0) {   // the many-sequence mode has the sparse matcher only
    ctx->last_error = "svo_track_multi_step_dev: depth_source must be 0";
    return SVO_E_INVALID;
  }
  hipSetDevice(ctx->device);
  int rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, n_seq, 2 * n_seq);
  if (rc) return rc;
  if ((rc = svo_launch_stereo(ctx, d_grayL, d_grayR, stride, n_seq, &ctx->cam))) return rc;
  if ((rc = tail_launch(ctx, 0, d_results, n_seq))) return rc;
  ctx->track_frame++;
  return SVO_OK;
}

extern "C" int svo_track_batch_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR,
                                   int stride, int B, svo_track_result* d_results) {
  if (!ctx || !d_grayL || !d_grayR || !d_results || B < 1 || stride < ctx->g.W) return SVO_E_INVALID;
  if (B > ctx->max_batch) return SVO_E_CAPACITY;
  if (!ctx->d_track || ctx->n_seq != 1) return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  // the batched mode carries no detection boxes (the offline box files are a per-frame host input)
  SVO_HIP(ctx, hipMemsetAsync(&reinterpret_cast<TrackState*>(ctx->d_track)->n_boxes, 0, 4, ctx->stream));
  int rc;
  if (ctx->opt_depth_source == 1) {
    // dense ELAS maps for the B frames (svo_elas_batch_dev), then the reference's per-keypoint lookups
    const size_t n = (size_t)ctx->g.W * ctx->g.H;
    if ((rc = dense_reserve(ctx, B))) return rc;
    float* dD1 = ctx->d_dense;
    float* dD2 = dD1 + n * (size_t)B;
    int32_t* d_prod = reinterpret_cast<int32_t*>(dD2 + n * (size_t)B);
    if ((rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, B, B))) return rc;   // left images only
    svo_elas_params ep;
    svo_elas_default_params(0, &ep);
    std::vector<int32_t> prod(B, 0);
    if ((rc = svo_elas_batch_dev(ctx, d_grayL, d_grayR, stride, ctx->g.W, ctx->g.H, B, &ep, dD1, dD2, prod.data()))) return rc;
    SVO_HIP(ctx, hipMemcpyAsync(d_prod, prod.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_tk_dense_depth, dim3((ctx->max_kp + 255) / 256, B), dim3(256), 0, ctx->stream, ctx->d_kp, ctx->d_nkp,
                       dD1, ctx->g.W, ctx->cam.bf, ctx->d_uR, ctx->d_depth, ctx->max_kp, n, d_prod);
    SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `prod` is on this stack frame
  } else if (ctx->opt_depth_source == 2) {
    // MSA maps, up to eight frames in flight (most of a solve is the host tree builds)
    const size_t n = (size_t)ctx->g.W * ctx->g.H;
    if ((rc = dense_reserve(ctx, B))) return rc;
    if ((rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, B, B))) return rc;   // left images only
    if ((rc = svo_msa_run_many_dev(ctx, d_grayL, d_grayR, stride, (size_t)ctx->g.H * stride, ctx->g.W, ctx->g.H, 48, B,
                                   ctx->d_dense)))
      return rc;
    hipLaunchKernelGGL(k_tk_dense_depth, dim3((ctx->max_kp + 255) / 256, B), dim3(256), 0, ctx->stream, ctx->d_kp, ctx->d_nkp,
                       ctx->d_dense, ctx->g.W, ctx->cam.bf, ctx->d_uR, ctx->d_depth, ctx->max_kp, n, (const int32_t*)nullptr);
  } else {
    if ((rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, B, 2 * B))) return rc;
    if ((rc = svo_launch_stereo(ctx, d_grayL, d_grayR, stride, B, &ctx->cam))) return rc;
  }
  for (int f = 0; f < B; ++f) {
    rc = tail_launch(ctx, f, d_results + f);
    if (rc) return rc;
  }
  ctx->track_frame += B;
  return SVO_OK;
}

extern "C" int svo_debug_track_matches(svo_ctx* ctx, int32_t* cur_mp) {
  if (!ctx || !cur_mp || !ctx->d_track) return SVO_E_INVALID;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  SVO_HIP(ctx, hipMemcpyAsync(cur_mp, st->dbg_cur_mp, sizeof(int32_t) * ctx->max_kp,
                              hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_debug_track_gate(svo_ctx* ctx, double F[9], int32_t* n_vetoed) {
  if (!ctx || !ctx->d_track) return SVO_E_INVALID;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  if (F) SVO_HIP(ctx, hipMemcpyAsync(F, st->F, 72, hipMemcpyDeviceToHost, ctx->stream));
  if (n_vetoed) SVO_HIP(ctx, hipMemcpyAsync(n_vetoed, &st->n_vetoed, 4, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}
