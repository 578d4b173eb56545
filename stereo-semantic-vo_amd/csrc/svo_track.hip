// svo_track.hip - Tracking::Track (reference src/Tracking.cc:180-252) with all state in HBM.
//
// The ordered tail of a frame is TWO chains that meet only through map-point identities:
//
//   index chain (pose-free)                         pose chain
//   -----------------------                         ----------
//   Tracking::init            src/Tracking.cc:42-97   gather 3D-2D correspondences   src/pnpmatch.cc:216-224
//   pnpmatch pass 1 + veto    src/pnpmatch.cc:61-156  cv::solvePnPRansac             src/pnpmatch.cc:227
//   pnpmatch pass 2           src/pnpmatch.cc:159-199 Optimizer::PoseOptimization    src/Optimizer.cc:15-86
//   frame::createmappoint     src/frame.cc:182-238    SetPose, UnprojectStereo of    src/frame.cc:66-73,166-180
//   local-map cull            src/Tracking.cc:239-250   the points created this frame
//
// In the reference, matching never reads the pose: pnpmatch::poseEstimationPnP compares descriptors and
// `MapPoints[j]` only (Rcw/tcw at src/pnpmatch.cc:56-57 are unused), the epipolar veto takes F from brute-force
// matches, and which map points exist / are culled depends on descriptors, depth > 0, boxes and frame ids.  The pose
// only decides WHERE a new map point lies - which only later PnP / LM calls consume.  So the index chain runs ahead
// on its own HIP stream (two launches per frame: k_ti_lists, k_ti_resolve) and hands each frame's correspondences to
// the pose chain as map-point ids (`TrackWork`); the pose chain (ONE launch per frame: k_tp_frame) follows on the
// context stream and keeps the positions in a table indexed by map-point id.  Results are what the single ordered
// chain produces, record for record.
//
// Index chain, per frame:
//   k_ti_lists    every live map point (pool row) against the frame's keypoints: Hamming distances, and per row its
//                 packed ENTRIES - the columns with distance < 30 in column order (the only ones the row can ever
//                 claim: pass 2 accepts best < 30, pass 1 best < 15), each with up to three columns before it whose
//                 distance can hold its runner-up at or below 2 x best (the ratio test `second / best > 2`).  On
//                 synthetic KITTI-like frames a row has 0.6 entries on average.
//   k_ti_resolve  one workgroup: frame begin, both order-dependent greedy passes, createmappoint, cull, compaction.
//                 The greedy passes (rows in order, each claims its best unclaimed column) are resolved in ROUNDS,
//                 see ti_resolve_pass; on real frames a pass takes 2-5 rounds instead of hundreds of serial steps.
// The map-point pool is a structure of arrays kept in creation order (the deterministic stand-in for the reference's
// std::set<mappoint*> address order); after each frame it is stably compacted into the other half of a ping-pong
// buffer, so "local map point r" is simply row r.
// The reference also stores match_score[i] = second/best for every pass-1 row (src/pnpmatch.cc:99); nothing ever
// reads it (its only use is commented out at src/Optimizer.cc:54), so the tracker does not materialise it -
// svo_match_greedy does.
// Offline detection boxes (main.cpp:82-95) gate the path exactly where the reference uses them: the +-5 px creation
// gates (src/Tracking.cc:61-66 with its never-reset flag, src/frame.cc:198-203) and the +-10 px epipolar veto of
// pass 1 (src/pnpmatch.cc:101-144) with F from the 8-point algorithm over brute-force matches
// (src/pnpmatch.cc:302-337): two more launches in front of the index chain of a frame that carries boxes - k_tg_bf
// (nearest last-frame descriptor of every keypoint) and k_tg_fmat (one wave: match filter, Hartley normalisation, normal
// matrix, wave-parallel Jacobi, rank-2 projection; svo_fmat_dev.h).  Boxes arrive as HBM arrays in every mode
// (svo_boxes_dev), nothing of a gated frame touches the host.
#include <chrono>
#include <cstddef>
#include <initializer_list>
#include <string>
#include <vector>
#include <algorithm>
#include <cstring>
#include <mutex>

#include "svo_internal.h"
#include "svo_wave.h"
#include "svo_gate.h"
#include "svo_pose_dev.h"
#include "svo_fmat_dev.h"

#define TRK_MAXKP 512
#define TRK_CAP 4096
#define TRK_ROWS_MAX 3072        // live rows k_ti_lists can meet: 4 frames of local points + the last frame's, + slack
#define TRK_LCAP 8               // packed entries (claimable columns) a pool row keeps; more -> the row is "dense"
#define TRK_GPOS (1 << 20)       // map-point position table: ring over map-point ids
#define TRK_DNC 24                // dense rows whose distance row is cached in LDS during a pass
#define TRK_DENSE 0xFF           // ncand marker: more than `lcap` candidates, the row's distances are in D

struct TrackPool {
  alignas(16) uint32_t desc[TRK_CAP * 8];
  int32_t create_id[TRK_CAP];
  int32_t gid[TRK_CAP];          // map-point id = creation sequence number; positions live in gpos[gid % TRK_GPOS]
  uint8_t bad[TRK_CAP];
  uint8_t in_local[TRK_CAP];
};

// Samples of the fused pose launch whose completion the frame's workgroup waits for one by one: cv::solvePnPRansac's adaptive bound
// ends within the first 8 samples on 96 % of the frames (a median of 4), and then the other 92 need not be waited for.
#define TP_EARLY 8
#define TP_FLAGS 16   // samples that announce themselves one by one (work->hyp_early)
// What the index chain hands to the pose chain for one frame.
// Written by k_ti_resolve with agent-scope (sc1) stores and published through `ready`, read by the pose kernels with agent-scope
// loads after they have seen the tag: the pose chain does not wait on stream events for the index chain (see tp_wait_work).
struct TrackWork {
  int32_t ready;                 // = frame index + 1 once the record of that frame is complete (cleared by svo_track_reset)
  int32_t frame_id, nkp, n_stereo, n_pass1, n_pass2, n_new, n_local, skip_match;
  long long ts[8];               // diagnostics: s_memtime at begin / pass 1 / pass 2 / frame end / done, dense rows, late rows
  long long rt[6];               // diagnostics: s_memrealtime (100 MHz, one clock for the chip) at k_ti_resolve start / end,
                                 // k_tp_hyp start (its workgroup 0), k_tp_frame end, k_tp_frame start, (unused)
  int32_t diag[2];               // [0] rows of pass 1 | rounds << 16, [1] rows of pass 2 | rounds << 16
  int32_t n_edges;               // 3D-2D correspondences of the frame (src/pnpmatch.cc:216-224), in keypoint order:
  int32_t edge_gid[TRK_MAXKP];   //   id of the map point (CurrentFrame->MapPoints[j]->...) ...
  int32_t edge_kp[TRK_MAXKP];    //   ... and the keypoint j it is matched to
  int32_t new_gid[TRK_MAXKP];    // per keypoint: id of the map point created from it at the frame's end, or -1
  // written by the pose chain (k_tp_frame), for svo_debug_track_frames: cv::solvePnPRansac's outcome and pose
  int32_t pnp_best, pnp_iterations, pnp_inliers, pnp_ok;
  double T_pnp[16];
  int32_t hyp_done, pad_hyp;     // fused pose launch (k_tp_tail_ord): RANSAC samples of this frame that have stored their result
  int32_t hyp_early[TP_FLAGS];   // ... and, for the first TP_FLAGS samples, frame id + 1 once that sample's result is stored (ids only grow
                                 // between two resets and a reset clears the records: a stale value never equals the current one)
};

struct TrackState {
  // ---- index chain -------------------------------------------------------------------------
  int32_t frame_num, npool, lastN, cur;      // cur: active half of the pool ping-pong
  int32_t next_gid, overflow, n_vetoed, n_boxes;
  int32_t epnp_fallbacks, pad_counters[3];   // RANSAC samples of the default solver that took its sequential fallback (sticky count)
  int32_t last_mp[TRK_MAXKP];
  int32_t dbg_cur_mp[TRK_MAXKP];
  svo_camera cam;
  int32_t boxes[SVO_MAX_BOXES * 4];    // {left, right, top, bottom} of the current frame
  double F[9];                         // fundamental matrix cur <- last (row-major)
  float last_xy[TRK_MAXKP * 2];        // LastFrame.keypoints_l[i].pt
  alignas(16) uint32_t last_desc[TRK_MAXKP * 8];   // LastFrame.f_descriptor
  int32_t bf_idx[TRK_MAXKP], bf_dist[TRK_MAXKP], bf_min;
  uint8_t bf_keep[TRK_MAXKP];
  // ---- pose chain --------------------------------------------------------------------------
  float lastTcw[16];
  double Xw[TRK_MAXKP * 3], obs[TRK_MAXKP * 2], K[4], Tprior[16], T[16];
  svo_lm_stats lm;
  svo_pnp_stats pnp;
  PnpHyp hyp[PNP_HYP];                 // the frame's RANSAC samples (k_tp_hyp)
  long long pose_ts[16];               // diagnostics: s_memtime stamps of k_tp_hyp (0..7) and k_tp_frame (8..15)
  // ---- large arrays (not cleared by a reset) -----------------------------------------------
  TrackPool pool[2];
  uint16_t rowmin[TRK_CAP];                // min over ALL current keypoints of the row's distances
  uint8_t ncand[TRK_CAP];                  // packed entries of the row, or TRK_DENSE
  uint32_t cand[(size_t)TRK_CAP * 2 * TRK_LCAP];   // packed entries, two words each (k_ti_lists), columns ascending
  uint16_t D[(size_t)TRK_CAP * 512];       // full distance rows (written for every row that has an entry)
  float gpos[(size_t)TRK_GPOS * 3];        // pose chain: world position of map point gid
};

// ---- block-wide exclusive scan (blockDim.x = 256 or 1024) -------------------------------------
// wave64 inclusive scan on DPP (gfx9 row shifts + row broadcasts): six dependent VALU steps, no LDS crossbar
__device__ __forceinline__ int tk_wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8  -> scan inside each row of 16
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ int block_excl_scan(int v, int* sm /*[16], 16-byte aligned*/, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int incl = tk_wave_incl_scan(v);
  const int nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 63) sm[wv] = incl;
  __syncthreads();
  const int4 a = reinterpret_cast<const int4*>(sm)[0], b = reinterpret_cast<const int4*>(sm)[1],
             c = reinterpret_cast<const int4*>(sm)[2], d = reinterpret_cast<const int4*>(sm)[3];
  const int t[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    off += w < wv ? t[w] : 0;
    tot += w < nw ? t[w] : 0;
  }
  *total = tot;
  return off + incl - v;
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

// ---- hand-over index chain -> pose chain without stream events -----------------------------------------------------------
// The pose chain used to wait on one stream event per GROUP of frames (a wait costs ~3 us on the pose stream): whenever the
// index chain's lead shrank below a group - runs of frames whose passes need 30-47 rounds - the pose chain of frames
// g .. g + 3 stood still until frame g + 3 was matched (200-490 us per group, tools/gap_probe.py).  Now k_ti_resolve
// publishes every frame's record itself: all of its fields go out with agent-scope (sc1) stores, every thread waits for its
// own stores (s_waitcnt), and thread 0 then stores the frame's tag; the pose kernels are enqueued without any dependency on
// the index stream, poll the tag with an agent-scope load and read the record with agent-scope loads.  No release / acquire
// fence on either side: on gfx950 a release fence is buffer_wbl2 - a write-back of everything dirty in that XCD's L2, which
// holds the pyramid data of the front end running beside the tail - and sc1 accesses reach the device-coherent level by
// themselves.  The poll is bounded (~1 s): a lost hand-over sets the tracker's error flag instead of hanging the GPU.
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld_agent(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define TP_STORES_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// ================================================================================================
// Index chain 1/2: distances of every live pool row to the frame's keypoints -> sparse candidate lists
// ================================================================================================
// One wave per row.  ROWS map-point rows per workgroup: 4 (one per wave) for a single sequence, where the chain's
// latency counts; 16 when many sequences are advanced together and the staging of the keypoint descriptors is worth
// sharing.  blockIdx.y = sequence (TrackState number, frame slot of the front-end buffers).
template <int ROWS>
__global__ __launch_bounds__(256) void k_ti_lists(TrackState* st, const uint32_t* desc, const int32_t* nkp_p, int kstride,
                                                  int lcap, int nblk) {
  // descriptors of the frame's keypoints, TRANSPOSED: word k of keypoint j at td[k * TRK_MAXKP + j], so that the 64 lanes
  // of a wave (64 consecutive keypoints) read 64 consecutive LDS words
  __shared__ uint32_t td[8 * TRK_MAXKP];
  __shared__ uint16_t acol[4][TRK_LCAP];
  __shared__ uint8_t adist[4][TRK_LCAP];
  st += blockIdx.y; desc += (size_t)blockIdx.y * kstride * 8; nkp_p += blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int npool = st->npool;
  const int row0 = blockIdx.x * ROWS;
  if (row0 >= npool) return;
  const int nkp = min(*nkp_p, TRK_MAXKP);
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
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (int rr = 0; rr < ROWS / 4; ++rr) {
    const int row = row0 + wv * (ROWS / 4) + rr;
    if (row >= npool) break;
    if (P.bad[row]) {
      if (lane == 0) { st->rowmin[row] = 0x7fff; st->ncand[row] = 0; }
      continue;
    }
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = P.desc[8 * row + k];
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
    }
    mn = wave_min_u32_dpp(mn);
    // A entries: the columns this row could ever claim (distance < 30), in column order - piece t holds columns
    // 64 t .. 64 t + 63, lanes ascending
    int nA = 0;
    if (mn < 30) {   // (uniform: the wave's minimum) two rows in three have no column below 30 - nothing to list
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const bool c = dcol[t] < 30;
        const uint64_t m = __ballot(c);
        const int pos = nA + __popcll(m & lt_mask);
        if (c && pos < TRK_LCAP) { acol[wv][pos] = (uint16_t)(lane + 64 * t); adist[wv][pos] = (uint8_t)dcol[t]; }
        nA += __popcll(m);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the wave reads back what its lanes just stored
    __builtin_amdgcn_wave_barrier();
    const bool dense = nA > lcap;
    uint32_t* lst = st->cand + (size_t)row * (2 * TRK_LCAP);
    if (!dense) {
      // per entry with distance >= 15: up to three columns BEFORE it whose distance lies in [30, 2 d] - the columns
      // outside the A list that can hold its runner-up at or below 2 d (ratio test); `more` says there are others
      for (int i = 0; i < nA; ++i) {
        const int c = acol[wv][i], d = adist[wv][i];
        uint32_t w0 = (uint32_t)c | ((uint32_t)d << 9), w1 = 0;
        if (d >= 15) {
          int nb = 0, total = 0;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            uint64_t m = __ballot(lane + 64 * t < c && dcol[t] >= 30 && dcol[t] <= 2 * d);
            total += __popcll(m);
            while (m && nb < nblk) {
              const int bcol = 64 * t + __ffsll((long long)m) - 1;
              w1 |= (uint32_t)bcol << (9 * nb);
              ++nb;
              m &= m - 1;
            }
          }
          w0 |= ((uint32_t)nb << 14) | (total > nb ? 1u << 16 : 0u);
        }
        if (lane == 0) { lst[2 * i] = w0; lst[2 * i + 1] = w1; }
      }
    }
    if (nA > 0) {   // the whole row, for the (rare) evaluations the packed entries cannot answer
#pragma unroll
      for (int t = 0; t < 8; ++t) st->D[(size_t)row * 512 + lane + 64 * t] = (uint16_t)dcol[t];
    }
    if (lane == 0) {
      st->rowmin[row] = (uint16_t)mn;
      st->ncand[row] = (uint8_t)(dense ? TRK_DENSE : nA);
    }
  }
}

// ================================================================================================
// Index chain 2/2: everything order-dependent of one frame, one 1024-thread workgroup per sequence
// ================================================================================================
struct TiLds {
  alignas(16) int sm[32];
  int32_t cur_mp[TRK_MAXKP];       // CurrentFrame->MapPoints as pool rows
  uint32_t minrow[2][TRK_MAXKP];   // per column: lowest unresolved row that could still claim it (rounds alternate between
                                   // the two copies: the idle one is wiped while the other is read)
  uint16_t claimer[TRK_MAXKP];     // who took the column: 0 = taken before the pass (or beyond nkp), k + 1 = active row k of
                                   // this pass, 0xffff = free.  Row k sees a column as free iff claimer > k + 1: a LATER row's
                                   // claim must stay invisible to the earlier rows that are still unresolved
  uint16_t act_m[TRK_CAP];         // active row k -> pool row
  uint16_t act_i[TRK_MAXKP];       // pass 1: active row k -> last-frame keypoint index
  uint8_t act_n[TRK_CAP];          // its list length / TRK_DENSE
  uint8_t fin[TRK_CAP];            // row resolved
  uint8_t observed[TRK_CAP];       // pool row matched in pass 1 (observations.count(CurrentFrame))
  uint8_t ref[TRK_CAP];            // pool row referenced by the frame's keypoints (kept alive for the next pass 1)
  int16_t remap[TRK_CAP];          // pool row -> row after compaction
  int cnt_acc, cnt_veto, cnt_late, nd, flag[2];
  int32_t boxes[SVO_MAX_BOXES * 4];   // the frame's detection boxes {left, right, top, bottom}
  double F[9];                        // the frame's fundamental matrix (epipolar veto of pass 1)
  uint16_t dn[TRK_CAP];            // dense active rows of the running pass
  uint16_t dnD[TRK_DNC][512];      // distance rows of the first TRK_DNC of them (fetched once per pass)
};

// ---- one greedy pass, resolved in rounds -----------------------------------------------------------
// A row's packed entries (k_ti_lists): the columns it could claim (distance < 30, column order), each with up to three
// columns outside that list that can hold its runner-up down.  The reference's scan
//   `if (dist < best) { second = best; best = dist; idx = j; }`  over the unclaimed columns in index order
// (src/pnpmatch.cc:75-94) ends with best = the first minimum and second = the minimum over the unclaimed columns
// BEFORE it; the accept rule `best < max_dist [&& (float)second / (float)best > 2]` therefore reads: the best entry
// has no unclaimed column before it with distance <= 2 * best ("blocker").  Blockers with distance < 30 are entries
// of the same list; the others are what k_ti_lists stored next to the entry (or, rarely, found in the full row D).
//
// Round: (1) every unresolved row publishes the columns it could still claim (minrow[column] = lowest such row);
// (2) a row is FINAL when no earlier unresolved row can change its outcome: its best column cannot be claimed by an
// earlier row, and - if the ratio test rejects it - one of its blockers cannot be claimed by an earlier row either (the
// row then stays rejected whatever else happens).  Accepted rows claim their column; the claim carries the row's rank,
// and a row only sees the claims of EARLIER rows (ti_avail): a later row that is resolved first must not change what
// an earlier, still unresolved row sees.  The first unresolved row is always final, so every round makes progress.
//
// Thread k owns active row k (its entries live in registers for the whole pass).  Rows with more claimable columns than
// entries, and active rows beyond 1024, are "dense": a whole wave evaluates such a row from its distance row in D.
__device__ __forceinline__ bool ti_avail(const TiLds& S, int j, int k) { return (int)S.claimer[j] > k + 1; }

// resolve row k (pool row m) given its evaluation; returns true when the row is final.  tally: three 10-bit counters of
// the calling lane - accepted, vetoed, resolved after the first round (summed per wave at the end of the pass; a lane
// resolves at most 1 + TRK_CAP / 16 rows)
template <int PASS>
__device__ __forceinline__ bool ti_finalize(TiLds& S, TrackState* st, TrackPool& P, int k, int m, int bj, uint32_t minrow_bj,
                                            bool ok, bool perm, int rounds, const svo_kp* kp, int n_boxes, int& tally) {
  if (minrow_bj < (uint32_t)k) return false;   // an earlier unresolved row may still take the best column
  if (!ok && !perm) return false;                 // rejected, but every blocker may still be claimed away
  if (rounds > 0) tally += 1 << 20;
  if (!ok) return true;
  if (PASS == 1 && n_boxes > 0) {
    // epipolar veto (src/pnpmatch.cc:103-144): the match lands in a padded box and is off the epipolar
    // line -> the map point is marked bad and claims nothing
    const svo_kp kc = kp[bj];
    const int i_last = S.act_i[k];
    if (svo_in_boxes(kc.x, kc.y, S.boxes, n_boxes, 10) &&
        svo_epipolar_distance(S.F, st->last_xy[2 * i_last], st->last_xy[2 * i_last + 1], kc.x, kc.y) > 0.1) {
      P.bad[m] = 1;
      tally += 1 << 10;
      return true;
    }
  }
  S.cur_mp[bj] = m;     // rows finalised in one round never share a best column
  S.claimer[bj] = (uint16_t)(k + 1);   // visible at once: earlier rows ignore it by rank, later rows may use it (it is final)
  S.observed[m] = 1;
  tally += 1;
  return true;
}

// ---- a dense row (active row kk, pool row m), evaluated by one wave: lane l holds the distances of columns 8 l .. 8 l + 7
// phase 1: which of my columns are still free for this row (returned as a mask: claims only change between rounds, so
// phase 2 reuses it), and the claimable ones published in minrow
__device__ __forceinline__ uint32_t ti_dense_publish(TiLds& S, uint32_t* minrow, const uint32_t (&dd)[8], int kk, int lane, int nkp, int max_dist) {
  uint32_t cl[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) cl[c] = S.claimer[lane * 8 + c];     // all eight reads in flight together
  uint32_t am = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (lane * 8 + c < nkp && (int)cl[c] > kk + 1) am |= 1u << c;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (((am >> c) & 1u) && (int)dd[c] < max_dist) atomicMin(&minrow[lane * 8 + c], (uint32_t)kk);
  return am;
}
// phase 2: best free column, ratio test against the free columns before it, finality.  Returns (wave-uniform) whether
// the row is resolved.
template <int PASS>
__device__ __forceinline__ bool ti_dense_decide(TiLds& S, const uint32_t* minrow, TrackState* st, TrackPool& P, const uint32_t (&dd)[8], uint32_t am,
                                                int kk, int m, int lane, int rounds, const svo_kp* kp, int n_boxes, int& tally) {
  constexpr int max_dist = PASS == 1 ? 15 : 30;
  uint32_t mr[8];
  if (PASS == 2) {
#pragma unroll
    for (int c = 0; c < 8; ++c) mr[c] = minrow[lane * 8 + c];    // requested before the reduction below needs the wave
  }
  uint32_t key = 0xffffffffu;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (((am >> c) & 1u) && (int)dd[c] < max_dist) key = min(key, (dd[c] << 16) | (uint32_t)(lane * 8 + c));
  key = wave_min_u32_dpp(key);
  if (key == 0xffffffffu) return true;      // claims only remove columns: never accepted
  const int bj = (int)(key & 0xffffu), bd = (int)(key >> 16);
  const uint32_t mrbj = minrow[bj];
  bool blk = false, pb = false;
  if (PASS == 2) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (((am >> c) & 1u) && lane * 8 + c < bj && (int)dd[c] <= 2 * bd) { blk = true; pb = pb || mr[c] >= (uint32_t)kk; }
  }
  const bool ok = __ballot(blk) == 0, perm = __ballot(pb) != 0;
  bool fin = false;
  if (lane == 0) fin = ti_finalize<PASS>(S, st, P, kk, m, bj, mrbj, ok, perm, rounds, kp, n_boxes, tally);
  return __ballot(fin) != 0;
}
__device__ __forceinline__ void ti_unpack8(const uint4& v, uint32_t (&dd)[8]) {
  dd[0] = v.x & 0xffff; dd[1] = v.x >> 16; dd[2] = v.y & 0xffff; dd[3] = v.y >> 16;
  dd[4] = v.z & 0xffff; dd[5] = v.z >> 16; dd[6] = v.w & 0xffff; dd[7] = v.w >> 16;
}

// lane's eight columns (8 lane .. 8 lane + 7) of dense-list row q (pool row m): from the LDS cache once filled
__device__ __forceinline__ uint4 ti_dense_piece(TiLds& S, const TrackState* st, int q, int m, int lane, int ncached) {
  if (q < ncached) return *reinterpret_cast<const uint4*>(&S.dnD[q][lane * 8]);
  const uint4 v = *reinterpret_cast<const uint4*>(st->D + (size_t)m * 512 + lane * 8);
  if (q < TRK_DNC) *reinterpret_cast<uint4*>(&S.dnD[q][lane * 8]) = v;   // only this wave touches row q until the next barrier
  return v;
}

// Dense rows are dealt round-robin to the 16 waves (list position q -> wave q % 16).  A wave keeps its FIRST dense row
// (q = wave id: all of them on ordinary frames) in registers for the whole pass - row ids, the 16 bytes of distances per
// lane, the resolved flag - so that a round costs it one LDS round trip per phase; the rows after that go through the
// LDS cache dnD (list positions 16 .. 16 + TRK_DNC - 1) or, beyond it, the distance rows in HBM.
template <int PASS>
__device__ __forceinline__ int ti_resolve_pass(TiLds& S, TrackState* st, TrackPool& P, int n_act, int nkp,
                                               const svo_kp* kp, int n_boxes, int* n_acc, int* n_veto, int* n_late) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int max_dist = PASS == 1 ? 15 : 30;
  // ---- set-up: my row's entries into registers; dense rows into a list ------------------------
  const int k = tid;
  int nc = 0, my_m = 0;
  uint32_t w0[TRK_LCAP], w1[TRK_LCAP];
#pragma unroll
  for (int i = 0; i < TRK_LCAP; ++i) { w0[i] = 0; w1[i] = 0; }
  bool unresolved = false;
  if (tid == 0) { S.nd = 0; S.flag[0] = 0; S.flag[1] = 0; S.cnt_acc = 0; S.cnt_veto = 0; S.cnt_late = 0; }
  if (tid < TRK_MAXKP) { S.minrow[0][tid] = 0xffffffffu; S.minrow[1][tid] = 0xffffffffu; }
  __syncthreads();
  for (int kk = tid; kk < n_act; kk += 1024) {
    const int n = S.act_n[kk];
    if (n == TRK_DENSE || kk >= 1024) {
      S.dn[atomicAdd(&S.nd, 1)] = (uint16_t)kk;
      S.fin[kk] = 0;
    } else if (kk == k) {
      nc = n;
      my_m = S.act_m[kk];
      const uint4* src = reinterpret_cast<const uint4*>(st->cand + (size_t)my_m * 2 * TRK_LCAP);
      uint4 v[TRK_LCAP / 2];
#pragma unroll
      for (int q = 0; q < TRK_LCAP / 2; ++q) v[q] = src[q];
#pragma unroll
      for (int q = 0; q < TRK_LCAP / 2; ++q) { w0[2 * q] = v[q].x; w1[2 * q] = v[q].y; w0[2 * q + 1] = v[q].z; w1[2 * q + 1] = v[q].w; }
      unresolved = true;
    }
  }
  __syncthreads();
  int rounds = 0, tally = 0;
  int ncached = 0;                 // list positions 16 .. 16 + ncached - 1 are in dnD
  int r_kk = -1, r_m = 0;          // the wave's register-resident dense row (list position wv): active row, pool row
  bool r_fin = false;
  uint32_t r_dd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r_am = 0;
  // A round = phase 1, barrier, phase 2, barrier.  Claims are written straight into `claimer` in phase 2 (see
  // ti_finalize); minrow and the continue-flag exist twice, the copy of the NEXT round being wiped during phase 2 (last
  // read one barrier ago, next written one barrier ahead).
  for (;;) {
    uint32_t* minrow = S.minrow[rounds & 1];
    const int nd = S.nd;   // rows may join the dense list during a round
    const int nd1 = min(max(nd - 16, 0), TRK_DNC);   // cached once phase 1 of this round has run
    // ---- phase 1: every unresolved row publishes the columns it could still claim -------------
    uint32_t avail = 0;   // bit i: entry i not claimed by an earlier row (and below max_dist)
    if (unresolved) {
      uint32_t cl[TRK_LCAP];
#pragma unroll
      for (int i = 0; i < TRK_LCAP; ++i) cl[i] = S.claimer[w0[i] & 511u];   // all reads in flight together
#pragma unroll
      for (int i = 0; i < TRK_LCAP; ++i) {
        const int d = (int)((w0[i] >> 9) & 31u);
        if (i < nc && d < max_dist && (int)cl[i] > k + 1) avail |= 1u << i;
      }
#pragma unroll
      for (int i = 0; i < TRK_LCAP; ++i)
        if ((avail >> i) & 1u) atomicMin(&minrow[w0[i] & 511u], (uint32_t)k);
      S.flag[rounds & 1] = 1;
    }
    if (wv < nd) {
      if (r_kk < 0) {   // first sight of the wave's own dense row: fetch it once
        r_kk = S.dn[wv];
        r_m = S.act_m[r_kk];
        const uint4 v = *reinterpret_cast<const uint4*>(st->D + (size_t)r_m * 512 + lane * 8);
        ti_unpack8(v, r_dd);
      }
      if (!r_fin) {
        r_am = ti_dense_publish(S, minrow, r_dd, r_kk, lane, nkp, max_dist);
        if (lane == 0) S.flag[rounds & 1] = 1;
      }
    }
    for (int q = wv + 16; q < nd; q += 16) {
      const int kk = S.dn[q];
      if (S.fin[kk]) continue;
      uint32_t dd[8];
      ti_unpack8(ti_dense_piece(S, st, q - 16, S.act_m[kk], lane, ncached), dd);
      ti_dense_publish(S, minrow, dd, kk, lane, nkp, max_dist);
      if (lane == 0) S.flag[rounds & 1] = 1;
    }
    __syncthreads();
    if (!S.flag[rounds & 1]) break;
    if (tid < TRK_MAXKP) S.minrow[(rounds & 1) ^ 1][tid] = 0xffffffffu;
    if (tid == 0) S.flag[(rounds & 1) ^ 1] = 0;
    // ---- phase 2: rows whose outcome no earlier unresolved row can change are final -------------
    if (unresolved) {
      // the best free entry (first minimum in column order); everything the decision may need from LDS is requested
      // together: minrow of the entries, claimer and minrow of the best entry's stored blockers
      uint32_t bd = 255u, w0b = 0, w1b = 0;
#pragma unroll
      for (int i = 0; i < TRK_LCAP; ++i) {
        const uint32_t d = ((avail >> i) & 1u) ? ((w0[i] >> 9) & 31u) : 255u;
        const bool better = d < bd;
        bd = better ? d : bd; w0b = better ? w0[i] : w0b; w1b = better ? w1[i] : w1b;
      }
      const uint32_t bj = w0b & 511u;
      const uint32_t mrbj = minrow[bj];
      uint32_t mr[TRK_LCAP], bc[3], bm[3];
      if (PASS == 2) {
#pragma unroll
        for (int i = 0; i < TRK_LCAP; ++i) mr[i] = minrow[w0[i] & 511u];
#pragma unroll
        for (int q = 0; q < 3; ++q) { const uint32_t j = (w1b >> (9 * q)) & 511u; bc[q] = S.claimer[j]; bm[q] = minrow[j]; }
      }
      // integer 0 / 1 logic on purpose: no divergent branches in here
      uint32_t blocked = 0, perm = 0, all_taken = 1;
      if (PASS == 2) {
#pragma unroll
        for (int i = 0; i < TRK_LCAP; ++i) {   // blockers inside the list (entries are in column order)
          const uint32_t j = w0[i] & 511u, d = (w0[i] >> 9) & 31u;
          const uint32_t c = ((avail >> i) & 1u) & (uint32_t)(j < bj) & (uint32_t)(d <= 2 * bd);
          blocked |= c; perm |= c & (uint32_t)(mr[i] >= (uint32_t)k);
        }
        const uint32_t nb = (w0b >> 14) & 3u;
#pragma unroll
        for (int q = 0; q < 3; ++q) {   // stored blockers outside the list (distance in [30, 2 * best])
          const uint32_t c = (uint32_t)((uint32_t)q < nb) & (uint32_t)((int)bc[q] > k + 1);
          blocked |= c; perm |= c & (uint32_t)(bm[q] >= (uint32_t)k);
        }
        all_taken = blocked ^ 1u;   // (no list blocker either: only consulted when the row would be accepted)
      }
      const bool none = bd == 255u;                       // claims only remove columns: never accepted
      const bool to_dense = PASS == 2 && !none && !blocked && all_taken && ((w0b >> 16) & 1u);
      if (to_dense) {
        // every stored blocker has been claimed and there were more (rare): from now on a wave evaluates this row
        // from its full distance row
        S.fin[k] = 0;
        S.dn[atomicAdd(&S.nd, 1)] = (uint16_t)k;
      }
      if (none || to_dense) unresolved = false;
      else if (ti_finalize<PASS>(S, st, P, k, my_m, (int)bj, mrbj, !blocked, perm != 0, rounds, kp, n_boxes, tally)) unresolved = false;
    }
    if (wv < nd && r_kk >= 0 && !r_fin)
      r_fin = ti_dense_decide<PASS>(S, minrow, st, P, r_dd, r_am, r_kk, r_m, lane, rounds, kp, n_boxes, tally);
    for (int q = wv + 16; q < nd; q += 16) {
      const int kk = S.dn[q];
      if (S.fin[kk]) continue;
      const int m = S.act_m[kk];
      uint32_t dd[8], cl[8];
      ti_unpack8(ti_dense_piece(S, st, q - 16, m, lane, nd1), dd);
#pragma unroll
      for (int c = 0; c < 8; ++c) cl[c] = S.claimer[lane * 8 + c];
      uint32_t am = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (lane * 8 + c < nkp && (int)cl[c] > kk + 1) am |= 1u << c;
      if (ti_dense_decide<PASS>(S, minrow, st, P, dd, am, kk, m, lane, rounds, kp, n_boxes, tally) && lane == 0) S.fin[kk] = 1;
    }
    __syncthreads();
    ncached = nd1;
    ++rounds;
  }
  // what this pass claimed is simply taken for the next pass
  if (tid < TRK_MAXKP && S.claimer[tid] != 0xffffu) S.claimer[tid] = 0;
  // tallies: one LDS atomic per wave and counter
  {
    const int a = wave_sum_i32_dpp(tally & 1023), v = wave_sum_i32_dpp((tally >> 10) & 1023), l = wave_sum_i32_dpp(tally >> 20);
    if (lane == 0) {
      if (a) atomicAdd(&S.cnt_acc, a);
      if (v) atomicAdd(&S.cnt_veto, v);
      if (l) atomicAdd(&S.cnt_late, l);
    }
  }
  __syncthreads();
  *n_acc = S.cnt_acc; *n_veto = S.cnt_veto; *n_late = S.cnt_late;
  __syncthreads();
  return rounds;
}

extern __shared__ __attribute__((aligned(16))) unsigned char tk_smem[];

// boxes / nboxes (nullable): the detection boxes of this frame - of sequence blockIdx.y: `bstride` boxes further on each
__global__ __launch_bounds__(1024) void k_ti_resolve(TrackState* st, TrackWork* work, const svo_kp* kp,
                                                     const uint32_t* desc, const int32_t* nkp_p, const float* depth,
                                                     int kstride, const int32_t* boxes, const int32_t* nboxes, int bstride,
                                                     const double* Fpre) {
  TiLds& S = *reinterpret_cast<TiLds*>(tk_smem);
  st += blockIdx.y; work += blockIdx.y; kp += (size_t)blockIdx.y * kstride;
  desc += (size_t)blockIdx.y * kstride * 8; nkp_p += blockIdx.y; depth += (size_t)blockIdx.y * kstride;
  const int tid = threadIdx.x;
  const int nkp = min(*nkp_p, TRK_MAXKP);
  const int n_boxes = (boxes && nboxes) ? min(max(nboxes[blockIdx.y], 0), SVO_MAX_BOXES) : 0;
  if (tid < 4 * n_boxes) S.boxes[tid] = boxes[(size_t)blockIdx.y * bstride * 4 + tid];   // visible after the first barrier below
  // F of this frame: solved in front of this kernel into st->F (k_tg_fmat), or ahead of the chain for a whole group of frames
  // (k_tg_fmat_group: Fpre) - then st->F is brought up to date here (svo_track_fundamental reads it)
  if (tid >= 64 && tid < 73) {
    const double v = Fpre ? Fpre[tid - 64] : st->F[tid - 64];
    S.F[tid - 64] = v;
    if (Fpre) st->F[tid - 64] = v;
  }
  const int id = st->frame_num;
  const int np_start = st->npool, lastN = st->lastN, gid0 = st->next_gid;
  TrackPool& P = st->pool[st->cur];
  TrackPool& Q = st->pool[st->cur ^ 1];
  long long ts0 = clock64(), ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
  const long long rt0 = wall_clock64();
  // ---- frame begin ---------------------------------------------------------------------------
  if (tid < TRK_MAXKP) { S.cur_mp[tid] = -1; S.claimer[tid] = tid >= nkp ? 0 : 0xffffu; }
  for (int r = tid; r < TRK_CAP; r += 1024) { S.observed[r] = 0; S.ref[r] = 0; }
  const bool has_depth = tid < nkp && depth[tid] > 0.f;
  const int n_stereo = __syncthreads_count(has_depth);
  int edge_gid = -1;          // map-point id matched to keypoint `tid`
  int npool = np_start, n_new0 = 0, next_gid = gid0;
  int n_pass1 = 0, n_pass2 = 0, n_act1 = 0, n_act2 = 0, rounds1 = 0, rounds2v = 0, late2 = 0, n_veto = 0;
  if (id == 0) {
    // Tracking::init (src/Tracking.cc:42-97): one map point per keypoint with depth.  Its `dynamic` flag is declared
    // outside the keypoint loop and never reset (:44): once a keypoint falls into a padded box, every later one is
    // skipped as well.
    bool create0 = has_depth;
    if (n_boxes > 0) {
      const svo_kp k = kp[min(tid, max(nkp - 1, 0))];
      const bool inb = tid < nkp && svo_in_boxes(k.x, k.y, S.boxes, n_boxes, 5);
      int tot_in;
      const int before = block_excl_scan(inb ? 1 : 0, S.sm, &tot_in);
      if (before + (inb ? 1 : 0) > 0) create0 = false;
    }
    int total;
    const int rank = block_excl_scan(create0 ? 1 : 0, S.sm, &total);
    if (create0) {
      const int m = npool + rank;
#pragma unroll
      for (int w = 0; w < 8; ++w) P.desc[8 * m + w] = desc[8 * tid + w];
      P.bad[m] = 0; P.in_local[m] = 1; P.create_id[m] = 0; P.gid[m] = next_gid + rank;
      S.cur_mp[tid] = m;
      edge_gid = next_gid + rank;    // the pose chain places these with the identity pose before frame 0's LM
    }
    npool += total; next_gid += total; n_new0 = total;
    __syncthreads();
  } else {
    // ---- pass 1 (src/pnpmatch.cc:61-156): last frame's map points, in keypoint order --------
    ts1 = clock64();
    {
      int m = -1;
      bool act = false;
      if (tid < lastN) {
        m = st->last_mp[tid];
        act = m >= 0 && !P.bad[m] && st->rowmin[m] < 15;
      }
      const int k = block_excl_scan(act ? 1 : 0, S.sm, &n_act1);
      if (act) { S.act_m[k] = (uint16_t)m; S.act_i[k] = (uint16_t)tid; S.act_n[k] = st->ncand[m]; }
      __syncthreads();
      int late1;
      rounds1 = ti_resolve_pass<1>(S, st, P, n_act1, nkp, kp, n_boxes, &n_pass1, &n_veto, &late1);
    }
    // ---- pass 2 (src/pnpmatch.cc:159-199): local map points not observed by this frame -------
    ts2 = clock64();
    {
      // rows 4 tid .. 4 tid + 3: flags as 4-byte / 8-byte vector loads, all in flight together
      const uint32_t loc4 = *reinterpret_cast<const uint32_t*>(&P.in_local[4 * tid]);
      const uint32_t bad4 = *reinterpret_cast<const uint32_t*>(&P.bad[4 * tid]);
      const uint32_t obs4 = *reinterpret_cast<const uint32_t*>(&S.observed[4 * tid]);
      const uint2 rm4 = *reinterpret_cast<const uint2*>(&st->rowmin[4 * tid]);
      const uint32_t nc4 = *reinterpret_cast<const uint32_t*>(&st->ncand[4 * tid]);
      const uint32_t rmv[4] = {rm4.x & 0xffffu, rm4.x >> 16, rm4.y & 0xffffu, rm4.y >> 16};
      int cnt = 0;
      uint32_t bits = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * tid + q;
        if (r < npool && ((loc4 >> (8 * q)) & 0xffu) && !((bad4 >> (8 * q)) & 0xffu) && !((obs4 >> (8 * q)) & 0xffu) && rmv[q] < 30u) {
          bits |= 1u << q; ++cnt;
        }
      }
      int k = block_excl_scan(cnt, S.sm, &n_act2);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((bits >> q) & 1u) {
          S.act_m[k] = (uint16_t)(4 * tid + q); S.act_n[k] = (uint8_t)((nc4 >> (8 * q)) & 0xffu);
          ++k;
        }
      __syncthreads();
      int veto2;
      rounds2v = ti_resolve_pass<2>(S, st, P, n_act2, nkp, kp, 0, &n_pass2, &veto2, &late2);
    }
    if (tid < nkp && S.cur_mp[tid] >= 0) edge_gid = P.gid[S.cur_mp[tid]];
  }
  if (tid < TRK_MAXKP) st->dbg_cur_mp[tid] = tid < nkp ? S.cur_mp[tid] : -1;
  int n_edges;
  {
    const int e = block_excl_scan(edge_gid >= 0 ? 1 : 0, S.sm, &n_edges);
    if (edge_gid >= 0) { st_agent(&work->edge_gid[e], edge_gid); st_agent(&work->edge_kp[e], tid); }
  }
  ts3 = clock64();
  // ---- frame end: createmappoint (src/frame.cc:182-238) for keypoints without a map point ------
  int m_cur = tid < nkp ? S.cur_mp[tid] : -1;
  bool create = tid < nkp && m_cur < 0 && has_depth;
  uint4 dk0 = {0, 0, 0, 0}, dk1 = dk0;   // this keypoint's descriptor: two 16-byte loads, used twice below
  if (tid < nkp) {
    const svo_kp k = kp[tid];
    dk0 = reinterpret_cast<const uint4*>(desc + 8 * tid)[0]; dk1 = reinterpret_cast<const uint4*>(desc + 8 * tid)[1];
    if (create && n_boxes > 0 && svo_in_boxes(k.x, k.y, S.boxes, n_boxes, 5)) create = false;
    st->last_xy[2 * tid] = k.x; st->last_xy[2 * tid + 1] = k.y;
    reinterpret_cast<uint4*>(st->last_desc + 8 * tid)[0] = dk0; reinterpret_cast<uint4*>(st->last_desc + 8 * tid)[1] = dk1;
  }
  int n_new;
  const int rank = block_excl_scan(create ? 1 : 0, S.sm, &n_new);
  int new_gid = -1;
  if (create && npool + rank < TRK_CAP) {
    const int m = npool + rank;
    reinterpret_cast<uint4*>(&P.desc[8 * m])[0] = dk0; reinterpret_cast<uint4*>(&P.desc[8 * m])[1] = dk1;
    P.bad[m] = 0; P.in_local[m] = 1; P.create_id[m] = id; P.gid[m] = next_gid + rank;
    new_gid = next_gid + rank;
    m_cur = m;
  }
  if (tid < TRK_MAXKP) st_agent(&work->new_gid[tid], new_gid);
  const int np1 = min(npool + n_new, TRK_CAP);
  const bool overflow = npool + n_new > TRK_CAP;
  next_gid += n_new;
  if (tid < nkp && m_cur >= 0) S.ref[m_cur] = 1;
  __syncthreads();
  // ---- cull (src/Tracking.cc:239-250) + stable compaction into the other pool half -------------
  // live = still in the local map, or referenced by this frame's keypoints (next frame's pass 1 needs it)
  int live_cnt = 0, oldest = 0x7fffffff, nl = 0;
  uint32_t live_bits = 0, loc_bits = 0;
  // rows 4 tid .. 4 tid + 3 of the pool, every field as one vector load, the descriptors (8 x 16 bytes) with them
  // (only by the threads whose rows exist: the pool holds ~1,300 of its 4,096 rows on these frames, and one CU pulls
  // 172 bytes per thread at ~11 bytes per clock)
  const bool mine = 4 * tid < np1;
  uint32_t loc4 = 0, bad4 = 0;
  int4 cid4 = {0, 0, 0, 0}, gid4 = {0, 0, 0, 0};
  uint4 dsc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) dsc[q] = uint4{0, 0, 0, 0};
  if (mine) {
    loc4 = *reinterpret_cast<const uint32_t*>(&P.in_local[4 * tid]);
    bad4 = *reinterpret_cast<const uint32_t*>(&P.bad[4 * tid]);
    cid4 = *reinterpret_cast<const int4*>(&P.create_id[4 * tid]);
    gid4 = *reinterpret_cast<const int4*>(&P.gid[4 * tid]);
#pragma unroll
    for (int q = 0; q < 8; ++q) dsc[q] = reinterpret_cast<const uint4*>(&P.desc[32 * tid])[q];
  }
  const uint32_t ref4 = *reinterpret_cast<const uint32_t*>(&S.ref[4 * tid]);
  const int cidv[4] = {cid4.x, cid4.y, cid4.z, cid4.w}, gidv[4] = {gid4.x, gid4.y, gid4.z, gid4.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = tid * 4 + q;
    if (r < np1) {
      bool loc = ((loc4 >> (8 * q)) & 0xffu) != 0;
      if (id >= 4 && cidv[q] <= id - 4) loc = false;   // cull
      if (loc) { loc_bits |= 1u << q; }
      if (loc || ((ref4 >> (8 * q)) & 0xffu)) { live_bits |= 1u << q; ++live_cnt; oldest = min(oldest, gidv[q]); nl += loc ? 1 : 0; }
    }
  }
  int total_live;
  int base = block_excl_scan(live_cnt, S.sm, &total_live);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = tid * 4 + q;
    if ((live_bits >> q) & 1u) {
      const int d = base++;
      uint4* dst = reinterpret_cast<uint4*>(&Q.desc[8 * d]);
      dst[0] = dsc[2 * q]; dst[1] = dsc[2 * q + 1];
      Q.create_id[d] = cidv[q];
      Q.gid[d] = gidv[q];
      Q.bad[d] = (uint8_t)((bad4 >> (8 * q)) & 0xffu);
      Q.in_local[d] = (uint8_t)((loc_bits >> q) & 1u);
      S.remap[r] = (int16_t)d;
    } else if (r < np1) {
      S.remap[r] = -1;
    }
  }
  int nl_total;
  block_excl_scan(nl, S.sm, &nl_total);
  // a live point whose id is about to be lapped by the position ring would alias a newer one: flag it (sticky)
  const int lapped = __syncthreads_or(oldest != 0x7fffffff && next_gid - oldest > TRK_GPOS - 2 * TRK_MAXKP);
  if (tid < TRK_MAXKP) st->last_mp[tid] = (tid < nkp && m_cur >= 0) ? S.remap[m_cur] : -1;
  if (tid == 0) {
    st_agent(&work->frame_id, id); st_agent(&work->nkp, nkp); st_agent(&work->n_stereo, n_stereo); st_agent(&work->n_edges, n_edges);
    st_agent(&work->n_pass1, n_pass1); st_agent(&work->n_pass2, n_pass2);
    st_agent(&work->n_new, n_new0 + n_new); st_agent(&work->n_local, nl_total); st_agent(&work->skip_match, id == 0 ? 1 : 0);
    ts4 = clock64();
    work->ts[0] = ts0; work->ts[1] = ts1; work->ts[2] = ts2; work->ts[3] = ts3; work->ts[4] = ts4;
    work->ts[5] = S.nd; work->ts[6] = late2; work->ts[7] = 0;
    st_agent(&work->rt[0], rt0); st_agent(&work->rt[1], (long long)wall_clock64());
    st_agent(&work->diag[0], n_act1 | (rounds1 << 16));
    st_agent(&work->diag[1], n_act2 | (rounds2v << 16));
    st_agent(&work->hyp_done, 0);
    st->n_vetoed = n_veto;
    st->lastN = nkp;
    st->npool = total_live;
    st->next_gid = next_gid;
    if (overflow || lapped) st->overflow = 1;
    st->cur ^= 1;
    st->frame_num = id + 1;
  }
  // publish the record: every thread's stores have reached the coherent level, then the tag
  TP_STORES_DONE();
  __syncthreads();
  if (tid == 0) st_agent(&work->ready, id + 1);
}

// ================================================================================================
// Semantic gating, in front of the index chain of a frame that carries detection boxes
// ================================================================================================
// pnpmatch::find_feature_matches (src/pnpmatch.cc:253-300): cv BruteForce-Hamming match(desc_cur, desc_last) - for every
// keypoint of the current frame the nearest descriptor of the last frame (first minimum).  One wave per current keypoint,
// the last frame's descriptors transposed in LDS as in k_ti_lists.  blockIdx.y = sequence.
__global__ __launch_bounds__(256) void k_tg_bf(TrackState* st, const uint32_t* desc, const int32_t* nkp_p, int kstride,
                                               const int32_t* nboxes) {
  __shared__ uint32_t td[8 * TRK_MAXKP];
  st += blockIdx.y; desc += (size_t)blockIdx.y * kstride * 8; nkp_p += blockIdx.y;
  if (nboxes[blockIdx.y] <= 0 || st->frame_num == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nkp = min(*nkp_p, TRK_MAXKP), lastN = min(st->lastN, TRK_MAXKP);
  const int i = blockIdx.x * 4 + wv;
  if (blockIdx.x * 4 >= nkp) return;
  {
    const uint4* d4 = reinterpret_cast<const uint4*>(st->last_desc);
    uint4 v[TRK_MAXKP * 2 / 256];
#pragma unroll
    for (int k = 0; k < TRK_MAXKP * 2 / 256; ++k) v[k] = d4[min(tid + 256 * k, max(2 * lastN - 1, 0))];
#pragma unroll
    for (int k = 0; k < TRK_MAXKP * 2 / 256; ++k) {
      const int q = tid + 256 * k;
      if (q < 2 * lastN) {
        const int j = q >> 1, w = 4 * (q & 1);
        td[(w + 0) * TRK_MAXKP + j] = v[k].x; td[(w + 1) * TRK_MAXKP + j] = v[k].y;
        td[(w + 2) * TRK_MAXKP + j] = v[k].z; td[(w + 3) * TRK_MAXKP + j] = v[k].w;
      }
    }
  }
  __syncthreads();
  if (i >= nkp) return;
  uint32_t qd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qd[k] = desc[8 * i + k];
  uint32_t key = 0xffffffffu;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int j = lane + 64 * t;
    if (j < lastN) {
      uint32_t d = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ td[k * TRK_MAXKP + j]);
      key = min(key, (d << 16) | (uint32_t)j);   // ties: the lowest train index
    }
  }
  key = wave_min_u32_dpp(key);
  if (lane == 0) {
    st->bf_idx[i] = key == 0xffffffffu ? -1 : (int)(key & 0xffffu);
    st->bf_dist[i] = key == 0xffffffffu ? -1 : (int)(key >> 16);
  }
}

// pnpmatch::poseEstimation2D_2D (src/pnpmatch.cc:302-337) on one wave: keep the matches with distance <= max(2 min, 30)
// (:281-299), drop those whose CURRENT point lies in a box padded by 10 px (:318-328), 8-point F from the rest (:336).
__global__ __launch_bounds__(64) void k_tg_fmat(TrackState* st, const svo_kp* kp, const int32_t* nkp_p, int kstride,
                                                const int32_t* boxes, const int32_t* nboxes, int bstride) {
  __shared__ EpnpWaveLds S;
  st += blockIdx.y; kp += (size_t)blockIdx.y * kstride; nkp_p += blockIdx.y;
  const int lane = threadIdx.x;
  const int n_boxes = min(max(nboxes[blockIdx.y], 0), SVO_MAX_BOXES);
  boxes += (size_t)blockIdx.y * bstride * 4;
  double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (n_boxes > 0 && st->frame_num > 0) {
    const int nkp = min(*nkp_p, TRK_MAXKP);
    int idx[8], dist[8];
    uint32_t gmin = 0x7fffffffu;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = lane + 64 * t;
      idx[t] = i < nkp ? st->bf_idx[i] : -1;
      dist[t] = i < nkp ? st->bf_dist[i] : -1;
      if (idx[t] >= 0) gmin = min(gmin, (uint32_t)dist[t]);
    }
    gmin = wave_min_u32_dpp(gmin);
    const double md = (double)min(gmin, 10000u);
    const double thr = 2 * md > 30.0 ? 2 * md : 30.0;
    uint32_t keep = 0;
    double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = lane + 64 * t;
      x1[t] = y1[t] = x2[t] = y2[t] = 0.0;
      if (idx[t] >= 0 && (double)dist[t] <= thr) {
        const svo_kp k = kp[i];
        if (!svo_in_boxes(k.x, k.y, boxes, n_boxes, 10)) {
          keep |= 1u << t;
          x1[t] = (double)k.x; y1[t] = (double)k.y;
          x2[t] = (double)st->last_xy[2 * idx[t]]; y2[t] = (double)st->last_xy[2 * idx[t] + 1];
        }
      }
    }
    fmat8_wave(S, keep, x1, y1, x2, y2, F);
  }
  if (lane < 9) {
    double v = F[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) v = lane == k ? F[k] : v;
    st->F[lane] = v;
  }
}

// The same two steps for a GROUP of consecutive frames of one sequence in one launch each, ahead of the index chain: the
// brute-force matches and F of frame f need nothing but the front-end results of frames f - 1 and f and frame f's boxes -
// no tracker state - so the 11 + 42 us they cost per gated frame leave the index chain (which they had made longer than
// the pose chain: 122 against 117 us per frame in configs[4]).  blockIdx.y = frame of the group; the group's first frame must
// not be the sequence's first (the caller starts groups at frame >= 1 of a call; frame 0 of a call keeps k_tg_bf / k_tg_fmat,
// whose "last frame" is in the tracker state).  Results per frame in a GatePre record.
struct GatePre {
  int32_t bf_idx[TRK_MAXKP], bf_dist[TRK_MAXKP];
  double F[9];
};
__global__ __launch_bounds__(256) void k_tg_bf_group(const uint32_t* desc, const int32_t* nkp_p, int kstride, const int32_t* nboxes,
                                                     GatePre* pre) {
  __shared__ uint32_t td[8 * TRK_MAXKP];
  const int f = blockIdx.y;   // relative to the pointers: they address the group's first frame, row -1 is its predecessor
  if (nboxes[f] <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nkp = min(nkp_p[f], TRK_MAXKP), lastN = min(nkp_p[f - 1], TRK_MAXKP);
  const int i = blockIdx.x * 4 + wv;
  if (blockIdx.x * 4 >= nkp) return;
  const uint32_t* cur = desc + (ptrdiff_t)f * kstride * 8;
  {
    const uint4* d4 = reinterpret_cast<const uint4*>(desc + (ptrdiff_t)(f - 1) * kstride * 8);
    uint4 v[TRK_MAXKP * 2 / 256];
#pragma unroll
    for (int k = 0; k < TRK_MAXKP * 2 / 256; ++k) v[k] = d4[min(tid + 256 * k, max(2 * lastN - 1, 0))];
#pragma unroll
    for (int k = 0; k < TRK_MAXKP * 2 / 256; ++k) {
      const int q = tid + 256 * k;
      if (q < 2 * lastN) {
        const int j = q >> 1, w = 4 * (q & 1);
        td[(w + 0) * TRK_MAXKP + j] = v[k].x; td[(w + 1) * TRK_MAXKP + j] = v[k].y;
        td[(w + 2) * TRK_MAXKP + j] = v[k].z; td[(w + 3) * TRK_MAXKP + j] = v[k].w;
      }
    }
  }
  __syncthreads();
  if (i >= nkp) return;
  uint32_t qd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qd[k] = cur[8 * i + k];
  uint32_t key = 0xffffffffu;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int j = lane + 64 * t;
    if (j < lastN) {
      uint32_t d = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ td[k * TRK_MAXKP + j]);
      key = min(key, (d << 16) | (uint32_t)j);   // ties: the lowest train index
    }
  }
  key = wave_min_u32_dpp(key);
  if (lane == 0) {
    pre[f].bf_idx[i] = key == 0xffffffffu ? -1 : (int)(key & 0xffffu);
    pre[f].bf_dist[i] = key == 0xffffffffu ? -1 : (int)(key >> 16);
  }
}

__global__ __launch_bounds__(64) void k_tg_fmat_group(const svo_kp* kp, const int32_t* nkp_p, int kstride, const int32_t* boxes,
                                                      const int32_t* nboxes, int bstride, GatePre* pre) {
  __shared__ EpnpWaveLds S;
  const int f = blockIdx.y, lane = threadIdx.x;
  const int n_boxes = min(max(nboxes[f], 0), SVO_MAX_BOXES);
  boxes += (ptrdiff_t)f * bstride * 4;
  const svo_kp* cur = kp + (ptrdiff_t)f * kstride;
  const svo_kp* last = kp + (ptrdiff_t)(f - 1) * kstride;
  double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (n_boxes > 0) {
    const int nkp = min(nkp_p[f], TRK_MAXKP);
    int idx[8], dist[8];
    uint32_t gmin = 0x7fffffffu;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = lane + 64 * t;
      idx[t] = i < nkp ? pre[f].bf_idx[i] : -1;
      dist[t] = i < nkp ? pre[f].bf_dist[i] : -1;
      if (idx[t] >= 0) gmin = min(gmin, (uint32_t)dist[t]);
    }
    gmin = wave_min_u32_dpp(gmin);
    const double md = (double)min(gmin, 10000u);
    const double thr = 2 * md > 30.0 ? 2 * md : 30.0;
    uint32_t keep = 0;
    double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = lane + 64 * t;
      x1[t] = y1[t] = x2[t] = y2[t] = 0.0;
      if (idx[t] >= 0 && (double)dist[t] <= thr) {
        const svo_kp k = cur[i];
        if (!svo_in_boxes(k.x, k.y, boxes, n_boxes, 10)) {
          keep |= 1u << t;
          const svo_kp kl = last[idx[t]];
          x1[t] = (double)k.x; y1[t] = (double)k.y;
          x2[t] = (double)kl.x; y2[t] = (double)kl.y;
        }
      }
    }
    fmat8_wave(S, keep, x1, y1, x2, y2, F);
  }
  if (lane < 9) {
    double v = F[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) v = lane == k ? F[k] : v;
    pre[f].F[lane] = v;
  }
}

// ================================================================================================
// Pose chain: one launch per frame and sequence
// ================================================================================================
// k_tp_hyp: RANSAC samples of cv::solvePnPRansac (src/pnpmatch.cc:227), 4 waves per workgroup, one EPnP and one consensus
// count per wave (svo_pose_dev.h, svo_epnp_dev.h).  Every workgroup gathers the frame's correspondences itself.
// One sequence: all 100 samples in one launch, one single-wave workgroup each (the chip is idle, latency counts, and a
// CU's float64 pipeline serves one wave 2.7x faster than each of four).  Many sequences together:
// two launches per step - samples 0..15 first; then samples 16..99, whose workgroups first replay the adaptive rule over
// the first sixteen and leave at once when the iteration bound (log 0.01 / log(1 - w^5): 12 at 80 % inliers) says the loop
// can never reach their samples - six times less EPnP work on ordinary frames (61 k -> 74 k frames/s with 64 sequences).
// Wait (thread 0 polls, the workgroup follows through the barrier) until the index chain has published the record of the frame
// with tag `tag`; tag <= 0: the caller ordered the streams itself (many sequences, latency entry).  Bounded: see above.
__device__ __forceinline__ bool tp_wait_work(TrackState* st, const TrackWork* work, int tag) {
  __shared__ int s_lost;
  if (tag > 0) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while (ld_agent(&work->ready) != tag && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(2); ++spins; }
      s_lost = spins >= (1 << 22) ? 1 : 0;
      if (s_lost) st->overflow = 4;                  // lost hand-over: svo_track_overflowed reports it
    }
    __syncthreads();
    // what the record points at (keypoints, depths: written by front-end kernels on other CUs and L2s, with plain stores, before
    // the index chain published the tag) is read with plain loads below: acquire at agent scope, so that no line cached before
    // the tag was seen is served
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (s_lost) return false;                        // a stale record must not reach the results: the kernel leaves
  }
  return true;
}

#define TP_HYP_FIRST 16
#define TP_HYP_PRE 32   // most samples a later launch replays the adaptive rule over
struct TpHypLds {
  double Xw[TRK_MAXKP * 3], uv[TRK_MAXKP * 2];
  EpnpWaveLds ws[4];
  int cnt[TP_HYP_PRE], ok[TP_HYP_PRE], bound;
};

__global__ __launch_bounds__(256) void k_tp_hyp(TrackState* st, const TrackWork* work, const svo_kp* kp, const uint16_t* subsets,
                                                int kstride, int hyp_base, int tag) {
  TpHypLds& S = *reinterpret_cast<TpHypLds*>(tk_smem);
  st += blockIdx.y; work += blockIdx.y; kp += (size_t)blockIdx.y * kstride;
  if (!tp_wait_work(st, work, tag)) return;
  const long long t_start = clock64();
  if (hyp_base == 0 && blockIdx.x == 0 && threadIdx.x == 0) const_cast<TrackWork*>(work)->rt[2] = wall_clock64();
  // the record's header and this thread's first correspondence in ONE round trip to the coherent level (the entry is read
  // whether or not it exists - a stale one is a valid index - and dropped below if e >= n): a dependent round trip less
  // in front of every frame's RANSAC
  const int n = ld_agent(&work->n_edges), skip0 = ld_agent(&work->skip_match);
  const int gid0 = ld_agent(&work->edge_gid[threadIdx.x]), j0 = min(max(ld_agent(&work->edge_kp[threadIdx.x]), 0), kstride - 1);
  if (skip0 || n < 5) return;
  const int first = hyp_base + (int)blockIdx.x * (int)(blockDim.x >> 6);
  if (hyp_base > 0) {
    const int pre = min(hyp_base, TP_HYP_PRE);
    if ((int)threadIdx.x < pre) { S.cnt[threadIdx.x] = st->hyp[threadIdx.x].cnt; S.ok[threadIdx.x] = st->hyp[threadIdx.x].ok; }
    __syncthreads();
    if (threadIdx.x == 0) S.bound = pnp_bound_after(S.cnt, S.ok, n, pre);
    __syncthreads();
    if (first >= S.bound) return;
  }
  const float* gpos = st->gpos;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const bool first_e = e == (int)threadIdx.x;
    const float* gp = gpos + 3 * (size_t)((first_e ? gid0 : ld_agent(&work->edge_gid[e])) & (TRK_GPOS - 1));
    const svo_kp k = kp[first_e ? j0 : ld_agent(&work->edge_kp[e])];
    S.Xw[3 * e] = (double)gp[0]; S.Xw[3 * e + 1] = (double)gp[1]; S.Xw[3 * e + 2] = (double)gp[2];
    S.uv[2 * e] = (double)k.x; S.uv[2 * e + 1] = (double)k.y;
  }
  __syncthreads();
  const double K[4] = {(double)st->cam.fx, (double)st->cam.fy, (double)st->cam.cx, (double)st->cam.cy};
  const long long t_gather = clock64();
  pnp_hyp_block(S.ws, S.Xw, S.uv, n, K, subsets + (size_t)min(n, 512) * 500, st->hyp, first);
  if (first == 0 && threadIdx.x == 0) {
    st->pose_ts[0] = t_start; st->pose_ts[1] = t_gather; st->pose_ts[2] = S.ws[0].stamp[0]; st->pose_ts[3] = S.ws[0].stamp[4];
    st->pose_ts[4] = clock64();
    st->pose_ts[5] = S.ws[0].stamp[1]; st->pose_ts[6] = S.ws[0].stamp[2]; st->pose_ts[7] = S.ws[0].stamp[3]; st->pose_ts[12] = S.ws[0].sweeps;
    st->pose_ts[13] = S.ws[0].stamp[5]; st->pose_ts[14] = S.ws[0].stamp[6]; st->pose_ts[15] = S.ws[0].stamp[7];
  }
}

// The RANSAC samples in the parity mode (svo_set_option "epnp_exact"): one single-wave workgroup per sample, lane 0 walks
// OpenCV's loops in their own order (svo_epnp_exact_dev.h), the wave counts the consensus.  Every workgroup gathers the
// frame's correspondences into LDS itself, like k_tp_hyp.
struct TpHypExactLds { double Xw[TRK_MAXKP * 3], uv[TRK_MAXKP * 2]; PnpExactLds ex; };
__global__ __launch_bounds__(64) void k_tp_hyp_exact(TrackState* st, TrackWork* work, const svo_kp* kp, const uint16_t* subsets,
                                                     int kstride, int tag) {
  TpHypExactLds& S = *reinterpret_cast<TpHypExactLds*>(tk_smem);
  st += blockIdx.y; work += blockIdx.y; kp += (size_t)blockIdx.y * kstride;
  if (!tp_wait_work(st, work, tag)) return;
  if (threadIdx.x == 0 && blockIdx.x == 0) work->rt[2] = wall_clock64();
  const int n = ld_agent(&work->n_edges);
  if (ld_agent(&work->skip_match) || n < 5) return;
  const float* gpos = st->gpos;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const float* gp = gpos + 3 * (size_t)(ld_agent(&work->edge_gid[e]) & (TRK_GPOS - 1));
    const svo_kp k = kp[ld_agent(&work->edge_kp[e])];
    S.Xw[3 * e] = (double)gp[0]; S.Xw[3 * e + 1] = (double)gp[1]; S.Xw[3 * e + 2] = (double)gp[2];
    S.uv[2 * e] = (double)k.x; S.uv[2 * e + 1] = (double)k.y;
  }
  __syncthreads();
  const double K[4] = {(double)st->cam.fx, (double)st->cam.fy, (double)st->cam.cx, (double)st->cam.cy};
  pnp_hyp_exact_wave(S.ex, S.Xw, S.uv, n, K, subsets + (size_t)min(n, 512) * 500, st->hyp, blockIdx.x);
}

// The RANSAC samples in the order-preserving wave mode ("epnp_exact" = 2, the default): one single-wave workgroup per sample,
// OpenCV's operations over the wavefront with their rounding kept (svo_epnp_ord_dev.h); the wave counts the consensus.
struct TpHypOrdLds { double Xw[TRK_MAXKP * 3], uv[TRK_MAXKP * 2]; PnpOrdLds ord; int cnt[TP_HYP_PRE], ok[TP_HYP_PRE], bound; };
// One sample (called by the 64 lanes of the workgroup's only live wave).  FUSED: the sample belongs to a k_tp_tail_ord launch - its
// result goes out with agent-scope stores and is announced in work->hyp_done (the frame's workgroup of the SAME launch reads it).
// SEMI > 0 (FUSED only): the launch holds the samples from SEMI on, the first SEMI were solved by the launch before it on the same
// stream ("tail_semi", below): a sample the adaptive bound after those can no longer reach announces itself and leaves; one beyond
// 2 SEMI first waits for samples SEMI .. 2 SEMI - 1 (its own launch, dispatched before it) and applies the rule again.
template <bool FUSED>
__device__ __forceinline__ void tp_hyp_ord_body(TpHypOrdLds& S, TrackState* st, TrackWork* work, const svo_kp* kp, const uint16_t* subsets,
                                                int kstride, int hyp_base, int sample, int tag, int force_seq, int semi = 0) {
  if (!tp_wait_work(st, work, tag)) return;
  const long long t_start = clock64();
  if (threadIdx.x == 0 && sample == 0) work->rt[2] = wall_clock64();
  const int n = ld_agent(&work->n_edges);
  const int frame_tag = FUSED ? ld_agent(&work->frame_id) + 1 : 0;
  if (ld_agent(&work->skip_match) || n < 5) return;
  if (FUSED && (force_seq >> 8) == sample + 1) return;   // test switch "debug_lose_sample": this sample never reports (the frame part's bounded wait)
  if (FUSED && semi > 0) {
    const int tid = threadIdx.x;
    if (tid < semi) { S.cnt[tid] = ld_agent(&st->hyp[tid].cnt); S.ok[tid] = ld_agent(&st->hyp[tid].ok); }
    __syncthreads();
    if (tid == 0) S.bound = pnp_bound_after(S.cnt, S.ok, n, semi);
    __syncthreads();
    bool leave = sample >= S.bound;
    if (!leave && sample >= 2 * semi) {   // (the bound after `semi` samples reaches beyond 2 semi: all of semi .. 2 semi - 1 are being solved)
      if (tid >= semi && tid < 2 * semi) {
        int spins = 0;
        while (ld_agent(&work->hyp_early[tid]) != frame_tag && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(2); ++spins; }
        // (a sample that never reported: this one is solved whatever the rule would have said - solving more samples than the rule
        // visits changes no outcome; the frame part's own bounded wait reports the loss)
        const bool late = spins >= (1 << 22);
        S.cnt[tid] = late ? 0 : ld_agent(&st->hyp[tid].cnt); S.ok[tid] = late ? 0 : ld_agent(&st->hyp[tid].ok);
      }
      __syncthreads();
      if (tid == 0) S.bound = pnp_bound_after(S.cnt, S.ok, n, 2 * semi);
      __syncthreads();
      leave = sample >= S.bound;
    }
    if (leave) {
      if (tid == 0) {
        if (sample < TP_FLAGS) st_agent(&work->hyp_early[sample], frame_tag);
        __hip_atomic_fetch_add(&work->hyp_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
  }
  if (hyp_base > 0) {   // second launch of a many-sequence step: only the samples the adaptive bound can still reach (see k_tp_hyp)
    const int pre = min(hyp_base, TP_HYP_PRE);   // the rule over the samples of the earlier launches (all of them done: same stream)
    if ((int)threadIdx.x < pre) { S.cnt[threadIdx.x] = st->hyp[threadIdx.x].cnt; S.ok[threadIdx.x] = st->hyp[threadIdx.x].ok; }
    __syncthreads();
    if (threadIdx.x == 0) S.bound = pnp_bound_after(S.cnt, S.ok, n, pre);
    __syncthreads();
    if (sample >= S.bound) return;
  }
  const float* gpos = st->gpos;
  for (int e = threadIdx.x; e < n; e += 64) {
    const float* gp = gpos + 3 * (size_t)(ld_agent(&work->edge_gid[e]) & (TRK_GPOS - 1));
    const svo_kp k = kp[ld_agent(&work->edge_kp[e])];
    S.Xw[3 * e] = (double)gp[0]; S.Xw[3 * e + 1] = (double)gp[1]; S.Xw[3 * e + 2] = (double)gp[2];
    S.uv[2 * e] = (double)k.x; S.uv[2 * e + 1] = (double)k.y;
  }
  __syncthreads();
  const double K[4] = {(double)st->cam.fx, (double)st->cam.fy, (double)st->cam.cx, (double)st->cam.cy};
  if (!FUSED && sample == 0) {   // the frame kernel's copy of the correspondences (it skips its own gather: k_tp_frame, `pre_gathered`)
    for (int i = threadIdx.x; i < 3 * n; i += 64) st->Xw[i] = S.Xw[i];
    for (int i = threadIdx.x; i < 2 * n; i += 64) st->obs[i] = S.uv[i];
    if (threadIdx.x < 4) st->K[threadIdx.x] = K[threadIdx.x];
  }
  const long long t_gather = clock64();
  pnp_hyp_ord_wave<FUSED>(S.ord, S.Xw, S.uv, n, K, subsets + (size_t)min(n, 512) * 500, st->hyp, sample, (force_seq & 0xff) != 0);
  if (FUSED) {
    TP_STORES_DONE();   // (lane 0's agent-scope stores of the sample's result have reached the coherent level)
    if (threadIdx.x == 0) {
      if (sample < TP_FLAGS) st_agent(&work->hyp_early[sample], frame_tag);
      __hip_atomic_fetch_add(&work->hyp_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0 && S.ord.S.flag) atomicAdd(&st->epnp_fallbacks, 1);
  if (sample == 0 && threadIdx.x == 0) {
    const long long* sp = S.ord.S.stamp;
    st->pose_ts[0] = t_start; st->pose_ts[1] = t_gather; st->pose_ts[2] = sp[0]; st->pose_ts[3] = sp[7];
    st->pose_ts[4] = clock64();
    st->pose_ts[5] = sp[2]; st->pose_ts[6] = sp[3]; st->pose_ts[7] = sp[4]; st->pose_ts[12] = S.ord.S.flag;
    st->pose_ts[13] = sp[4]; st->pose_ts[14] = sp[5]; st->pose_ts[15] = sp[6];
  }
}
__global__ __launch_bounds__(64) void k_tp_hyp_ord(TrackState* st, TrackWork* work, const svo_kp* kp, const uint16_t* subsets,
                                                   int kstride, int hyp_base, int tag, int force_seq) {
  TpHypOrdLds& S = *reinterpret_cast<TpHypOrdLds*>(tk_smem);
  st += blockIdx.y; work += blockIdx.y; kp += (size_t)blockIdx.y * kstride;
  tp_hyp_ord_body<false>(S, st, work, kp, subsets, kstride, hyp_base, hyp_base + (int)blockIdx.x, tag, force_seq);
}
// the first samples of a frame whose other samples and frame part follow in a k_tp_tail_ord launch ("tail_semi"): results out with
// agent-scope stores, each sample announced
__global__ __launch_bounds__(64) void k_tp_hyp_ord_first(TrackState* st, TrackWork* work, const svo_kp* kp, const uint16_t* subsets,
                                                         int kstride, int tag, int force_seq) {
  TpHypOrdLds& S = *reinterpret_cast<TpHypOrdLds*>(tk_smem);
  st += blockIdx.y; work += blockIdx.y; kp += (size_t)blockIdx.y * kstride;
  tp_hyp_ord_body<true>(S, st, work, kp, subsets, kstride, 0, (int)blockIdx.x, tag, force_seq);
}

// k_tp_frame: RANSAC's acceptance rule over the samples, Optimizer::PoseOptimization, SetPose, the positions of the
// map points created this frame, the frame's record.
struct TpLds {
  PoseLds pose;
  alignas(16) int sm[16];
  float sT[16], sRwc[9], stwc[3];
  int cnt[PNP_HYP], ok[PNP_HYP], upd_r[PNP_HYP], best, good, iters;
  int late[TP_EARLY], lost;   // fused launch: a wait for the samples ran into its bound (the frame then counts as a PnP failure)
  int recf[8];   // the record's counters, fetched at the kernel's start
  double upd_ld[PNP_HYP];
};

// TPF_NT threads per sequence.  (Measured with ONE wave, TPF_NT = 64: k_tp_frame 30 -> 43 us under rocprofv3, 12.5 k -> 10.9 k
// frames/s - the rows of the Gram matrix and the MFMA accumulation spread over four SIMDs are worth more than the wave-local
// barriers a single wave would buy.  The LM stays templated on the thread count, pose_opt_block<NT>.)
#define TPF_NT 256
// FUSED: the frame's workgroup of a k_tp_tail_ord launch - the RANSAC samples run beside it in the SAME launch; it prepares
// everything that does not depend on them, then waits for work->hyp_done and reads their results with agent-scope loads.
template <bool FUSED>
__device__ __forceinline__ void tp_frame_body(TpLds& S, TrackState* st, TrackWork* work, const svo_kp* kp, const float* depth,
                                              svo_track_result* res_out, int kstride, int use_mfma, int tag, int pre_gathered) {
  const int tid = threadIdx.x;
  if (!tp_wait_work(st, work, tag)) return;
  if (!FUSED && tid == 0) work->rt[4] = wall_clock64();   // (FUSED: stamped when the samples are done, below)
  const long long tf0 = clock64();
  const int id = ld_agent(&work->frame_id), nkp = ld_agent(&work->nkp), skip = ld_agent(&work->skip_match), n_edges = ld_agent(&work->n_edges);
  // (this thread's first correspondence with the header, in one round trip: see k_tp_hyp)
  const int gid0 = ld_agent(&work->edge_gid[tid]), j0 = min(max(ld_agent(&work->edge_kp[tid]), 0), kstride - 1);
  float* gpos = st->gpos;
  // what the end of the kernel needs and the pose does not decide, requested now (one round trip to the coherent level beside
  // the work below instead of a chain of them behind the LM): the record's counters, and this thread's share of the keypoints
  // that become map points
  if (tid >= 64 && tid < 71) {
    const int* src = tid == 64 ? &work->n_stereo : tid == 65 ? &work->n_pass1 : tid == 66 ? &work->n_pass2 : tid == 67 ? &work->n_new
                   : tid == 68 ? &work->n_local : tid == 69 ? &work->diag[0] : &work->diag[1];
    S.recf[tid - 64] = ld_agent(src);
  }
  constexpr int NEWS = (TRK_MAXKP + TPF_NT - 1) / TPF_NT;
  int new_g[NEWS]; svo_kp new_k[NEWS]; float new_d[NEWS];
#pragma unroll
  for (int q = 0; q < NEWS; ++q) {
    const int j = tid + q * TPF_NT;
    new_g[q] = ld_agent(&work->new_gid[min(j, TRK_MAXKP - 1)]);   // (entries beyond nkp are stale: dropped below)
    new_k[q] = kp[min(j, kstride - 1)];
    new_d[q] = depth[min(j, kstride - 1)];
  }
  // ---- 3D-2D correspondences, ordered by keypoint index (src/pnpmatch.cc:216-224) ---------------
  // (with pre_gathered, frames the RANSAC kernel ran on arrive with st->Xw / obs / K filled in by its first workgroup)
  const bool have_corr = pre_gathered && !skip && n_edges >= 5 && id != 0;
  for (int e = tid; e < (have_corr ? 0 : n_edges); e += TPF_NT) {
    const int j = e == tid ? j0 : ld_agent(&work->edge_kp[e]);
    const svo_kp k = kp[j];
    float* gp = gpos + 3 * (size_t)((e == tid ? gid0 : ld_agent(&work->edge_gid[e])) & (TRK_GPOS - 1));
    float xyz[3];
    if (id == 0) {   // Tracking::init: the points of frame 0 are placed with the identity pose, before its LM
      const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
      tk_unproject(st->cam, k.x, k.y, depth[j], I3, z3, xyz);
      gp[0] = xyz[0]; gp[1] = xyz[1]; gp[2] = xyz[2];
    } else {
      xyz[0] = gp[0]; xyz[1] = gp[1]; xyz[2] = gp[2];
    }
    st->Xw[3 * e] = (double)xyz[0]; st->Xw[3 * e + 1] = (double)xyz[1]; st->Xw[3 * e + 2] = (double)xyz[2];
    st->obs[2 * e] = (double)k.x; st->obs[2 * e + 1] = (double)k.y;
  }
  if (tid < 4 && !have_corr) st->K[tid] = (double)((const float*)&st->cam)[tid];
  // ---- PnP initial pose (src/pnpmatch.cc:212-247): no prior; if solvePnPRansac fails the last pose stays ----------
  const bool ran = !skip && n_edges >= 5;
  bool decided_early = false;
  if (FUSED && ran) {
    // The samples of this launch.  First the TP_EARLY lowest ones, each awaited by a thread of its own (which fetches the sample's
    // consensus and prepares its pow / log terms as soon as it sees it): if the rule ends within them - it nearly always does -
    // the slowest of the other samples is not waited for.  (Bounded waits, as tp_wait_work's.)
    if (tid < TP_EARLY) {
      int spins = 0;
      while (ld_agent(&work->hyp_early[tid]) != id + 1 && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(1); ++spins; }
      // A sample that did not report (forward progress of a workgroup that waits for others of its launch rests on in-order
      // dispatch, which the hardware does and HIP does not promise): NOTHING of this launch's samples is used - the frame is
      // treated as cv::solvePnPRansac returning false (the last pose stays), the record says so (n_pnp_inliers = -1), the
      // sticky flag becomes 4 and the next svo_sync / svo_track_overflowed reports it (SVO_E_TIMEOUT) and switches the
      // context to the two-launch pose chain ("tail_fused" = 0).
      const bool late = spins >= (1 << 22);
      if (late) st->overflow = 4;
      S.late[tid] = late ? 1 : 0;
      const int c = late ? 0 : ld_agent(&st->hyp[tid].cnt), o = late ? 0 : ld_agent(&st->hyp[tid].ok);
      S.cnt[tid] = c; S.ok[tid] = o;
      double ld = 1.0; int r = 0;
      if (o && c > 4 && n_edges > 5) pnp_update_terms(c, n_edges, &ld, &r);
      S.upd_ld[tid] = ld; S.upd_r[tid] = r;
    } else if (tid < PNP_HYP) {
      S.cnt[tid] = 0; S.ok[tid] = 0; S.upd_ld[tid] = 1.0; S.upd_r[tid] = 0;   // (never visited when the rule ends early)
    }
    __syncthreads();
    if (tid == 0) {
      int l = 0;
      for (int k = 0; k < TP_EARLY; ++k) l |= S.late[k];
      S.lost = l;
      S.iters = l ? 0 : pnp_select_pre_bound(S.cnt, S.ok, S.upd_ld, S.upd_r, n_edges, TP_EARLY);
    }
    __syncthreads();
    decided_early = S.lost || S.iters <= TP_EARLY;   // (lost: nothing more is waited for)
    if (!decided_early) {   // all PNP_HYP results stored
      if (tid == 0) {
        int spins = 0;
        while (ld_agent(&work->hyp_done) < PNP_HYP && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(1); ++spins; }
        if (spins >= (1 << 22)) { st->overflow = 4; S.lost = 1; }
      }
      __syncthreads();
    }
  } else if (tid == 0) {
    S.lost = 0;
  }
  if (FUSED && tid == 0) work->rt[4] = wall_clock64();   // the frame part proper starts here (rt[2] .. rt[4]: the samples)
  if (ran && !decided_early && !(FUSED && S.lost))
    for (int h = tid; h < PNP_HYP; h += TPF_NT) {
      // every sample's thread prepares the pow / log terms the iteration bound would need if that sample became the best
      const int c = FUSED ? ld_agent(&st->hyp[h].cnt) : st->hyp[h].cnt, o = FUSED ? ld_agent(&st->hyp[h].ok) : st->hyp[h].ok;
      S.cnt[h] = c; S.ok[h] = o;
      double ld = 1.0; int r = 0;
      if (o && c > 4 && n_edges > 5) pnp_update_terms(c, n_edges, &ld, &r);
      S.upd_ld[h] = ld; S.upd_r[h] = r;
    }
  __syncthreads();
  if (tid == 0) {
    int good = 0, iters = 0;
    S.best = (ran && !S.lost) ? pnp_select_pre(S.cnt, S.ok, S.upd_ld, S.upd_r, n_edges, &good, &iters) : -1;
    S.good = good; S.iters = iters;
  }
  __syncthreads();
  if (tid < 16) {
    double v = (double)st->lastTcw[tid];
    if (S.best >= 0) {
      const PnpHyp& h = st->hyp[S.best];
      const int r = tid >> 2, c = tid & 3;
      if (FUSED) {
        const double* src = r == 3 ? nullptr : (c == 3 ? &h.t[r] : &h.R[3 * r + c]);
        v = src ? __builtin_bit_cast(double, ld_agent(reinterpret_cast<const long long*>(src))) : (c == 3 ? 1.0 : 0.0);
      } else {
        v = r == 3 ? (c == 3 ? 1.0 : 0.0) : (c == 3 ? h.t[r] : h.R[3 * r + c]);
      }
    }
    st->T[tid] = v;
    work->T_pnp[tid] = v;
  }
  if (tid == 0) {
    st->pnp.n_points = n_edges; st->pnp.n_inliers = S.best >= 0 ? S.good : 0; st->pnp.best_hypothesis = S.best;
    st->pnp.ok = S.best >= 0 ? 1 : 0; st->pnp.iterations = S.iters;
    work->pnp_best = S.best; work->pnp_iterations = S.iters; work->pnp_inliers = S.best >= 0 ? S.good : 0;
    work->pnp_ok = S.best >= 0 ? 1 : 0;
  }
  __syncthreads();
  const long long tf1 = clock64();
  // ---- Optimizer::PoseOptimization (src/Optimizer.cc:15-86) from the CV_32F-stored PnP pose ----------------------
  pose_opt_block<TPF_NT>(S.pose, st->Xw, st->obs, n_edges, st->K, st->T, &st->lm, 1, use_mfma);
  __syncthreads();
  const long long tf2 = clock64();
  // ---- SetPose (CV_32F, src/Optimizer.cc:82-83), positions of the points created this frame -------
  if (tid < 16) S.sT[tid] = (float)st->T[tid];
  __syncthreads();
  if (tid < 9) S.sRwc[tid] = S.sT[4 * (tid % 3) + tid / 3];
  __syncthreads();
  if (tid < 3) {
    const double acc = (double)S.sRwc[3 * tid] * (double)S.sT[3] + (double)S.sRwc[3 * tid + 1] * (double)S.sT[7] +
                       (double)S.sRwc[3 * tid + 2] * (double)S.sT[11];
    S.stwc[tid] = (float)(-acc);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NEWS; ++q) {
    const int g = tid + q * TPF_NT < nkp ? new_g[q] : -1;
    if (g < 0) continue;
    float xyz[3];
    tk_unproject(st->cam, new_k[q].x, new_k[q].y, new_d[q], S.sRwc, S.stwc, xyz);
    float* gp = gpos + 3 * (size_t)(g & (TRK_GPOS - 1));
    gp[0] = xyz[0]; gp[1] = xyz[1]; gp[2] = xyz[2];
  }
  if (tid == 0) {
    svo_track_result r;
    for (int i = 0; i < 16; ++i) { r.Tcw[i] = S.sT[i]; st->lastTcw[i] = S.sT[i]; }
    r.frame_id = id; r.n_kp = nkp; r.n_stereo = S.recf[0];
    r.n_match_pass1 = S.recf[1]; r.n_match_pass2 = S.recf[2];
    r.n_pnp_inliers = skip ? 0 : (S.lost ? -1 : st->pnp.n_inliers);   // (-1: the samples of a fused launch did not report in time)
    r.n_lm_edges = n_edges;
    r.n_new_mappoints = S.recf[3];
    r.n_local_map = S.recf[4];
    r.lm_iterations = st->lm.iterations;
    // diagnostics: rows of pass 1 / pass 2 that could match at all.  (The ROUNDS a pass took are not part of the record: a
    // dense row may or may not see a claim made earlier in the same phase - the outcome is the same either way, the
    // number of rounds is not, and records are compared byte for byte.  svo_debug_track_frames reports them.)
    r.reserved[0] = S.recf[5] & 0xffff;
    r.reserved[1] = S.recf[6] & 0xffff;
    *res_out = r;
    st->pose_ts[8] = tf0; st->pose_ts[9] = tf1; st->pose_ts[10] = tf2; st->pose_ts[11] = clock64();
    work->rt[3] = wall_clock64();
  }
}
__global__ __launch_bounds__(TPF_NT) void k_tp_frame(TrackState* st, TrackWork* work, const svo_kp* kp,
                                                  const float* depth, svo_track_result* res_out, int kstride,
                                                  int use_mfma, int tag, int pre_gathered) {
  TpLds& S = *reinterpret_cast<TpLds*>(tk_smem);
  st += blockIdx.y; work += blockIdx.y; kp += (size_t)blockIdx.y * kstride;
  depth += (size_t)blockIdx.y * kstride; res_out += blockIdx.y;
  tp_frame_body<false>(S, st, work, kp, depth, res_out, kstride, use_mfma, tag, pre_gathered);
}

// The pose chain of one frame in ONE launch ("tail_fused", the default with the order-preserving solver and one sequence):
// workgroups 0 .. PNP_HYP - 1 are the RANSAC samples (k_tp_hyp_ord's: their first wave solves, the other three leave at once - a
// sample wants a CU's float64 pipeline to itself), workgroup PNP_HYP is the frame's (k_tp_frame's).  Workgroups are dispatched in
// index order, so the frame's workgroup - the only one that waits for others of its launch - is placed last and cannot keep a
// sample from starting.  Saved against two launches: the boundary between them (completion signal, barrier packet, dispatch:
// 2.6 us between two empty kernels of a stream, tools/microbench/anyorder_probe) and the frame kernel's own start-up (hand-over
// poll, record and keypoint loads, the gather of the correspondences), which now runs in the shadow of the samples.
union TpTailLds { TpHypOrdLds hyp; TpLds frame; };
__global__ __launch_bounds__(TPF_NT) void k_tp_tail_ord(TrackState* st, TrackWork* work, const svo_kp* kp, const float* depth,
                                                     const uint16_t* subsets, svo_track_result* res_out, int kstride, int use_mfma,
                                                     int tag, int force_seq, int first_sample) {
  st += blockIdx.y; work += blockIdx.y; kp += (size_t)blockIdx.y * kstride;
  if (blockIdx.x < PNP_HYP - first_sample) {
    if (threadIdx.x >= 64) return;
    tp_hyp_ord_body<true>(*reinterpret_cast<TpHypOrdLds*>(tk_smem), st, work, kp, subsets, kstride, 0, first_sample + (int)blockIdx.x, tag, force_seq,
                          first_sample);
  } else {
    depth += (size_t)blockIdx.y * kstride; res_out += blockIdx.y;
    tp_frame_body<true>(*reinterpret_cast<TpLds*>(tk_smem), st, work, kp, depth, res_out, kstride, use_mfma, tag, 0);
  }
}

// --------------------------------------------------------------------------------------------
// Host side
// --------------------------------------------------------------------------------------------
static void shard_gather_free(svo_ctx* ctx);
static void track_batch_sets_free(svo_ctx* ctx) {
  if (ctx->tb_kp) hipFree(ctx->tb_kp);
  if (ctx->tb_desc) hipFree(ctx->tb_desc);
  if (ctx->tb_nkp) hipFree(ctx->tb_nkp);
  if (ctx->tb_uR) hipFree(ctx->tb_uR);
  if (ctx->tb_depth) hipFree(ctx->tb_depth);
  if (ctx->tb_sad) hipFree(ctx->tb_sad);
  ctx->tb_kp = nullptr; ctx->tb_desc = nullptr; ctx->tb_nkp = nullptr; ctx->tb_uR = nullptr; ctx->tb_depth = nullptr; ctx->tb_sad = nullptr;
}
void svo_track_release(svo_ctx* ctx) {
  shard_gather_free(ctx);
  track_batch_sets_free(ctx);
  for (int q = 0; q < 2; ++q) {
    if (ctx->tb_done[q]) { hipEventDestroy(ctx->tb_done[q]); ctx->tb_done[q] = nullptr; }
    ctx->tb_used[q] = false;
  }
  ctx->tb_parity = 0;
  if (ctx->d_track) { hipFree(ctx->d_track); ctx->d_track = nullptr; }
  if (ctx->d_work) { hipFree(ctx->d_work); ctx->d_work = nullptr; }
  if (ctx->d_gate_pre) { hipFree(ctx->d_gate_pre); ctx->d_gate_pre = nullptr; }
  ctx->n_seq = 0; ctx->work_cap = 0;
  for (hipEvent_t e : ctx->ev_frame) hipEventDestroy(e);
  ctx->ev_frame.clear();
  if (ctx->ev_frontend) { hipEventDestroy(ctx->ev_frontend); ctx->ev_frontend = nullptr; }
  if (ctx->stream_idx) { hipStreamDestroy(ctx->stream_idx); ctx->stream_idx = nullptr; }
  for (hipEvent_t e : ctx->ev_sub) hipEventDestroy(e);
  ctx->ev_sub.clear();
  if (ctx->stream_fe) { hipStreamDestroy(ctx->stream_fe); ctx->stream_fe = nullptr; }
  if (ctx->stream_fe_batch) { hipStreamDestroy(ctx->stream_fe_batch); ctx->stream_fe_batch = nullptr; }
  if (ctx->stream_dense) { hipStreamDestroy(ctx->stream_dense); ctx->stream_dense = nullptr; }
  if (ctx->h_prod) { hipHostFree(ctx->h_prod); ctx->h_prod = nullptr; }
  for (int p = 0; p < 2; ++p) {
    if (ctx->ms_kp[p]) hipFree(ctx->ms_kp[p]);
    if (ctx->ms_desc[p]) hipFree(ctx->ms_desc[p]);
    if (ctx->ms_nkp[p]) hipFree(ctx->ms_nkp[p]);
    if (ctx->ms_depth[p]) hipFree(ctx->ms_depth[p]);
    ctx->ms_kp[p] = nullptr; ctx->ms_desc[p] = nullptr; ctx->ms_nkp[p] = nullptr; ctx->ms_depth[p] = nullptr;
    if (ctx->ms_fe_done[p]) { hipEventDestroy(ctx->ms_fe_done[p]); ctx->ms_fe_done[p] = nullptr; }
    if (ctx->ms_tail_done[p]) { hipEventDestroy(ctx->ms_tail_done[p]); ctx->ms_tail_done[p] = nullptr; }
    ctx->ms_tail_recorded[p] = false;
  }
  ctx->ms_cap = 0; ctx->ms_parity = 0;
}

// The index chain's stream: high priority, side by side with the pose chain's (and with the batched front end's, if that exists).
static int track_index_stream(svo_ctx* ctx) {
  return svo_pick_stream(ctx, [](hipStream_t* s) { return svo_stream_create(s, +1); }, {ctx->stream, ctx->stream_fe_batch}, &ctx->stream_idx,
                         &ctx->idx_probe_attempts, &ctx->idx_probe_spins, {}, {ctx->stream_fe_batch, ctx->stream_fe, ctx->stream_dense});
}
static void track_stream_diag(svo_ctx* ctx);
// The batched tracker's front-end stream ("fe_cu_percent" of the CUs): side by side with both chains of the tail.
int svo_track_fe_batch_stream(svo_ctx* ctx) {
  if (ctx->stream_fe_batch) return SVO_OK;
  const int dev = ctx->device, pct = ctx->opt_fe_cu_percent;
  int attempts = 0, percent = 0;
  const int rc = svo_pick_stream(ctx, [dev, pct](hipStream_t* s) { return pct < 100 ? svo_stream_create_masked(s, dev, pct) : svo_stream_create(s, -1); },
                                 {ctx->stream, ctx->stream_idx}, &ctx->stream_fe_batch, &attempts, &percent,
                                 {ctx->stream, ctx->stream_idx, ctx->stream_dense}, {ctx->stream_dense});   // (its grids wait for their eighth of the CUs: no chain, and not the dense stage, behind them)
  if (rc == SVO_OK) { ctx->stream_diag_done = true; track_stream_diag(ctx); }
  return rc;
}

// SVO_STREAM_DIAG=1: how long a grid of many small workgroups (the index chain's shape) takes on each of the tracker's streams, alone
// and while front-end-like grids keep the front end's stream busy - printed once per context (docs/NEXT_ROUNDS.md, process states).
__global__ void k_diag_spin(long long cycles) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(2);
}
static void track_stream_diag(svo_ctx* ctx) {
  static const bool on = []() { const char* e = getenv("SVO_STREAM_DIAG"); return e && e[0] == '1'; }();
  if (!on || !ctx->stream_idx || !ctx->stream_fe_batch) return;
  hipStream_t ss[3] = {ctx->stream, ctx->stream_idx, ctx->stream_fe_batch};
  const char* nm[3] = {"pose", "index", "front end"};
  auto grid_us = [](hipStream_t q, int wgs, int threads, long long cyc) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(q);
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_diag_spin, dim3(wgs), dim3(threads), 0, q, cyc);
      hipStreamSynchronize(q);
      best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best;
  };
  for (int k = 0; k < 3; ++k) {
    const double one = grid_us(ss[k], 1, 64, 4000), many = grid_us(ss[k], 3072, 64, 4000), big = grid_us(ss[k], 2048, 256, 40000);
    double beside = 0;
    if (k < 2) {
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_diag_spin, dim3(4096), dim3(256), 0, ss[2], 100000LL);
      beside = grid_us(ss[k], 3072, 64, 4000);
      hipStreamSynchronize(ss[2]);
    }
    fprintf(stderr, "[svo stream diag] ctx %p %-9s stream %p: 1 workgroup %.0f us, 3072 x 64 threads %.0f us, 2048 x 256 threads (18 us each) %.0f us, 3072 x 64 beside front-end grids %.0f us\n",
            (void*)ctx, nm[k], (void*)ss[k], one, many, big, beside);
  }
  (void)hipGetLastError();
}

static size_t tail_lds_bytes() {
  static const size_t v = []() { const char* e = getenv("SVO_TAIL_LDS_KB"); const size_t kb = e ? (size_t)atoi(e) : 0; return std::max(sizeof(TpTailLds), kb << 10); }();
  return v;
}
// streams, events, work records and the kernels' LDS opt-ins for `frames` frames per call of `nseq` sequences
static int track_resources(svo_ctx* ctx, int frames, int nseq) {
  if (!ctx->stream_idx) { const int rcs = track_index_stream(ctx); if (rcs) return rcs; }
  if (!ctx->stream_diag_done) { ctx->stream_diag_done = true; track_stream_diag(ctx); }   // (SVO_STREAM_DIAG=1; once per context)
  if (!ctx->ev_frontend) SVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_frontend, hipEventDisableTiming));
  while ((int)ctx->ev_frame.size() < frames) {
    hipEvent_t e;
    SVO_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->ev_frame.push_back(e);
  }
  const int need = std::max(frames, nseq);
  if (ctx->work_cap < need) {
    SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_work) hipFree(ctx->d_work);
    if (ctx->d_gate_pre) hipFree(ctx->d_gate_pre);
    ctx->d_work = nullptr; ctx->d_gate_pre = nullptr; ctx->work_cap = 0;
    if (hipMalloc(&ctx->d_gate_pre, sizeof(GatePre) * (size_t)need * 2) != hipSuccess) return SVO_E_NOMEM;
    if (hipMalloc(&ctx->d_work, sizeof(TrackWork) * (size_t)need * 2) != hipSuccess) return SVO_E_NOMEM;   // two halves (see tail_enqueue)
    // no stale `ready` tags.  On the ctx stream and waited for: hipMemset runs on the null stream, which the tracker's
    // non-blocking streams do not order against - the index chain's first records were zeroed by it when the chain started
    // within microseconds of this call (svo_track_sharded_dev with sub-batched front ends, fresh context)
    SVO_HIP(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(TrackWork) * (size_t)need * 2, ctx->stream));
    SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->work_cap = need;
  }
  if (ctx->track_lds_state == 0) {
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ti_resolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sizeof(TiLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_frame), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(TpLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_hyp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(TpHypLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_hyp_ord), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(TpHypOrdLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_hyp_ord_first), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(TpHypOrdLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_tail_ord), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)tail_lds_bytes()) == hipSuccess;
    ctx->track_lds_state = ok ? 1 : -1;
    if (!ok) ctx->last_error = std::string("hipFuncSetAttribute(tracker kernels): ") + hipGetErrorString(hipGetLastError());
  }
  return ctx->track_lds_state > 0 ? SVO_OK : SVO_E_HIP;
}

// The ordered tail for front-end results that are already in HBM (`kstride` keypoints per frame slot):
//   nseq == 1: `frames` consecutive frames of the one sequence, record f into d_res[f];
//   nseq  > 1: one frame of each of nseq sequences (frames == 1), sequence q from frame slot q into d_res[q].
// The index chain is enqueued on ctx->stream_idx and runs ahead; the pose chain follows on ctx->stream.
// `fe_events` (may be null): fe_events[f] != nullptr is an event the index chain must wait for before frame f (its front-end
// results are produced on another stream, in sub-batches).
// `bx` (may be null): detection boxes in HBM - frame f's (nseq == 1) or sequence q's (nseq > 1) at bx->boxes + 4 * bx->stride * f.
static int tail_enqueue(svo_ctx* ctx, const svo_kp* kp, const uint8_t* desc8, const int32_t* nkp, const float* depth,
                        int kstride, int frames, int nseq, svo_track_result* d_res, const svo_boxes_dev* bx,
                        const hipEvent_t* fe_events = nullptr, const int* row_of_frame = nullptr, int work_half = 0,
                        hipEvent_t idx_wait = nullptr, bool fe_elsewhere = false) {
  // work_half: which half of the TrackWork records this call uses (svo_track_batch_dev alternates, so that the index chain of
  // call c + 1 may start while the pose chain of call c still reads its records); idx_wait: what the index chain waits for
  // before it starts - by default everything the ctx stream holds (the front end of this call, the previous call's pose
  // chain), with fe_elsewhere (front end on its own stream, announced through fe_events) only `idx_wait`, if any
  if (bx && (!bx->boxes || !bx->n || bx->stride < 1)) bx = nullptr;
  int rc = track_resources(ctx, frames, nseq);
  if (rc) return rc;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  TrackWork* work = reinterpret_cast<TrackWork*>(ctx->d_work) + (size_t)work_half * ctx->work_cap;
  ctx->work_last_half = work_half;
  hipStream_t s0 = ctx->stream, s1 = ctx->stream_idx;
  const unsigned ny = (unsigned)nseq;
  const uint32_t* desc = reinterpret_cast<const uint32_t*>(desc8);
  if (!fe_elsewhere) {
    SVO_HIP(ctx, hipEventRecord(ctx->ev_frontend, s0));            // keypoints, descriptors, depths are ready ...
    SVO_HIP(ctx, hipStreamWaitEvent(s1, ctx->ev_frontend, 0));    // ... and the previous call's pose chain has read its records
  } else if (idx_wait) {
    SVO_HIP(ctx, hipStreamWaitEvent(s1, idx_wait, 0));            // the pose chain that last read this half of the records
  }
  // per-kernel HIP-event timing (svo_profile_enable) costs ~2.5 us per event pair on the host - more than a tail kernel's
  // launch; the tail is therefore SAMPLED: every 32nd frame of a call is timed (an event pair around a kernel also holds the chain up by ~5 us), the others run untimed
  const bool prof = ctx->profiling;
  // frame f's front-end results sit in row f of the arrays unless the caller says otherwise (svo_track_sharded_dev)
  auto row = [&](int f) { return (size_t)(row_of_frame ? row_of_frame[f] : f); };
  // gated frames of ONE sequence whose front-end rows are consecutive: brute-force matches + F for groups of frames in one
  // launch each, ahead of the chain (k_tg_bf_group)
  const bool gate_group = bx && nseq == 1 && frames > 1 && !row_of_frame && ctx->opt_gate_group;
  GatePre* gpre = reinterpret_cast<GatePre*>(ctx->d_gate_pre) + (size_t)work_half * ctx->work_cap;
  auto enqueue_index = [&](int f) {
    const svo_kp* kpf = kp + row(f) * kstride;
    const uint32_t* descf = desc + row(f) * kstride * 8;
    const float* depf = depth + row(f) * kstride;
    const int32_t* nkpf = nkp + row(f);
    ctx->profiling = prof && (f % 32 == 0 || frames < 32);
    if (fe_events && fe_events[f]) hipStreamWaitEvent(s1, fe_events[f], 0);
    const bool pre_f = gate_group && f >= 1;   // this frame's F is solved ahead of the chain, with its group
    if (gate_group && (f == 0 || (fe_events && fe_events[f]))) {
      // a group: the frames whose front-end results this wait (or the call's start) made available, except frame 0 of the call
      int g1 = f + 1;
      while (g1 < frames && !(fe_events && fe_events[g1])) ++g1;
      const int a = std::max(f, 1), n = g1 - a;
      if (n > 0) {
        {
          SvoTimer t(ctx, "k_tg_bf_group", s1);
          hipLaunchKernelGGL(k_tg_bf_group, dim3(TRK_MAXKP / 4, n), dim3(256), 0, s1, desc + (size_t)a * kstride * 8, nkp + a, kstride,
                             bx->n + a, gpre + a);
        }
        {
          SvoTimer t(ctx, "k_tg_fmat_group", s1);
          hipLaunchKernelGGL(k_tg_fmat_group, dim3(1, n), dim3(64), 0, s1, kp + (size_t)a * kstride, nkp + a, kstride,
                             bx->boxes + (size_t)a * bx->stride * 4, bx->n + a, bx->stride, gpre + a);
        }
      }
    }
    // frame f's boxes (one sequence) / the sequences' boxes of this step (many): the kernels add blockIdx.y themselves
    const int32_t* bxf = bx ? bx->boxes + (nseq == 1 ? (size_t)f * bx->stride * 4 : 0) : nullptr;
    const int32_t* nbf = bx ? bx->n + (nseq == 1 ? f : 0) : nullptr;
    const int bstride = bx ? bx->stride : 0;
    if (bx && !pre_f) {   // F for the epipolar veto (src/pnpmatch.cc:302-337): brute-force matches cur -> last, then the 8-point solve
      {
        SvoTimer t(ctx, "k_tg_bf", s1);
        hipLaunchKernelGGL(k_tg_bf, dim3(TRK_MAXKP / 4, ny), dim3(256), 0, s1, st, descf, nkpf, kstride, nbf);
      }
      {
        SvoTimer t(ctx, "k_tg_fmat", s1);
        hipLaunchKernelGGL(k_tg_fmat, dim3(1, ny), dim3(64), 0, s1, st, kpf, nkpf, kstride, bxf, nbf, bstride);
      }
    }
    {
      SvoTimer t(ctx, "k_ti_lists", s1);
      if (ny >= 8) hipLaunchKernelGGL(k_ti_lists<16>, dim3(TRK_ROWS_MAX / 16, ny), dim3(256), 0, s1, st, descf, nkpf, kstride, ctx->opt_track_lcap, ctx->opt_track_nblk);
      else hipLaunchKernelGGL(k_ti_lists<4>, dim3(TRK_ROWS_MAX / 4, ny), dim3(256), 0, s1, st, descf, nkpf, kstride, ctx->opt_track_lcap, ctx->opt_track_nblk);
    }
    {
      SvoTimer t(ctx, "k_ti_resolve", s1);
      hipLaunchKernelGGL(k_ti_resolve, dim3(1, ny), dim3(1024), sizeof(TiLds), s1, st, work + f, kpf, descf, nkpf, depf, kstride,
                         bxf, nbf, bstride, pre_f ? gpre[f].F : nullptr);
    }
  };
  // one sequence: the pose kernels find out by themselves when their frame's record is there (tp_wait_work) - no stream event
  // between the chains; tag of frame f = its index in the sequence + 1
  const bool flagged = ny == 1 && ctx->opt_pose_flag;
  auto tag_of = [&](int f) { return flagged ? ctx->track_frame + f + 1 : 0; };
  auto enqueue_pose = [&](int f) {
    const svo_kp* kpf = kp + row(f) * kstride;
    const float* depf = depth + row(f) * kstride;
    ctx->profiling = prof && (f % 32 == 0 || frames < 32);
    if (ctx->opt_epnp_exact == 2 && ny == 1 && !ctx->hyp_two_launch && ctx->opt_tail_fused && (ctx->opt_depth_source == 0 || ctx->opt_tail_fused == 2)) {
      // one sequence, the default solver: samples and frame part in ONE launch (k_tp_tail_ord).  Not beside a dense stereo stage
      // (depth_source 1 / 2): a fused launch's sample workgroups carry the frame part's 67 KB of LDS and three idle waves each,
      // which the dense kernels running on the same CUs pay for (configs[4]: 6.05 k frames/s fused, 6.85 k with two launches)
      SvoTimer t(ctx, "k_tp_tail_ord");
      hipLaunchKernelGGL(k_tp_tail_ord, dim3(PNP_HYP + 1, 1), dim3(TPF_NT), tail_lds_bytes(), s0, st, work + f, kpf, depf, ctx->d_pnp_subsets,
                         d_res + f, kstride, ctx->opt_pose_mfma, tag_of(f), ctx->opt_epnp_force_seq | (ctx->opt_debug_lose_sample << 8), 0);
      return;
    }
    if (ctx->opt_epnp_exact == 2 && ctx->opt_tail_fused && ctx->opt_tail_semi && ((ny == 1 && ctx->opt_depth_source != 0) || (ny >= 8 && ctx->opt_tail_semi == 2))) {
      // beside a dense stage (or, "tail_semi" = 2, many sequences): the first TP_EARLY samples, then the other samples and the frame
      // part in one launch - 8 CUs' float64 pipelines per ordinary frame instead of 100 taken from the dense kernels, and no more launches.  cv::solvePnPRansac ends within 8 samples on 96 % of the frames: the second
      // launch's sample workgroups replay the rule and leave at once, the frame part starts with them.
      const int fs = ctx->opt_epnp_force_seq | (ctx->opt_debug_lose_sample << 8);
      {
        SvoTimer t(ctx, "k_tp_hyp_ord");
        hipLaunchKernelGGL(k_tp_hyp_ord_first, dim3(TP_EARLY, ny), dim3(64), sizeof(TpHypOrdLds), s0, st, work + f, kpf, ctx->d_pnp_subsets, kstride, tag_of(f), fs);
      }
      SvoTimer t(ctx, "k_tp_tail_ord");
      hipLaunchKernelGGL(k_tp_tail_ord, dim3(PNP_HYP - TP_EARLY + 1, ny), dim3(TPF_NT), tail_lds_bytes(), s0, st, work + f, kpf, depf, ctx->d_pnp_subsets,
                         d_res + f, kstride, ctx->opt_pose_mfma, tag_of(f), fs, TP_EARLY);
      return;
    }
    if (ctx->opt_epnp_exact == 2) {
      SvoTimer t(ctx, "k_tp_hyp_ord");
      if (ny >= 8 || ctx->hyp_two_launch) {
        // many sequences (or a dense stage beside the tail): throughput counts.  cv::solvePnPRansac visits a median of 4 samples
        // on these sequences, 96 % of the frames 8 or fewer, 99.9 % 16 or fewer: F samples per sequence first ("hyp_first",
        // default 8), then F more, then the rest - the workgroups of a later launch replay the adaptive rule over the earlier
        // samples and leave at once when the loop can never reach theirs
        const int F = std::max(4, std::min(ctx->opt_hyp_first, TP_HYP_PRE / 2));
        const int base[4] = {0, F, 2 * F, PNP_HYP};
        for (int k = 0; k < 3; ++k)
          hipLaunchKernelGGL(k_tp_hyp_ord, dim3(base[k + 1] - base[k], ny), dim3(64), sizeof(TpHypOrdLds), s0, st, work + f, kpf,
                             ctx->d_pnp_subsets, kstride, base[k], k == 0 ? tag_of(f) : 0, ctx->opt_epnp_force_seq);
      } else {
        hipLaunchKernelGGL(k_tp_hyp_ord, dim3(PNP_HYP, ny), dim3(64), sizeof(TpHypOrdLds), s0, st, work + f, kpf, ctx->d_pnp_subsets, kstride,
                           0, tag_of(f), ctx->opt_epnp_force_seq);
      }
    } else if (ctx->opt_epnp_exact) {
      SvoTimer t(ctx, "k_tp_hyp_exact");
      hipLaunchKernelGGL(k_tp_hyp_exact, dim3(PNP_HYP, ny), dim3(64), sizeof(TpHypExactLds), s0, st, work + f, kpf, ctx->d_pnp_subsets, kstride,
                         tag_of(f));
    } else {
      SvoTimer t(ctx, "k_tp_hyp");
      if (ny >= 8) {
        // many sequences: throughput counts - solve sixteen samples per sequence, then only those the adaptive bound can reach
        hipLaunchKernelGGL(k_tp_hyp, dim3(TP_HYP_FIRST / 4, ny), dim3(256), sizeof(TpHypLds), s0, st, work + f, kpf, ctx->d_pnp_subsets,
                           kstride, 0, 0);
        hipLaunchKernelGGL(k_tp_hyp, dim3((PNP_HYP - TP_HYP_FIRST) / 4, ny), dim3(256), sizeof(TpHypLds), s0, st, work + f, kpf,
                           ctx->d_pnp_subsets, kstride, TP_HYP_FIRST, 0);
      } else {
        // one sequence: latency counts and the chip is idle - all 100 samples at once, ONE wave per workgroup (= per CU: the
        // four SIMDs of a CU share its float64 pipeline, four solves side by side on a CU run 2.7x slower each)
        hipLaunchKernelGGL(k_tp_hyp, dim3(PNP_HYP, ny), dim3(64), sizeof(TpHypLds), s0, st, work + f, kpf, ctx->d_pnp_subsets,
                           kstride, 0, tag_of(f));
      }
    }
    {
      SvoTimer t(ctx, "k_tp_frame");
      hipLaunchKernelGGL(k_tp_frame, dim3(1, ny), dim3(TPF_NT), sizeof(TpLds), s0, st, work + f, kpf, depf, d_res + f, kstride,
                         ctx->opt_pose_mfma, tag_of(f), ctx->opt_epnp_exact == 2 ? 1 : 0);
    }
  };
  // The pose chain takes the frames over in groups: ONE event (a barrier packet on the pose stream, ~3 us even when long
  // signalled) per group instead of one per frame.  Group sizes ramp up 1, 1, 2, 4, 4, ... (option "track_group": 1 -> 11.80 k, 4 -> 12.01 k, 8 -> 11.94 k, 16 -> 11.69 k frames/s) so that the first poses of
  // a call do not wait for a long run of the index chain; the index chain is the faster one and stays ahead after that.
  // The host enqueues the index kernels one group ahead of the pose kernels.
  int g0 = 0, gsize = 1, prev0 = -1, prev1 = -1, ngroup = 0;
  while (g0 < frames || prev0 >= 0) {
    int cur0 = -1, cur1 = -1;
    if (g0 < frames) {
      cur0 = g0; cur1 = std::min(frames, g0 + gsize);
      for (int f = cur0; f < cur1; ++f) enqueue_index(f);
      if (!flagged) hipEventRecord(ctx->ev_frame[cur0], s1);
      g0 = cur1;
      ++ngroup;
      if (ngroup >= 2) gsize = std::min(2 * gsize, ctx->opt_track_group > 0 ? ctx->opt_track_group : 1);
    }
    if (prev0 >= 0) {
      if (!flagged) hipStreamWaitEvent(s0, ctx->ev_frame[prev0], 0);
      for (int f = prev0; f < prev1; ++f) enqueue_pose(f);   // (always AFTER the index kernels of the same frames were enqueued)
    }
    prev0 = cur0; prev1 = cur1;
  }
  ctx->profiling = prof;
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

// (Re)allocate `nseq` tracker states and reset them: pose I, empty map, frame counter 0.
static int track_reset_n(svo_ctx* ctx, const svo_camera* cam, int nseq) {
  if (!ctx || !cam || nseq < 1) return SVO_E_INVALID;
  if (ctx->max_kp > TRK_MAXKP) return SVO_E_CAPACITY;
  hipSetDevice(ctx->device);
  if (ctx->stream_idx) SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_idx));
  if (ctx->stream_fe) SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_fe));
  if (ctx->stream_fe_batch) SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_fe_batch));
  if (ctx->stream_dense) SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_dense));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  { const int rcf = svo_hostfeed_flush(ctx); if (rcf) return rcf; }   // (records of host-fed calls of the sequence that ends here)
  ctx->ms_parity = 0; ctx->ms_tail_recorded[0] = false; ctx->ms_tail_recorded[1] = false;
  ctx->tb_parity = 0; ctx->tb_used[0] = false; ctx->tb_used[1] = false;   // (all streams are idle here)
  if (ctx->d_work) SVO_HIP(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(TrackWork) * (size_t)ctx->work_cap * 2, ctx->stream));   // frame tags restart at 1 (the ctx stream is waited for below)
  if (!ctx->d_track || ctx->n_seq != nseq) {
    if (ctx->d_track) { hipFree(ctx->d_track); ctx->d_track = nullptr; }
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
  { const int rcq = svo_track_quiesce(ctx); if (rcq) return rcq; }
  const SvoGeom& g = ctx->g;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  uint8_t* dL = ctx->d_stage;
  uint8_t* dR = ctx->d_stage + (size_t)g.H * ctx->stage_pitch;
  int rc = svo_upload_image(ctx, grayL, strideL, 0);
  if (rc) return rc;
  if ((rc = svo_upload_image(ctx, grayR, strideR, 1))) return rc;
  // the frame's boxes go to HBM with the images (pinned staging: the copies are asynchronous, the caller's array may be
  // pageable); from there on a gated frame is device work only
  int32_t* h_box = reinterpret_cast<int32_t*>(ctx->h_pinned + ctx->pinned_bytes - 4096);   // the buffer's last page: reserved for this (svo_msa.hip stays below it)
  h_box[0] = n_boxes;
  if (n_boxes > 0) memcpy(h_box + 4, boxes, 16 * (size_t)n_boxes);
  SVO_HIP(ctx, hipMemcpyAsync(&st->n_boxes, h_box, 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_boxes > 0)
    SVO_HIP(ctx, hipMemcpyAsync(st->boxes, h_box + 4, 16 * (size_t)n_boxes, hipMemcpyHostToDevice, ctx->stream));
  const svo_boxes_dev bx{st->boxes, &st->n_boxes, SVO_MAX_BOXES};
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
  svo_track_result* d_res = reinterpret_cast<svo_track_result*>(ctx->d_scratch);
  rc = tail_enqueue(ctx, ctx->d_kp, ctx->d_desc, ctx->d_nkp, ctx->d_depth, ctx->max_kp, 1, 1, d_res, n_boxes > 0 ? &bx : nullptr);
  if (rc) return rc;
  SVO_HIP(ctx, hipMemcpyAsync(res, d_res, sizeof *res, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->track_frame++;
  if (res->n_pnp_inliers == -1) return svo_track_check_timeout(ctx);   // (the record is valid: the frame was tracked as a PnP failure)
  return SVO_OK;
}

// One time step of n_seq independent sequences: pair q is the next frame of sequence q.
extern "C" int svo_track_multi_step_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR,
                                        int stride, int n_seq, const svo_boxes_dev* boxes, svo_track_result* d_results) {
  if (!ctx || !d_grayL || !d_grayR || !d_results || stride < ctx->g.W) return SVO_E_INVALID;
  if (!ctx->d_track || n_seq != ctx->n_seq) return SVO_E_INVALID;   // svo_track_multi_reset(n_seq) first
  if (ctx->opt_depth_source != 0) {   // the many-sequence mode has the sparse matcher only
    ctx->last_error = "svo_track_multi_step_dev: depth_source must be 0";
    return SVO_E_INVALID;
  }
  hipSetDevice(ctx->device);
  { const int rcs = svo_shard_quiesce(ctx); if (rcs) return rcs; }   // (a sharded call of this context may still be in flight: staging sets, work records)
  int rc;
  if (ctx->opt_multi_pipeline) {
    // Pipelined steps (svo_set_option("multi_pipeline", 1)): the front end is stateless, so step t + 1's may run while
    // step t's tail is still busy - on its own stream, into the other of two private output sets.  Contract: between
    // consecutive steps nothing else is enqueued on this context (svo_sync and reading results are fine).
    const size_t K = ctx->max_kp, I = 2 * (size_t)n_seq;
    if (ctx->ms_cap < n_seq) {
      SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->stream_fe) SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_fe));
      for (int p = 0; p < 2; ++p) {
        if (ctx->ms_kp[p]) hipFree(ctx->ms_kp[p]);
        if (ctx->ms_desc[p]) hipFree(ctx->ms_desc[p]);
        if (ctx->ms_nkp[p]) hipFree(ctx->ms_nkp[p]);
        if (ctx->ms_depth[p]) hipFree(ctx->ms_depth[p]);
        ctx->ms_kp[p] = nullptr; ctx->ms_desc[p] = nullptr; ctx->ms_nkp[p] = nullptr; ctx->ms_depth[p] = nullptr;
        ctx->ms_cap = 0;   // a failed allocation below leaves null pointers and no capacity behind, never freed memory
        if (hipMalloc(reinterpret_cast<void**>(&ctx->ms_kp[p]), sizeof(svo_kp) * I * K) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx->ms_desc[p]), 32 * I * K) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx->ms_nkp[p]), sizeof(int32_t) * I) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx->ms_depth[p]), sizeof(float) * (size_t)n_seq * K) != hipSuccess) {
          (void)hipGetLastError();
          for (int q = 0; q < 2; ++q) {
            if (ctx->ms_kp[q]) hipFree(ctx->ms_kp[q]);
            if (ctx->ms_desc[q]) hipFree(ctx->ms_desc[q]);
            if (ctx->ms_nkp[q]) hipFree(ctx->ms_nkp[q]);
            if (ctx->ms_depth[q]) hipFree(ctx->ms_depth[q]);
            ctx->ms_kp[q] = nullptr; ctx->ms_desc[q] = nullptr; ctx->ms_nkp[q] = nullptr; ctx->ms_depth[q] = nullptr;
          }
          return SVO_E_NOMEM;
        }
        if (!ctx->ms_fe_done[p]) SVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ms_fe_done[p], hipEventDisableTiming));
        if (!ctx->ms_tail_done[p]) SVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ms_tail_done[p], hipEventDisableTiming));
        ctx->ms_tail_recorded[p] = false;
      }
      ctx->ms_cap = n_seq;
    }
    if ((rc = track_resources(ctx, 1, n_seq))) return rc;   // streams and events exist from here on
    if (!ctx->stream_fe) {   // (confining it to a share of the CUs only costs: 88 % -6 %, 75 % -8 %, 50 % -34 % with 64 sequences)
      int attempts = 0, percent = 0;
      const int rcp = svo_pick_stream(ctx, [](hipStream_t* q) { return svo_stream_create(q, -1); }, {ctx->stream, ctx->stream_idx}, &ctx->stream_fe,
                                      &attempts, &percent, {ctx->stream, ctx->stream_idx}, {});
      if (rcp) return rcp;
    }
    const int p = ctx->ms_parity;
    if (ctx->ms_tail_recorded[p]) {
      SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream_fe, ctx->ms_tail_done[p], 0));   // the tail that read this set two steps ago
    } else {
      SVO_HIP(ctx, hipEventRecord(ctx->ev_frontend, ctx->stream));                  // first use: after whatever came before
      SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream_fe, ctx->ev_frontend, 0));
    }
    svo_kp* kp0 = ctx->d_kp; uint8_t* desc0 = ctx->d_desc; int32_t* nkp0 = ctx->d_nkp; float* depth0 = ctx->d_depth;
    hipStream_t s_main = ctx->stream;
    ctx->d_kp = ctx->ms_kp[p]; ctx->d_desc = ctx->ms_desc[p]; ctx->d_nkp = ctx->ms_nkp[p]; ctx->d_depth = ctx->ms_depth[p];
    ctx->stream = ctx->stream_fe;
    rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, n_seq, 2 * n_seq);
    if (rc == SVO_OK) rc = svo_launch_stereo(ctx, d_grayL, d_grayR, stride, n_seq, &ctx->cam);
    ctx->stream = s_main;
    ctx->d_kp = kp0; ctx->d_desc = desc0; ctx->d_nkp = nkp0; ctx->d_depth = depth0;
    if (rc) return rc;
    SVO_HIP(ctx, hipEventRecord(ctx->ms_fe_done[p], ctx->stream_fe));
    SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ms_fe_done[p], 0));
    if ((rc = tail_enqueue(ctx, ctx->ms_kp[p], ctx->ms_desc[p], ctx->ms_nkp[p], ctx->ms_depth[p], ctx->max_kp, 1, n_seq, d_results, boxes))) return rc;
    SVO_HIP(ctx, hipEventRecord(ctx->ms_tail_done[p], ctx->stream));
    ctx->ms_tail_recorded[p] = true;
    ctx->ms_parity ^= 1;
    ctx->track_frame++;
    return SVO_OK;
  }
  rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, n_seq, 2 * n_seq);
  if (rc) return rc;
  if ((rc = svo_launch_stereo(ctx, d_grayL, d_grayR, stride, n_seq, &ctx->cam))) return rc;
  if ((rc = tail_enqueue(ctx, ctx->d_kp, ctx->d_desc, ctx->d_nkp, ctx->d_depth, ctx->max_kp, 1, n_seq, d_results, boxes))) return rc;
  ctx->track_frame++;
  return SVO_OK;
}

extern "C" int svo_track_batch_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR,
                                   int stride, int B, const svo_boxes_dev* boxes, svo_track_result* d_results) {
  if (!ctx || !d_grayL || !d_grayR || !d_results || B < 1 || stride < ctx->g.W) return SVO_E_INVALID;
  if (B > ctx->max_batch) return SVO_E_CAPACITY;
  if (!ctx->d_track || ctx->n_seq != 1) return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  { const int rcs = svo_shard_quiesce(ctx); if (rcs) return rcs; }   // (a sharded call of this context may still be in flight: staging sets, work records)
  int rc;
  if (ctx->opt_depth_source == 1) {
    // BASELINE configs[4] as a pipeline: the dense front end (ORB on the left images, ELAS maps, the reference's per-keypoint
    // lookups - src/Tracking.cc:225-228, src/frame.cc:122-164) runs on a stream of its own; svo_elas_batch_dev moves the call's
    // pairs through its GPU -> host -> GPU stages in chunks, and as soon as a chunk's last GPU phase is enqueued the hook
    // below hangs that chunk's depth lookups behind it and enqueues the ordered tail of its frames on the tail's streams - the
    // tail of chunk c runs while chunks c + 1 ... are still in the dense stage (whose host phases - support-point filter,
    // Delaunay - block this thread, not the GPU).  Every chunk uses its own half of the work records, like consecutive calls
    // of the sparse path.
    const size_t n = (size_t)ctx->g.W * ctx->g.H, K = ctx->max_kp;
    if ((rc = dense_reserve(ctx, B))) return rc;
    if ((rc = track_resources(ctx, B, 1))) return rc;
    if (!ctx->stream_dense) {   // the dense stage's stream: beside both chains of the tail
      // (confined: a hardware queue of its own - which may still sit behind the dispatch pipe of one of the chains or of the front
      // end's queue: picked by measurement either way)
      const int dev = ctx->device, pct = ctx->opt_dense_cu_percent;
      int attempts = 0, percent = 0;
      const int rcp = svo_pick_stream(ctx, [dev, pct](hipStream_t* q) { return pct < 100 ? svo_stream_create_masked(q, dev, pct) : svo_stream_create(q, 0); },
                                      {ctx->stream, ctx->stream_idx}, &ctx->stream_dense, &attempts, &percent,
                                      {ctx->stream, ctx->stream_idx, ctx->stream_fe_batch}, {ctx->stream_fe_batch});
      if (rcp) return rcp;
    }
    float* dD1 = ctx->d_dense;
    float* dD2 = dD1 + n * (size_t)ctx->dense_cap;
    int32_t* d_prod = reinterpret_cast<int32_t*>(dD2 + n * (size_t)ctx->dense_cap);
    if (ctx->h_prod_cap < B) {
      if (ctx->h_prod) { SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_dense)); hipHostFree(ctx->h_prod); ctx->h_prod = nullptr; ctx->h_prod_cap = 0; }
      if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_prod), sizeof(int32_t) * (size_t)B) != hipSuccess) { (void)hipGetLastError(); return SVO_E_NOMEM; }
      ctx->h_prod_cap = B;
    }
    svo_elas_params ep;
    svo_elas_default_params(0, &ep);
    for (int q = 0; q < 2; ++q)
      if (!ctx->tb_done[q]) SVO_HIP(ctx, hipEventCreateWithFlags(&ctx->tb_done[q], hipEventDisableTiming));
    // the dense stage writes the ctx's own result arrays and maps: everything the ctx stream holds (an earlier call's tail) first
    SVO_HIP(ctx, hipEventRecord(ctx->ev_frontend, ctx->stream));
    SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream_dense, ctx->ev_frontend, 0));
    struct Hook {
      svo_ctx* ctx; hipStream_t s_main; const svo_boxes_dev* boxes; svo_track_result* d_results; float* dD1; int32_t* d_prod;
      size_t n, K; int chunk;
      static int run(void* u, int f0, int b) {
        Hook& h = *static_cast<Hook*>(u);
        svo_ctx* ctx = h.ctx;
        hipStream_t sd = ctx->stream_dense;
        while ((int)ctx->ev_sub.size() <= h.chunk) {
          hipEvent_t e;
          if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return SVO_E_HIP;
          ctx->ev_sub.push_back(e);
        }
        if (hipMemcpyAsync(h.d_prod + f0, ctx->h_prod + f0, sizeof(int32_t) * (size_t)b, hipMemcpyHostToDevice, sd) != hipSuccess) return SVO_E_HIP;
        hipLaunchKernelGGL(k_tk_dense_depth, dim3((ctx->max_kp + 255) / 256, b), dim3(256), 0, sd, ctx->d_kp + f0 * h.K, ctx->d_nkp + f0,
                           h.dD1 + h.n * f0, ctx->g.W, ctx->cam.bf, ctx->d_uR + f0 * h.K, ctx->d_depth + f0 * h.K, ctx->max_kp, h.n, h.d_prod + f0);
        if (hipEventRecord(ctx->ev_sub[h.chunk], sd) != hipSuccess) return SVO_E_HIP;
        const int p = ctx->tb_parity;
        std::vector<hipEvent_t> wait(b, nullptr);
        wait[0] = ctx->ev_sub[h.chunk];
        svo_boxes_dev bj{nullptr, nullptr, 0};
        if (h.boxes && h.boxes->boxes && h.boxes->n) bj = svo_boxes_dev{h.boxes->boxes + (size_t)f0 * h.boxes->stride * 4, h.boxes->n + f0, h.boxes->stride};
        ctx->stream = h.s_main;          // (the dense stage runs with the ctx stream swapped for its own)
        int rc = tail_enqueue(ctx, ctx->d_kp + f0 * h.K, ctx->d_desc + f0 * h.K * 32, ctx->d_nkp + f0, ctx->d_depth + f0 * h.K, ctx->max_kp, b, 1,
                              h.d_results + f0, bj.boxes ? &bj : nullptr, wait.data(), nullptr, p,
                              ctx->tb_used[p] ? ctx->tb_done[p] : ctx->ev_frontend, true);
        if (rc == SVO_OK && hipEventRecord(ctx->tb_done[p], ctx->stream) != hipSuccess) rc = SVO_E_HIP;   // the pose chain is the last reader of this half
        ctx->stream = sd;
        if (rc) return rc;
        ctx->tb_used[p] = true;
        ctx->tb_parity ^= 1;
        ctx->track_frame += b;
        ++h.chunk;
        return SVO_OK;
      }
    } hook{ctx, ctx->stream, boxes, d_results, dD1, d_prod, n, K, 0};
    if (ctx->feed_pair_event) SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream_dense, ctx->feed_pair_event[B - 1], 0));   // host-fed: the call's uploads
    ctx->stream = ctx->stream_dense;
    ctx->hyp_two_launch = ctx->opt_dense_two_launch != 0;   // (the hook's tail_enqueue calls: 16 CUs per frame instead of 100 on ordinary frames)
    rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, B, B);   // left images only
    if (rc == SVO_OK)
      rc = svo_elas_batch_dev_hooked(ctx, d_grayL, d_grayR, stride, ctx->g.W, ctx->g.H, B, &ep, dD1, dD2, ctx->h_prod, &Hook::run, &hook);
    ctx->hyp_two_launch = false;
    ctx->stream = hook.s_main;
    return rc;
  } else if (ctx->opt_depth_source == 2) {
    // MSA maps, up to eight frames in flight (most of a solve is the host tree builds)
    const size_t n = (size_t)ctx->g.W * ctx->g.H;
    if ((rc = dense_reserve(ctx, B))) return rc;
    if (ctx->feed_pair_event) SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->feed_pair_event[B - 1], 0));   // host-fed: the call's uploads
    if ((rc = svo_launch_orb(ctx, d_grayL, d_grayR, stride, B, B))) return rc;   // left images only
    if ((rc = svo_msa_run_many_dev(ctx, d_grayL, d_grayR, stride, (size_t)ctx->g.H * stride, ctx->g.W, ctx->g.H, 48, B,
                                   ctx->d_dense)))
      return rc;
    hipLaunchKernelGGL(k_tk_dense_depth, dim3((ctx->max_kp + 255) / 256, B), dim3(256), 0, ctx->stream, ctx->d_kp, ctx->d_nkp,
                       ctx->d_dense, ctx->g.W, ctx->cam.bf, ctx->d_uR, ctx->d_depth, ctx->max_kp, n, (const int32_t*)nullptr);
  } else {
    // The front end runs in sub-batches on its own stream, so that only the first sub-batch stands in front of the
    // ordered tail: while the tail works through sub-batch j, the front end of j + 1 runs beside it.  Sub-batch j writes the
    // keypoints of its LEFT images to frame slots f0 .. f0 + b - 1 (what the tail reads); its right images use the slots
    // behind them, which the next sub-batch's left images overwrite later on the same stream.
    rc = track_resources(ctx, B, 1);
    if (rc) return rc;
    const int SUB = 32, nsub = (B + SUB - 1) / SUB;
    // The front end of a batched call runs beside the ordered tail, and its kernels are large enough to fill every CU: the
    // tail's small dependent kernels (100 single-wave RANSAC workgroups that want a CU's float64 pipe each) then queue for
    // slots and run at a fraction of their speed - 7 us per frame on average (tools/option_sweep.py: 13.2 k frames/s with the
    // front end on all CUs, 14.1 k on a quarter of them, 14.4 k on 32 CUs; stream priorities did not change that).  So the
    // batched tracker's front-end stream is confined to a share of the compute units (default an eighth: four CUs of every XCD): the
    // front end needs ~7 us per pair on the whole chip against the tail's ~70 us per frame - an eighth of the chip keeps up.
    { const int rcf = svo_track_fe_batch_stream(ctx); if (rcf) return rcf; }
    while ((int)ctx->ev_sub.size() < nsub) {
      hipEvent_t e;
      SVO_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->ev_sub.push_back(e);
    }
    // Output sets alternate between calls: this call's front end writes set p while the tail of the previous call still reads
    // the other one - it only has to wait for the tail of the call BEFORE that (first use of a set: for whatever the ctx
    // stream holds).  The front end's working set (pyramids, corner lists) is shared: stream_fe runs the calls' sub-batches
    // one after the other anyway.
    const int p = ctx->tb_parity;
    if (p == 1 && !ctx->tb_kp) {
      const size_t I = (size_t)ctx->max_images, Kk = (size_t)ctx->max_kp, Bm = (size_t)ctx->max_batch;
      if (hipMalloc(reinterpret_cast<void**>(&ctx->tb_kp), sizeof(svo_kp) * I * Kk) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&ctx->tb_desc), 32 * I * Kk) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&ctx->tb_nkp), 4 * I) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&ctx->tb_uR), 4 * Bm * Kk) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&ctx->tb_depth), 4 * Bm * Kk) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&ctx->tb_sad), 4 * Bm * Kk) != hipSuccess) {
        (void)hipGetLastError();
        track_batch_sets_free(ctx);
        return SVO_E_NOMEM;
      }
    }
    for (int q = 0; q < 2; ++q)
      if (!ctx->tb_done[q]) SVO_HIP(ctx, hipEventCreateWithFlags(&ctx->tb_done[q], hipEventDisableTiming));
    if (ctx->tb_used[p]) {
      SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream_fe_batch, ctx->tb_done[p], 0));
    } else {
      SVO_HIP(ctx, hipEventRecord(ctx->ev_frontend, ctx->stream));
      SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream_fe_batch, ctx->ev_frontend, 0));
    }
    std::vector<hipEvent_t> wait(B, nullptr);
    svo_kp* const own_kp = ctx->d_kp; uint8_t* const own_desc = ctx->d_desc; int32_t* const own_nkp = ctx->d_nkp;
    float* const own_uR = ctx->d_uR; float* const own_depth = ctx->d_depth; int32_t* const own_sad = ctx->d_sad;
    svo_kp* kp0 = p ? ctx->tb_kp : own_kp; uint8_t* desc0 = p ? ctx->tb_desc : own_desc; int32_t* nkp0 = p ? ctx->tb_nkp : own_nkp;
    float* uR0 = p ? ctx->tb_uR : own_uR; float* depth0 = p ? ctx->tb_depth : own_depth; int32_t* sad0 = p ? ctx->tb_sad : own_sad;
    hipStream_t s_main = ctx->stream;
    const size_t K = ctx->max_kp, img = (size_t)ctx->g.H * stride;
    ctx->stream = ctx->stream_fe_batch;
    for (int j = 0; j < nsub && rc == SVO_OK; ++j) {
      const int f0 = j * SUB, b = std::min(SUB, B - f0);
      ctx->d_kp = kp0 + f0 * K; ctx->d_desc = desc0 + f0 * K * 32; ctx->d_nkp = nkp0 + f0;
      ctx->d_uR = uR0 + f0 * K; ctx->d_depth = depth0 + f0 * K; ctx->d_sad = sad0 + f0 * K;
      if (ctx->feed_pair_event && hipStreamWaitEvent(ctx->stream_fe_batch, ctx->feed_pair_event[f0 + b - 1], 0) != hipSuccess) rc = SVO_E_HIP;   // host-fed: this sub-batch's uploads
      if (rc) break;
      rc = svo_launch_orb(ctx, d_grayL + f0 * img, d_grayR + f0 * img, stride, b, 2 * b);
      if (rc == SVO_OK) rc = svo_launch_stereo(ctx, d_grayL + f0 * img, d_grayR + f0 * img, stride, b, &ctx->cam);
      if (rc == SVO_OK && hipEventRecord(ctx->ev_sub[j], ctx->stream_fe_batch) != hipSuccess) rc = SVO_E_HIP;
      wait[f0] = ctx->ev_sub[j];
    }
    ctx->stream = s_main;
    ctx->d_kp = own_kp; ctx->d_desc = own_desc; ctx->d_nkp = own_nkp; ctx->d_uR = own_uR; ctx->d_depth = own_depth; ctx->d_sad = own_sad;
    if (rc) return rc;
    // (the index chain: its inputs arrive through the sub-batch events; its half of the work records was last read by the pose
    // chain of the call before the previous one - the same event the front end waited for)
    if ((rc = tail_enqueue(ctx, kp0, desc0, nkp0, depth0, ctx->max_kp, B, 1, d_results, boxes, wait.data(), nullptr, p,
                           ctx->tb_used[p] ? ctx->tb_done[p] : ctx->ev_frontend, true)))
      return rc;
    SVO_HIP(ctx, hipEventRecord(ctx->tb_done[p], ctx->stream));   // the pose chain is the last reader of this call's set
    ctx->tb_used[p] = true;
    ctx->tb_parity ^= 1;
    ctx->track_frame += B;
    return SVO_OK;
  }
  if ((rc = tail_enqueue(ctx, ctx->d_kp, ctx->d_desc, ctx->d_nkp, ctx->d_depth, ctx->max_kp, B, 1, d_results, boxes))) return rc;
  ctx->track_frame += B;
  return SVO_OK;
}

// The ordered tail alone, for front-end results produced elsewhere (another context, another GPU): frame f's keypoints
// at d_kp + f * kp_stride, descriptors at d_desc + f * kp_stride * 32, count d_n[f], depths d_depth + f * kp_stride.
extern "C" int svo_track_tail_dev(svo_ctx* ctx, const svo_kp* d_kp, const uint8_t* d_desc, const int32_t* d_n,
                                  const float* d_depth, int kp_stride, int B, const svo_boxes_dev* boxes,
                                  svo_track_result* d_results) {
  if (!ctx || !d_kp || !d_desc || !d_n || !d_depth || !d_results || B < 1 || kp_stride < 1) return SVO_E_INVALID;
  if (!ctx->d_track || ctx->n_seq != 1) return SVO_E_INVALID;   // svo_track_reset first
  hipSetDevice(ctx->device);
  { const int rcs = svo_shard_quiesce(ctx); if (rcs) return rcs; }   // (a sharded call of this context may still be in flight: staging sets, work records)
  int rc = tail_enqueue(ctx, d_kp, d_desc, d_n, d_depth, kp_stride, B, 1, d_results, boxes);
  if (rc) return rc;
  ctx->track_frame += B;
  return SVO_OK;
}

// svo_sync's look at the sticky flag of a single-sequence tracker (the stream is idle here): code 4 = a wait inside the pose chain
// ran into its bound (k_tp_tail_ord's frame part waiting for the samples of its own launch, or a kernel polling for a hand-over
// record with "pose_flag").  The frame concerned was treated as a PnP failure (record: n_pnp_inliers = -1) - nothing stale was
// consumed -; the context falls back to the two-launch pose chain, whose kernels never wait for each other, and says so.
int svo_track_check_timeout(svo_ctx* ctx) {
  if (!ctx->d_track || ctx->n_seq != 1 || ctx->timeout_reported) return SVO_OK;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  int32_t v = 0;
  SVO_HIP(ctx, svo_memcpy_sync(ctx, &v, &st->overflow, 4, hipMemcpyDeviceToHost));
  if (v != 4) return SVO_OK;
  ctx->timeout_reported = true;
  ctx->opt_tail_fused = 0;
  ctx->opt_pose_flag = 0;
  ctx->last_error = "tracker: a wait inside the pose chain timed out (sticky flag 4): the frame was treated as a PnP failure (n_pnp_inliers = -1); "
                    "the context now runs the two-launch pose chain (tail_fused = 0, pose_flag = 0)";
  return SVO_E_TIMEOUT;
}

// Sticky flag of the tracker (0 = fine).  1: a frame wanted more than 4096 live map points or a map point outlived the position
// table (2^20 ids) - results after that are not the reference's.  4: a bounded wait inside the pose chain timed out (above).
extern "C" int svo_track_overflowed(svo_ctx* ctx, int32_t* flag) {
  if (!ctx || !flag || !ctx->d_track) return SVO_E_INVALID;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int32_t any = 0;
  for (int q = 0; q < ctx->n_seq; ++q) {
    int32_t v = 0;
    SVO_HIP(ctx, svo_memcpy_sync(ctx, &v, &st[q].overflow, 4, hipMemcpyDeviceToHost));
    any |= v;
  }
  *flag = any;
  return SVO_OK;
}

// How the index chain's stream was chosen (svo_stream_burst, or track_index_stream with SVO_POOLED_QUEUES=1): out[0] = candidates
// tried (0: probe switched off), out[1] = "two chains together / one alone" on the pose and the index stream in per cent
// (~105 side by side, ~200: the two chains of the tail take turns instead of overlapping).
extern "C" int svo_debug_stream_probe(svo_ctx* ctx, int32_t out[2]) {
  if (!ctx || !out) return SVO_E_INVALID;
  out[0] = ctx->idx_probe_attempts; out[1] = ctx->idx_probe_spins;
  return SVO_OK;
}

// Do the front end's / the dense stage's queued grids hold the tail's two streams up?  (include/svo.h; svo_probe_block_percent)
extern "C" int svo_debug_stream_pipes(svo_ctx* ctx, int32_t out[4]) {
  if (!ctx || !out) return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  { const int rcq = svo_track_quiesce(ctx, true); if (rcq) return rcq; }
  hipStream_t fill[2] = {ctx->stream_fe_batch, ctx->stream_dense}, tail[2] = {ctx->stream, ctx->stream_idx};
  for (int f = 0; f < 2; ++f)
    for (int t = 0; t < 2; ++t)
      out[2 * f + t] = fill[f] && tail[t] && fill[f] != tail[t] ? svo_probe_block_percent(fill[f], tail[t]) : -1;
  return SVO_OK;
}

// How many RANSAC samples of the order-preserving EPnP (epnp_exact = 2) were handed to its sequential fallback since
// svo_track_reset (a zero or repeated singular value, 25 Jacobi sweeps): same results, ~10x the time of the frame concerned.
extern "C" int svo_track_epnp_fallbacks(svo_ctx* ctx, int64_t* count) {
  if (!ctx || !count || !ctx->d_track) return SVO_E_INVALID;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int64_t total = 0;
  for (int q = 0; q < ctx->n_seq; ++q) {
    int32_t v = 0;
    SVO_HIP(ctx, svo_memcpy_sync(ctx, &v, &st[q].epnp_fallbacks, 4, hipMemcpyDeviceToHost));
    total += v;
  }
  *count = total;
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

// Parity probe: map-point identities, RANSAC outcome and PnP pose of n frames of the last batched call (work slots first ..).
extern "C" int svo_debug_track_frames(svo_ctx* ctx, int first, int n, svo_track_debug* out) {
  if (!ctx || !out || !ctx->d_work || first < 0 || n < 1 || first + n > ctx->work_cap) return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<TrackWork> w((size_t)n);
  SVO_HIP(ctx, svo_memcpy_sync(ctx, w.data(), reinterpret_cast<TrackWork*>(ctx->d_work) + (size_t)ctx->work_last_half * ctx->work_cap + first,
                         sizeof(TrackWork) * (size_t)n, hipMemcpyDeviceToHost));
  for (int f = 0; f < n; ++f) {
    const TrackWork& q = w[f];
    svo_track_debug& o = out[f];
    for (int j = 0; j < TRK_MAXKP; ++j) { o.match_gid[j] = -1; o.new_gid[j] = q.new_gid[j]; }
    for (int e = 0; e < q.n_edges && e < TRK_MAXKP; ++e) o.match_gid[q.edge_kp[e]] = q.edge_gid[e];
    o.frame_id = q.frame_id;
    o.pnp_best = q.pnp_best; o.pnp_iterations = q.pnp_iterations; o.pnp_inliers = q.pnp_inliers; o.pnp_ok = q.pnp_ok;
    o.active_rows[0] = q.diag[0] & 0xffff; o.active_rows[1] = q.diag[1] & 0xffff;
    o.rounds[0] = q.diag[0] >> 16; o.rounds[1] = q.diag[1] >> 16;
    o.resolve_us = (int32_t)((q.rt[1] - q.rt[0]) / 100);   // s_memrealtime: 100 MHz
    for (int i = 0; i < 6; ++i) o.rt[i] = q.rt[i];
    memcpy(o.T_pnp, q.T_pnp, sizeof o.T_pnp);
  }
  return SVO_OK;
}

// Diagnostics: s_memtime stamps (shader clock) of k_ti_resolve's phases for work slot `slot` of the last call:
// begin, pass 1, pass 2, frame end, done.
extern "C" int svo_debug_track_stamps(svo_ctx* ctx, int slot, int64_t ts[8]) {
  if (!ctx || !ts || !ctx->d_work || slot < 0 || slot >= ctx->work_cap) return SVO_E_INVALID;
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TrackWork* w = reinterpret_cast<TrackWork*>(ctx->d_work) + (size_t)ctx->work_last_half * ctx->work_cap + slot;
  SVO_HIP(ctx, svo_memcpy_sync(ctx, ts, w->ts, sizeof(long long) * 8, hipMemcpyDeviceToHost));
  return SVO_OK;
}

// Diagnostics: s_memrealtime stamps (100 MHz) of work slot `slot` of the last batched call: k_ti_resolve start / end,
// k_tp_hyp start, k_tp_frame end - where the two chains of the tail wait for each other (tools/chain_times.py).
extern "C" int svo_debug_track_realtime(svo_ctx* ctx, int slot, int64_t rt[4]) {
  if (!ctx || !rt || !ctx->d_work || slot < 0 || slot >= ctx->work_cap) return SVO_E_INVALID;
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TrackWork* w = reinterpret_cast<TrackWork*>(ctx->d_work) + (size_t)ctx->work_last_half * ctx->work_cap + slot;
  SVO_HIP(ctx, svo_memcpy_sync(ctx, rt, w->rt, sizeof(long long) * 4, hipMemcpyDeviceToHost));   // (the first four stamps)
  return SVO_OK;
}

// --------------------------------------------------------------------------------------------
// ONE sequence over G GPUs in one process (BASELINE configs[3]): stereo pair k goes to context
// k mod G - the stateless front end (ORB on both images + sparse stereo) shards by pair with nothing to exchange - and
// the strict temporal chain (src/Tracking.cc:231-250) runs in frame order on context 0, which pulls every frame's
// ~34 KB of keypoints / descriptors / depths from where they were produced (device-to-device copies on its own
// stream, no collective, no host round trip).  d_grayL[g] / d_grayR[g]: the pairs of context g, on ITS device, in
// order of increasing k (local index k / G).  B frames per call, records into d_results (on context 0's device).
// Records are identical to svo_track_batch_dev on one context.  Does not synchronise.
struct ShardGather {
  // staging on the tail context's device: region g (frames g, g + G, ... of a call, `per` rows) holds what context g produced
  svo_kp* kp = nullptr; uint8_t* desc = nullptr; int32_t* n = nullptr; float* depth = nullptr;
  // TWO sets of staging rows (set p at row p * per * G): call c + 1 gathers into one while the tail of call c reads the other
  int per = 0, G = 0;
  std::vector<hipEvent_t> ev;         // front end (and staging copies) of context g finished; created ON context g's device
  std::vector<int> ev_dev;            //   (an event is recorded on a stream of its own device only) - that device
  hipEvent_t ev_prev = nullptr;       // what the tail context's stream held when the (first) call began
  hipStream_t gs = nullptr;           // the gather's own stream on the tail context's device (not the pose chain's)
  std::vector<hipEvent_t> ev_sub;     // sub-batch j of the latest call gathered (the index chain waits for it before the sub-batch's first frame)
  hipEvent_t ev_gathered = nullptr;   // the latest call's gather finished: the contexts' result buffers / the bounce buffer are free again
  hipEvent_t done[2] = {nullptr, nullptr};   // the pose chain finished with staging set / work half p
  bool used[2] = {false, false};
  int parity = 0;
  std::vector<svo_ctx*> producers;    // contexts whose shard_wait points at ev_gathered
  uint8_t* h_stage = nullptr;         // pinned bounce buffer for contexts whose device the tail's device cannot read directly
  size_t h_bytes = 0;
  std::vector<int> row_of_frame;
};
static std::vector<std::pair<svo_ctx*, ShardGather*>> g_gathers;   // per tail context, freed by svo_track_release
static std::mutex g_gathers_mu;                                      // contexts may live on different host threads

static ShardGather* shard_gather(svo_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_gathers_mu);
  for (auto& p : g_gathers)
    if (p.first == ctx) return p.second;
  g_gathers.emplace_back(ctx, new ShardGather());
  return g_gathers.back().second;
}
static void shard_gather_buffers_free(ShardGather* g) {
  if (g->kp) hipFree(g->kp);
  if (g->desc) hipFree(g->desc);
  if (g->n) hipFree(g->n);
  if (g->depth) hipFree(g->depth);
  g->kp = nullptr; g->desc = nullptr; g->n = nullptr; g->depth = nullptr;
  g->per = 0; g->G = 0;
}
static void shard_gather_free(svo_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_gathers_mu);
  for (auto& pr : g_gathers) {   // a context that goes away is nobody's producer any more
    auto& v = pr.second->producers;
    v.erase(std::remove(v.begin(), v.end(), ctx), v.end());
  }
  ctx->shard_wait = nullptr;
  for (size_t i = 0; i < g_gathers.size(); ++i)
    if (g_gathers[i].first == ctx) {
      ShardGather* g = g_gathers[i].second;
      for (svo_ctx* pc : g->producers) if (pc->shard_wait == g->ev_gathered) pc->shard_wait = nullptr;
      shard_gather_buffers_free(g);
      for (hipEvent_t e : g->ev) hipEventDestroy(e);
      if (g->ev_prev) hipEventDestroy(g->ev_prev);
      if (g->ev_gathered) hipEventDestroy(g->ev_gathered);
      for (hipEvent_t e : g->ev_sub) hipEventDestroy(e);
      for (int q = 0; q < 2; ++q) if (g->done[q]) hipEventDestroy(g->done[q]);
      if (g->gs) { hipStreamSynchronize(g->gs); hipStreamDestroy(g->gs); }
      if (g->h_stage) hipHostFree(g->h_stage);
      delete g;
      g_gathers.erase(g_gathers.begin() + i);
      return;
    }
}

// What a sharded call left in flight and may still read of this context: as a producer, the gather of its result arrays (and of
// the bounce buffer); as the tail context, the pose chains of the last two calls (staging sets, work records) - other entry
// points of the context wait for them (svo_track_quiesce).
int svo_shard_quiesce(svo_ctx* ctx) {
  if (ctx->shard_wait) {
    hipEvent_t e = ctx->shard_wait;
    ctx->shard_wait = nullptr;
    SVO_HIP(ctx, hipEventSynchronize(e));
  }
  ShardGather* sg = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_gathers_mu);
    for (auto& pr : g_gathers)
      if (pr.first == ctx) sg = pr.second;
  }
  if (sg)
    for (int q = 0; q < 2; ++q)
      if (sg->used[q] && sg->done[q]) SVO_HIP(ctx, hipEventSynchronize(sg->done[q]));
  return SVO_OK;
}

extern "C" int svo_track_sharded_dev(svo_ctx* const* ctxs, int G, const uint8_t* const* d_grayL, const uint8_t* const* d_grayR,
                                     int stride, int B, const svo_boxes_dev* boxes, svo_track_result* d_results) {
  if (!ctxs || G < 1 || !d_grayL || !d_grayR || B < 1 || !d_results) return SVO_E_INVALID;
  svo_ctx* c0 = ctxs[0];
  if (!c0 || !c0->d_track || c0->n_seq != 1) return SVO_E_INVALID;   // svo_track_reset(ctxs[0]) first
  const int K = c0->max_kp;
  for (int g = 0; g < G; ++g) {
    if (!ctxs[g] || !d_grayL[g] || !d_grayR[g] || ctxs[g]->max_kp != K || ctxs[g]->g.W != c0->g.W || ctxs[g]->g.H != c0->g.H)
      return SVO_E_INVALID;
    if ((B - g + G - 1) / G > ctxs[g]->max_batch) return SVO_E_CAPACITY;
  }
  ShardGather* sg = shard_gather(c0);
  SVO_HIP(c0, hipSetDevice(c0->device));
  { const int rcf = svo_track_fe_batch_stream(c0); if (rcf) return rcf; }
  if (!sg->gs) {   // the gather's copies feed the tail sub-batch by sub-batch: beside its chains, and never behind the front end's grids
    int attempts = 0, percent = 0;
    const int rcp = svo_pick_stream(c0, [](hipStream_t* q) { return svo_stream_create(q, 0); }, {c0->stream, c0->stream_idx, c0->stream_fe_batch},
                                    &sg->gs, &attempts, &percent, {}, {c0->stream_fe_batch});
    if (rcp) return rcp;
  }
  if (!sg->ev_gathered) SVO_HIP(c0, hipEventCreateWithFlags(&sg->ev_gathered, hipEventDisableTiming));
  for (int q = 0; q < 2; ++q)
    if (!sg->done[q]) SVO_HIP(c0, hipEventCreateWithFlags(&sg->done[q], hipEventDisableTiming));
  const int per = (B + G - 1) / G;
  const size_t wk = sizeof(svo_kp) * (size_t)K, wd = 32 * (size_t)K, wf = 4 * (size_t)K;
  if (sg->per < per || sg->G < G) {
    SVO_HIP(c0, hipStreamSynchronize(sg->gs));
    if (c0->stream_idx) SVO_HIP(c0, hipStreamSynchronize(c0->stream_idx));
    SVO_HIP(c0, hipStreamSynchronize(c0->stream));
    shard_gather_buffers_free(sg);
    const size_t rows = (size_t)per * G * 2;
    if (hipMalloc(reinterpret_cast<void**>(&sg->kp), wk * rows) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&sg->desc), wd * rows) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&sg->n), 4 * rows) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&sg->depth), wf * rows) != hipSuccess) {
      (void)hipGetLastError();
      shard_gather_buffers_free(sg);
      return SVO_E_NOMEM;
    }
    sg->per = per; sg->G = G;
    sg->used[0] = sg->used[1] = false;
  }
  const int rper = sg->per;   // rows per region (may be larger than this call needs)
  const int G0 = sg->G;       // regions per set
  for (int g = 0; g < G; ++g) {
    if (g < (int)sg->ev.size() && sg->ev_dev[g] == ctxs[g]->device) continue;
    hipSetDevice(ctxs[g]->device);
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { hipSetDevice(c0->device); return SVO_E_HIP; }
    if (g < (int)sg->ev.size()) { hipEventDestroy(sg->ev[g]); sg->ev[g] = e; sg->ev_dev[g] = ctxs[g]->device; }
    else { sg->ev.push_back(e); sg->ev_dev.push_back(ctxs[g]->device); }
  }
  hipSetDevice(c0->device);
  if (!sg->ev_prev) SVO_HIP(c0, hipEventCreateWithFlags(&sg->ev_prev, hipEventDisableTiming));
  // How the tail's device reaches each producer's results: the same device, a direct peer read over xGMI (checked on every
  // call - the contexts of a call may change), or a bounce through pinned host memory when the platform offers no peer access.
  std::vector<int> direct(G, 1);
  bool need_stage = false;
  for (int g = 1; g < G; ++g) {
    if (ctxs[g]->device == c0->device) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, c0->device, ctxs[g]->device) != hipSuccess) { can = 0; (void)hipGetLastError(); }
    if (can) {
      const hipError_t e = hipDeviceEnablePeerAccess(ctxs[g]->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
      (void)hipGetLastError();
    }
    direct[g] = can;
    need_stage = need_stage || !can;
  }
  if (c0->opt_shard_force_staged) {   // test switch (SVO_SHARD_FORCE_STAGED at svo_create, or the option): the bounce path even where a direct read exists
    for (int g = 1; g < G; ++g) direct[g] = 0;
    need_stage = G > 1;
  }
  const size_t region_bytes = (wk + wd + wf + 4) * (size_t)rper;
  if (need_stage && sg->h_bytes < region_bytes * G) {
    SVO_HIP(c0, hipStreamSynchronize(sg->gs));
    SVO_HIP(c0, hipStreamSynchronize(c0->stream));
    if (sg->h_stage) hipHostFree(sg->h_stage);
    sg->h_stage = nullptr; sg->h_bytes = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&sg->h_stage), region_bytes * G, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return SVO_E_NOMEM; }   // (portable: every producer's device copies into it)
    sg->h_bytes = region_bytes * G;
  }
  // The pipeline over consecutive calls: the front ends of call c + 1 run while the tail of call c is in flight.  Nothing of a
  // front end sits on the pose chain's stream: context g > 0 uses its own stream (its own GPU), context 0 its front-end
  // stream (a share of the CUs, "fe_cu_percent", like svo_track_batch_dev); the gather runs on a stream of its own into the
  // staging set the tail of call c does not read.  A front end waits only for the previous call's GATHER (the last reader of
  // the context's result buffers and of the bounce buffer), the gather for the pose chain of call c - 1 (the last reader of
  // its staging set), the index chain for the gather.
  const int p = sg->parity;
  const bool first_call = !(sg->used[0] || sg->used[1]);
  if (first_call) SVO_HIP(c0, hipEventRecord(sg->ev_prev, c0->stream));   // after whatever the tail context's stream holds
  // A front end on the TAIL's device (context 0's; in tests and one-GPU runs every context's) runs on the context's CU-confined
  // front-end stream ("fe_cu_percent"), like svo_track_batch_dev's: on all CUs at the main stream's high priority it took the
  // tail's CUs (measured with two contexts on one GPU: 5.4 k frames/s against 8.1 k).  Every context works through its pairs in
  // sub-batches (its left images' results end up contiguous, row i = its pair i; a sub-batch's right images use the slots
  // behind it, which the next sub-batch overwrites - as in svo_track_batch_dev), so that the tail can start on the first
  // sub-batch while the rest of the call's front end still runs.
  // sub-batch j of every context: its pairs sub_off[j] .. sub_off[j + 1]; 8, 8, 16, then 32 at a time - the tail of a call that
  // finds the chip idle starts after 8 pairs per context instead of 32 (the confined stream takes ~50 us per pair)
  std::vector<int> sub_off(1, 0);
  while (sub_off.back() < per) {
    const int j = (int)sub_off.size() - 1;
    sub_off.push_back(std::min(per, sub_off.back() + (j < 2 ? 8 : j == 2 ? 16 : 32)));
  }
  const int nsub = (int)sub_off.size() - 1;
  std::vector<char> confined(G, 0);
  for (int g = 0; g < G; ++g) {
    svo_ctx* c = ctxs[g];
    hipSetDevice(c->device);
    while ((int)c->ev_sub.size() < nsub) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { hipSetDevice(c0->device); return SVO_E_HIP; }
      c->ev_sub.push_back(e);
    }
    if (c->device == c0->device) confined[g] = 1;   // ... on context 0's front-end stream, one context after the other: two
    // confined streams kept that share of the CUs busy without a gap, and the RANSAC workgroups the dispatcher places there
    // waited for room (pose chain 111 -> 147 us per frame with two contexts on one GPU)
  }
  hipSetDevice(c0->device);
  while ((int)sg->ev_sub.size() < nsub) {
    hipEvent_t e;
    SVO_HIP(c0, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    sg->ev_sub.push_back(e);
  }
  const size_t img = (size_t)c0->g.H * stride;
  int rc = SVO_OK;
  // sub-batch by sub-batch, every context's share of it (contexts on the tail's device share ONE stream: all of context 0's
  // sub-batches in front of context 1's first would hold the tail up for the whole of context 0's front end)
  struct Own { svo_kp* kp; uint8_t* desc; int32_t* nkp; float* uR; float* depth; int32_t* sad; hipStream_t stream; };
  std::vector<Own> own(G);
  std::vector<hipStream_t> fsv(G, nullptr);
  for (int g = 0; g < G && rc == SVO_OK; ++g) {
    svo_ctx* c = ctxs[g];
    own[g] = Own{c->d_kp, c->d_desc, c->d_nkp, c->d_uR, c->d_depth, c->d_sad, c->stream};
    if ((B - g + G - 1) / G <= 0) continue;
    hipSetDevice(c->device);
    { const int rcq = svo_track_quiesce(c, false); if (rcq) { hipSetDevice(c0->device); return rcq; } }   // (a batched call of this context may still read its result arrays; its own earlier calls are ordered by the stream waits below)
    fsv[g] = confined[g] ? c0->stream_fe_batch : c->stream;
    bool waited = false;   // (contexts that share a stream: one wait is enough)
    for (int q = 0; q < g; ++q) waited = waited || fsv[q] == fsv[g];
    if (!waited && hipStreamWaitEvent(fsv[g], first_call ? sg->ev_prev : sg->ev_gathered, 0) != hipSuccess) rc = SVO_E_HIP;
  }
  for (int j = 0; j < nsub && rc == SVO_OK; ++j) {
    for (int g = 0; g < G && rc == SVO_OK; ++g) {
      const int nb = (B - g + G - 1) / G;
      const int f0 = sub_off[j], b = std::min(sub_off[j + 1], nb) - f0;
      if (b <= 0) continue;
      svo_ctx* c = ctxs[g];
      hipStream_t fs = fsv[g];
      hipSetDevice(c->device);
      c->stream = fs;
      c->d_kp = own[g].kp + (size_t)f0 * K; c->d_desc = own[g].desc + (size_t)f0 * wd; c->d_nkp = own[g].nkp + f0;
      c->d_uR = own[g].uR + (size_t)f0 * K; c->d_depth = own[g].depth + (size_t)f0 * K; c->d_sad = own[g].sad + (size_t)f0 * K;
      if (c->feed_pair_event && hipStreamWaitEvent(fs, c->feed_pair_event[f0 + b - 1], 0) != hipSuccess) rc = SVO_E_HIP;   // host-fed: this sub-batch's uploads (context g's copy stream, its device)
      if (rc == SVO_OK) rc = svo_launch_orb(c, d_grayL[g] + f0 * img, d_grayR[g] + f0 * img, stride, b, 2 * b);
      if (rc == SVO_OK) rc = svo_launch_stereo(c, d_grayL[g] + f0 * img, d_grayR[g] + f0 * img, stride, b, &c0->cam);
      if (rc == SVO_OK && !direct[g]) {
        // bounce, first half: this sub-batch's results into the context's region of the pinned buffer, on its own stream
        uint8_t* h = sg->h_stage + region_bytes * g;
        if (hipMemcpyAsync(h + wk * f0, c->d_kp, wk * b, hipMemcpyDeviceToHost, fs) != hipSuccess ||
            hipMemcpyAsync(h + wk * rper + wd * f0, c->d_desc, wd * b, hipMemcpyDeviceToHost, fs) != hipSuccess ||
            hipMemcpyAsync(h + (wk + wd) * rper + wf * f0, c->d_depth, wf * b, hipMemcpyDeviceToHost, fs) != hipSuccess ||
            hipMemcpyAsync(h + (wk + wd + wf) * rper + 4 * (size_t)f0, c->d_nkp, 4 * (size_t)b, hipMemcpyDeviceToHost, fs) != hipSuccess)
          rc = SVO_E_HIP;
      }
      if (rc == SVO_OK && hipEventRecord(c->ev_sub[j], fs) != hipSuccess) rc = SVO_E_HIP;
      c->stream = own[g].stream;
      c->d_kp = own[g].kp; c->d_desc = own[g].desc; c->d_nkp = own[g].nkp; c->d_uR = own[g].uR; c->d_depth = own[g].depth; c->d_sad = own[g].sad;
    }
  }
  hipSetDevice(c0->device);
  if (rc) { if (rc == SVO_E_HIP) c0->last_error = std::string("svo_track_sharded_dev: ") + hipGetErrorString(hipGetLastError()); return rc; }
  // gather on its own stream, sub-batch by sub-batch: rows of region g of staging set p <- context g's results
  const size_t set0 = (size_t)p * rper * G0;
  if (sg->used[p]) SVO_HIP(c0, hipStreamWaitEvent(sg->gs, sg->done[p], 0));
  std::vector<hipEvent_t> wait(B, nullptr);
  for (int j = 0; j < nsub; ++j) {
    bool any = false;
    for (int g = 0; g < G; ++g) {
      const int nb = (B - g + G - 1) / G;
      const int f0 = sub_off[j], b = std::min(sub_off[j + 1], nb) - f0;
      if (b <= 0) continue;
      any = true;
      svo_ctx* c = ctxs[g];
      SVO_HIP(c0, hipStreamWaitEvent(sg->gs, c->ev_sub[j], 0));
      const size_t r0 = set0 + (size_t)g * rper + f0;
      if (!direct[g]) {
        const uint8_t* h = sg->h_stage + region_bytes * g;
        SVO_HIP(c0, hipMemcpyAsync(sg->kp + r0 * K, h + wk * f0, wk * b, hipMemcpyHostToDevice, sg->gs));
        SVO_HIP(c0, hipMemcpyAsync(sg->desc + r0 * wd, h + wk * rper + wd * f0, wd * b, hipMemcpyHostToDevice, sg->gs));
        SVO_HIP(c0, hipMemcpyAsync(sg->depth + r0 * K, h + (wk + wd) * rper + wf * f0, wf * b, hipMemcpyHostToDevice, sg->gs));
        SVO_HIP(c0, hipMemcpyAsync(sg->n + r0, h + (wk + wd + wf) * rper + 4 * (size_t)f0, 4 * (size_t)b, hipMemcpyHostToDevice, sg->gs));
        continue;
      }
      const bool same = c->device == c0->device;
      auto pull = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return same ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, sg->gs)
                    : hipMemcpyPeerAsync(dst, c0->device, src, c->device, bytes, sg->gs);
      };
      SVO_HIP(c0, pull(sg->kp + r0 * K, c->d_kp + (size_t)f0 * K, wk * b));
      SVO_HIP(c0, pull(sg->desc + r0 * wd, c->d_desc + (size_t)f0 * wd, wd * b));
      SVO_HIP(c0, pull(sg->depth + r0 * K, c->d_depth + (size_t)f0 * K, wf * b));
      SVO_HIP(c0, pull(sg->n + r0, c->d_nkp + f0, 4 * (size_t)b));
    }
    if (!any) break;
    SVO_HIP(c0, hipEventRecord(sg->ev_sub[j], sg->gs));
    if (sub_off[j] * G < B) wait[(size_t)sub_off[j] * G] = sg->ev_sub[j];   // frames sub_off[j] G .. of the call need sub-batch j of every context
  }
  SVO_HIP(c0, hipEventRecord(sg->ev_gathered, sg->gs));
  // the ordered tail reads frame k = g + G i from row g * rper + i of set p; its index chain starts when the gather is done
  // (which came after the pose chain that last read this set and this half of the work records)
  sg->row_of_frame.resize(B);
  for (int k = 0; k < B; ++k) sg->row_of_frame[k] = (int)set0 + (k % G) * rper + k / G;
  rc = tail_enqueue(c0, sg->kp, sg->desc, sg->n, sg->depth, K, B, 1, d_results, boxes, wait.data(), sg->row_of_frame.data(), p,
                    nullptr, true);
  if (rc) return rc;
  SVO_HIP(c0, hipEventRecord(sg->done[p], c0->stream));   // the pose chain is the last reader of this set
  for (int g = 0; g < G; ++g) {
    ctxs[g]->shard_wait = sg->ev_gathered;
    if (std::find(sg->producers.begin(), sg->producers.end(), ctxs[g]) == sg->producers.end()) sg->producers.push_back(ctxs[g]);
  }
  sg->used[p] = true;
  sg->parity ^= 1;
  c0->track_frame += B;
  return SVO_OK;
}

// Parity probe: cv::solvePnPRansac's outcome for the frame just tracked (winning sample, consensus, samples visited)
// and the pose it handed to PoseOptimization (before the CV_32F rounding), row-major 4x4.
extern "C" int svo_debug_track_pnp(svo_ctx* ctx, svo_pnp_stats* stats, double T_pnp[16]) {
  if (!ctx || !ctx->d_track) return SVO_E_INVALID;
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (stats) SVO_HIP(ctx, svo_memcpy_sync(ctx, stats, &st->pnp, sizeof *stats, hipMemcpyDeviceToHost));
  if (T_pnp) {
    PnpHyp h;
    svo_pnp_stats s;
    SVO_HIP(ctx, svo_memcpy_sync(ctx, &s, &st->pnp, sizeof s, hipMemcpyDeviceToHost));
    if (s.best_hypothesis < 0) return SVO_E_INVALID;
    SVO_HIP(ctx, svo_memcpy_sync(ctx, &h, &st->hyp[s.best_hypothesis], sizeof h, hipMemcpyDeviceToHost));
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T_pnp[4 * r + c] = h.R[3 * r + c]; T_pnp[4 * r + 3] = h.t[r]; }
    T_pnp[12] = 0; T_pnp[13] = 0; T_pnp[14] = 0; T_pnp[15] = 1;
  }
  return SVO_OK;
}

// Diagnostics: s_memtime stamps of the pose chain for the frame just tracked: [0] k_tp_hyp start, [1] correspondences
// gathered, [2] EPnP start, [3] EPnP done, [4] consensus counted; [8] k_tp_frame start, [9] RANSAC rule applied,
// [10] PoseOptimization done, [11] record written.
extern "C" int svo_debug_track_pose_stamps(svo_ctx* ctx, int64_t ts[16]) {
  if (!ctx || !ts || !ctx->d_track) return SVO_E_INVALID;
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TrackState* st = reinterpret_cast<TrackState*>(ctx->d_track);
  SVO_HIP(ctx, svo_memcpy_sync(ctx, ts, st->pose_ts, sizeof(long long) * 16, hipMemcpyDeviceToHost));
  return SVO_OK;
}
