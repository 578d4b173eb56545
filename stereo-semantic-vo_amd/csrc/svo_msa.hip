// svo_msa.hip - stages of the reference's MSA dense stereo built so far.
//
//  svo_ctmf : Thirdparty/MB/ctmf.h:7 `ctmf(src, dst, width, height, src_step, dst_step, r, channels, memsize)`,
//             the constant-time median filter MSA runs on the colour images (r = 1, 3 channels,
//             MSA.cpp:58-59) and on the winner-take-all disparity map (r = 2, 1 channel, MSA.cpp:1006).
//             What ctmf.c computes (ctmf.c:222-320): per channel the true median (rank 2r^2+2r of the
//             (2r+1)^2 window) with the window clamped to the image - rows and columns are REPLICATED at the
//             borders (`MAX(0, ..)`/`MIN(m-1, ..)` and the `r`-fold first column), not zero-padded as its
//             comment says.  The stripe decomposition (`memsize`) is a cache optimisation of the CPU code
//             and does not change results; it has no counterpart here.
//             GPU: one thread per (pixel, channel); the median is found by bisection on the 8-bit value
//             range (8 counting passes over the window held in registers) - O(r^2) per pixel, which for the
//             radii MSA uses (1, 2) is 9 / 25 elements.
//  svo_msa_init : `MSA::init` (MSA.cpp:22-63) - gray conversion (:42-47), `gradient` (:65-76), `getCost` (:78-108:
//             the two cost volumes, cost = 0.11 * min(mean |dBGR|, 7) + 0.89 * min(|dgrad|, 2) in float64, stored
//             as float), `ctmf` r = 1 on both colour images and `gradient_after_ctmf` (:110-139).  All per-pixel
//             work; the right-reference volume is the left one re-indexed (costR[j][d] = costL[j + d'][d'],
//             d' = min(d, m - 1 - j): the reference's "copy the previous disparity" chain in closed form).
//  svo_msa_tree_dp : `MSA::TreeDp` (MSA.cpp:929-990) - the two-pass cost aggregation over the spanning tree, level by
//             level (a node after all its children on the way up, after its parent on the way down), one thread
//             per (node, disparity); the children are added in the order the reference's adjacency chain yields
//             them, each `+=` rounded to float as there.
//  svo_msa_wta / svo_msa_lrcheck : `MSA::WTA` (:992-1006, first minimum + 5x5 ctmf) and `MSA::LRcheck` (:1027-1105).
// Not built: the host-side graph stages that produce the tree (build, Tarjan arborescence, region Kruskal, BFS
// orders; MSA.cpp:152-926) and therefore MSA::solve as a whole.
#include <math.h>

#include <algorithm>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "svo_internal.h"

namespace {

template <int R>
__global__ __launch_bounds__(256) void k_ctmf(const uint8_t* src, uint8_t* dst, int width, int height,
                                              int src_step, int dst_step, int cn) {
  const int xc = blockIdx.x * 256 + threadIdx.x;   // column * cn + channel
  const int y = blockIdx.y;
  if (xc >= width * cn) return;
  const int x = xc / cn, c = xc - x * cn;
  constexpr int N = (2 * R + 1) * (2 * R + 1);
  uint8_t v[N];
#pragma unroll
  for (int dy = -R; dy <= R; ++dy) {
    const uint8_t* row = src + (size_t)min(max(y + dy, 0), height - 1) * src_step;
#pragma unroll
    for (int dx = -R; dx <= R; ++dx) v[(dy + R) * (2 * R + 1) + dx + R] = row[min(max(x + dx, 0), width - 1) * cn + c];
  }
  // smallest value m with #{v <= m} > t, t = 2r^2 + 2r  (ctmf.c:271-312)
  int lo = 0, hi = 255;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int mid = (lo + hi) >> 1;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) cnt += v[k] <= mid;
    if (cnt > 2 * R * R + 2 * R) hi = mid; else lo = mid + 1;
  }
  dst[(size_t)y * dst_step + xc] = (uint8_t)lo;
}

__device__ __forceinline__ uint8_t msa_gray(const uint8_t* bgr) {   // MSA.cpp:44-45
  return (uint8_t)(int)(0.299 * bgr[2] + 0.587 * bgr[1] + 0.114 * bgr[0] + 0.5);
}
__global__ void k_msa_gray(const uint8_t* bgr, int n, uint8_t* gray) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) gray[i] = msa_gray(bgr + 3 * (size_t)i);
}
// central difference along x (`along_rows`) or y, one-sided at the borders, plus `offset` (MSA.cpp:65-76, 118-138)
__global__ void k_msa_gradient(const uint8_t* img, int n, int m, int along_rows, double offset, double* gra) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= m) return;
  const int t = i * m + j;
  const int len = along_rows ? m : n, pos = along_rows ? j : i, st = along_rows ? 1 : m;
  double g;
  if (pos == 0) g = (double)img[t + st] - (double)img[t];
  else if (pos == len - 1) g = (double)img[t] - (double)img[t - st];
  else g = ((double)img[t + st] - (double)img[t - st]) * 0.5;
  gra[t] = g + offset;
}
__global__ void k_msa_cost(const uint8_t* bgrL, const uint8_t* bgrR, const double* graL, const double* graR, int n, int m,
                           int disp, float* costL) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // (pixel, d), d fastest
  if (idx >= n * m * disp) return;
  const int t = idx / disp, d = idx - t * disp, i = t / m, j = t - i * m;
  const int o = (j - d >= 0) ? t - d : i * m;
  const double dif_gra = fmin(fabs(graL[t] - graR[o]), 2.0);
  double dif_col = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) dif_col += abs((int)bgrL[3 * (size_t)t + k] - (int)bgrR[3 * (size_t)o + k]);
  dif_col = fmin(dif_col / 3, 7.0);
  costL[idx] = (float)(0.11 * dif_col + (1 - 0.11) * dif_gra);
}
__global__ void k_msa_cost_right(const float* costL, int n, int m, int disp, float* costR) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * m * disp) return;
  const int t = idx / disp, d = idx - t * disp, j = t % m;
  const int dd = min(d, m - 1 - j);
  costR[idx] = costL[(size_t)(t + dd) * disp + dd];
}

// leaves -> root: nodes of one level, all their children are final
__global__ void k_msa_dp_up(const int32_t* nodes, int cnt, int D, const int32_t* child_ptr, const int32_t* child,
                            const uint8_t* child_c, const double* Exp, float* up) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= cnt * D) return;
  const int u = nodes[idx / D], d = idx % D;
  float acc = up[(size_t)u * D + d];
  for (int e = child_ptr[u]; e < child_ptr[u + 1]; ++e)
    acc = (float)((double)acc + Exp[child_c[e]] * (double)up[(size_t)child[e] * D + d]);
  up[(size_t)u * D + d] = acc;
}
// root -> leaves: nodes of one level, their parent is final
__global__ void k_msa_dp_down(const int32_t* nodes, int cnt, int D, const int32_t* parent, const uint8_t* parent_c,
                              const double* Exp, const float* up, float* A) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= cnt * D) return;
  const int v = nodes[idx / D], d = idx % D, u = parent[v];
  if (u < 0) { A[(size_t)v * D + d] = up[(size_t)v * D + d]; return; }
  const double w = Exp[parent_c[v]];
  A[(size_t)v * D + d] = (float)(w * (double)A[(size_t)u * D + d] + (1 - w * w) * (double)up[(size_t)v * D + d]);
}
// The same two sweeps for several frames at once (blockIdx.y = frame): the level sweep of one frame is a chain of
// ~2000 tiny dependent launches, so a batch shares them - level l of every frame in one launch.
struct MsaTreeTab {   // one tree on the device
  const int32_t *nodes, *level_ptr, *child_ptr, *child, *parent;
  const uint8_t *child_c, *parent_c;
  int32_t levels;
};
__global__ void k_msa_dp_up_many(const MsaTreeTab* tabs, int l, int D, const double* Exp, float* up_slab, size_t V) {
  const MsaTreeTab t = tabs[blockIdx.y];
  if (l >= t.levels) return;
  const int beg = t.level_ptr[l], cnt = t.level_ptr[l + 1] - beg;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= cnt * D) return;
  float* up = up_slab + (size_t)blockIdx.y * V;
  const int u = t.nodes[beg + idx / D], d = idx % D;
  float acc = up[(size_t)u * D + d];
  for (int e = t.child_ptr[u]; e < t.child_ptr[u + 1]; ++e)
    acc = (float)((double)acc + Exp[t.child_c[e]] * (double)up[(size_t)t.child[e] * D + d]);
  up[(size_t)u * D + d] = acc;
}
__global__ void k_msa_dp_down_many(const MsaTreeTab* tabs, int l, int D, const double* Exp, const float* up_slab, float* A_slab,
                                   size_t V) {
  const MsaTreeTab t = tabs[blockIdx.y];
  if (l >= t.levels) return;
  const int beg = t.level_ptr[l], cnt = t.level_ptr[l + 1] - beg;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= cnt * D) return;
  const float* up = up_slab + (size_t)blockIdx.y * V;
  float* A = A_slab + (size_t)blockIdx.y * V;
  const int v = t.nodes[beg + idx / D], d = idx % D, u = t.parent[v];
  if (u < 0) { A[(size_t)v * D + d] = up[(size_t)v * D + d]; return; }
  const double w = Exp[t.parent_c[v]];
  A[(size_t)v * D + d] = (float)(w * (double)A[(size_t)u * D + d] + (1 - w * w) * (double)up[(size_t)v * D + d]);
}

// ---- TreeDp as ONE launch per aggregation -------------------------------------------------------------------------
// The recurrences run per disparity: nothing couples A[.][d] to A[.][d'].  So one workgroup takes ONE disparity of one
// tree and walks all ~2000 levels itself - leaves to root, then root to leaves - with a workgroup barrier per level
// instead of a kernel launch per level (2 x 2000 launches per aggregation before).  Nodes are addressed by their
// position g in level order (children of a node are consecutive there, MsaBfsRec::cpos), the per-disparity values live in
// a transposed slab upT[d][g] (coalesced), and the values of the level just finished - all the next level needs -
// stay in LDS: a level costs one LDS round trip and a barrier, the HBM reads of the next level's own terms are issued a
// level ahead.  Arithmetic and its order are those of k_msa_dp_up / k_msa_dp_down (the child's weighted term
// Exp[c] * (double)up[child] is formed by the child, added by the parent in child order).
// (MsaBfsRec - first child's position | weight to parent + (children << 8) | parent's position | pixel - is declared in svo_internal.h:
// the host-side tree builder writes the records itself)
struct MsaBfsTab {   // one tree and the volumes of one aggregation over it
  const MsaBfsRec* rec; const int32_t* level_ptr; int32_t levels, N;
  const float* cost;   // [pixel][D] in
  float* upT;          // [D][position] work
  float* A;            // [pixel][D] out
};

// upT[d][g] = cost[node(g)][d]  (64 positions x D disparities per workgroup, transposed through LDS)
__global__ __launch_bounds__(256) void k_msa_bfs_gather(const MsaBfsTab* tabs, int D) {
  extern __shared__ float msa_tile[];   // 64 x (D + 1)
  const MsaBfsTab t = tabs[blockIdx.y];
  const int g0 = blockIdx.x * 64;
  if (g0 >= t.N) return;
  const float* cost = t.cost;
  float* upT = t.upT;
  for (int idx = threadIdx.x; idx < 64 * D; idx += 256) {
    const int gi = idx / D, d = idx - gi * D;
    if (g0 + gi < t.N) msa_tile[gi * (D + 1) + d] = cost[(size_t)t.rec[g0 + gi].node * D + d];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * D; idx += 256) {
    const int d = idx >> 6, gi = idx & 63;
    if (g0 + gi < t.N) upT[(size_t)d * t.N + g0 + gi] = msa_tile[gi * (D + 1) + d];
  }
}

template <int T>
__global__ __launch_bounds__(T) void k_msa_dp_bfs(const MsaBfsTab* tabs, int D, const double* Exp, int maxw, int maxl) {
  extern __shared__ double msa_lds[];   // Exp[256] | two level buffers of maxw doubles | level_ptr (maxl + 2 ints)
  const MsaBfsTab t = tabs[blockIdx.y];
  const int d = blockIdx.x, tid = threadIdx.x, L = t.levels;
  double* E = msa_lds;
  double* cur = msa_lds + 256;
  double* prev = cur + maxw;
  int* lp = reinterpret_cast<int*>(prev + maxw);
  for (int i = tid; i < 256; i += T) E[i] = Exp[i];
  for (int i = tid; i <= L; i += T) lp[i] = t.level_ptr[i];
  __syncthreads();
  float* up = t.upT + (size_t)d * t.N;
  float* A = t.A;
  const MsaBfsRec* rec = t.rec;
  MsaBfsRec r_n = {0, 0, -1, 0};
  float own_n = 0.f;
  // ---- leaves -> root ------------------------------------------------------------------------------------------
  if (L > 0 && tid < lp[L] - lp[L - 1]) { r_n = rec[lp[L - 1] + tid]; own_n = up[lp[L - 1] + tid]; }
  for (int l = L - 1; l >= 0; --l) {
    const int beg = lp[l], cbeg = lp[l + 1], cnt = cbeg - beg;
    MsaBfsRec r = r_n;
    float own = own_n;
    if (l > 0 && tid < beg - lp[l - 1]) { r_n = rec[lp[l - 1] + tid]; own_n = up[lp[l - 1] + tid]; }   // next level's terms: in flight now
    for (int i = tid; i < cnt; i += T) {
      if (i != tid) { r = rec[beg + i]; own = up[beg + i]; }
      float acc = own;
      const int c0 = r.cpos - cbeg, nch = (r.meta >> 8) & 255;
      for (int c = 0; c < nch; ++c) acc = (float)((double)acc + prev[c0 + c]);
      up[beg + i] = acc;
      cur[i] = E[r.meta & 255] * (double)acc;
    }
    __syncthreads();
    double* x = cur; cur = prev; prev = x;
  }
  // ---- root -> leaves ------------------------------------------------------------------------------------------
  float* curA = reinterpret_cast<float*>(cur);
  float* prevA = reinterpret_cast<float*>(prev);
  if (L > 0 && tid < lp[1] - lp[0]) { r_n = rec[lp[0] + tid]; own_n = up[lp[0] + tid]; }
  for (int l = 0; l < L; ++l) {
    const int beg = lp[l], cnt = lp[l + 1] - beg, pbeg = l > 0 ? lp[l - 1] : 0;
    MsaBfsRec r = r_n;
    float upv = own_n;
    if (l + 1 < L && tid < lp[l + 2] - lp[l + 1]) { r_n = rec[lp[l + 1] + tid]; own_n = up[lp[l + 1] + tid]; }
    for (int i = tid; i < cnt; i += T) {
      if (i != tid) { r = rec[beg + i]; upv = up[beg + i]; }
      float a = upv;
      if (r.ppos >= 0) {
        const double w = E[r.meta & 255];
        a = (float)(w * (double)prevA[r.ppos - pbeg] + (1 - w * w) * (double)upv);
      }
      A[(size_t)r.node * D + d] = a;
      curA[i] = a;
    }
    __syncthreads();
    float* x = curA; curA = prevA; prevA = x;
  }
}
__global__ void k_msa_argmin(const float* costA, int N, int D, uint8_t* disp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int k = 0;
  float best = costA[(size_t)i * D];
  for (int j = 1; j < D; ++j) { const float c = costA[(size_t)i * D + j]; if (c < best) { best = c; k = j; } }
  disp[i] = (uint8_t)k;
}
__global__ void k_msa_lrcheck(const uint8_t* d1, const uint8_t* d2, int n, int m, int D, float* cost, uint8_t* mask) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * m * D) return;
  const int t = idx / D, d = idx - t * D, j = t % m, dd = d1[t];
  const bool stable = j - dd >= 0 && dd > 0 && d2[t - dd] == dd;
  if (d == 0) mask[t] = stable ? 1 : 0;
  cost[idx] = stable ? (float)abs(d - dd) : 0.0f;
}

struct DevBuf {   // device allocations of one solve; reset() hands the same buffers out again in the same order
  struct Slot { void* p; size_t bytes; };
  std::vector<Slot> slots;
  size_t cursor = 0;
  ~DevBuf() { for (Slot& q : slots) hipFree(q.p); }
  void reset() { cursor = 0; }
  template <typename T> T* get(size_t count) {
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (cursor == slots.size()) slots.push_back(Slot{nullptr, 0});
    Slot& q = slots[cursor++];
    if (q.bytes < bytes) {
      if (q.p) hipFree(q.p);
      q.p = nullptr; q.bytes = 0;
      if (hipMalloc(&q.p, bytes) != hipSuccess) { q.p = nullptr; return nullptr; }
      q.bytes = bytes;
    }
    return reinterpret_cast<T*>(q.p);
  }
};

// the buffers of svo_msa_solve / the tracker's MSA mode live as long as the ctx (index 0: the ctx stream; 1..: the
// streams of svo_msa_run_many_dev) - a solve allocates ~0.5 GB at KITTI size and hipFree synchronises the device
struct MsaHostStore;   // host-side buffers of the batched solve (kept for the same reason: page faults)
struct MsaArenas { std::vector<DevBuf*> a; std::mutex m; MsaHostStore* host[2] = {nullptr, nullptr}; };
DevBuf& msa_arena(svo_ctx* ctx, int idx) {
  if (!ctx->msa_arenas) ctx->msa_arenas = new MsaArenas();
  MsaArenas* A = static_cast<MsaArenas*>(ctx->msa_arenas);
  std::lock_guard<std::mutex> lock(A->m);
  while ((int)A->a.size() <= idx) A->a.push_back(new DevBuf());
  A->a[idx]->reset();
  return *A->a[idx];
}

}  // namespace

// Level-order records of a tree for k_msa_dp_bfs.  nodes: pixels in level order; returns false when some node's children are
// not consecutive positions in child order (cannot happen for a breadth-first `seq`; the caller then keeps the
// level-by-level launches).
static bool msa_bfs_records(int N, const int32_t* nodes, const std::vector<int32_t>& level_ptr, const int32_t* child_ptr,
                            const int32_t* child, const int32_t* parent, const uint8_t* parent_c, std::vector<int32_t>& pos,
                            std::vector<MsaBfsRec>& rec, int* maxw) {
  pos.resize(N);
  for (int g = 0; g < N; ++g) pos[nodes[g]] = g;
  rec.resize(N);
  int cpos = 1;
  bool ok = true;
  for (int g = 0; g < N; ++g) {
    const int u = nodes[g], nch = child_ptr[u + 1] - child_ptr[u];
    for (int j = 0; j < nch && ok; ++j) ok = cpos + j < N && nodes[cpos + j] == child[child_ptr[u] + j];
    rec[g] = MsaBfsRec{cpos, (int32_t)parent_c[u] | (nch << 8), parent[u] >= 0 ? pos[parent[u]] : -1, u};
    if (nch > 255) ok = false;
    cpos += nch;
  }
  int w = 0;
  for (size_t l = 0; l + 1 < level_ptr.size(); ++l) w = std::max(w, level_ptr[l + 1] - level_ptr[l]);
  *maxw = w;
  return ok && cpos == N;
}

// k_msa_dp_bfs applicable?  (levels fit its LDS buffers, opt-in above 64 KB granted; SVO_MSA_LEVEL_LAUNCHES forces the old path)
constexpr int kMsaDpThreads = 512;
static size_t msa_dp_bfs_lds(int maxw, int maxl) { return sizeof(double) * (256 + 2 * (size_t)maxw) + sizeof(int) * ((size_t)maxl + 2); }
static bool msa_dp_bfs_ok(svo_ctx* ctx, int D, int maxw, int maxl) {
  const size_t lds = msa_dp_bfs_lds(maxw, maxl);
  if (getenv("SVO_MSA_LEVEL_LAUNCHES") || lds > 150 * 1024 || (size_t)64 * (D + 1) * sizeof(float) > 64 * 1024) return false;
  if (ctx->msa_lds_state == 0)
    ctx->msa_lds_state = hipFuncSetAttribute(reinterpret_cast<const void*>(k_msa_dp_bfs<kMsaDpThreads>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess ? 1 : -1;
  return ctx->msa_lds_state > 0 || lds <= 64 * 1024;
}
// One aggregation (both sweeps) of `ntrees` trees: two launches
static void msa_dp_bfs_launch(hipStream_t s, const MsaBfsTab* d_tabs, int ntrees, int N, int D, const double* d_Exp, int maxw, int maxl) {
  hipLaunchKernelGGL(k_msa_bfs_gather, dim3((unsigned)((N + 63) / 64), ntrees), dim3(256), (size_t)64 * (D + 1) * sizeof(float), s, d_tabs, D);
  hipLaunchKernelGGL(k_msa_dp_bfs<kMsaDpThreads>, dim3(D, ntrees), dim3(kMsaDpThreads), msa_dp_bfs_lds(maxw, maxl), s, d_tabs, D, d_Exp,
                     maxw, maxl);
}

extern "C" int svo_msa_tree_dp(svo_ctx* ctx, const float* cost, int N, int D, const int32_t* seq, const int32_t* child_ptr,
                               const int32_t* child, const uint8_t* child_c, int root, double o, float* costA) {
  if (!ctx) return SVO_E_INVALID;
  if (!cost || !seq || !child_ptr || !child || !child_c || !costA || N < 1 || D < 1 || D > 256 || root < 0 || root >= N ||
      !(o > 0) || seq[0] != root || child_ptr[0] != 0 || child_ptr[N] != N - 1) {
    ctx->last_error = "svo_msa_tree_dp: invalid argument (need a spanning tree in BFS order from `root`)";
    return SVO_E_INVALID;
  }
  // parents, edge weights towards the parent, depths (seq is a BFS order: parents come before children)
  std::vector<int32_t> parent(N, -2), depth(N, 0), level_ptr, nodes(N);
  std::vector<uint8_t> parent_c(N, 0);
  parent[root] = -1;
  int max_depth = 0;
  for (int k = 0; k < N; ++k) {
    const int u = seq[k];
    if (u < 0 || u >= N || parent[u] == -2) { ctx->last_error = "svo_msa_tree_dp: seq is not a BFS order of the tree"; return SVO_E_INVALID; }
    for (int e = child_ptr[u]; e < child_ptr[u + 1]; ++e) {
      const int v = child[e];
      if (v < 0 || v >= N || parent[v] != -2) { ctx->last_error = "svo_msa_tree_dp: not a tree"; return SVO_E_INVALID; }
      parent[v] = u; parent_c[v] = child_c[e]; depth[v] = depth[u] + 1;
      max_depth = std::max(max_depth, depth[v]);
    }
  }
  level_ptr.assign(max_depth + 2, 0);
  for (int v = 0; v < N; ++v) ++level_ptr[depth[v] + 1];
  for (int l = 0; l <= max_depth; ++l) level_ptr[l + 1] += level_ptr[l];
  {
    std::vector<int32_t> fill(level_ptr.begin(), level_ptr.end() - 1);
    for (int k = 0; k < N; ++k) nodes[fill[depth[seq[k]]]++] = seq[k];
  }
  double Exp[256];
  for (int i = 0; i <= 255; ++i) Exp[i] = exp(-i * 1.0 / o / 255);   // MSA::setExp, MSA.cpp:1126-1130
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf buf;
  const size_t V = (size_t)N * D;
  float* d_up = buf.get<float>(V); float* d_A = buf.get<float>(V);
  {
    // both sweeps in one launch (k_msa_dp_bfs) whenever the tree's levels fit its LDS buffers
    std::vector<MsaBfsRec> rec;
    std::vector<int32_t> pos;
    int maxw = 0;
    if (msa_bfs_records(N, nodes.data(), level_ptr, child_ptr, child, parent.data(), parent_c.data(), pos, rec, &maxw) &&
        msa_dp_bfs_ok(ctx, D, maxw, max_depth + 1)) {
      float* d_cost = buf.get<float>(V);
      MsaBfsRec* d_rec = buf.get<MsaBfsRec>(N);
      int32_t* d_lp = buf.get<int32_t>(level_ptr.size());
      MsaBfsTab* d_tab = buf.get<MsaBfsTab>(1);
      double* d_E = buf.get<double>(256);
      if (!d_up || !d_A || !d_cost || !d_rec || !d_lp || !d_tab || !d_E) { ctx->last_error = "svo_msa_tree_dp: hipMalloc"; return SVO_E_NOMEM; }
      hipStream_t s = ctx->stream;
      const MsaBfsTab tab{d_rec, d_lp, max_depth + 1, N, d_cost, d_up, d_A};
      SVO_HIP(ctx, hipMemcpyAsync(d_cost, cost, V * sizeof(float), hipMemcpyHostToDevice, s));
      SVO_HIP(ctx, hipMemcpyAsync(d_rec, rec.data(), (size_t)N * sizeof(MsaBfsRec), hipMemcpyHostToDevice, s));
      SVO_HIP(ctx, hipMemcpyAsync(d_lp, level_ptr.data(), level_ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
      SVO_HIP(ctx, hipMemcpyAsync(d_tab, &tab, sizeof tab, hipMemcpyHostToDevice, s));
      SVO_HIP(ctx, hipMemcpyAsync(d_E, Exp, sizeof Exp, hipMemcpyHostToDevice, s));
      {
        SvoTimer t(ctx, "k_msa_tree_dp");
        msa_dp_bfs_launch(s, d_tab, 1, N, D, d_E, maxw, max_depth + 1);
      }
      SVO_HIP(ctx, hipMemcpyAsync(costA, d_A, V * sizeof(float), hipMemcpyDeviceToHost, s));
      SVO_HIP(ctx, hipStreamSynchronize(s));
      SVO_HIP(ctx, hipGetLastError());
      return SVO_OK;
    }
  }
  int32_t* d_nodes = buf.get<int32_t>(N); int32_t* d_cptr = buf.get<int32_t>(N + 1); int32_t* d_child = buf.get<int32_t>(N);
  int32_t* d_parent = buf.get<int32_t>(N);
  uint8_t* d_cc = buf.get<uint8_t>(N); uint8_t* d_pc = buf.get<uint8_t>(N);
  double* d_Exp = buf.get<double>(256);
  if (!d_up || !d_A || !d_nodes || !d_cptr || !d_child || !d_parent || !d_cc || !d_pc || !d_Exp) { ctx->last_error = "svo_msa_tree_dp: hipMalloc"; return SVO_E_NOMEM; }
  hipStream_t s = ctx->stream;
  SVO_HIP(ctx, hipMemcpyAsync(d_up, cost, V * sizeof(float), hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(d_nodes, nodes.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(d_cptr, child_ptr, (N + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (N > 1) {
    SVO_HIP(ctx, hipMemcpyAsync(d_child, child, (N - 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(d_cc, child_c, (N - 1), hipMemcpyHostToDevice, s));
  }
  SVO_HIP(ctx, hipMemcpyAsync(d_parent, parent.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(d_pc, parent_c.data(), N, hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(d_Exp, Exp, sizeof Exp, hipMemcpyHostToDevice, s));
  {
    SvoTimer t(ctx, "k_msa_tree_dp");
    for (int l = max_depth; l >= 0; --l) {
      const int cnt = level_ptr[l + 1] - level_ptr[l];
      hipLaunchKernelGGL(k_msa_dp_up, dim3((unsigned)(((size_t)cnt * D + 255) / 256)), dim3(256), 0, s, d_nodes + level_ptr[l], cnt, D,
                         d_cptr, d_child, d_cc, d_Exp, d_up);
    }
    for (int l = 0; l <= max_depth; ++l) {
      const int cnt = level_ptr[l + 1] - level_ptr[l];
      hipLaunchKernelGGL(k_msa_dp_down, dim3((unsigned)(((size_t)cnt * D + 255) / 256)), dim3(256), 0, s, d_nodes + level_ptr[l], cnt, D,
                         d_parent, d_pc, d_Exp, d_up, d_A);
    }
  }
  SVO_HIP(ctx, hipMemcpyAsync(costA, d_A, V * sizeof(float), hipMemcpyDeviceToHost, s));
  SVO_HIP(ctx, hipStreamSynchronize(s));
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

extern "C" int svo_msa_wta(svo_ctx* ctx, const float* costA, int width, int height, int D, uint8_t* disparity) {
  if (!ctx) return SVO_E_INVALID;
  if (!costA || !disparity || width < 5 || height < 5 || D < 1 || D > 256) { ctx->last_error = "svo_msa_wta: invalid argument"; return SVO_E_INVALID; }
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf buf;
  const size_t N = (size_t)width * height;
  float* d_A = buf.get<float>(N * D); uint8_t* d_raw = buf.get<uint8_t>(N); uint8_t* d_med = buf.get<uint8_t>(N);
  if (!d_A || !d_raw || !d_med) { ctx->last_error = "svo_msa_wta: hipMalloc"; return SVO_E_NOMEM; }
  hipStream_t s = ctx->stream;
  SVO_HIP(ctx, hipMemcpyAsync(d_A, costA, N * D * sizeof(float), hipMemcpyHostToDevice, s));
  {
    SvoTimer t(ctx, "k_msa_wta");
    hipLaunchKernelGGL(k_msa_argmin, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_A, (int)N, D, d_raw);
    hipLaunchKernelGGL(k_ctmf<2>, dim3((width + 255) / 256, height), dim3(256), 0, s, d_raw, d_med, width, height, width, width, 1);
  }
  SVO_HIP(ctx, hipMemcpyAsync(disparity, d_med, N, hipMemcpyDeviceToHost, s));
  SVO_HIP(ctx, hipStreamSynchronize(s));
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

extern "C" int svo_msa_lrcheck(svo_ctx* ctx, const uint8_t* d1, const uint8_t* d2, int width, int height, int D, float* cost,
                               uint8_t* mask) {
  if (!ctx) return SVO_E_INVALID;
  if (!d1 || !d2 || !cost || !mask || width < 1 || height < 1 || D < 1 || D > 256) { ctx->last_error = "svo_msa_lrcheck: invalid argument"; return SVO_E_INVALID; }
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf buf;
  const size_t N = (size_t)width * height;
  uint8_t* a = buf.get<uint8_t>(N); uint8_t* b = buf.get<uint8_t>(N); uint8_t* mk = buf.get<uint8_t>(N);
  float* c = buf.get<float>(N * D);
  if (!a || !b || !mk || !c) { ctx->last_error = "svo_msa_lrcheck: hipMalloc"; return SVO_E_NOMEM; }
  hipStream_t s = ctx->stream;
  SVO_HIP(ctx, hipMemcpyAsync(a, d1, N, hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(b, d2, N, hipMemcpyHostToDevice, s));
  {
    SvoTimer t(ctx, "k_msa_lrcheck");
    hipLaunchKernelGGL(k_msa_lrcheck, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, s, a, b, height, width, D, c, mk);
  }
  SVO_HIP(ctx, hipMemcpyAsync(cost, c, N * D * sizeof(float), hipMemcpyDeviceToHost, s));
  SVO_HIP(ctx, hipMemcpyAsync(mask, mk, N, hipMemcpyDeviceToHost, s));
  SVO_HIP(ctx, hipStreamSynchronize(s));
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

// n rows x m columns (the reference's naming), BGR interleaved, `step` bytes per row.
extern "C" int svo_msa_init(svo_ctx* ctx, const uint8_t* bgrL, const uint8_t* bgrR, int width, int height, int step, int disp,
                            float* costL, float* costR, uint8_t* m_img3L, uint8_t* m_img3R, double* r_graL, double* c_graL,
                            double* r_graR, double* c_graR) {
  if (!ctx) return SVO_E_INVALID;
  const int n = height, m = width;
  if (!bgrL || !bgrR || !costL || !costR || !m_img3L || !m_img3R || !r_graL || !c_graL || !r_graR || !c_graR || n < 2 ||
      m < 2 || disp < 1 || disp > 256 || step < 3 * m || (size_t)n * m * disp > (size_t)1 << 30) {
    ctx->last_error = "svo_msa_init: invalid argument";
    return SVO_E_INVALID;
  }
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)n * m, V = N * disp;
  // one arena: 2 colour images, 2 median images, 2 gray, 2 gray-of-median, 6 gradient maps, 2 cost volumes
  const size_t bytes = 4 * 3 * N + 4 * N + 6 * N * sizeof(double) + 2 * V * sizeof(float) + 64;
  uint8_t* base = nullptr;
  SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&base), bytes));
  double* g = reinterpret_cast<double*>(base);                 // graL graR r_graL c_graL r_graR c_graR
  float* cL = reinterpret_cast<float*>(g + 6 * N);
  float* cR = cL + V;
  uint8_t* img3[2] = {reinterpret_cast<uint8_t*>(cR + V), reinterpret_cast<uint8_t*>(cR + V) + 3 * N};
  uint8_t* med3[2] = {img3[1] + 3 * N, img3[1] + 6 * N};
  uint8_t* gray[2] = {med3[1] + 3 * N, med3[1] + 4 * N};
  uint8_t* mgray[2] = {gray[1] + N, gray[1] + 2 * N};
  hipStream_t s = ctx->stream;
  int rc = SVO_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == SVO_OK) { rc = SVO_E_HIP; ctx->last_error = hipGetErrorString(e); } };
  chk(hipMemcpy2DAsync(img3[0], 3 * (size_t)m, bgrL, step, 3 * (size_t)m, n, hipMemcpyHostToDevice, s));
  chk(hipMemcpy2DAsync(img3[1], 3 * (size_t)m, bgrR, step, 3 * (size_t)m, n, hipMemcpyHostToDevice, s));
  if (rc == SVO_OK) {
    const dim3 px((m + 255) / 256, n);
    const unsigned nbN = (unsigned)((N + 255) / 256), nbV = (unsigned)((V + 255) / 256);
    for (int side = 0; side < 2; ++side) {
      SvoTimer t(ctx, "k_msa_gray_gradient");
      hipLaunchKernelGGL(k_msa_gray, dim3(nbN), dim3(256), 0, s, img3[side], (int)N, gray[side]);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, gray[side], n, m, 1, 127.5, g + side * N);
    }
    {
      SvoTimer t(ctx, "k_msa_cost");
      hipLaunchKernelGGL(k_msa_cost, dim3(nbV), dim3(256), 0, s, img3[0], img3[1], g, g + N, n, m, disp, cL);
      hipLaunchKernelGGL(k_msa_cost_right, dim3(nbV), dim3(256), 0, s, cL, n, m, disp, cR);
    }
    for (int side = 0; side < 2; ++side) {
      {
        SvoTimer t(ctx, "k_ctmf");
        hipLaunchKernelGGL(k_ctmf<1>, dim3((3 * m + 255) / 256, n), dim3(256), 0, s, img3[side], med3[side], m, n, 3 * m, 3 * m, 3);
      }
      SvoTimer t(ctx, "k_msa_gray_gradient");
      hipLaunchKernelGGL(k_msa_gray, dim3(nbN), dim3(256), 0, s, med3[side], (int)N, mgray[side]);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, mgray[side], n, m, 1, 0.0, g + (2 + 2 * side) * N);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, mgray[side], n, m, 0, 0.0, g + (3 + 2 * side) * N);
    }
    chk(hipMemcpyAsync(costL, cL, V * sizeof(float), hipMemcpyDeviceToHost, s));
    chk(hipMemcpyAsync(costR, cR, V * sizeof(float), hipMemcpyDeviceToHost, s));
    chk(hipMemcpyAsync(m_img3L, med3[0], 3 * N, hipMemcpyDeviceToHost, s));
    chk(hipMemcpyAsync(m_img3R, med3[1], 3 * N, hipMemcpyDeviceToHost, s));
    double* outs[4] = {r_graL, c_graL, r_graR, c_graR};
    for (int k = 0; k < 4; ++k) chk(hipMemcpyAsync(outs[k], g + (2 + k) * N, N * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  chk(hipStreamSynchronize(s));
  chk(hipGetLastError());
  hipFree(base);
  return rc;
}

extern "C" int svo_ctmf(svo_ctx* ctx, const uint8_t* src, uint8_t* dst, int width, int height, int src_step,
                        int dst_step, int r, int channels) {
  if (!ctx) return SVO_E_INVALID;
  if (!src || !dst || width < 1 || height < 1 || channels < 1 || channels > 4 || r < 1 || r > 3 ||
      src_step < width * channels || dst_step < width * channels) {
    ctx->last_error = "svo_ctmf: invalid argument (radius 1..3, 1..4 channels)";
    return SVO_E_INVALID;
  }
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sbytes = (size_t)src_step * height, dbytes = (size_t)dst_step * height;
  uint8_t *d_src = nullptr, *d_dst = nullptr;
  SVO_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d_src), sbytes));
  if (hipMalloc(reinterpret_cast<void**>(&d_dst), dbytes) != hipSuccess) { hipFree(d_src); ctx->last_error = "svo_ctmf: hipMalloc"; return SVO_E_HIP; }
  int rc = SVO_OK;
  hipStream_t s = ctx->stream;
  const dim3 grid((width * channels + 255) / 256, height);
  if (hipMemcpyAsync(d_src, src, sbytes, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_dst, dst, dbytes, hipMemcpyHostToDevice, s) != hipSuccess) rc = SVO_E_HIP;   // keep dst padding
  if (rc == SVO_OK) {
    SvoTimer t(ctx, "k_ctmf");
    if (r == 1) hipLaunchKernelGGL(k_ctmf<1>, grid, dim3(256), 0, s, d_src, d_dst, width, height, src_step, dst_step, channels);
    else if (r == 2) hipLaunchKernelGGL(k_ctmf<2>, grid, dim3(256), 0, s, d_src, d_dst, width, height, src_step, dst_step, channels);
    else hipLaunchKernelGGL(k_ctmf<3>, grid, dim3(256), 0, s, d_src, d_dst, width, height, src_step, dst_step, channels);
  }
  if (rc == SVO_OK && (hipMemcpyAsync(dst, d_dst, dbytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
                       hipStreamSynchronize(s) != hipSuccess)) rc = SVO_E_HIP;
  hipFree(d_src); hipFree(d_dst);
  if (rc) ctx->last_error = "svo_ctmf: HIP error";
  return rc;
}

// ------------------------------------------------------------------------------------------------------------
// MSA::solve (MSA.cpp:1132-1169) end to end: everything per pixel / per node on the GPU and resident there
// (the four cost volumes never leave HBM), the two aggregation trees on two host threads in between.
// ------------------------------------------------------------------------------------------------------------
namespace {

__global__ void k_msa_scale(const uint8_t* d, int n, int scale, uint8_t* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (uint8_t)(d[i] * scale);
}

}  // namespace
extern "C" int svo_msa_tree(const uint8_t* m_img3, const double* r_gra, const double* c_gra, int width, int height, int32_t* seq,
                            int32_t* child_ptr, int32_t* child, uint8_t* child_c, int32_t* root);
namespace {
struct HostTree {                       // what svo_msa_tree returns, plus what the level-by-level sweep needs
  std::vector<int32_t> seq, child_ptr, child, parent, nodes, level_ptr, depth_, fill_;
  std::vector<uint8_t> child_c, parent_c;
  int32_t root = -1;
  int rc = SVO_OK;
  // What k_msa_dp_bfs needs - the records by level-order position, the level boundaries, the widest level - in ONE pass over
  // the breadth-first sequence: `seq` IS the level order (a node's children sit at consecutive positions right after those of
  // the nodes before it), so a node's position, its parent's position and its depth are known when its parent is visited.
  // The by-pixel arrays of the level-launch fallback (nodes, parent, parent_c) are left to levels_full(); bfs_ok = false
  // sends the caller there (cannot happen for a breadth-first seq).  Round 4: the five passes of levels_full() were 12 % of a
  // tree's host time.
  void levels(int N) {
    rec.resize(N);
    ppos_.resize(N); pc_.resize(N); dpos_.resize(N);
    ppos_[0] = -1; pc_[0] = 0; dpos_[0] = 0;
    level_ptr.clear(); level_ptr.push_back(0);
    int cpos = 1, w = 0, level_start = 0;
    bool ok = true;
    for (int g = 0; g < N; ++g) {
      if (g >= cpos) { ok = false; break; }          // position g was never handed out: not a tree in breadth-first order
      const int u = seq[g], e0 = child_ptr[u], nch = child_ptr[u + 1] - e0;
      if (dpos_[g] != dpos_[level_start]) {          // first node of the next level
        w = std::max(w, g - level_start);
        level_ptr.push_back(g);
        level_start = g;
      }
      if (nch > 255 || cpos + nch > N) { ok = false; break; }
      for (int j = 0; j < nch; ++j) {
        if (seq[cpos + j] != child[e0 + j]) { ok = false; break; }
        ppos_[cpos + j] = g; pc_[cpos + j] = child_c[e0 + j]; dpos_[cpos + j] = dpos_[g] + 1;
      }
      if (!ok) break;
      rec[g] = MsaBfsRec{cpos, (int32_t)pc_[g] | (nch << 8), ppos_[g], u};
      cpos += nch;
    }
    if (ok && cpos == N) {
      w = std::max(w, N - level_start);
      level_ptr.push_back(N);
      maxw = w;
      bfs_ok = true;
      full_ = false;
      return;
    }
    levels_full(N);
  }
  // The tree of one image, ready for k_msa_dp_bfs: svo_msa_tree_rec writes the level-order records and the level boundaries
  // WHILE it walks the tree breadth-first (round 5: the breadth-first sequence, the child lists in CSR form and levels() were
  // three passes over the tree with random access each - 15 % of a tree's host time).  The by-pixel arrays (seq, child lists)
  // exist only on the fallback path (SVO_MSA_LEVEL_LAUNCHES, or a node with more than 255 children).
  void build(const uint8_t* med3, const double* r_gra, const double* c_gra, int m, int n) {
    const int N = n * m;
    static const bool level_launches = getenv("SVO_MSA_LEVEL_LAUNCHES") != nullptr;
    if (!level_launches) {
      rec.resize(N);
      rc = svo_msa_tree_rec(med3, r_gra, c_gra, m, n, rec.data(), &level_ptr, &maxw, &root);
      if (rc == SVO_OK) { bfs_ok = true; full_ = false; have_csr_ = false; return; }
      if (rc != SVO_E_CAPACITY) return;
    }
    seq.resize(N); child_ptr.resize(N + 1); child.resize(N); child_c.resize(N);
    rc = svo_msa_tree(med3, r_gra, c_gra, m, n, seq.data(), child_ptr.data(), child.data(), child_c.data(), &root);
    have_csr_ = true;
    if (rc == SVO_OK) levels(N);
  }
  bool have_csr_ = false;
  // the level-launch fallback's arrays, on demand (the child lists by pixel are read back from the records if need be)
  void need_full(int N) {
    if (full_) return;
    if (!have_csr_) {
      seq.resize(N); child_ptr.assign(N + 1, 0); child.resize(N); child_c.resize(N);
      for (int g = 0; g < N; ++g) { seq[g] = rec[g].node; child_ptr[rec[g].node + 1] = (rec[g].meta >> 8) & 255; }
      for (int u = 0; u < N; ++u) child_ptr[u + 1] += child_ptr[u];
      for (int g = 0; g < N; ++g) {
        const int e0 = child_ptr[rec[g].node], nch = (rec[g].meta >> 8) & 255;
        for (int j = 0; j < nch; ++j) { child[e0 + j] = rec[rec[g].cpos + j].node; child_c[e0 + j] = (uint8_t)(rec[rec[g].cpos + j].meta & 255); }
      }
      have_csr_ = true;
    }
    levels_full(N);
  }
  void levels_full(int N) {
    full_ = true;
    parent.assign(N, -1); parent_c.assign(N, 0);
    std::vector<int32_t>& depth = depth_;
    depth.assign(N, 0);
    int max_depth = 0;
    for (int k = 0; k < N; ++k) {
      const int u = seq[k];
      for (int e = child_ptr[u]; e < child_ptr[u + 1]; ++e) {
        const int v = child[e];
        parent[v] = u; parent_c[v] = child_c[e]; depth[v] = depth[u] + 1;
        max_depth = std::max(max_depth, depth[v]);
      }
    }
    level_ptr.assign(max_depth + 2, 0);
    for (int v = 0; v < N; ++v) ++level_ptr[depth[v] + 1];
    for (int l = 0; l <= max_depth; ++l) level_ptr[l + 1] += level_ptr[l];
    fill_.assign(level_ptr.begin(), level_ptr.end() - 1);
    nodes.resize(N);
    for (int k = 0; k < N; ++k) nodes[fill_[depth[seq[k]]]++] = seq[k];
    bfs_ok = msa_bfs_records(N, nodes.data(), level_ptr, child_ptr.data(), child.data(), parent.data(), parent_c.data(), fill_, rec, &maxw);
  }
  std::vector<MsaBfsRec> rec;   // the tree by level-order position (k_msa_dp_bfs)
  int maxw = 0;                 // widest level
  bool bfs_ok = false;          // children are consecutive in level order (always, for a breadth-first seq)
  bool full_ = false;           // nodes / parent / parent_c are filled in
  std::vector<int32_t> ppos_, dpos_;
  std::vector<uint8_t> pc_;
};

struct DevTree { int32_t *nodes, *child_ptr, *child, *parent; uint8_t *child_c, *parent_c; };
struct MsaHostStore {
  std::vector<HostTree> tree;
  std::vector<std::vector<uint8_t>> med;
  std::vector<std::vector<double>> gra;
};

}  // namespace

// img3[side]: packed BGR images on the device (3 * m bytes per row); d_out: n * m bytes on the device.
static int msa_solve_device(svo_ctx* ctx, hipStream_t s, DevBuf& buf, uint8_t* const img3[2], int n, int m, int d, int scale,
                            uint8_t* d_out) {
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* name) {   // wall-clock profile entries of the host-visible stages
    const auto now = std::chrono::steady_clock::now();
    if (ctx->profiling) {
      const double ms = std::chrono::duration<double, std::milli>(now - t_last).count();
      bool found = false;
      for (auto& e : ctx->prof)
        if (e.name == name) { e.total_ms += ms; e.launches += 1; found = true; break; }
      if (!found) { SvoProfileEntry e; e.name = name; e.total_ms = ms; e.launches = 1; ctx->prof.push_back(e); }
    }
    t_last = now;
  };
  const int D = d + 1;
  const size_t N = (size_t)n * m, V = N * D;
  double* g = buf.get<double>(6 * N);                 // graL graR r_graL c_graL r_graR c_graR
  float* cost[2] = {buf.get<float>(V), buf.get<float>(V)};   // costL, costR
  float* d_up = buf.get<float>(V); float* d_A = buf.get<float>(V);
  uint8_t* med3[2] = {buf.get<uint8_t>(3 * N), buf.get<uint8_t>(3 * N)};
  uint8_t* gray = buf.get<uint8_t>(N);
  uint8_t* d_disp[2] = {buf.get<uint8_t>(N), buf.get<uint8_t>(N)};   // d0 (left), d1 (right)
  uint8_t* d_raw = buf.get<uint8_t>(N); uint8_t* d_mask = buf.get<uint8_t>(N);
  double* d_Exp = buf.get<double>(256);
  DevTree dt[2];
  for (int s2 = 0; s2 < 2; ++s2) {
    dt[s2].nodes = buf.get<int32_t>(N); dt[s2].child_ptr = buf.get<int32_t>(N + 1); dt[s2].child = buf.get<int32_t>(N);
    dt[s2].parent = buf.get<int32_t>(N); dt[s2].child_c = buf.get<uint8_t>(N); dt[s2].parent_c = buf.get<uint8_t>(N);
  }
  MsaBfsRec* d_rec[2] = {buf.get<MsaBfsRec>(N), buf.get<MsaBfsRec>(N)};     // the trees in level order (k_msa_dp_bfs)
  int32_t* d_lp[2] = {buf.get<int32_t>(N + 1), buf.get<int32_t>(N + 1)};
  MsaBfsTab* d_tabs = buf.get<MsaBfsTab>(2);
  double* d_Exp2 = buf.get<double>(256);
  float* d_up1 = buf.get<float>(V); float* d_A1 = buf.get<float>(V); uint8_t* d_raw1 = buf.get<uint8_t>(N);   // second work volume
  if (!g || !cost[0] || !cost[1] || !d_up || !d_A || !med3[0] || !med3[1] || !gray || !d_disp[0] || !d_disp[1] || !d_raw ||
      !d_mask || !d_Exp || !dt[1].parent_c || !d_rec[0] || !d_rec[1] || !d_lp[0] || !d_lp[1] || !d_tabs || !d_Exp2 ||
      !d_up1 || !d_A1 || !d_raw1) {
    ctx->last_error = "svo_msa_solve: hipMalloc";
    return SVO_E_NOMEM;
  }
  mark("host_msa_alloc");
  const dim3 px((m + 255) / 256, n);
  const unsigned nbN = (unsigned)((N + 255) / 256), nbV = (unsigned)((V + 255) / 256);

  // 1. MSA::init on the device
  {
    SvoTimer t(ctx, "k_msa_init");
    for (int side = 0; side < 2; ++side) {
      hipLaunchKernelGGL(k_msa_gray, dim3(nbN), dim3(256), 0, s, img3[side], (int)N, gray);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, gray, n, m, 1, 127.5, g + side * N);
    }
    hipLaunchKernelGGL(k_msa_cost, dim3(nbV), dim3(256), 0, s, img3[0], img3[1], g, g + N, n, m, D, cost[0]);
    hipLaunchKernelGGL(k_msa_cost_right, dim3(nbV), dim3(256), 0, s, cost[0], n, m, D, cost[1]);
    for (int side = 0; side < 2; ++side) {
      hipLaunchKernelGGL(k_ctmf<1>, dim3((3 * m + 255) / 256, n), dim3(256), 0, s, img3[side], med3[side], m, n, 3 * m, 3 * m, 3);
      hipLaunchKernelGGL(k_msa_gray, dim3(nbN), dim3(256), 0, s, med3[side], (int)N, gray);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, gray, n, m, 1, 0.0, g + (2 + 2 * side) * N);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, gray, n, m, 0, 0.0, g + (3 + 2 * side) * N);
    }
  }
  std::vector<uint8_t> h_med[2] = {std::vector<uint8_t>(3 * N), std::vector<uint8_t>(3 * N)};
  std::vector<double> h_gra(4 * N);
  for (int side = 0; side < 2; ++side) SVO_HIP(ctx, hipMemcpyAsync(h_med[side].data(), med3[side], 3 * N, hipMemcpyDeviceToHost, s));
  SVO_HIP(ctx, hipMemcpyAsync(h_gra.data(), g + 2 * N, 4 * N * sizeof(double), hipMemcpyDeviceToHost, s));
  SVO_HIP(ctx, hipStreamSynchronize(s));
  mark("host_msa_init_and_download");

  // 2. the two aggregation trees, one host thread each (the reference builds them one after the other)
  HostTree tree[2];
  auto grow = [&](int side) {
    HostTree& t = tree[side];
    t.build(h_med[side].data(), h_gra.data() + 2 * side * N, h_gra.data() + (2 * side + 1) * N, m, n);
  };
  {
    std::thread right(grow, 1);
    grow(0);
    right.join();
  }
  mark("host_msa_trees");
  for (int side = 0; side < 2; ++side)
    if (tree[side].rc) { ctx->last_error = "svo_msa_solve: tree construction failed"; return tree[side].rc; }
  // one launch per aggregation (k_msa_dp_bfs) when the widest level fits its LDS buffers - always, on real images
  int maxw = 0, maxl = 0;
  for (int side = 0; side < 2; ++side) { maxw = std::max(maxw, tree[side].maxw); maxl = std::max(maxl, (int)tree[side].level_ptr.size() - 1); }
  const bool bfs = tree[0].bfs_ok && tree[1].bfs_ok && msa_dp_bfs_ok(ctx, D, maxw, maxl);
  MsaBfsTab h_tabs[2];
  for (int side = 0; side < 2; ++side) {
    const HostTree& t = tree[side];
    if (bfs) {
      SVO_HIP(ctx, hipMemcpyAsync(d_rec[side], t.rec.data(), N * sizeof(MsaBfsRec), hipMemcpyHostToDevice, s));
      SVO_HIP(ctx, hipMemcpyAsync(d_lp[side], t.level_ptr.data(), t.level_ptr.size() * 4, hipMemcpyHostToDevice, s));
      h_tabs[side] = MsaBfsTab{d_rec[side], d_lp[side], (int32_t)t.level_ptr.size() - 1, (int32_t)N, cost[side], side ? d_up1 : d_up,
                               side ? d_A1 : d_A};
      continue;
    }
    tree[side].need_full((int)N);
    SVO_HIP(ctx, hipMemcpyAsync(dt[side].nodes, t.nodes.data(), N * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(dt[side].child_ptr, t.child_ptr.data(), (N + 1) * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(dt[side].child, t.child.data(), (N - 1) * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(dt[side].parent, t.parent.data(), N * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(dt[side].child_c, t.child_c.data(), N - 1, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(dt[side].parent_c, t.parent_c.data(), N, hipMemcpyHostToDevice, s));
  }
  if (bfs) SVO_HIP(ctx, hipMemcpyAsync(d_tabs, h_tabs, sizeof h_tabs, hipMemcpyHostToDevice, s));

  // 3. TreeDp + WTA per image, L/R check, TreeDp + WTA again with the sharper weights
  double Exp[2][256];
  for (int i = 0; i <= 255; ++i) { Exp[0][i] = exp(-i * 1.0 / 0.1 / 255); Exp[1][i] = exp(-i * 1.0 / (0.1 / 2) / 255); }
  auto aggregate = [&](int side, const float* c, const double* E, uint8_t* out_disp) -> int {
    const HostTree& t = tree[side];
    const int levels = (int)t.level_ptr.size() - 1;
    SVO_HIP(ctx, hipMemcpyAsync(d_Exp, E, 256 * sizeof(double), hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(d_up, c, V * sizeof(float), hipMemcpyDeviceToDevice, s));
    {
      SvoTimer tm(ctx, "k_msa_tree_dp");
      for (int l = levels - 1; l >= 0; --l) {
        const int cnt = t.level_ptr[l + 1] - t.level_ptr[l];
        hipLaunchKernelGGL(k_msa_dp_up, dim3((unsigned)(((size_t)cnt * D + 255) / 256)), dim3(256), 0, s, dt[side].nodes + t.level_ptr[l],
                           cnt, D, dt[side].child_ptr, dt[side].child, dt[side].child_c, d_Exp, d_up);
      }
      for (int l = 0; l < levels; ++l) {
        const int cnt = t.level_ptr[l + 1] - t.level_ptr[l];
        hipLaunchKernelGGL(k_msa_dp_down, dim3((unsigned)(((size_t)cnt * D + 255) / 256)), dim3(256), 0, s, dt[side].nodes + t.level_ptr[l],
                           cnt, D, dt[side].parent, dt[side].parent_c, d_Exp, d_up, d_A);
      }
    }
    SvoTimer tm(ctx, "k_msa_wta");
    hipLaunchKernelGGL(k_msa_argmin, dim3(nbN), dim3(256), 0, s, d_A, (int)N, D, d_raw);
    hipLaunchKernelGGL(k_ctmf<2>, dim3((m + 255) / 256, n), dim3(256), 0, s, d_raw, out_disp, m, n, m, m, 1);
    return SVO_OK;
  };
  int rc;
  if (bfs) {
    // the two first aggregations (right image as base, left image as base) in ONE launch pair, the refinement in another
    SVO_HIP(ctx, hipMemcpyAsync(d_Exp, Exp[0], 256 * sizeof(double), hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(d_Exp2, Exp[1], 256 * sizeof(double), hipMemcpyHostToDevice, s));
    {
      SvoTimer tm(ctx, "k_msa_tree_dp");
      msa_dp_bfs_launch(s, d_tabs, 2, (int)N, D, d_Exp, maxw, maxl);
    }
    {
      SvoTimer tm(ctx, "k_msa_wta");
      hipLaunchKernelGGL(k_msa_argmin, dim3(nbN), dim3(256), 0, s, d_A1, (int)N, D, d_raw1);
      hipLaunchKernelGGL(k_ctmf<2>, dim3((m + 255) / 256, n), dim3(256), 0, s, d_raw1, d_disp[1], m, n, m, m, 1);
      hipLaunchKernelGGL(k_msa_argmin, dim3(nbN), dim3(256), 0, s, d_A, (int)N, D, d_raw);
      hipLaunchKernelGGL(k_ctmf<2>, dim3((m + 255) / 256, n), dim3(256), 0, s, d_raw, d_disp[0], m, n, m, m, 1);
    }
    {
      SvoTimer tm(ctx, "k_msa_lrcheck");
      hipLaunchKernelGGL(k_msa_lrcheck, dim3(nbV), dim3(256), 0, s, d_disp[0], d_disp[1], n, m, D, cost[0], d_mask);
    }
    {
      SvoTimer tm(ctx, "k_msa_tree_dp");
      msa_dp_bfs_launch(s, d_tabs, 1, (int)N, D, d_Exp2, maxw, maxl);   // refine: left tree, sharper weights, checked costs
    }
    {
      SvoTimer tm(ctx, "k_msa_wta");
      hipLaunchKernelGGL(k_msa_argmin, dim3(nbN), dim3(256), 0, s, d_A, (int)N, D, d_raw);
      hipLaunchKernelGGL(k_ctmf<2>, dim3((m + 255) / 256, n), dim3(256), 0, s, d_raw, d_disp[0], m, n, m, m, 1);
    }
  } else {
  // (levels too wide for k_msa_dp_bfs: one launch per level)
  // The first two aggregations (right image as base image, left image as base image) are independent until the L/R
  // check: their level sweeps - two chains of ~2 x 1000 small dependent launches - run side by side on two streams, with
  // their own work volumes, instead of one after the other.
  {
    if (!ctx->stream_fe) SVO_HIP(ctx, svo_stream_create(&ctx->stream_fe, -1));
    hipStream_t s2 = ctx->stream_fe;
    hipEvent_t e0, e1;
    SVO_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    SVO_HIP(ctx, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    SVO_HIP(ctx, hipMemcpyAsync(d_Exp, Exp[0], 256 * sizeof(double), hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(d_up, cost[0], V * sizeof(float), hipMemcpyDeviceToDevice, s));
    SVO_HIP(ctx, hipEventRecord(e0, s));
    SVO_HIP(ctx, hipStreamWaitEvent(s2, e0, 0));
    SVO_HIP(ctx, hipMemcpyAsync(d_up1, cost[1], V * sizeof(float), hipMemcpyDeviceToDevice, s2));
    const int lv[2] = {(int)tree[0].level_ptr.size() - 1, (int)tree[1].level_ptr.size() - 1};
    float* ups[2] = {d_up, d_up1}; float* As[2] = {d_A, d_A1};
    hipStream_t st2[2] = {s, s2};
    {
      SvoTimer tm(ctx, "k_msa_tree_dp");
      const int lmax = std::max(lv[0], lv[1]);
      for (int l = lmax - 1; l >= 0; --l)
        for (int side = 0; side < 2; ++side) {
          if (l >= lv[side]) continue;
          const HostTree& t = tree[side];
          const int cnt = t.level_ptr[l + 1] - t.level_ptr[l];
          hipLaunchKernelGGL(k_msa_dp_up, dim3((unsigned)(((size_t)cnt * D + 255) / 256)), dim3(256), 0, st2[side],
                             dt[side].nodes + t.level_ptr[l], cnt, D, dt[side].child_ptr, dt[side].child, dt[side].child_c, d_Exp, ups[side]);
        }
      for (int l = 0; l < lmax; ++l)
        for (int side = 0; side < 2; ++side) {
          if (l >= lv[side]) continue;
          const HostTree& t = tree[side];
          const int cnt = t.level_ptr[l + 1] - t.level_ptr[l];
          hipLaunchKernelGGL(k_msa_dp_down, dim3((unsigned)(((size_t)cnt * D + 255) / 256)), dim3(256), 0, st2[side],
                             dt[side].nodes + t.level_ptr[l], cnt, D, dt[side].parent, dt[side].parent_c, d_Exp, ups[side], As[side]);
        }
    }
    {
      SvoTimer tm(ctx, "k_msa_wta");
      hipLaunchKernelGGL(k_msa_argmin, dim3(nbN), dim3(256), 0, s2, d_A1, (int)N, D, d_raw1);
      hipLaunchKernelGGL(k_ctmf<2>, dim3((m + 255) / 256, n), dim3(256), 0, s2, d_raw1, d_disp[1], m, n, m, m, 1);
      hipLaunchKernelGGL(k_msa_argmin, dim3(nbN), dim3(256), 0, s, d_A, (int)N, D, d_raw);
      hipLaunchKernelGGL(k_ctmf<2>, dim3((m + 255) / 256, n), dim3(256), 0, s, d_raw, d_disp[0], m, n, m, m, 1);
    }
    SVO_HIP(ctx, hipEventRecord(e1, s2));
    SVO_HIP(ctx, hipStreamWaitEvent(s, e1, 0));
    hipEventDestroy(e0); hipEventDestroy(e1);   // destruction is deferred until the recorded work has completed
    (void)rc;
  }
  {
    SvoTimer tm(ctx, "k_msa_lrcheck");
    hipLaunchKernelGGL(k_msa_lrcheck, dim3(nbV), dim3(256), 0, s, d_disp[0], d_disp[1], n, m, D, cost[0], d_mask);
  }
  if ((rc = aggregate(0, cost[0], Exp[1], d_disp[0]))) return rc;   // refine
  }
  hipLaunchKernelGGL(k_msa_scale, dim3(nbN), dim3(256), 0, s, d_disp[0], (int)N, scale, d_out);
  mark("host_msa_enqueue_aggregation");
  SVO_HIP(ctx, hipStreamSynchronize(s));
  mark("host_msa_wait_aggregation");   // Exp[][] and the trees are read by copies until here
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

extern "C" int svo_msa_solve(svo_ctx* ctx, const uint8_t* bgrL, const uint8_t* bgrR, int width, int height, int step, int d,
                             int scale, uint8_t* disparity) {
  if (!ctx) return SVO_E_INVALID;
  const int n = height, m = width, D = d + 1;
  if (!bgrL || !bgrR || !disparity || n < 5 || m < 5 || d < 0 || D > 256 || step < 3 * m || scale < 1 ||
      (size_t)n * m * D > (size_t)1 << 30 || (int64_t)n * m > (1 << 24)) {
    ctx->last_error = "svo_msa_solve: invalid argument";
    return SVO_E_INVALID;
  }
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)n * m;
  DevBuf& buf = msa_arena(ctx, 0);
  uint8_t* img3[2] = {buf.get<uint8_t>(3 * N), buf.get<uint8_t>(3 * N)};
  uint8_t* d_out = buf.get<uint8_t>(N);
  if (!img3[0] || !img3[1] || !d_out) { ctx->last_error = "svo_msa_solve: hipMalloc"; return SVO_E_NOMEM; }
  hipStream_t s = ctx->stream;
  if (ctx->h_pinned && 6 * N + 4096 <= ctx->pinned_bytes) {   // (the buffer's last page belongs to svo_track_frame's boxes)
    // rows gathered into the pinned buffer, one linear copy per image (a 2-D copy from pageable memory costs ~10 ms)
    const uint8_t* src[2] = {bgrL, bgrR};
    for (int side = 0; side < 2; ++side) {
      uint8_t* h = reinterpret_cast<uint8_t*>(ctx->h_pinned) + 3 * N * side;
      for (int y = 0; y < n; ++y) memcpy(h + (size_t)y * 3 * m, src[side] + (size_t)y * step, 3 * (size_t)m);
      SVO_HIP(ctx, hipMemcpyAsync(img3[side], h, 3 * N, hipMemcpyHostToDevice, s));
    }
  } else {
    SVO_HIP(ctx, hipMemcpy2DAsync(img3[0], 3 * (size_t)m, bgrL, step, 3 * (size_t)m, n, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpy2DAsync(img3[1], 3 * (size_t)m, bgrR, step, 3 * (size_t)m, n, hipMemcpyHostToDevice, s));
  }
  const int rc = msa_solve_device(ctx, s, buf, img3, n, m, d, scale, d_out);
  if (rc) return rc;
  SVO_HIP(ctx, svo_memcpy_sync(ctx, disparity, d_out, N, hipMemcpyDeviceToHost));
  return SVO_OK;
}

namespace {
__global__ void k_msa_gray_to_bgr(const uint8_t* gray, int stride, int m, int n, uint8_t* bgr) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= m) return;
  const uint8_t v = gray[(size_t)i * stride + j];
  uint8_t* o = bgr + ((size_t)i * m + j) * 3;
  o[0] = v; o[1] = v; o[2] = v;
}
__global__ void k_msa_to_float(const uint8_t* d, int n, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)d[i];
}
}  // namespace

// frame::MB (src/frame.cc:82-91) for the tracker's dense-depth mode: gray images already on the device (a grayscale
// file read as colour has B = G = R), disparity as CV_32F like `disp_img.convertTo(disp_32f, CV_32F, 1)`.
static int msa_run_on_stream(svo_ctx* ctx, hipStream_t st, int arena, const uint8_t* dL, const uint8_t* dR, int pitch, int W,
                             int H, int d, float* d_disp) {
  if (H < 5 || W < 5 || d < 0 || d > 255) return SVO_E_INVALID;
  const size_t N = (size_t)W * H;
  DevBuf& buf = msa_arena(ctx, arena);
  uint8_t* img3[2] = {buf.get<uint8_t>(3 * N), buf.get<uint8_t>(3 * N)};
  uint8_t* d_out = buf.get<uint8_t>(N);
  if (!img3[0] || !img3[1] || !d_out) { ctx->last_error = "svo_msa_run_dev: hipMalloc"; return SVO_E_NOMEM; }
  const dim3 px((W + 255) / 256, H);
  hipLaunchKernelGGL(k_msa_gray_to_bgr, px, dim3(256), 0, st, dL, pitch, W, H, img3[0]);
  hipLaunchKernelGGL(k_msa_gray_to_bgr, px, dim3(256), 0, st, dR, pitch, W, H, img3[1]);
  const int rc = msa_solve_device(ctx, st, buf, img3, H, W, d, 1, d_out);
  if (rc) return rc;
  hipLaunchKernelGGL(k_msa_to_float, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_out, (int)N, d_disp);
  SVO_HIP(ctx, hipStreamSynchronize(st));   // the arena is handed out again by the next solve
  return SVO_OK;
}

int svo_msa_run_dev(svo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int pitch, int W, int H, int d, float* d_disp) {
  return msa_run_on_stream(ctx, ctx->stream, 0, dL, dR, pitch, W, H, d, d_disp);
}

// MSA::solve for C frames together (gray frames on the device, float maps out): init per frame, all 2C trees on host
// threads, then every stage of the aggregation once for the whole chunk - the level sweeps as one launch per level for
// all frames, the per-pixel stages over the C maps stacked into one tall image.
// `lane` (0 / 1) selects the device arena and host store; with two lanes alternating over the chunks, host_phase is held
// over the tree builders and gpu_phase from the tree upload to the end, so that the trees of one chunk grow while the GPU
// aggregates the previous one and initialises the next.
static int msa_many_device(svo_ctx* ctx, hipStream_t s, const uint8_t* dL, const uint8_t* dR, int pitch, size_t frame_stride,
                           int m, int n, int d, int C, float* d_disp_out, int lane, std::mutex* host_phase,
                           std::mutex* gpu_phase) {
  const bool dbg = getenv("SVO_MSA_DEBUG") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (dbg) fprintf(stderr, "[msa x%d] %-28s %8.2f ms\n", C, what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  const int D = d + 1;
  const size_t N = (size_t)n * m, V = N * D;
  std::unique_lock<std::mutex> host_lock;   // taken for the tree builders only (below): a lane's init kernels and downloads - its own
  // arena, its own host store - run while the other lane builds its trees
  DevBuf& buf = msa_arena(ctx, 1 + lane);
  uint8_t* img3[2] = {buf.get<uint8_t>(3 * N), buf.get<uint8_t>(3 * N)};
  uint8_t* med3[2] = {buf.get<uint8_t>(3 * N), buf.get<uint8_t>(3 * N)};
  uint8_t* gray = buf.get<uint8_t>(N);
  double* g = buf.get<double>(6 * N);
  float* cost[2] = {buf.get<float>(V * C), buf.get<float>(V * C)};   // [side][frame][pixel][disparity]
  float* d_up = buf.get<float>(V * C); float* d_A = buf.get<float>(V * C);
  uint8_t* d_disp[2] = {buf.get<uint8_t>(N * C), buf.get<uint8_t>(N * C)};
  uint8_t* d_raw = buf.get<uint8_t>(N * C); uint8_t* d_mask = buf.get<uint8_t>(N * C);
  double* d_Exp = buf.get<double>(512);
  int32_t* d_int = buf.get<int32_t>((size_t)2 * C * (5 * N + 4));     // nodes, level_ptr, child_ptr, child, parent per tree
  uint8_t* d_byte = buf.get<uint8_t>((size_t)2 * C * 2 * N);          // child_c, parent_c per tree
  MsaTreeTab* d_tabs = buf.get<MsaTreeTab>((size_t)2 * C);
  MsaBfsRec* d_rec = buf.get<MsaBfsRec>((size_t)2 * C * N);           // the trees in level order (k_msa_dp_bfs)
  MsaBfsTab* d_btabs = buf.get<MsaBfsTab>((size_t)2 * C);
  if (!d_rec || !d_btabs) { ctx->last_error = "svo_msa (batched): hipMalloc"; return SVO_E_NOMEM; }
  if (!img3[0] || !img3[1] || !med3[0] || !med3[1] || !gray || !g || !cost[0] || !cost[1] || !d_up || !d_A || !d_disp[0] ||
      !d_disp[1] || !d_raw || !d_mask || !d_Exp || !d_int || !d_byte || !d_tabs) {
    ctx->last_error = "svo_msa (batched): hipMalloc";
    return SVO_E_NOMEM;
  }
  mark("device buffers");
  const dim3 px((m + 255) / 256, n);
  const unsigned nbN = (unsigned)((N + 255) / 256), nbV = (unsigned)((V + 255) / 256);
  // 1. MSA::init per frame; the median images and their gradients go to the host for the tree builders
  MsaArenas* arenas = static_cast<MsaArenas*>(ctx->msa_arenas);
  if (!arenas->host[lane]) arenas->host[lane] = new MsaHostStore();
  MsaHostStore& hs = *arenas->host[lane];
  if ((int)hs.tree.size() < 2 * C) { hs.tree.resize(2 * C); hs.med.resize(2 * C); }
  if ((int)hs.gra.size() < C) hs.gra.resize(C);
  for (int k = 0; k < 2 * C; ++k) hs.med[k].resize(3 * N);
  for (int b = 0; b < C; ++b) hs.gra[b].resize(4 * N);
  std::vector<std::vector<uint8_t>>& h_med = hs.med;
  std::vector<std::vector<double>>& h_gra = hs.gra;
  mark("host buffers");
  for (int b = 0; b < C; ++b) {
    hipLaunchKernelGGL(k_msa_gray_to_bgr, px, dim3(256), 0, s, dL + b * frame_stride, pitch, m, n, img3[0]);
    hipLaunchKernelGGL(k_msa_gray_to_bgr, px, dim3(256), 0, s, dR + b * frame_stride, pitch, m, n, img3[1]);
    for (int side = 0; side < 2; ++side) {
      hipLaunchKernelGGL(k_msa_gray, dim3(nbN), dim3(256), 0, s, img3[side], (int)N, gray);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, gray, n, m, 1, 127.5, g + side * N);
    }
    hipLaunchKernelGGL(k_msa_cost, dim3(nbV), dim3(256), 0, s, img3[0], img3[1], g, g + N, n, m, D, cost[0] + (size_t)b * V);
    hipLaunchKernelGGL(k_msa_cost_right, dim3(nbV), dim3(256), 0, s, cost[0] + (size_t)b * V, n, m, D, cost[1] + (size_t)b * V);
    for (int side = 0; side < 2; ++side) {
      hipLaunchKernelGGL(k_ctmf<1>, dim3((3 * m + 255) / 256, n), dim3(256), 0, s, img3[side], med3[side], m, n, 3 * m, 3 * m, 3);
      hipLaunchKernelGGL(k_msa_gray, dim3(nbN), dim3(256), 0, s, med3[side], (int)N, gray);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, gray, n, m, 1, 0.0, g + (2 + 2 * side) * N);
      hipLaunchKernelGGL(k_msa_gradient, px, dim3(256), 0, s, gray, n, m, 0, 0.0, g + (3 + 2 * side) * N);
      SVO_HIP(ctx, hipMemcpyAsync(h_med[2 * b + side].data(), med3[side], 3 * N, hipMemcpyDeviceToHost, s));
    }
    SVO_HIP(ctx, hipMemcpyAsync(h_gra[b].data(), g + 2 * N, 4 * N * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  SVO_HIP(ctx, hipStreamSynchronize(s));
  mark("init + download");
  if (host_phase) host_lock = std::unique_lock<std::mutex>(*host_phase);
  // 2. the 2C trees, on as many host threads as are sensible (tree 2b: left image of frame b, 2b + 1: right)
  std::vector<HostTree>& tree = hs.tree;
  std::atomic<long long> dbg_tree_us{0}, dbg_levels_us{0};
  {
    std::atomic<int> next(0);
    auto work = [&]() {
      for (int k = next.fetch_add(1); k < 2 * C; k = next.fetch_add(1)) {
        HostTree& t = tree[k];
        const int b = k >> 1, side = k & 1;
        const auto ta = std::chrono::steady_clock::now();
        t.build(h_med[k].data(), h_gra[b].data() + 2 * side * N, h_gra[b].data() + (2 * side + 1) * N, m, n);
        const auto tb = std::chrono::steady_clock::now();
        if (dbg) { dbg_tree_us += (long long)std::chrono::duration<double, std::micro>(tb - ta).count(); dbg_levels_us += (long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tb).count(); }
      }
    };
    const int T = std::max(1, std::min(std::min(2 * C, 32), 2 * svo_host_cpus()));   // (measured on a 16-CPU quota: 32 builders 58 frames/s, 16: 51, 64: 36)
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
  }
  mark("trees");
  if (dbg) fprintf(stderr, "[msa x%d]   per tree (thread time): svo_msa_tree %.1f ms, level records %.1f ms\n", C, dbg_tree_us / 1e3 / (2 * C), dbg_levels_us / 1e3 / (2 * C));
  for (int k = 0; k < 2 * C; ++k)
    if (tree[k].rc) { ctx->last_error = "svo_msa (batched): tree construction failed"; return tree[k].rc; }
  if (host_lock.owns_lock()) host_lock.unlock();
  std::unique_lock<std::mutex> gpu_lock;
  if (gpu_phase) gpu_lock = std::unique_lock<std::mutex>(*gpu_phase);
  // 3. trees to the device: tabs[side * C + b]
  std::vector<MsaTreeTab> h_tabs(2 * C);
  std::vector<MsaBfsTab> h_btabs(2 * C);
  int Lmax[2] = {0, 0};
  int maxw = 0, maxl = 0;
  bool bfs = true;
  for (int k = 0; k < 2 * C; ++k) { bfs = bfs && tree[k].bfs_ok; maxw = std::max(maxw, tree[k].maxw); maxl = std::max(maxl, (int)tree[k].level_ptr.size() - 1); }
  bfs = bfs && msa_dp_bfs_ok(ctx, D, maxw, maxl);
  for (int k = 0; k < 2 * C; ++k) {
    const HostTree& t = tree[k];
    const int b = k >> 1, side = k & 1;
    int32_t* ip = d_int + (size_t)k * (5 * N + 4);
    uint8_t* bp = d_byte + (size_t)k * 2 * N;
    const int levels = (int)t.level_ptr.size() - 1;
    SVO_HIP(ctx, hipMemcpyAsync(ip + N, t.level_ptr.data(), (size_t)(levels + 1) * 4, hipMemcpyHostToDevice, s));
    Lmax[side] = std::max(Lmax[side], levels);
    if (bfs) {
      SVO_HIP(ctx, hipMemcpyAsync(d_rec + (size_t)k * N, t.rec.data(), N * sizeof(MsaBfsRec), hipMemcpyHostToDevice, s));
      h_btabs[side * C + b] = MsaBfsTab{d_rec + (size_t)k * N, ip + N, levels, (int32_t)N, cost[side] + (size_t)b * V, d_up + (size_t)b * V,
                                        d_A + (size_t)b * V};
      continue;
    }
    tree[k].need_full((int)N);
    SVO_HIP(ctx, hipMemcpyAsync(ip, t.nodes.data(), N * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(ip + 2 * N + 2, t.child_ptr.data(), (N + 1) * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(ip + 3 * N + 3, t.child.data(), (N - 1) * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(ip + 4 * N + 3, t.parent.data(), N * 4, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(bp, t.child_c.data(), N - 1, hipMemcpyHostToDevice, s));
    SVO_HIP(ctx, hipMemcpyAsync(bp + N, t.parent_c.data(), N, hipMemcpyHostToDevice, s));
    h_tabs[side * C + b] = MsaTreeTab{ip, ip + N, ip + 2 * N + 2, ip + 3 * N + 3, ip + 4 * N + 3, bp, bp + N, levels};
  }
  if (bfs) SVO_HIP(ctx, hipMemcpyAsync(d_btabs, h_btabs.data(), sizeof(MsaBfsTab) * 2 * C, hipMemcpyHostToDevice, s));
  SVO_HIP(ctx, hipMemcpyAsync(d_tabs, h_tabs.data(), sizeof(MsaTreeTab) * 2 * C, hipMemcpyHostToDevice, s));
  double Exp[512];
  for (int i = 0; i <= 255; ++i) { Exp[i] = exp(-i * 1.0 / 0.1 / 255); Exp[256 + i] = exp(-i * 1.0 / (0.1 / 2) / 255); }
  SVO_HIP(ctx, hipMemcpyAsync(d_Exp, Exp, sizeof Exp, hipMemcpyHostToDevice, s));
  mark("tree upload (enqueue)");
  // widest level over the frames, per level and side: the grid of that level's launch
  std::vector<int> width[2];
  for (int side = 0; side < 2; ++side) {
    width[side].assign(Lmax[side], 0);
    for (int b = 0; b < C; ++b) {
      const HostTree& t = tree[2 * b + side];
      for (int l = 0; l + 1 < (int)t.level_ptr.size(); ++l) width[side][l] = std::max(width[side][l], t.level_ptr[l + 1] - t.level_ptr[l]);
    }
  }
  // 4. TreeDp + WTA for the right images, the left images, L/R check, TreeDp + WTA again with the sharper weights
  auto aggregate = [&](int side, const float* c, const double* E, uint8_t* out_disp) -> int {
    const MsaTreeTab* tabs = d_tabs + side * C;
    if (bfs) {
      msa_dp_bfs_launch(s, d_btabs + side * C, C, (int)N, D, E, maxw, maxl);   // reads the costs from where they are
    } else {
    SVO_HIP(ctx, hipMemcpyAsync(d_up, c, V * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int l = Lmax[side] - 1; l >= 0; --l)
      hipLaunchKernelGGL(k_msa_dp_up_many, dim3((unsigned)(((size_t)width[side][l] * D + 255) / 256), C), dim3(256), 0, s, tabs, l, D, E,
                         d_up, V);
    for (int l = 0; l < Lmax[side]; ++l)
      hipLaunchKernelGGL(k_msa_dp_down_many, dim3((unsigned)(((size_t)width[side][l] * D + 255) / 256), C), dim3(256), 0, s, tabs, l, D,
                         E, d_up, d_A, V);
    }
    hipLaunchKernelGGL(k_msa_argmin, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, s, d_A, (int)(N * C), D, d_raw);
    for (int b = 0; b < C; ++b)
      hipLaunchKernelGGL(k_ctmf<2>, dim3((m + 255) / 256, n), dim3(256), 0, s, d_raw + (size_t)b * N, out_disp + (size_t)b * N, m, n, m, m, 1);
    return SVO_OK;
  };
  int rc;
  if ((rc = aggregate(1, cost[1], d_Exp, d_disp[1]))) return rc;
  if ((rc = aggregate(0, cost[0], d_Exp, d_disp[0]))) return rc;
  hipLaunchKernelGGL(k_msa_lrcheck, dim3((unsigned)((V * C + 255) / 256)), dim3(256), 0, s, d_disp[0], d_disp[1], n * C, m, D, cost[0],
                     d_mask);   // the C maps as one tall image: the check only looks along a row
  if ((rc = aggregate(0, cost[0], d_Exp + 256, d_disp[0]))) return rc;
  hipLaunchKernelGGL(k_msa_to_float, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, s, d_disp[0], (int)(N * C), d_disp_out);
  mark("aggregation (enqueue)");
  SVO_HIP(ctx, hipStreamSynchronize(s));
  mark("aggregation (wait)");   // host trees and tables are read by copies until here
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

// B frames (frame b at dL + b * frame_stride bytes, map b at d_disp + b * W * H floats), solved in chunks that share
// their launches and build their trees side by side.
int svo_msa_run_many_dev(svo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int pitch, size_t frame_stride, int W, int H, int d,
                         int B, float* d_disp) {
  const size_t N = (size_t)W * H;
  if (B <= 1 || ctx->profiling || (int64_t)N * (d + 1) * 16 > ((int64_t)1 << 31) || getenv("SVO_MSA_FRAME_BY_FRAME")) {
    for (int b = 0; b < B; ++b) {
      const int rc = svo_msa_run_dev(ctx, dL + b * frame_stride, dR + b * frame_stride, pitch, W, H, d, d_disp + b * N);
      if (rc) return rc;
    }
    return SVO_OK;
  }
  // chunks of up to 16 frames: ~0.4 GB of HBM per frame at KITTI size (four cost volumes)
  const int nchunk = (B + 15) / 16;
  if (nchunk == 1)
    return msa_many_device(ctx, ctx->stream, dL, dR, pitch, frame_stride, W, H, d, B, d_disp, 0, nullptr, nullptr);
  // two lanes (threads, streams, arenas) take the chunks alternately
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));   // whatever produced the frames
  std::mutex host_phase, gpu_phase;
  hipStream_t st[2] = {nullptr, nullptr};
  for (int t = 0; t < 2; ++t) SVO_HIP(ctx, hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking));
  int rcs[2] = {SVO_OK, SVO_OK};
  auto lane_work = [&](int t) {
    if (hipSetDevice(ctx->device) != hipSuccess) { rcs[t] = SVO_E_HIP; return; }
    for (int c = t; c < nchunk; c += 2) {
      const int b0 = 16 * c, C = std::min(16, B - b0);
      const int rc = msa_many_device(ctx, st[t], dL + b0 * frame_stride, dR + b0 * frame_stride, pitch, frame_stride, W, H, d, C,
                                     d_disp + b0 * N, t, &host_phase, &gpu_phase);
      if (rc) { rcs[t] = rc; return; }
    }
  };
  {
    std::thread other(lane_work, 1);
    lane_work(0);
    other.join();
  }
  for (int t = 0; t < 2; ++t) hipStreamDestroy(st[t]);
  return rcs[0] ? rcs[0] : rcs[1];
}

extern "C" int svo_msa_batch_dev(svo_ctx* ctx, const uint8_t* d_L, const uint8_t* d_R, int stride, int width, int height, int B,
                                 int d, float* d_disp) {
  if (!ctx) return SVO_E_INVALID;
  if (!d_L || !d_R || !d_disp || B < 1 || width < 5 || height < 5 || stride < width || d < 0 || d > 255 ||
      (int64_t)width * height > (1 << 24)) {
    ctx->last_error = "svo_msa_batch_dev: invalid argument";
    return SVO_E_INVALID;
  }
  SVO_HIP(ctx, hipSetDevice(ctx->device));
  return svo_msa_run_many_dev(ctx, d_L, d_R, stride, (size_t)height * stride, width, height, d, B, d_disp);
}

void svo_msa_release(svo_ctx* ctx) {
  if (!ctx || !ctx->msa_arenas) return;
  MsaArenas* A = static_cast<MsaArenas*>(ctx->msa_arenas);
  for (DevBuf* b : A->a) delete b;
  delete A->host[0];
  delete A->host[1];
  delete A;
  ctx->msa_arenas = nullptr;
}
