// svo_hostfeed.hip - the pipelined host-fed tracker entries (include/svo.h: svo_track_batch_host, svo_track_sharded_host,
// svo_frontend_batch_host).  The reference reads one stereo pair from disk per Tracking::Track (main.cpp:159-195); SURVEY.md
// section 8e prices the GPU path at "H2D 2 P bytes, D2H ~30 KB" per pair with pair k uploaded to GPU k mod G.  These entries are
// that path: the caller's images start in HOST memory, are uploaded on a copy stream of their own a few pairs ahead of the front
// end (eight pairs per event), and the records come back to host memory on the way out - no synchronisation inside a call.
//
//   host images --(pinned: copied where they lie | pageable: rows gathered into a pinned ring by worker threads)-->
//   copy stream: H2D into image set p (two sets alternate between calls)  --event per 8 pairs-->
//   front-end stream: sub-batches wait for the event of their last pair (svo_track_batch_dev / svo_track_sharded_dev, unchanged)
//   ... ordered tail ...  pose stream: records D2H into a pinned buffer --> the caller's array at svo_sync (or two calls later)
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "svo_internal.h"
#include "svo_gate.h"

namespace {

constexpr int FEED_CHUNK = 8;   // pairs per upload event (7.4 MB at 1241x376: long enough for the link, short enough to start early)

struct PendingOut {             // records that sit in the pinned buffer of set p and still have to reach the caller's array
  svo_track_result* user = nullptr;
  int n = 0;
};

struct HostFeed {
  hipStream_t copy = nullptr;
  int cap = 0;                                   // pairs per image set
  uint8_t* d_img[2] = {nullptr, nullptr};        // set p: cap left images, then cap right images, H x stage_pitch each
  uint8_t* h_img[2] = {nullptr, nullptr};        // pinned staging of the same layout (allocated with the first pageable source)
  std::vector<hipEvent_t> ev_up[2];              // upload events of set p, one per FEED_CHUNK pairs
  std::vector<hipEvent_t> pair_ev;               // the call being enqueued: pair i is resident after pair_ev[i]
  hipEvent_t img_free[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // the readers of set p are done with it
  int n_free[2] = {0, 0};
  bool used[2] = {false, false};
  int parity = 0;
  svo_track_result* d_res[2] = {nullptr, nullptr};
  svo_track_result* h_res[2] = {nullptr, nullptr};
  hipEvent_t res_done[2] = {nullptr, nullptr};
  PendingOut pending[2];
  int32_t* d_box[2] = {nullptr, nullptr};        // cap x SVO_MAX_BOXES x 4 boxes, then cap counts
  int32_t* h_box[2] = {nullptr, nullptr};
  // svo_frontend_batch_host: result staging (device + pinned), allocated on first use
  uint8_t* d_fe[2] = {nullptr, nullptr};
  uint8_t* h_fe[2] = {nullptr, nullptr};
  hipEvent_t fe_done[2] = {nullptr, nullptr};
  struct FeOut { svo_kp* kp; uint8_t* desc; int32_t* n; float* uR; float* depth; int B; } fe_pending[2] = {};
};

HostFeed* feed_of(svo_ctx* ctx) { return reinterpret_cast<HostFeed*>(ctx->hostfeed); }

size_t img_bytes(const svo_ctx* ctx) { return (size_t)ctx->g.H * ctx->stage_pitch; }

bool is_pinned(const void* p) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

int feed_reserve(svo_ctx* ctx, int pairs) {
  if (!ctx->hostfeed) ctx->hostfeed = new HostFeed();
  HostFeed* hf = feed_of(ctx);
  if (!hf->copy) {
    // the copy stream runs beside everything: picked like the sharded tracker's gather stream (its packets must not sit on the
    // dispatch pipe of one of the tail's queues, and never behind the front end's grids)
    int attempts = 0, percent = 0;
    const int rc = svo_pick_stream(ctx, [](hipStream_t* q) { return svo_stream_create(q, 0); }, {ctx->stream, ctx->stream_idx, ctx->stream_fe_batch},
                                   &hf->copy, &attempts, &percent, {}, {ctx->stream_fe_batch});
    if (rc) return rc;
  }
  if (hf->cap >= pairs) return SVO_OK;
  // grow: nothing of the old sets may be in flight
  SVO_HIP(ctx, hipStreamSynchronize(hf->copy));
  { const int rcq = svo_track_quiesce(ctx); if (rcq) return rcq; }
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int rcf = svo_hostfeed_flush(ctx);
  if (rcf) return rcf;
  for (int p = 0; p < 2; ++p) {
    if (hf->d_img[p]) hipFree(hf->d_img[p]);
    if (hf->h_img[p]) hipHostFree(hf->h_img[p]);
    if (hf->d_res[p]) hipFree(hf->d_res[p]);
    if (hf->h_res[p]) hipHostFree(hf->h_res[p]);
    if (hf->d_box[p]) hipFree(hf->d_box[p]);
    if (hf->h_box[p]) hipHostFree(hf->h_box[p]);
    if (hf->d_fe[p]) hipFree(hf->d_fe[p]);
    if (hf->h_fe[p]) hipHostFree(hf->h_fe[p]);
    hf->d_img[p] = nullptr; hf->h_img[p] = nullptr; hf->d_res[p] = nullptr; hf->h_res[p] = nullptr; hf->d_box[p] = nullptr; hf->h_box[p] = nullptr;
    hf->d_fe[p] = nullptr; hf->h_fe[p] = nullptr;
    hf->used[p] = false;
  }
  hf->cap = 0;
  const size_t box_bytes = (size_t)pairs * (SVO_MAX_BOXES * 16 + 4);
  for (int p = 0; p < 2; ++p) {
    if (hipMalloc(reinterpret_cast<void**>(&hf->d_img[p]), 2 * (size_t)pairs * img_bytes(ctx)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&hf->d_res[p]), sizeof(svo_track_result) * (size_t)pairs) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&hf->h_res[p]), sizeof(svo_track_result) * (size_t)pairs) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&hf->d_box[p]), box_bytes) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&hf->h_box[p]), box_bytes) != hipSuccess) {
      (void)hipGetLastError();
      ctx->last_error = "host feed: out of memory for the image sets";
      return SVO_E_NOMEM;
    }
    // (rows are W bytes wide in a pitch of stage_pitch: the padding is never read as image content, but keep it defined)
    SVO_HIP(ctx, hipMemsetAsync(hf->d_img[p], 0, 2 * (size_t)pairs * img_bytes(ctx), hf->copy));
    if (!hf->res_done[p]) SVO_HIP(ctx, hipEventCreateWithFlags(&hf->res_done[p], hipEventDisableTiming));
    if (!hf->fe_done[p]) SVO_HIP(ctx, hipEventCreateWithFlags(&hf->fe_done[p], hipEventDisableTiming));
    for (int k = 0; k < 4; ++k)
      if (!hf->img_free[p][k]) SVO_HIP(ctx, hipEventCreateWithFlags(&hf->img_free[p][k], hipEventDisableTiming));
  }
  SVO_HIP(ctx, hipStreamSynchronize(hf->copy));
  hf->cap = pairs;
  return SVO_OK;
}

// Pair i of this call (i = 0 .. n - 1) is the caller's frame `first + i * step` (sharded: first = g, step = G).  Uploads them
// into image set p in chunks of FEED_CHUNK pairs on the copy stream; hf->pair_ev[i] = the event after which pair i is resident.
int feed_upload(svo_ctx* ctx, int p, const uint8_t* grayL, const uint8_t* grayR, int stride, int first, int step, int n) {
  HostFeed* hf = feed_of(ctx);
  const int W = ctx->g.W, H = ctx->g.H, pitch = ctx->stage_pitch;
  const size_t ib = img_bytes(ctx), fb = (size_t)H * stride;
  const int nchunk = (n + FEED_CHUNK - 1) / FEED_CHUNK;
  while ((int)hf->ev_up[p].size() < nchunk) {
    hipEvent_t e;
    SVO_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hf->ev_up[p].push_back(e);
  }
  // the set's last readers (front end / dense stage of the call two back) first
  for (int k = 0; k < hf->n_free[p]; ++k) SVO_HIP(ctx, hipStreamWaitEvent(hf->copy, hf->img_free[p][k], 0));
  uint8_t* dL = hf->d_img[p];
  uint8_t* dR = hf->d_img[p] + (size_t)hf->cap * ib;
  const bool pinned = is_pinned(grayL) && is_pinned(grayR);
  hf->pair_ev.assign((size_t)n, nullptr);
  if (pinned) {
    // copied where they lie: one linear copy per chunk and side when the caller's frames are consecutive and pitched like the
    // set, one 2-D copy per chunk and side when only the pitch differs (rows of consecutive frames continue at `stride`),
    // one per image otherwise
    for (int c = 0; c < nchunk; ++c) {
      const int i0 = c * FEED_CHUNK, b = std::min(FEED_CHUNK, n - i0);
      const uint8_t* sL = grayL + (size_t)(first + (size_t)i0 * step) * fb;
      const uint8_t* sR = grayR + (size_t)(first + (size_t)i0 * step) * fb;
      if (step == 1 && stride == pitch) {
        SVO_HIP(ctx, hipMemcpyAsync(dL + i0 * ib, sL, ib * b, hipMemcpyHostToDevice, hf->copy));
        SVO_HIP(ctx, hipMemcpyAsync(dR + i0 * ib, sR, ib * b, hipMemcpyHostToDevice, hf->copy));
      } else if (step == 1) {
        SVO_HIP(ctx, hipMemcpy2DAsync(dL + i0 * ib, pitch, sL, stride, W, (size_t)H * b, hipMemcpyHostToDevice, hf->copy));
        SVO_HIP(ctx, hipMemcpy2DAsync(dR + i0 * ib, pitch, sR, stride, W, (size_t)H * b, hipMemcpyHostToDevice, hf->copy));
      } else {
        for (int i = i0; i < i0 + b; ++i) {
          const size_t k = (size_t)first + (size_t)i * step;
          if (stride == pitch) {
            SVO_HIP(ctx, hipMemcpyAsync(dL + i * ib, grayL + k * fb, ib, hipMemcpyHostToDevice, hf->copy));
            SVO_HIP(ctx, hipMemcpyAsync(dR + i * ib, grayR + k * fb, ib, hipMemcpyHostToDevice, hf->copy));
          } else {
            SVO_HIP(ctx, hipMemcpy2DAsync(dL + i * ib, pitch, grayL + k * fb, stride, W, H, hipMemcpyHostToDevice, hf->copy));
            SVO_HIP(ctx, hipMemcpy2DAsync(dR + i * ib, pitch, grayR + k * fb, stride, W, H, hipMemcpyHostToDevice, hf->copy));
          }
        }
      }
      SVO_HIP(ctx, hipEventRecord(hf->ev_up[p][c], hf->copy));
      for (int i = i0; i < i0 + b; ++i) hf->pair_ev[i] = hf->ev_up[p][c];
    }
    return SVO_OK;
  }
  // Pageable source: rows gathered into the pinned ring by worker threads (a copy straight from pageable memory blocks the
  // caller and moves ~14 ms per image on this stack, docs/NEXT_ROUNDS.md), chunk by chunk; this thread enqueues each chunk's
  // linear H2D as soon as it is staged - staging of chunk c + 1 runs beside the upload of chunk c.
  if (!hf->h_img[p]) {
    if (hipHostMalloc(reinterpret_cast<void**>(&hf->h_img[p]), 2 * (size_t)hf->cap * ib) != hipSuccess) {
      (void)hipGetLastError();
      ctx->last_error = "host feed: out of pinned memory for the staging ring";
      return SVO_E_NOMEM;
    }
    memset(hf->h_img[p], 0, 2 * (size_t)hf->cap * ib);
  } else if (hf->used[p] && !hf->ev_up[p].empty()) {
    // the uploads that last read this half of the ring (two calls back)
    for (hipEvent_t e : hf->ev_up[p]) SVO_HIP(ctx, hipEventSynchronize(e));
  }
  uint8_t* hL = hf->h_img[p];
  uint8_t* hR = hf->h_img[p] + (size_t)hf->cap * ib;
  std::vector<std::atomic<int>> staged((size_t)nchunk);
  for (auto& s : staged) s.store(0, std::memory_order_relaxed);
  const int T = std::max(1, std::min({4, svo_host_cpus() - 1, nchunk}));   // staging threads (this thread enqueues)
  auto worker = [&](int t) {
    for (int c = t; c < nchunk; c += T) {
      const int i0 = c * FEED_CHUNK, b = std::min(FEED_CHUNK, n - i0);
      for (int i = i0; i < i0 + b; ++i) {
        const size_t k = (size_t)first + (size_t)i * step;
        const uint8_t* sL = grayL + k * fb;
        const uint8_t* sR = grayR + k * fb;
        uint8_t* tL = hL + i * ib;
        uint8_t* tR = hR + i * ib;
        if (stride == pitch) {
          memcpy(tL, sL, ib);
          memcpy(tR, sR, ib);
        } else {
          for (int y = 0; y < H; ++y) {
            memcpy(tL + (size_t)y * pitch, sL + (size_t)y * stride, W);
            memcpy(tR + (size_t)y * pitch, sR + (size_t)y * stride, W);
          }
        }
      }
      staged[c].store(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  pool.reserve((size_t)T);
  int started = 0;
  // (std::thread's constructor may throw; an exception must not unwind past joinable threads: the shares of the threads that
  // could not be started are staged here, before the uploads)
  try { for (; started < T; ++started) pool.emplace_back(worker, started); } catch (...) {}
  for (int t = started; t < T; ++t) worker(t);
  int rc = SVO_OK;
  for (int c = 0; c < nchunk; ++c) {
    while (!staged[c].load(std::memory_order_acquire)) std::this_thread::yield();
    if (rc == SVO_OK) {
      const int i0 = c * FEED_CHUNK, b = std::min(FEED_CHUNK, n - i0);
      if (hipMemcpyAsync(dL + i0 * ib, hL + i0 * ib, ib * b, hipMemcpyHostToDevice, hf->copy) != hipSuccess ||
          hipMemcpyAsync(dR + i0 * ib, hR + i0 * ib, ib * b, hipMemcpyHostToDevice, hf->copy) != hipSuccess ||
          hipEventRecord(hf->ev_up[p][c], hf->copy) != hipSuccess) {
        ctx->last_error = std::string("host feed upload: ") + hipGetErrorString(hipGetLastError());
        rc = SVO_E_HIP;
      }
      for (int i = i0; i < i0 + b; ++i) hf->pair_ev[i] = hf->ev_up[p][c];
    }
  }
  for (auto& th : pool) th.join();
  return rc;
}

// the caller's host boxes -> set p's device arrays (on the copy stream, in front of the images they belong to)
int feed_boxes(svo_ctx* ctx, int p, const svo_boxes_host* boxes, int B, svo_boxes_dev* out) {
  HostFeed* hf = feed_of(ctx);
  out->boxes = nullptr; out->n = nullptr; out->stride = 0;
  if (!boxes || !boxes->boxes || !boxes->n || boxes->stride < 1) return SVO_OK;
  int any = 0;
  for (int f = 0; f < B; ++f) {
    if (boxes->n[f] < 0 || boxes->n[f] > SVO_MAX_BOXES || boxes->n[f] > boxes->stride) return SVO_E_INVALID;
    any |= boxes->n[f];
  }
  if (!any) return SVO_OK;
  if (hf->used[p]) SVO_HIP(ctx, hipEventSynchronize(hf->res_done[p]));   // the chain that last read this set's boxes (two calls back)
  int32_t* hb = hf->h_box[p];
  int32_t* hn = hb + (size_t)hf->cap * SVO_MAX_BOXES * 4;
  for (int f = 0; f < B; ++f) {
    hn[f] = boxes->n[f];
    memcpy(hb + (size_t)f * SVO_MAX_BOXES * 4, boxes->boxes + (size_t)f * boxes->stride * 4, 16 * (size_t)boxes->n[f]);
  }
  int32_t* db = hf->d_box[p];
  int32_t* dn = db + (size_t)hf->cap * SVO_MAX_BOXES * 4;
  SVO_HIP(ctx, hipMemcpyAsync(db, hb, (size_t)B * SVO_MAX_BOXES * 16, hipMemcpyHostToDevice, hf->copy));
  SVO_HIP(ctx, hipMemcpyAsync(dn, hn, (size_t)B * 4, hipMemcpyHostToDevice, hf->copy));
  out->boxes = db; out->n = dn; out->stride = SVO_MAX_BOXES;
  return SVO_OK;
}

// records of set p: D2H behind the pose chain; straight into the caller's array when that is pinned, through the set's pinned
// buffer (copied out by svo_hostfeed_flush) otherwise
int feed_results_out(svo_ctx* ctx, int p, int B, svo_track_result* results) {
  HostFeed* hf = feed_of(ctx);
  const bool direct = is_pinned(results);
  SVO_HIP(ctx, hipMemcpyAsync(direct ? results : hf->h_res[p], hf->d_res[p], sizeof(svo_track_result) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipEventRecord(hf->res_done[p], ctx->stream));
  hf->pending[p].user = direct ? nullptr : results;
  hf->pending[p].n = B;
  return SVO_OK;
}

int flush_set(svo_ctx* ctx, HostFeed* hf, int p) {
  if (hf->pending[p].user) {
    SVO_HIP(ctx, hipEventSynchronize(hf->res_done[p]));
    memcpy(hf->pending[p].user, hf->h_res[p], sizeof(svo_track_result) * (size_t)hf->pending[p].n);
    hf->pending[p].user = nullptr;
  }
  HostFeed::FeOut& fo = hf->fe_pending[p];
  if (fo.B > 0) {
    SVO_HIP(ctx, hipEventSynchronize(hf->fe_done[p]));
    const size_t K = ctx->max_kp, B = (size_t)fo.B;
    const uint8_t* h = hf->h_fe[p];
    const size_t o_desc = sizeof(svo_kp) * K * hf->cap, o_n = o_desc + 32 * K * hf->cap, o_uR = o_n + 4 * (size_t)hf->cap, o_dep = o_uR + 4 * K * hf->cap;
    if (fo.kp) memcpy(fo.kp, h, sizeof(svo_kp) * K * B);
    if (fo.desc) memcpy(fo.desc, h + o_desc, 32 * K * B);
    if (fo.n) memcpy(fo.n, h + o_n, 4 * B);
    if (fo.uR) memcpy(fo.uR, h + o_uR, 4 * K * B);
    if (fo.depth) memcpy(fo.depth, h + o_dep, 4 * K * B);
    fo.B = 0;
  }
  return SVO_OK;
}

}  // namespace

// Everything the host-fed entries still owe the caller: records (and front-end outputs) that wait in pinned buffers are copied
// into the arrays they were promised to.  Called by svo_sync, svo_track_reset, svo_destroy and before a set is reused.
int svo_hostfeed_flush(svo_ctx* ctx) {
  HostFeed* hf = feed_of(ctx);
  if (!hf) return SVO_OK;
  for (int p = 0; p < 2; ++p) {
    const int rc = flush_set(ctx, hf, p);
    if (rc) return rc;
  }
  return SVO_OK;
}

void svo_hostfeed_release(svo_ctx* ctx) {
  HostFeed* hf = feed_of(ctx);
  if (!hf) return;
  if (hf->copy) hipStreamSynchronize(hf->copy);
  // (records still waiting in the pinned buffers are DROPPED, not copied out: a caller that destroys the context without svo_sync
  // may have released its result arrays already - include/svo.h promises them after svo_sync only)
  for (int p = 0; p < 2; ++p) {
    if (hf->d_img[p]) hipFree(hf->d_img[p]);
    if (hf->h_img[p]) hipHostFree(hf->h_img[p]);
    if (hf->d_res[p]) hipFree(hf->d_res[p]);
    if (hf->h_res[p]) hipHostFree(hf->h_res[p]);
    if (hf->d_box[p]) hipFree(hf->d_box[p]);
    if (hf->h_box[p]) hipHostFree(hf->h_box[p]);
    if (hf->d_fe[p]) hipFree(hf->d_fe[p]);
    if (hf->h_fe[p]) hipHostFree(hf->h_fe[p]);
    for (hipEvent_t e : hf->ev_up[p]) hipEventDestroy(e);
    for (int k = 0; k < 4; ++k) if (hf->img_free[p][k]) hipEventDestroy(hf->img_free[p][k]);
    if (hf->res_done[p]) hipEventDestroy(hf->res_done[p]);
    if (hf->fe_done[p]) hipEventDestroy(hf->fe_done[p]);
  }
  if (hf->copy) hipStreamDestroy(hf->copy);
  delete hf;
  ctx->hostfeed = nullptr;
}

extern "C" int svo_track_batch_host(svo_ctx* ctx, const uint8_t* grayL, const uint8_t* grayR, int stride, int B,
                                    const svo_boxes_host* boxes, svo_track_result* results) {
  if (!ctx || !grayL || !grayR || !results || B < 1 || stride < ctx->g.W) return SVO_E_INVALID;
  if (B > ctx->max_batch) return SVO_E_CAPACITY;
  if (!ctx->d_track || ctx->n_seq != 1) return SVO_E_INVALID;   // svo_track_reset first
  hipSetDevice(ctx->device);
  { const int rcs = svo_track_fe_batch_stream(ctx); if (rcs) return rcs; }
  int rc = feed_reserve(ctx, ctx->max_batch);
  if (rc) return rc;
  HostFeed* hf = feed_of(ctx);
  const int p = hf->parity;
  if ((rc = flush_set(ctx, hf, p))) return rc;          // what the call two back left in this set's pinned buffers
  svo_boxes_dev bx;
  if ((rc = feed_boxes(ctx, p, boxes, B, &bx))) return rc;
  if ((rc = feed_upload(ctx, p, grayL, grayR, stride, 0, 1, B))) return rc;
  const size_t ib = img_bytes(ctx);
  ctx->feed_pair_event = hf->pair_ev.data();
  rc = svo_track_batch_dev(ctx, hf->d_img[p], hf->d_img[p] + (size_t)hf->cap * ib, ctx->stage_pitch, B, bx.boxes ? &bx : nullptr, hf->d_res[p]);
  ctx->feed_pair_event = nullptr;
  if (rc) return rc;
  // who read the images of this set: the front-end stream (sparse depth), the dense stage's streams and the main stream otherwise
  int nf = 0;
  if (ctx->opt_depth_source == 0) {
    SVO_HIP(ctx, hipEventRecord(hf->img_free[p][nf++], ctx->stream_fe_batch));
  } else {
    if (ctx->stream_dense) SVO_HIP(ctx, hipEventRecord(hf->img_free[p][nf++], ctx->stream_dense));
    if (ctx->stream_elas_a) SVO_HIP(ctx, hipEventRecord(hf->img_free[p][nf++], ctx->stream_elas_a));
    SVO_HIP(ctx, hipEventRecord(hf->img_free[p][nf++], ctx->stream));
  }
  hf->n_free[p] = nf;
  if ((rc = feed_results_out(ctx, p, B, results))) return rc;
  hf->used[p] = true;
  hf->parity ^= 1;
  return SVO_OK;
}

extern "C" int svo_track_sharded_host(svo_ctx* const* ctxs, int G, const uint8_t* grayL, const uint8_t* grayR, int stride, int B,
                                      const svo_boxes_host* boxes, svo_track_result* results) {
  if (!ctxs || G < 1 || G > 64 || !grayL || !grayR || !results || B < 1) return SVO_E_INVALID;
  svo_ctx* c0 = ctxs[0];
  if (!c0 || stride < c0->g.W || !c0->d_track || c0->n_seq != 1) return SVO_E_INVALID;
  for (int g = 0; g < G; ++g) {
    if (!ctxs[g] || ctxs[g]->g.W != c0->g.W || ctxs[g]->g.H != c0->g.H || ctxs[g]->max_kp != c0->max_kp) return SVO_E_INVALID;
    if ((B - g + G - 1) / G > ctxs[g]->max_batch) return SVO_E_CAPACITY;
  }
  hipSetDevice(c0->device);
  { const int rcs = svo_track_fe_batch_stream(c0); if (rcs) return rcs; }
  int rc = SVO_OK;
  // every context uploads ITS pairs (k = g, g + G, ...) to ITS device on ITS copy stream: SURVEY 8e's "pair k -> GPU k mod G"
  std::vector<const uint8_t*> dl((size_t)G), dr((size_t)G);
  std::vector<int> par((size_t)G);
  for (int g = 0; g < G && rc == SVO_OK; ++g) {
    svo_ctx* c = ctxs[g];
    hipSetDevice(c->device);
    // (the records and the boxes live in context 0's feed, which therefore holds at least the call's B frames)
    if ((rc = feed_reserve(c, g == 0 ? std::max(c->max_batch, B) : c->max_batch))) break;
    HostFeed* hf = feed_of(c);
    const int p = hf->parity;
    par[g] = p;
    if ((rc = flush_set(c, hf, p))) break;
    const int nb = (B - g + G - 1) / G;
    if (nb > 0 && (rc = feed_upload(c, p, grayL, grayR, stride, g, G, nb))) break;
    if (nb <= 0) hf->pair_ev.clear();
    dl[g] = hf->d_img[p];
    dr[g] = hf->d_img[p] + (size_t)hf->cap * img_bytes(c);
    c->feed_pair_event = hf->pair_ev.empty() ? nullptr : hf->pair_ev.data();
  }
  hipSetDevice(c0->device);
  HostFeed* h0 = feed_of(c0);
  svo_boxes_dev bx{nullptr, nullptr, 0};
  if (rc == SVO_OK) rc = feed_boxes(c0, par[0], boxes, B, &bx);
  if (rc == SVO_OK)
    rc = svo_track_sharded_dev(ctxs, G, dl.data(), dr.data(), c0->stage_pitch, B, bx.boxes ? &bx : nullptr, h0->d_res[par[0]]);
  for (int g = 0; g < G; ++g) if (ctxs[g]) ctxs[g]->feed_pair_event = nullptr;
  if (rc) return rc;
  for (int g = 0; g < G; ++g) {
    svo_ctx* c = ctxs[g];
    HostFeed* hf = feed_of(c);
    const int p = par[g];
    hf->n_free[p] = 0;
    if ((B - g + G - 1) / G > 0) {
      // context g's front end ran on context 0's confined stream when it shares the tail's device, on its own stream otherwise
      hipStream_t fs = c->device == c0->device ? c0->stream_fe_batch : c->stream;
      hipSetDevice(c->device);
      if (hipEventRecord(hf->img_free[p][0], fs) != hipSuccess) { hipSetDevice(c0->device); c0->last_error = "host feed: event record"; return SVO_E_HIP; }
      hf->n_free[p] = 1;
    }
    hf->used[p] = true;
    hf->parity ^= 1;
  }
  hipSetDevice(c0->device);
  return feed_results_out(c0, par[0], B, results);
}

// The stateless front end alone, host to host and pipelined: uploads on the copy stream, svo_frontend_batch_dev on the image set,
// results D2H behind it.  Outputs (host pointers, any may be NULL) are complete after svo_sync or two calls later.
extern "C" int svo_frontend_batch_host(svo_ctx* ctx, const uint8_t* grayL, const uint8_t* grayR, int stride, int B, const svo_camera* cam,
                                       svo_kp* kpL, uint8_t* descL, int32_t* nL, float* uR, float* depth) {
  if (!ctx || !grayL || !grayR || !cam || B < 1 || stride < ctx->g.W) return SVO_E_INVALID;
  if (B > ctx->max_batch) return SVO_E_CAPACITY;
  hipSetDevice(ctx->device);
  int rc = feed_reserve(ctx, ctx->max_batch);
  if (rc) return rc;
  HostFeed* hf = feed_of(ctx);
  const int p = hf->parity;
  if ((rc = flush_set(ctx, hf, p))) return rc;
  const size_t K = ctx->max_kp, cap = (size_t)hf->cap;
  const size_t o_desc = sizeof(svo_kp) * K * cap, o_n = o_desc + 32 * K * cap, o_uR = o_n + 4 * cap, o_dep = o_uR + 4 * K * cap, total = o_dep + 4 * K * cap;
  for (int q = 0; q < 2; ++q) {
    if (hf->d_fe[q]) continue;
    if (hipMalloc(reinterpret_cast<void**>(&hf->d_fe[q]), total) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&hf->h_fe[q]), total) != hipSuccess) {
      (void)hipGetLastError();
      return SVO_E_NOMEM;
    }
  }
  if ((rc = feed_upload(ctx, p, grayL, grayR, stride, 0, 1, B))) return rc;
  // svo_frontend_batch_dev runs its slices on the context's stream and the slice streams: all of them behind the uploads
  SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, hf->pair_ev[(size_t)B - 1], 0));
  uint8_t* d = hf->d_fe[p];
  const size_t ib = img_bytes(ctx);
  rc = svo_frontend_batch_dev(ctx, hf->d_img[p], hf->d_img[p] + cap * ib, ctx->stage_pitch, B, cam, reinterpret_cast<svo_kp*>(d), d + o_desc,
                              reinterpret_cast<int32_t*>(d + o_n), reinterpret_cast<float*>(d + o_uR), reinterpret_cast<float*>(d + o_dep));
  if (rc) return rc;
  SVO_HIP(ctx, hipEventRecord(hf->img_free[p][0], ctx->stream));   // (the slices' streams joined the context's stream again)
  hf->n_free[p] = 1;
  uint8_t* h = hf->h_fe[p];
  if (kpL) SVO_HIP(ctx, hipMemcpyAsync(h, d, sizeof(svo_kp) * K * B, hipMemcpyDeviceToHost, ctx->stream));
  if (descL) SVO_HIP(ctx, hipMemcpyAsync(h + o_desc, d + o_desc, 32 * K * B, hipMemcpyDeviceToHost, ctx->stream));
  if (nL) SVO_HIP(ctx, hipMemcpyAsync(h + o_n, d + o_n, 4 * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
  if (uR) SVO_HIP(ctx, hipMemcpyAsync(h + o_uR, d + o_uR, 4 * K * B, hipMemcpyDeviceToHost, ctx->stream));
  if (depth) SVO_HIP(ctx, hipMemcpyAsync(h + o_dep, d + o_dep, 4 * K * B, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipEventRecord(hf->fe_done[p], ctx->stream));
  hf->fe_pending[p] = HostFeed::FeOut{kpL, descL, nL, uR, depth, B};
  hf->used[p] = true;
  hf->parity ^= 1;
  return SVO_OK;
}
