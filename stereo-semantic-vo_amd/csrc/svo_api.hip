// svo_api.hip - context, geometry/tables and the C-ABI entry points of include/svo.h.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <functional>
#include <initializer_list>
#include "svo_internal.h"

// ---- profiling ---------------------------------------------------------------------
// Event pairs are recorded around a stage on the ctx stream and resolved lazily.
struct PendingEvt { int entry; hipEvent_t a, b; };

struct SvoProfState {
  std::vector<PendingEvt> pending;
  std::vector<hipEvent_t> pool;
};
static SvoProfState* prof_state(svo_ctx* ctx) {
  if (!ctx->prof_impl) ctx->prof_impl = new SvoProfState();
  return reinterpret_cast<SvoProfState*>(ctx->prof_impl);
}
static void prof_resolve(svo_ctx* ctx) {
  SvoProfState* ps = prof_state(ctx);
  for (auto& p : ps->pending) {
    hipEventSynchronize(p.b);
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->prof[p.entry].total_ms += ms;
      ctx->prof[p.entry].launches += 1;
    }
    ps->pool.push_back(p.a);
    ps->pool.push_back(p.b);
  }
  ps->pending.clear();
}
static hipEvent_t prof_event(SvoProfState* ps) {
  if (!ps->pool.empty()) {
    hipEvent_t e = ps->pool.back();
    ps->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
SvoTimer::SvoTimer(svo_ctx* c, const char* n, hipStream_t s) : ctx(c), name(n), stream(s ? s : c->stream) {
  if (!ctx->profiling) return;
  SvoProfState* ps = prof_state(ctx);
  if (ps->pending.size() > 8192) prof_resolve(ctx);
  int entry = -1;
  for (size_t i = 0; i < ctx->prof.size(); ++i)
    if (ctx->prof[i].name == n) entry = (int)i;
  if (entry < 0) {
    SvoProfileEntry e;
    e.name = n;
    ctx->prof.push_back(e);
    entry = (int)ctx->prof.size() - 1;
  }
  PendingEvt p{entry, prof_event(ps), prof_event(ps)};
  hipEventRecord(p.a, stream);
  ps->pending.push_back(p);
}
SvoTimer::~SvoTimer() {
  if (!ctx->profiling) return;
  SvoProfState* ps = prof_state(ctx);
  hipEventRecord(ps->pending.back().b, stream);
}

// ---- geometry + tables ---------------------------------------------------------------
static inline int cv_round_f(float v) { return (int)lrintf(v); }

static void build_geometry(SvoGeom& g, int W, int H, int nfeatures) {
  memset(&g, 0, sizeof g);
  g.W = W; g.H = H;
  for (int l = 0; l < SVO_NLEVELS; ++l) {
    g.scale[l] = (float)pow((double)1.2f, (double)l);
    const float inv = 1.0f / g.scale[l];
    g.w[l] = cv_round_f((float)W * inv);
    g.h[l] = cv_round_f((float)H * inv);
  }
  const float factor = (float)(1.0 / (double)1.2f);
  float ndesired = (float)nfeatures * (1.0f - factor) /
                   (1.0f - (float)pow((double)factor, (double)SVO_NLEVELS));
  int sum = 0;
  for (int l = 0; l < SVO_NLEVELS - 1; ++l) {
    g.quota[l] = cv_round_f(ndesired);
    sum += g.quota[l];
    ndesired *= factor;
  }
  g.quota[SVO_NLEVELS - 1] = std::max(nfeatures - sum, 0);
  int64_t off = 0, coff = 0;
  int xt = 0, yt = 0, tb = 0;
  for (int l = 0; l < SVO_NLEVELS; ++l) {
    g.pitch[l] = (g.w[l] + 63) / 64 * 64;
    g.loff[l] = off;
    if (l > 0) off += (int64_t)g.pitch[l] * g.h[l];
    off = (off + 255) / 256 * 256;
    const int iw = std::max(g.w[l] - 2 * SVO_EDGE, 0), ih = std::max(g.h[l] - 2 * SVO_EDGE, 0);
    g.cap[l] = (iw / 2 + 1) * (ih / 2 + 1);
    g.coff[l] = coff;
    coff += g.cap[l];
    g.tiles_x[l] = iw > 0 && ih > 0 ? (g.w[l] - SVO_EDGE - 28 + FAST_TW - 1) / FAST_TW : 0;
    g.tiles_y[l] = iw > 0 && ih > 0 ? (ih + FAST_TH - 1) / FAST_TH : 0;
    g.tile_base[l] = tb;
    tb += g.tiles_x[l] * g.tiles_y[l];
    g.xtab_off[l] = xt; g.ytab_off[l] = yt;
    xt += (g.w[l] + 3) / 4 * 4 + 4; yt += g.h[l];   // x tables padded for 16-byte loads
  }
  g.tile_base[SVO_NLEVELS] = tb;
  g.pyr_bytes = off;
  g.corner_entries = coff;
}

// cv::resize INTER_LINEAR coefficient tables (fixed point, 11 bits)
static void build_resize_tables(const SvoGeom& g, std::vector<int32_t>& xofs,
                                std::vector<int32_t>& xalpha, std::vector<int32_t>& yofs,
                                std::vector<int32_t>& ybeta) {
  int xt = 0, yt = 0;
  for (int l = 0; l < SVO_NLEVELS; ++l) { xt += (g.w[l] + 3) / 4 * 4 + 4; yt += g.h[l]; }
  xofs.assign(xt, 0); xalpha.assign(xt, 0); yofs.assign(yt, 0); ybeta.assign(yt, 0);
  for (int l = 1; l < SVO_NLEVELS; ++l) {
    const int sw = g.w[l - 1], sh = g.h[l - 1], dw = g.w[l], dh = g.h[l];
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    for (int dx = 0; dx < dw; ++dx) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = (int)floorf(fx);
      fx -= (float)sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
      const int a0 = (short)cv_round_f((1.f - fx) * 2048.f), a1 = (short)cv_round_f(fx * 2048.f);
      xofs[g.xtab_off[l] + dx] = sx;
      xalpha[g.xtab_off[l] + dx] = (a0 & 0xffff) | (a1 << 16);
    }
    // padding entries (read by the last 16-byte table load of a row) repeat the last column
    for (int dx = dw; dx < (dw + 3) / 4 * 4 + 4; ++dx) {
      xofs[g.xtab_off[l] + dx] = xofs[g.xtab_off[l] + dw - 1];
      xalpha[g.xtab_off[l] + dx] = xalpha[g.xtab_off[l] + dw - 1];
    }
    for (int dy = 0; dy < dh; ++dy) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = (int)floorf(fy);
      fy -= (float)sy;
      if (sy < 0) { fy = 0; sy = 0; }
      if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
      const int b0 = (short)cv_round_f((1.f - fy) * 2048.f), b1 = (short)cv_round_f(fy * 2048.f);
      yofs[g.ytab_off[l] + dy] = sy;
      ybeta[g.ytab_off[l] + dy] = (b0 & 0xffff) | (b1 << 16);
    }
  }
}

// ---- lifecycle -------------------------------------------------------------------------
extern "C" int svo_abi_version(void) { return SVO_ABI_VERSION; }

extern "C" const char* svo_strerror(int status) {
  switch (status) {
    case SVO_OK: return "ok";
    case SVO_E_INVALID: return "invalid argument";
    case SVO_E_NODEVICE: return "no usable HIP device";
    case SVO_E_NOMEM: return "out of memory";
    case SVO_E_HIP: return "HIP runtime error";
    case SVO_E_CAPACITY: return "capacity exceeded";
    case SVO_E_TIMEOUT: return "a wait inside the tracker's pose chain timed out";
    default: return "unknown status";
  }
}
extern "C" const char* svo_last_error(const svo_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

template <typename T>
static int dalloc(svo_ctx* ctx, T** p, size_t count) {
  void* v = nullptr;
  hipError_t e = hipMalloc(&v, std::max<size_t>(count * sizeof(T), 256));
  if (e != hipSuccess) {
    ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
    return SVO_E_NOMEM;
  }
  *p = reinterpret_cast<T*>(v);
  return SVO_OK;
}

extern "C" int svo_create(svo_ctx** out, int device, int W, int H, int max_kp, int max_batch) {
  // the environment's say is read per call (not once per process): SVO_POOLED_QUEUES=1 = svo_create_ex(..., SVO_CREATE_POOLED_STREAMS)
  const char* e = getenv("SVO_POOLED_QUEUES");
  return svo_create_ex(out, device, W, H, max_kp, max_batch, (e && e[0] == '1') ? SVO_CREATE_POOLED_STREAMS : 0u);
}

extern "C" int svo_stream_mode(const svo_ctx* ctx) {
  if (!ctx) return SVO_E_INVALID;
  return (ctx->streams_burst ? 1 : 0) | (ctx->streams_burst && !(ctx->create_flags & SVO_CREATE_TAIL_ALL_CUS) ? 2 : 0);
}

extern "C" int svo_create_ex(svo_ctx** out, int device, int W, int H, int max_kp, int max_batch, uint32_t flags) {
  if (flags & ~(uint32_t)(SVO_CREATE_POOLED_STREAMS | SVO_CREATE_TAIL_ALL_CUS)) return SVO_E_INVALID;
  if (!out || W < 96 || H < 96 || W > 4095 || H > 4095 || max_kp < 8 || max_kp > 512 ||
      max_batch < 1)
    return SVO_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return SVO_E_NODEVICE;
  if (hipSetDevice(device) != hipSuccess) return SVO_E_NODEVICE;
  svo_ctx* ctx = new svo_ctx();
  ctx->create_flags = flags;
  ctx->device = device;
  ctx->max_kp = max_kp;
  ctx->max_batch = max_batch;
  ctx->max_images = 2 * max_batch;
  build_geometry(ctx->g, W, H, max_kp);
  const SvoGeom& g = ctx->g;
  for (int l = 0; l < SVO_NLEVELS; ++l)
    if (g.quota[l] > SVO_QMAX) { delete ctx; return SVO_E_INVALID; }
  if (svo_stream_burst(ctx) != SVO_OK) {
    delete ctx;
    return SVO_E_NODEVICE;
  }
  std::vector<int32_t> xofs, xalpha, yofs, ybeta;
  build_resize_tables(g, xofs, xalpha, yofs, ybeta);
  {
    // the pyramid in three launches: levels (1, 2), (3, 4), (5, 6, 7) - each group's rows of the intermediate levels stay
    // in LDS (k_pyr_fused).  cap = the most rows of an intermediate level any strip needs.
    const int groups[3][3] = {{1, 2, 16}, {3, 2, 16}, {5, 3, 12}};   // first level, levels, rows of the last level per strip
    for (const auto& gr : groups) {
      SvoPyrGroup pg{gr[0], gr[1], gr[2], 0, 0, 0};
      if (pg.l0 + pg.nl > SVO_NLEVELS) continue;
      const int lt = pg.l0 + pg.nl - 1;
      for (int r0 = 0; r0 < g.h[lt]; r0 += pg.rt) {
        int lo = r0, hi = std::min(g.h[lt], r0 + pg.rt);
        for (int k = pg.nl - 2; k >= 0; --k) {
          const int child = pg.l0 + k + 1;
          const int nlo = lo == 0 ? 0 : yofs[g.ytab_off[child] + lo];
          const int nhi = hi == g.h[child] ? g.h[pg.l0 + k] : std::min(yofs[g.ytab_off[child] + hi - 1] + 2, g.h[pg.l0 + k]);
          lo = nlo; hi = nhi;
          (k == 0 ? pg.cap0 : pg.cap1) = std::max(k == 0 ? pg.cap0 : pg.cap1, hi - lo);
        }
      }
      pg.lds = 4 * ((size_t)pg.cap0 * (g.pitch[pg.l0] / 4) + 4 + (pg.nl == 3 ? (size_t)pg.cap1 * (g.pitch[pg.l0 + 1] / 4) + 4 : 0));
      ctx->pyr_plan.push_back(pg);
    }
    for (const SvoPyrGroup& pg : ctx->pyr_plan)
      if (pg.lds > 64 * 1024) { ctx->pyr_plan.clear(); break; }   // (very wide images: one launch per level)
  }
  const size_t I = (size_t)ctx->max_images;
  ctx->stage_pitch = (W + 63) / 64 * 64;
  ctx->scratch_bytes = 16u << 20;
  ctx->pinned_bytes = 8u << 20;
  ctx->opt_shard_force_staged = getenv("SVO_SHARD_FORCE_STAGED") != nullptr;
  int rc = SVO_OK;
#define TRY(x) if (rc == SVO_OK) rc = (x)
  TRY(dalloc(ctx, &ctx->d_xofs, xofs.size()));
  TRY(dalloc(ctx, &ctx->d_xalpha, xalpha.size()));
  TRY(dalloc(ctx, &ctx->d_yofs, yofs.size()));
  TRY(dalloc(ctx, &ctx->d_ybeta, ybeta.size()));
  TRY(dalloc(ctx, &ctx->d_stage, I * (size_t)H * ctx->stage_pitch));
  TRY(dalloc(ctx, &ctx->d_pyr, I * (size_t)g.pyr_bytes));
  TRY(dalloc(ctx, &ctx->d_corners, I * (size_t)g.corner_entries));
  TRY(dalloc(ctx, &ctx->d_counters, I * SVO_NLEVELS));
  TRY(dalloc(ctx, &ctx->d_hist, I * SVO_NLEVELS * 256));
  if (rc == SVO_OK && hipMemsetAsync(ctx->d_hist, 0, sizeof(int32_t) * I * SVO_NLEVELS * 256, ctx->stream) != hipSuccess) rc = SVO_E_HIP;   // k_select keeps it zero
  TRY(dalloc(ctx, &ctx->d_sel, I * SVO_NLEVELS * SVO_QMAX));
  TRY(dalloc(ctx, &ctx->d_selcnt, I * SVO_NLEVELS));
  TRY(dalloc(ctx, &ctx->d_kp, I * (size_t)max_kp));
  TRY(dalloc(ctx, &ctx->d_desc, I * (size_t)max_kp * 32));
  TRY(dalloc(ctx, &ctx->d_nkp, I));
  TRY(dalloc(ctx, &ctx->d_uR, (size_t)max_batch * max_kp));
  TRY(dalloc(ctx, &ctx->d_depth, (size_t)max_batch * max_kp));
  TRY(dalloc(ctx, &ctx->d_sad, (size_t)max_batch * max_kp));
  TRY(dalloc(ctx, &ctx->d_scratch, ctx->scratch_bytes));
  TRY(dalloc(ctx, &ctx->d_pnp_subsets, (size_t)513 * 500));
#undef TRY
  if (rc == SVO_OK && hipHostMalloc((void**)&ctx->h_stage, 2 * (size_t)H * ctx->stage_pitch) != hipSuccess)
    rc = SVO_E_HIP;
  if (rc == SVO_OK) memset(ctx->h_stage, 0, 2 * (size_t)H * ctx->stage_pitch);
  if (rc == SVO_OK && hipHostMalloc((void**)&ctx->h_pinned, ctx->pinned_bytes) != hipSuccess)
    rc = SVO_E_NOMEM;
  if (rc == SVO_OK) {
    hipMemcpyAsync(ctx->d_xofs, xofs.data(), xofs.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(ctx->d_xalpha, xalpha.data(), xalpha.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(ctx->d_yofs, yofs.data(), yofs.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(ctx->d_ybeta, ybeta.data(), ybeta.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    hipMemsetAsync(ctx->d_nkp, 0, I * 4, ctx->stream);
    {   // RANSAC sample indices of cv::solvePnPRansac for every possible point count (they depend on nothing else)
      std::vector<uint16_t> sub((size_t)513 * 500, 0);
      for (int n = 5; n <= 512; ++n) svo_pnp_subsets(0, n, &sub[(size_t)n * 500]);
      hipMemcpyAsync(ctx->d_pnp_subsets, sub.data(), sub.size() * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream);
      hipStreamSynchronize(ctx->stream);   // (`sub` lives in this block)
    }
    hipMemsetAsync(ctx->d_selcnt, 0, I * SVO_NLEVELS * 4, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = SVO_E_HIP;   // (tables and memsets ran on the context's own stream - no NULL-stream call - and are waited for: the other streams of the context do not order against it)
  }
  if (rc != SVO_OK) {
    svo_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return SVO_OK;
}

extern "C" void svo_destroy(svo_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  // Batched and sharded calls leave work in flight that reads this context's buffers from OTHER streams (the tail context's
  // gather stream and front-end stream when this context is a producer, a peer copy or a bounce through pinned memory):
  // hipFree's implicit synchronisation does not cover those.  Wait for all of it, then for every stream of the context.
  (void)svo_track_quiesce(ctx, true);
  if (ctx->stream_fe_batch) hipStreamSynchronize(ctx->stream_fe_batch);
  if (ctx->stream_dense) hipStreamSynchronize(ctx->stream_dense);
  for (hipStream_t st : ctx->fe_streams) hipStreamSynchronize(st);
  if (ctx->stream_idx) hipStreamSynchronize(ctx->stream_idx);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  (void)hipGetLastError();
  svo_hostfeed_release(ctx);
  svo_track_release(ctx);
  svo_elas_release(ctx);
  svo_msa_release(ctx);
  if (ctx->d_dense) hipFree(ctx->d_dense);
  void* ptrs[] = {ctx->d_xofs, ctx->d_xalpha, ctx->d_yofs, ctx->d_ybeta, ctx->d_stage, ctx->d_pyr,
                  ctx->d_corners, ctx->d_counters, ctx->d_hist, ctx->d_sel, ctx->d_selcnt,
                  ctx->d_kp, ctx->d_desc, ctx->d_nkp, ctx->d_uR, ctx->d_depth, ctx->d_sad,
                  ctx->d_scratch, ctx->d_track, ctx->d_pnp_subsets};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
  if (ctx->h_stage) hipHostFree(ctx->h_stage);
  if (ctx->prof_impl) {
    SvoProfState* ps = reinterpret_cast<SvoProfState*>(ctx->prof_impl);
    for (auto& p : ps->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : ps->pool) hipEventDestroy(e);
    delete ps;
  }
  for (hipStream_t st : ctx->fe_streams) hipStreamDestroy(st);
  for (hipEvent_t e : ctx->fe_events) hipEventDestroy(e);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  for (hipStream_t st : ctx->parked_streams) hipStreamDestroy(st);
  ctx->parked_streams.clear();
  delete ctx;
}

// svo_track_batch_dev leaves its tail in flight and lets the next call's front end run beside it (two alternating output sets,
// two halves of the work records).  Any OTHER entry that touches the streams, the result arrays or the options they were
// enqueued under first waits for those tails - the overlap is a contract between consecutive batch calls only.
int svo_track_quiesce(svo_ctx* ctx, bool shard_too) {
  if (shard_too) { const int rcs = svo_shard_quiesce(ctx); if (rcs) return rcs; }
  bool pending = false;
  for (int q = 0; q < 2; ++q) {
    if (!ctx->tb_used[q] || !ctx->tb_done[q]) continue;
    pending = true;
    SVO_HIP(ctx, hipEventSynchronize(ctx->tb_done[q]));
  }
  if (pending && ctx->stream_fe_batch) SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_fe_batch));
  if (pending && ctx->stream_dense) SVO_HIP(ctx, hipStreamSynchronize(ctx->stream_dense));
  return SVO_OK;
}

extern "C" int svo_set_option(svo_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return SVO_E_INVALID;
  // Options are read on the host when work is enqueued, so what is already in flight keeps the values it was enqueued under.
  // Only the two options that REPLACE a stream wait for the tails a batched / sharded call left on it.
  if (!strcmp(key, "fe_cu_percent") || !strcmp(key, "dense_cu_percent")) {
    if (value < 10 || value > 100) return SVO_E_INVALID;
    const int rcq = svo_track_quiesce(ctx);
    if (rcq) return rcq;
  }
  if (!strcmp(key, "pose_mfma")) { if (value < 0 || value > 2) return SVO_E_INVALID; ctx->opt_pose_mfma = value; return SVO_OK; }
  if (!strcmp(key, "fast_cand_cap")) {
    if (value < 0 || value > 2048) return SVO_E_INVALID;
    ctx->opt_fast_cand_cap = value;
    return SVO_OK;
  }
  if (!strcmp(key, "fe_cu_percent")) {
    if (value < 10 || value > 100) return SVO_E_INVALID;
    ctx->opt_fe_cu_percent = value;
    if (ctx->stream_fe_batch) { hipStreamSynchronize(ctx->stream_fe_batch); hipStreamDestroy(ctx->stream_fe_batch); ctx->stream_fe_batch = nullptr; }   // recreated on demand
    return SVO_OK;
  }
  if (!strcmp(key, "dense_cu_percent")) {
    if (value < 10 || value > 100) return SVO_E_INVALID;
    ctx->opt_dense_cu_percent = value;
    if (ctx->stream_dense) { hipStreamSynchronize(ctx->stream_dense); hipStreamDestroy(ctx->stream_dense); ctx->stream_dense = nullptr; }   // recreated on demand
    return SVO_OK;
  }
  if (!strcmp(key, "tail_semi")) { if (value < 0 || value > 2) return SVO_E_INVALID; ctx->opt_tail_semi = value; return SVO_OK; }
  if (!strcmp(key, "tail_fused")) { if (value < 0 || value > 2) return SVO_E_INVALID; ctx->opt_tail_fused = value; return SVO_OK; }   // 2: also beside a dense stage
  if (!strcmp(key, "pose_flag")) { ctx->opt_pose_flag = value != 0; return SVO_OK; }
  if (!strcmp(key, "gate_group")) { ctx->opt_gate_group = value != 0; return SVO_OK; }
  if (!strcmp(key, "hyp_first")) { if (value < 4 || value > 16 || (value & 3)) return SVO_E_INVALID; ctx->opt_hyp_first = value; return SVO_OK; }
  if (!strcmp(key, "dense_two_launch")) { ctx->opt_dense_two_launch = value != 0; return SVO_OK; }
  if (!strcmp(key, "epnp_exact")) {
    // 0 the statistical solver, 2 the order-preserving one (default); every other value is ABI 3's "non-zero": the one-lane checker
    ctx->opt_epnp_exact = value == 0 ? 0 : value == 2 ? 2 : 1;
    return SVO_OK;
  }
  if (!strcmp(key, "debug_lose_sample")) { if (value < 0 || value > 100) return SVO_E_INVALID; ctx->opt_debug_lose_sample = value; return SVO_OK; }
  if (!strcmp(key, "epnp_force_seq")) { ctx->opt_epnp_force_seq = value != 0; return SVO_OK; }
  if (!strcmp(key, "shard_force_staged")) { ctx->opt_shard_force_staged = value != 0; return SVO_OK; }
  if (!strcmp(key, "multi_pipeline")) { ctx->opt_multi_pipeline = value != 0; return SVO_OK; }
  if (!strcmp(key, "pyr_fused")) { ctx->opt_pyr_fused = value != 0; return SVO_OK; }
  if (!strcmp(key, "frontend_overlap")) {
    if (value < 0 || value > 8) return SVO_E_INVALID;
    ctx->opt_frontend_overlap = value;
    return SVO_OK;
  }
  if (!strcmp(key, "track_nblk")) {
    if (value < 0 || value > 3) return SVO_E_INVALID;
    ctx->opt_track_nblk = value;
    return SVO_OK;
  }
  if (!strcmp(key, "track_group")) {
    if (value < 1 || value > 64) return SVO_E_INVALID;
    ctx->opt_track_group = value;
    return SVO_OK;
  }
  if (!strcmp(key, "track_lcap")) {
    if (value < 1 || value > 8) return SVO_E_INVALID;
    ctx->opt_track_lcap = value;
    return SVO_OK;
  }
  if (!strcmp(key, "depth_source")) {
    if (value < 0 || value > 2) return SVO_E_INVALID;   // 0 sparse matcher, 1 ELAS map, 2 MSA map
    ctx->opt_depth_source = value;
    return SVO_OK;
  }
  return SVO_E_INVALID;
}
extern "C" int svo_sync(svo_ctx* ctx) {
  if (!ctx) return SVO_E_INVALID;
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int rcf = svo_hostfeed_flush(ctx);   // (records of host-fed calls: out of the pinned buffers into the caller's arrays)
  if (rcf) return rcf;
  return svo_track_check_timeout(ctx);
}
extern "C" void* svo_stream(svo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int svo_orb_geometry(const svo_ctx* ctx, int32_t w[8], int32_t h[8], float scale[8],
                                int32_t quota[8]) {
  if (!ctx) return SVO_E_INVALID;
  for (int l = 0; l < SVO_NLEVELS; ++l) {
    w[l] = ctx->g.w[l]; h[l] = ctx->g.h[l]; scale[l] = ctx->g.scale[l]; quota[l] = ctx->g.quota[l];
  }
  return SVO_OK;
}

// ---- host-path helpers -------------------------------------------------------------------
// Host image -> staging slot in HBM, through pinned memory: a 2-D copy straight from pageable memory
// takes ~14 ms per 1241x376 image on this stack, rows gathered into a pinned buffer + one linear copy
// ~0.1 ms.  The pinned buffer is reused by the next call; every host-path entry point synchronises
// before it returns.
int svo_upload_image(svo_ctx* ctx, const uint8_t* gray, int stride, int slot) {
  const size_t bytes = (size_t)ctx->g.H * ctx->stage_pitch;
  uint8_t* dst = ctx->d_stage + (size_t)slot * bytes;
  if (!ctx->h_stage || slot < 0 || slot >= 2) {
    SVO_HIP(ctx, hipMemcpy2DAsync(dst, ctx->stage_pitch, gray, stride, ctx->g.W, ctx->g.H,
                                  hipMemcpyHostToDevice, ctx->stream));
    return SVO_OK;
  }
  uint8_t* h = ctx->h_stage + (size_t)slot * bytes;
  for (int y = 0; y < ctx->g.H; ++y) memcpy(h + (size_t)y * ctx->stage_pitch, gray + (size_t)y * stride, ctx->g.W);
  SVO_HIP(ctx, hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, ctx->stream));
  return SVO_OK;
}
static int upload_image(svo_ctx* ctx, const uint8_t* gray, int stride, int slot) {
  return svo_upload_image(ctx, gray, stride, slot);
}
// scratch bump allocator (first half of d_scratch; second half belongs to the launchers)
struct Bump {
  svo_ctx* ctx; size_t off = 0;
  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~size_t(255);
    T* p = reinterpret_cast<T*>(ctx->d_scratch + off);
    off += count * sizeof(T);
    return p;
  }
  bool ok() const { return off <= ctx->scratch_bytes / 2; }
};

extern "C" int svo_orb_extract(svo_ctx* ctx, const uint8_t* gray, int stride, svo_kp* kp,
                               uint8_t* desc, int32_t* n) {
  if (!ctx || !gray || !kp || !desc || !n || stride < ctx->g.W) return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  int rc = upload_image(ctx, gray, stride, 0);
  if (rc) return rc;
  rc = svo_launch_orb(ctx, ctx->d_stage, ctx->d_stage, ctx->stage_pitch, 1, 1);
  if (rc) return rc;
  int32_t cnt = 0;
  SVO_HIP(ctx, hipMemcpyAsync(&cnt, ctx->d_nkp, 4, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(kp, ctx->d_kp, sizeof(svo_kp) * ctx->max_kp, hipMemcpyDeviceToHost,
                              ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(desc, ctx->d_desc, 32 * (size_t)ctx->max_kp, hipMemcpyDeviceToHost,
                              ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n = cnt;
  return SVO_OK;
}

extern "C" int svo_debug_pyramid_level(svo_ctx* ctx, int image_slot, int level, uint8_t* out) {
  if (!ctx || !out || level < 0 || level >= SVO_NLEVELS || image_slot < 0 ||
      image_slot >= ctx->max_images)
    return SVO_E_INVALID;
  const SvoGeom& g = ctx->g;
  const uint8_t* src;
  int pitch;
  if (level == 0) {
    src = ctx->d_stage + (size_t)image_slot * g.H * ctx->stage_pitch;
    pitch = ctx->stage_pitch;
  } else {
    src = ctx->d_pyr + (size_t)image_slot * g.pyr_bytes + g.loff[level];
    pitch = g.pitch[level];
  }
  SVO_HIP(ctx, hipMemcpy2DAsync(out, g.w[level], src, pitch, g.w[level], g.h[level],
                                hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_debug_fast_corners(svo_ctx* ctx, int image_slot, int level, int32_t* xy_score,
                                      int capacity, int32_t* n) {
  if (!ctx || !xy_score || !n || level < 0 || level >= SVO_NLEVELS || image_slot < 0 ||
      image_slot >= ctx->max_images)
    return SVO_E_INVALID;
  const SvoGeom& g = ctx->g;
  int32_t cnt = 0;
  SVO_HIP(ctx, svo_memcpy_sync(ctx, &cnt, ctx->d_counters + image_slot * SVO_NLEVELS + level, 4,
                         hipMemcpyDeviceToHost));
  cnt = std::min(cnt, capacity);
  std::vector<uint32_t> raw((size_t)std::max(cnt, 1));
  if (cnt > 0)
    SVO_HIP(ctx, svo_memcpy_sync(ctx, raw.data(),
                           ctx->d_corners + (size_t)image_slot * g.corner_entries + g.coff[level],
                           4 * (size_t)cnt, hipMemcpyDeviceToHost));
  for (int i = 0; i < cnt; ++i) {
    xy_score[3 * i] = raw[i] & 0xfff;
    xy_score[3 * i + 1] = (raw[i] >> 12) & 0xfff;
    xy_score[3 * i + 2] = raw[i] >> 24;
  }
  *n = cnt;
  return SVO_OK;
}

extern "C" int svo_stereo_frame_ex(svo_ctx* ctx, const uint8_t* grayL, int strideL,
                                   const uint8_t* grayR, int strideR, const svo_camera* cam,
                                   svo_kp* kpL, uint8_t* descL, int32_t* nL, float* uR,
                                   float* depth, svo_kp* kpR, uint8_t* descR, int32_t* nR) {
  if (!ctx || !grayL || !grayR || !cam || !kpL || !descL || !nL || !uR || !depth ||
      strideL < ctx->g.W || strideR < ctx->g.W)
    return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  // slots: left -> image 0, right -> image 1 (B = 1)
  int rc = upload_image(ctx, grayL, strideL, 0);
  if (rc) return rc;
  rc = upload_image(ctx, grayR, strideR, 1);
  if (rc) return rc;
  const uint8_t* dL = ctx->d_stage;
  const uint8_t* dR = ctx->d_stage + (size_t)ctx->g.H * ctx->stage_pitch;
  rc = svo_launch_orb(ctx, dL, dR, ctx->stage_pitch, 1, 2);
  if (rc) return rc;
  rc = svo_launch_stereo(ctx, dL, dR, ctx->stage_pitch, 1, cam);
  if (rc) return rc;
  int32_t cnt[2] = {0, 0};
  const size_t K = ctx->max_kp;
  SVO_HIP(ctx, hipMemcpyAsync(cnt, ctx->d_nkp, 8, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(kpL, ctx->d_kp, sizeof(svo_kp) * K, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(descL, ctx->d_desc, 32 * K, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(uR, ctx->d_uR, 4 * K, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipMemcpyAsync(depth, ctx->d_depth, 4 * K, hipMemcpyDeviceToHost, ctx->stream));
  if (kpR) SVO_HIP(ctx, hipMemcpyAsync(kpR, ctx->d_kp + K, sizeof(svo_kp) * K, hipMemcpyDeviceToHost, ctx->stream));
  if (descR) SVO_HIP(ctx, hipMemcpyAsync(descR, ctx->d_desc + 32 * K, 32 * K, hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *nL = cnt[0];
  if (nR) *nR = cnt[1];
  return SVO_OK;
}

extern "C" int svo_stereo_frame(svo_ctx* ctx, const uint8_t* grayL, int strideL,
                                const uint8_t* grayR, int strideR, const svo_camera* cam,
                                svo_kp* kpL, uint8_t* descL, int32_t* nL, float* uR, float* depth) {
  return svo_stereo_frame_ex(ctx, grayL, strideL, grayR, strideR, cam, kpL, descL, nL, uR, depth,
                             nullptr, nullptr, nullptr);
}

#define H2D(dst, src, bytes) SVO_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream))
#define D2H(dst, src, bytes) SVO_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream))

extern "C" int svo_disp2depth(svo_ctx* ctx, const float* disp, int count, float bf, float* depth) {
  if (!ctx || !disp || !depth || count < 0) return SVO_E_INVALID;
  if (count == 0) return SVO_OK;
  hipSetDevice(ctx->device);
  Bump b{ctx};
  float* d_in = b.take<float>(count);
  float* d_out = b.take<float>(count);
  if (!b.ok()) return SVO_E_CAPACITY;
  H2D(d_in, disp, 4 * (size_t)count);
  int rc = svo_launch_disp2depth(ctx, d_in, count, bf, d_out);
  if (rc) return rc;
  D2H(depth, d_out, 4 * (size_t)count);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_unproject(svo_ctx* ctx, const float* uvz, int n, const svo_camera* cam,
                             const float Rwc[9], const float twc[3], float* xyz) {
  if (!ctx || !uvz || !cam || !Rwc || !twc || !xyz || n < 0) return SVO_E_INVALID;
  if (n == 0) return SVO_OK;
  hipSetDevice(ctx->device);
  Bump b{ctx};
  float* d_in = b.take<float>(3 * (size_t)n);
  float* d_out = b.take<float>(3 * (size_t)n);
  float* d_R = b.take<float>(9);
  float* d_t = b.take<float>(3);
  if (!b.ok()) return SVO_E_CAPACITY;
  H2D(d_in, uvz, 12 * (size_t)n);
  H2D(d_R, Rwc, 36);
  H2D(d_t, twc, 12);
  int rc = svo_launch_unproject(ctx, d_in, n, cam, d_R, d_t, d_out);
  if (rc) return rc;
  D2H(xyz, d_out, 12 * (size_t)n);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_descriptor_distance(svo_ctx* ctx, const uint8_t* a, const uint8_t* b, int count,
                                       int32_t* dist) {
  if (!ctx || !a || !b || !dist || count < 0) return SVO_E_INVALID;
  if (count == 0) return SVO_OK;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  uint8_t* da = bm.take<uint8_t>(32 * (size_t)count);
  uint8_t* db = bm.take<uint8_t>(32 * (size_t)count);
  int32_t* dd = bm.take<int32_t>(count);
  if (!bm.ok()) return SVO_E_CAPACITY;
  H2D(da, a, 32 * (size_t)count);
  H2D(db, b, 32 * (size_t)count);
  int rc = svo_launch_descriptor_distance(ctx, da, db, count, dd);
  if (rc) return rc;
  D2H(dist, dd, 4 * (size_t)count);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_hamming_argmin(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                                  const uint8_t* t_mask, int32_t* best_idx, int32_t* best,
                                  int32_t* second) {
  if (!ctx || M < 0 || N < 0 || (M > 0 && (!q || !best_idx || !best || !second)) || (N > 0 && !t))
    return SVO_E_INVALID;
  if (M == 0) return SVO_OK;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  uint8_t* dq = bm.take<uint8_t>(32 * (size_t)M);
  uint8_t* dt = bm.take<uint8_t>(32 * (size_t)std::max(N, 1));
  uint8_t* dm = bm.take<uint8_t>(std::max(N, 1));
  int32_t* di = bm.take<int32_t>(M);
  int32_t* db = bm.take<int32_t>(M);
  int32_t* ds = bm.take<int32_t>(M);
  if (!bm.ok()) return SVO_E_CAPACITY;
  H2D(dq, q, 32 * (size_t)M);
  if (N > 0) H2D(dt, t, 32 * (size_t)N);
  if (t_mask && N > 0) H2D(dm, t_mask, N);
  int rc = svo_launch_hamming_argmin(ctx, dq, M, dt, N, t_mask ? dm : nullptr, di, db, ds);
  if (rc) return rc;
  D2H(best_idx, di, 4 * (size_t)M);
  D2H(best, db, 4 * (size_t)M);
  D2H(second, ds, 4 * (size_t)M);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_match_greedy(svo_ctx* ctx, const uint8_t* q, const uint8_t* q_skip, int M,
                                const uint8_t* t, int N, uint8_t* assigned, int max_dist,
                                float ratio, int32_t* best_idx, int32_t* best, int32_t* second,
                                uint8_t* accepted) {
  if (!ctx || M < 0 || N < 0 || (M > 0 && (!q || !best_idx || !best || !second || !accepted)) ||
      (N > 0 && (!t || !assigned)))
    return SVO_E_INVALID;
  if (M == 0) return SVO_OK;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  uint8_t* dq = bm.take<uint8_t>(32 * (size_t)M);
  uint8_t* dsk = bm.take<uint8_t>(M);
  uint8_t* dt = bm.take<uint8_t>(32 * (size_t)std::max(N, 1));
  uint8_t* das = bm.take<uint8_t>(std::max(N, 1));
  int32_t* di = bm.take<int32_t>(M);
  int32_t* db = bm.take<int32_t>(M);
  int32_t* ds = bm.take<int32_t>(M);
  uint8_t* dacc = bm.take<uint8_t>(M);
  if (!bm.ok()) return SVO_E_CAPACITY;
  H2D(dq, q, 32 * (size_t)M);
  if (q_skip) H2D(dsk, q_skip, M);
  if (N > 0) { H2D(dt, t, 32 * (size_t)N); H2D(das, assigned, N); }
  int rc = svo_launch_match_greedy(ctx, dq, q_skip ? dsk : nullptr, M, dt, N, das, max_dist, ratio,
                                   di, db, ds, dacc);
  if (rc) return rc;
  D2H(best_idx, di, 4 * (size_t)M);
  D2H(best, db, 4 * (size_t)M);
  D2H(second, ds, 4 * (size_t)M);
  D2H(accepted, dacc, M);
  if (N > 0) D2H(assigned, das, N);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_match_greedy_gated(svo_ctx* ctx, const uint8_t* q, const uint8_t* q_skip, int M,
                                      const uint8_t* t, int N, uint8_t* assigned, int max_dist,
                                      float ratio, const float* q_xy, const float* t_xy,
                                      const int32_t* boxes, int n_boxes, const double F[9],
                                      int32_t* best_idx, int32_t* best, int32_t* second,
                                      uint8_t* accepted, uint8_t* vetoed) {
  if (n_boxes <= 0) {
    if (vetoed && M > 0) memset(vetoed, 0, (size_t)M);
    return svo_match_greedy(ctx, q, q_skip, M, t, N, assigned, max_dist, ratio, best_idx, best, second, accepted);
  }
  if (!ctx || M < 0 || N < 0 || n_boxes > 64 || !boxes || !F || !q_xy || !t_xy || !vetoed ||
      (M > 0 && (!q || !best_idx || !best || !second || !accepted)) || (N > 0 && (!t || !assigned)))
    return SVO_E_INVALID;
  if (M == 0) return SVO_OK;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  uint8_t* dq = bm.take<uint8_t>(32 * (size_t)M);
  uint8_t* dsk = bm.take<uint8_t>(M);
  uint8_t* dt = bm.take<uint8_t>(32 * (size_t)std::max(N, 1));
  uint8_t* das = bm.take<uint8_t>(std::max(N, 1));
  int32_t* di = bm.take<int32_t>(M);
  int32_t* db = bm.take<int32_t>(M);
  int32_t* ds = bm.take<int32_t>(M);
  uint8_t* dacc = bm.take<uint8_t>(M);
  uint8_t* dvet = bm.take<uint8_t>(M);
  float* dqxy = bm.take<float>(2 * (size_t)M);
  float* dtxy = bm.take<float>(2 * (size_t)std::max(N, 1));
  int32_t* dbox = bm.take<int32_t>(4 * (size_t)n_boxes);
  double* dF = bm.take<double>(9);
  if (!bm.ok()) return SVO_E_CAPACITY;
  H2D(dq, q, 32 * (size_t)M);
  if (q_skip) H2D(dsk, q_skip, M);
  if (N > 0) { H2D(dt, t, 32 * (size_t)N); H2D(das, assigned, N); H2D(dtxy, t_xy, 8 * (size_t)N); }
  H2D(dqxy, q_xy, 8 * (size_t)M);
  H2D(dbox, boxes, 16 * (size_t)n_boxes);
  H2D(dF, F, 72);
  SVO_HIP(ctx, hipMemsetAsync(dvet, 0, M, ctx->stream));
  int rc = svo_launch_match_greedy(ctx, dq, q_skip ? dsk : nullptr, M, dt, N, das, max_dist, ratio, di, db,
                                   ds, dacc, dqxy, dtxy, dbox, n_boxes, dF, dvet);
  if (rc) return rc;
  D2H(best_idx, di, 4 * (size_t)M);
  D2H(best, db, 4 * (size_t)M);
  D2H(second, ds, 4 * (size_t)M);
  D2H(accepted, dacc, M);
  D2H(vetoed, dvet, M);
  if (N > 0) D2H(assigned, das, N);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_bf_match(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                            int32_t* train_idx, int32_t* dist, uint8_t* keep) {
  if (!ctx || M < 0 || N < 0 || (M > 0 && (!q || !train_idx || !dist || !keep)) || (N > 0 && !t))
    return SVO_E_INVALID;
  if (M == 0) return SVO_OK;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  uint8_t* dq = bm.take<uint8_t>(32 * (size_t)M);
  uint8_t* dt = bm.take<uint8_t>(32 * (size_t)std::max(N, 1));
  int32_t* di = bm.take<int32_t>(M);
  int32_t* dd = bm.take<int32_t>(M);
  uint8_t* dk = bm.take<uint8_t>(M);
  if (!bm.ok()) return SVO_E_CAPACITY;
  H2D(dq, q, 32 * (size_t)M);
  if (N > 0) H2D(dt, t, 32 * (size_t)N);
  int rc = svo_launch_bf_match(ctx, dq, M, dt, N, di, dd, dk);
  if (rc) return rc;
  D2H(train_idx, di, 4 * (size_t)M);
  D2H(dist, dd, 4 * (size_t)M);
  D2H(keep, dk, M);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_pose_opt(svo_ctx* ctx, const double* Xw, const double* obs, int n,
                            const double K[4], double T_cw[16], svo_lm_stats* stats) {
  if (!ctx || n < 0 || (n > 0 && (!Xw || !obs)) || !K || !T_cw) return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  double* dX = bm.take<double>(3 * (size_t)std::max(n, 1));
  double* dO = bm.take<double>(2 * (size_t)std::max(n, 1));
  double* dK = bm.take<double>(4);
  double* dT = bm.take<double>(16);
  svo_lm_stats* dS = bm.take<svo_lm_stats>(1);
  if (!bm.ok()) return SVO_E_CAPACITY;
  if (n > 0) { H2D(dX, Xw, 24 * (size_t)n); H2D(dO, obs, 16 * (size_t)n); }
  H2D(dK, K, 32);
  H2D(dT, T_cw, 128);
  int rc = svo_launch_pose_opt(ctx, dX, dO, n, dK, dT, dS);
  if (rc) return rc;
  D2H(T_cw, dT, 128);
  if (stats) D2H(stats, dS, sizeof(svo_lm_stats));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_pnp_ransac(svo_ctx* ctx, const double* Xw, const double* obs, int n,
                              const double K[4], const double T_fallback_cw[16], uint64_t rng_state,
                              double T_cw[16], uint8_t* inlier_mask, svo_pnp_stats* stats) {
  if (!ctx || n < 0 || (n > 0 && (!Xw || !obs)) || !K || !T_fallback_cw || !T_cw) return SVO_E_INVALID;
  if (n > 512) return SVO_E_CAPACITY;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  double* dX = bm.take<double>(3 * (size_t)std::max(n, 1));
  double* dO = bm.take<double>(2 * (size_t)std::max(n, 1));
  double* dK = bm.take<double>(4);
  double* dTp = bm.take<double>(16);
  double* dT = bm.take<double>(16);
  uint8_t* dM = bm.take<uint8_t>(std::max(n, 1));
  svo_pnp_stats* dS = bm.take<svo_pnp_stats>(1);
  uint16_t* dSub = bm.take<uint16_t>(512);
  void* dHyp = bm.take<uint8_t>(svo_pnp_hyp_bytes());
  if (!bm.ok()) return SVO_E_CAPACITY;
  if (n > 0) { H2D(dX, Xw, 24 * (size_t)n); H2D(dO, obs, 16 * (size_t)n); }
  H2D(dK, K, 32);
  H2D(dTp, T_fallback_cw, 128);
  const uint16_t* sub = ctx->d_pnp_subsets + (size_t)std::min(n, 512) * 500;   // cv::RNG((uint64)-1): the table made at svo_create
  uint16_t hsub[500];
  if (rng_state != 0 && rng_state != ~0ull) {   // another RNG state: this call's own samples
    svo_pnp_subsets(rng_state, n, hsub);
    H2D(dSub, hsub, sizeof hsub);
    SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));   // hsub lives on this stack frame
    sub = dSub;
  }
  int rc = svo_launch_pnp(ctx, dX, dO, n, dK, dTp, sub, dHyp, dT, dM, dS);
  if (rc) return rc;
  D2H(T_cw, dT, 128);
  if (inlier_mask && n > 0) D2H(inlier_mask, dM, n);
  if (stats) D2H(stats, dS, sizeof(svo_pnp_stats));
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_debug_epnp5(svo_ctx* ctx, const double Xw5[15], const double uv5[10], const double K[4], double R[9],
                               double t[3], double rep_err[3]) {
  if (!ctx || !Xw5 || !uv5 || !K || !R || !t) return SVO_E_INVALID;
  hipSetDevice(ctx->device);
  Bump bm{ctx};
  double* dX = bm.take<double>(15); double* dU = bm.take<double>(10); double* dK = bm.take<double>(4);
  double* dO = bm.take<double>(24);
  if (!bm.ok()) return SVO_E_CAPACITY;
  H2D(dX, Xw5, 120); H2D(dU, uv5, 80); H2D(dK, K, 32);
  const char* reps_env = getenv("SVO_EPNP_REPS");   // diagnostics: repeat inside the kernel, stamps of the last pass
  int rc = svo_launch_epnp5_probe(ctx, dX, dU, dK, dO, reps_env ? atoi(reps_env) : 1);
  if (rc) return rc;
  double o[24];
  D2H(o, dO, sizeof o);
  SVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(R, o, 72); memcpy(t, o + 9, 24);
  if (rep_err) memcpy(rep_err, o + 13, 24);
  if (getenv("SVO_EPNP_STAMPS")) fprintf(stderr, "epnp5 cycles: setup+MtM %.0f, eigen %.0f (%d sweeps), L/rho %.0f, branches %.0f (initial betas %.0f, Gauss-Newton %.0f, to the last SVD %.0f)\n", o[16], o[17], (int)o[20], o[18], o[19], o[21], o[22], o[23]);
  return o[12] != 0.0 ? SVO_OK : SVO_E_INVALID;
}

// ---- throughput mode ---------------------------------------------------------------------
// The front end of pairs p0 .. p0 + b - 1 of a batch on stream `st`, in image slots 2 p0 .. 2 p0 + 2 b - 1 of the working
// set (left images first, then the right ones), results copied to the caller's arrays.  The launchers read the
// context's working-set pointers and stream: they are pointed at the slice for the duration of the call.
static int frontend_slice(svo_ctx* ctx, hipStream_t st, const uint8_t* d_grayL, const uint8_t* d_grayR, int stride, int p0, int b,
                          const svo_camera* cam, svo_kp* d_kpL, uint8_t* d_descL, int32_t* d_nL, float* d_uR, float* d_depth) {
  const SvoGeom& g = ctx->g;
  const size_t K = ctx->max_kp, base = 2 * (size_t)p0, img = (size_t)g.H * stride;
  struct Saved {
    uint8_t* pyr; uint32_t* corners; int32_t *counters, *hist; SvoSel* sel; int32_t* selcnt; svo_kp* kp; uint8_t* desc; int32_t* nkp;
    float *uR, *depth; int32_t* sad; hipStream_t stream;
  } sv{ctx->d_pyr, ctx->d_corners, ctx->d_counters, ctx->d_hist, ctx->d_sel, ctx->d_selcnt, ctx->d_kp, ctx->d_desc, ctx->d_nkp,
       ctx->d_uR, ctx->d_depth, ctx->d_sad, ctx->stream};
  ctx->d_pyr += base * g.pyr_bytes; ctx->d_corners += base * g.corner_entries; ctx->d_counters += base * SVO_NLEVELS;
  ctx->d_hist += base * SVO_NLEVELS * 256; ctx->d_sel += base * SVO_NLEVELS * SVO_QMAX; ctx->d_selcnt += base * SVO_NLEVELS;
  ctx->d_kp += base * K; ctx->d_desc += base * K * 32; ctx->d_nkp += base;
  ctx->d_uR += (size_t)p0 * K; ctx->d_depth += (size_t)p0 * K; ctx->d_sad += (size_t)p0 * K;
  ctx->stream = st;
  int rc = svo_launch_orb(ctx, d_grayL + p0 * img, d_grayR + p0 * img, stride, b, 2 * b);
  if (rc == SVO_OK) rc = svo_launch_stereo(ctx, d_grayL + p0 * img, d_grayR + p0 * img, stride, b, cam);
  if (rc == SVO_OK) {
    auto d2d = [&](void* dst, const void* src, size_t bytes) {
      if (dst && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = SVO_E_HIP;
    };
    d2d(d_kpL ? d_kpL + (size_t)p0 * K : nullptr, ctx->d_kp, sizeof(svo_kp) * K * b);
    d2d(d_descL ? d_descL + (size_t)p0 * K * 32 : nullptr, ctx->d_desc, 32 * K * b);
    d2d(d_nL ? d_nL + p0 : nullptr, ctx->d_nkp, 4 * (size_t)b);
    d2d(d_uR ? d_uR + (size_t)p0 * K : nullptr, ctx->d_uR, 4 * K * b);
    d2d(d_depth ? d_depth + (size_t)p0 * K : nullptr, ctx->d_depth, 4 * K * b);
  }
  ctx->d_pyr = sv.pyr; ctx->d_corners = sv.corners; ctx->d_counters = sv.counters; ctx->d_hist = sv.hist; ctx->d_sel = sv.sel;
  ctx->d_selcnt = sv.selcnt; ctx->d_kp = sv.kp; ctx->d_desc = sv.desc; ctx->d_nkp = sv.nkp; ctx->d_uR = sv.uR; ctx->d_depth = sv.depth;
  ctx->d_sad = sv.sad; ctx->stream = sv.stream;
  return rc;
}

// Slices svo_frontend_batch_dev cuts a batch of B pairs into (1: one chain).  Slice k covers the pairs [k B / ns, (k + 1) B / ns)
// and keeps the results of its LEFT images at the context's slots 2 p0 .. (its right images behind them) - callers that read
// the context's own buffers (svo_track_sharded_dev) need the same rule.
int svo_frontend_nslices(const svo_ctx* ctx, int B) {
  const int ns = std::min(ctx->opt_frontend_overlap, B / 8);
  return (ns < 2 || ctx->profiling) ? 1 : ns;
}

extern "C" int svo_frontend_batch_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR,
                                      int stride, int B, const svo_camera* cam, svo_kp* d_kpL,
                                      uint8_t* d_descL, int32_t* d_nL, float* d_uR,
                                      float* d_depth) {
  if (!ctx || !d_grayL || !d_grayR || !cam || B < 1 || stride < ctx->g.W) return SVO_E_INVALID;
  if (B > ctx->max_batch) return SVO_E_CAPACITY;
  hipSetDevice(ctx->device);
  { const int rcq = svo_track_quiesce(ctx); if (rcq) return rcq; }   // (a batched tracker call's tail may still read the result arrays)
  // Slices of the batch side by side on their own streams: the kernels of the chain are a mix of arithmetic-bound ones (k_fast: the
  // vector ALUs 88 % busy) and latency-bound ones (k_select, k_stereo_*, the small pyramid levels: a few waves per CU
  // waiting on memory) - side by side they fill each other's gaps.  With the per-kernel timers on (svo_profile_enable)
  // the batch runs as one chain on one stream, so that a kernel's time is its own.
  const int ns = svo_frontend_nslices(ctx, B);
  if (ns < 2)
    return frontend_slice(ctx, ctx->stream, d_grayL, d_grayR, stride, 0, B, cam, d_kpL, d_descL, d_nL, d_uR, d_depth);
  while ((int)ctx->fe_streams.size() < ns - 1) {
    hipStream_t st = nullptr;
    hipEvent_t e;
    {   // a stream that runs beside the context's own (and beside the slices' streams made before it)
      int attempts = 0, percent = 0;
      const int rcp = svo_pick_stream(ctx, [](hipStream_t* s) { return svo_stream_create(s, 0); },
                                      {ctx->stream, ctx->fe_streams.empty() ? nullptr : ctx->fe_streams.back()}, &st, &attempts, &percent,
                                      {ctx->stream, ctx->fe_streams.empty() ? nullptr : ctx->fe_streams.back()},
                                      {ctx->stream, ctx->fe_streams.empty() ? nullptr : ctx->fe_streams.back()});   // (slices overlap: neither's grid in dispatch may hold the other's up)
      if (rcp) return rcp;
    }
    ctx->fe_streams.push_back(st);
    SVO_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->fe_events.push_back(e);
  }
  if (!ctx->ev_frontend) SVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_frontend, hipEventDisableTiming));
  SVO_HIP(ctx, hipEventRecord(ctx->ev_frontend, ctx->stream));            // whatever the ctx stream holds comes first
  int rc = SVO_OK;
  for (int k = ns - 1; k >= 0 && rc == SVO_OK; --k) {   // slice k: pairs [k B / ns, (k + 1) B / ns); slice 0 on the ctx stream
    const int p0 = (int)((int64_t)k * B / ns), p1 = (int)((int64_t)(k + 1) * B / ns);
    hipStream_t st = k ? ctx->fe_streams[k - 1] : ctx->stream;
    if (k) SVO_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_frontend, 0));
    rc = frontend_slice(ctx, st, d_grayL, d_grayR, stride, p0, p1 - p0, cam, d_kpL, d_descL, d_nL, d_uR, d_depth);
    if (k) SVO_HIP(ctx, hipEventRecord(ctx->fe_events[k - 1], st));
  }
  for (int k = 1; k < ns; ++k) SVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->fe_events[k - 1], 0));   // callers order against the ctx stream
  return rc;
}

// ---- profiling API -------------------------------------------------------------------------
extern "C" int svo_profile_enable(svo_ctx* ctx, int on) {
  if (!ctx) return SVO_E_INVALID;
  if (!on && ctx->profiling) prof_resolve(ctx);
  ctx->profiling = on != 0;
  return SVO_OK;
}
extern "C" int svo_profile_reset(svo_ctx* ctx) {
  if (!ctx) return SVO_E_INVALID;
  if (ctx->prof_impl) prof_resolve(ctx);
  ctx->prof.clear();
  return SVO_OK;
}
extern "C" int svo_profile_get(svo_ctx* ctx, int index, char* name, int name_cap, double* total_ms,
                               int64_t* launches) {
  if (!ctx || index < 0) return SVO_E_INVALID;
  if (ctx->prof_impl) prof_resolve(ctx);
  if (index >= (int)ctx->prof.size()) return SVO_E_INVALID;
  const SvoProfileEntry& e = ctx->prof[index];
  if (name && name_cap > 0) {
    strncpy(name, e.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (total_ms) *total_ms = e.total_ms;
  if (launches) *launches = e.launches;
  return SVO_OK;
}

// ---- streams that run side by side --------------------------------------------------------------
// Do two streams run side by side?  The tracker's tail is two chains on two streams that must overlap (svo_track.hip), the batched
// front end runs its slices on two streams, and the runtime maps a process's
// streams onto a few hardware queues (a pool per priority, least-used first: what the process created earlier decides what a
// new stream gets).  Two streams on ONE hardware queue serialise completely - measured with tools/microbench/queue_pair_probe
// (profiles/r05_queue_pairs.jsonl): of six high-priority streams made one after the other, the pairs (2, 5) and (3, 4) take
// the SUM of the two chains' times, every other pair the maximum; with three older high-priority streams in the process the
// first two new ones are such a pair - the tracker then ran at half its rate, with identical kernels.  The probe therefore
// measures the thing itself: a chain of eight short dependent kernels on A alone, then the same chain on A and on B together.
__global__ void k_probe_spin(long long cycles) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
}
static double probe_chain_us(hipStream_t A, hipStream_t B /* nullable */) {
  hipStreamSynchronize(A);
  if (B) hipStreamSynchronize(B);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 8; ++i) {
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, A, 60000LL);   // ~28 us each: device time, not the host's enqueue, decides
    if (B) hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, B, 60000LL);
  }
  hipStreamSynchronize(A);
  if (B) hipStreamSynchronize(B);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
// both chains together / one chain alone, in per cent (~100: side by side, ~200: one after the other)
static int probe_pair_percent(hipStream_t A, hipStream_t B) {
  (void)probe_chain_us(A, nullptr);                       // (first launches of the kernel, clocks)
  const double alone = probe_chain_us(A, nullptr);
  const double both = std::min(probe_chain_us(A, B), probe_chain_us(A, B));
  (void)hipGetLastError();
  return (int)std::min(999.0, 100.0 * both / std::max(alone, 1.0));
}
// The second way two streams get into each other's way: two hardware queues behind ONE dispatch pipe of the command processor.
// gfx950 places a process's hardware queues on four pipes in creation order (queue i and queue i + 4 share one).  Measured
// (tools/microbench/queue_block_probe, queue_prio_probe; profiles/r05_queue_block.jsonl, r05_queue_prio.jsonl):
//  * two normal-priority queues on one pipe: while the pipe is placing the workgroups of a grid that does not fit (64 KB of LDS per
//    workgroup, two per CU), a chain of one-wave kernels on the other queue takes 8-9x its time - it waits for the whole grid; beside
//    every queue of another pipe 1.0x.  Of twelve dedicated queues created one after the other exactly the pairs 4 and 8 places
//    apart do this, and pooled queues created in between shift the pattern as the rule predicts;
//  * a HIGH-priority queue on the pipe of the confined front end's queue is not stopped by such a grid, but by the real thing it is:
//    grids of 256-thread workgroups that fill every wave slot of the eighth of the CUs their stream owns make a 3,072-workgroup grid
//    on it take 119 us instead of 14 - on exactly one of the four pooled high-priority queues, the one whose creation index is
//    congruent to the front end's queue; which stream gets that queue depends on what the process created before.
// In the tracker this was the statistical solver's leg at 7.4 k instead of 14.2 k frames/s (64 sequences: 58 k instead of 102 k) in
// a process that had used eight torch streams before, with every pair of streams passing the chain-beside-chain probe above (it
// sees shared QUEUES, not shared pipes).  So a stream that carries large grids is also tested in that shape against the streams it
// must not hold up: a 3,072 x 64-thread grid on the other stream alone, then beside a queue of slot-filling grids on this one.
__global__ void k_probe_fill(long long cycles) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(2);
}
static double probe_grid_us(hipStream_t V) {
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_probe_fill, dim3(3072), dim3(64), 0, V, 4000LL);
  hipStreamSynchronize(V);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
int svo_probe_block_percent(hipStream_t blocker, hipStream_t victim) {
  hipStreamSynchronize(blocker); hipStreamSynchronize(victim);
  (void)probe_grid_us(victim);
  const double alone = std::min(probe_grid_us(victim), probe_grid_us(victim));
  // one round of the filling side: a single workgroup spinning as long as its workgroups do (~45 us), less the cost of a launch
  auto one_us = [blocker](long long cyc) {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_probe_fill, dim3(1), dim3(256), 0, blocker, cyc);
    hipStreamSynchronize(blocker);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  (void)one_us(100LL);   // (first launch on this stream)
  const double launch = one_us(100LL), round = std::max(10.0, one_us(100000LL) - launch);
  // the filling side: FOUR launches of 4,096 x 256 threads queued back to back (2 rounds each on a whole device, 16 on an eighth of
  // it).  It takes the queue of launches: ONE grid of any size in dispatch did not hold the other queue up (measured), a launch
  // waiting behind its predecessor on the queue does - which is what the front end's stream looks like all the time.
  for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_probe_fill, dim3(4096), dim3(256), 0, blocker, 100000LL);
  const double beside = probe_grid_us(victim);
  hipStreamSynchronize(blocker);
  const hipError_t err = hipGetLastError();
  // In rounds: a grid that merely has to find wave slots among the filling workgroups (a stream that owns every CU) waits for less
  // than one round (measured 0.7); held up behind the other queue's pipe it takes 2.3-2.6.  Reported as 100 + 100 x rounds.
  const double rounds = std::max(0.0, beside - alone) / round;
  static const bool dbg = getenv("SVO_PICK_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[svo block probe] filling grids on %p (%.0f us per round), 3072 x 64 threads on %p: alone %.0f us, beside them %.0f us = %.1f rounds late (%s)\n",
                   (void*)blocker, round, (void*)victim, alone, beside, rounds, hipGetErrorName(err));
  return (int)std::min(9999.0, 100.0 + 100.0 * rounds);
}
// A new stream (made by `make`) that runs side by side with every stream of `others`: up to six candidates; the rejected ones are
// kept until the choice is made, so that the pool hands out another queue each time.  None passes (a tool that serialises all
// dispatches, e.g. rocprofv3's kernel trace): the best one.  *attempts / *percent: candidates tried, the chosen one's worst ratio.
int svo_pick_stream(svo_ctx* ctx, const std::function<hipError_t(hipStream_t*)>& make, std::initializer_list<hipStream_t> others, hipStream_t* out,
                    int* attempts, int* percent, std::initializer_list<hipStream_t> holds_up, std::initializer_list<hipStream_t> held_up_by) {
  static const bool no_probe = []() { const char* e = getenv("SVO_NO_STREAM_PROBE"); return e && e[0] == '1'; }();
  static const bool no_block = []() { const char* e = getenv("SVO_NO_BLOCK_PROBE"); return e && e[0] == '1'; }();
  static const bool dbg = getenv("SVO_PICK_DEBUG") != nullptr;
  *out = nullptr; *attempts = 0; *percent = 0;
  if (no_probe) return make(out) == hipSuccess ? SVO_OK : SVO_E_HIP;
  std::vector<hipStream_t> rejected;
  hipStream_t best = nullptr;
  int best_pct = 100000, best_block = 0;
  // (a pipe holds every fourth queue: with the streams to keep clear of on at most three pipes, one of four consecutive new queues
  // is free of them all; pooled candidates may repeat a queue, hence six tries, the last two on queues of their own)
  for (int k = 0; k < 6; ++k) {
    hipStream_t cand = nullptr;
    hipError_t e = hipSuccess;
    if (k >= 4 && (holds_up.size() || held_up_by.size())) {
      int dev = 0; hipDeviceProp_t prop;
      e = hipGetDevice(&dev);
      if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
      if (e == hipSuccess) {
        std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
        if (holds_up.size() == 0 && mask.size() > 1) mask[0] = 0;   // a latency chain: off the CUs the confined front end keeps busy (masked streams have no priority)
        e = hipExtStreamCreateWithCUMask(&cand, (uint32_t)mask.size(), mask.data());
      }
    } else {
      e = make(&cand);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); break; }
    ++*attempts;
    int worst = 0, block = 0;
    for (hipStream_t o : others)
      if (o) worst = std::max(worst, probe_pair_percent(o, cand));
    if (!no_block && worst < 150) {
      for (hipStream_t o : holds_up)
        if (o) block = std::max(block, svo_probe_block_percent(cand, o));
      for (hipStream_t o : held_up_by)
        if (o) block = std::max(block, svo_probe_block_percent(o, cand));
    }
    if (dbg) fprintf(stderr, "[svo_pick_stream] ctx %p candidate %d (%p): chains %d %%, held-up chain %d %%\n", (void*)ctx, k, (void*)cand, worst, block);
    const int score = worst + (block >= 250 ? block : 0);   // (1.5 rounds late)
    if (score < best_pct) { if (best) rejected.push_back(best); best = cand; best_pct = score; best_block = block; }
    else rejected.push_back(cand);
    if (best_pct < 150) break;
  }
  // the rejected candidates stay until the context goes: a process's hardware queues sit on the dispatch pipes in creation order, and
  // destroying one moves every later queue - the stream just chosen included - one pipe down (tools/microbench/queue_block_probe
  // destroy).  (Beyond a dozen parked streams they are destroyed after all.)
  for (hipStream_t r : rejected) {
    if (ctx->parked_streams.size() < 12) ctx->parked_streams.push_back(r);
    else hipStreamDestroy(r);
  }
  *out = best; *percent = best ? std::min(best_pct, 999) : 0;
  if (dbg) fprintf(stderr, "[svo_pick_stream] ctx %p: %d candidate(s), chosen %p runs at %d %% beside %zu stream(s), held-up chain %d %%\n", (void*)ctx, *attempts, (void*)best, best_pct, others.size(), best_block);
  if (best && best_pct >= 150)
    ctx->last_error = "tracker: no candidate stream ran beside the pose chain's (" + std::to_string(best_pct) + " % of one chain's time for two): the tail runs at a reduced rate";
  return best ? SVO_OK : SVO_E_HIP;
}
// A stream that may only use the first `percent` per cent of the device's compute units (whole 32-bit mask words: on gfx950
// the first word of the mask stands for four CUs on EACH of the eight XCDs - tools/microbench/cu_mask_probe - so the stream keeps
// an even share of every XCD, not whole XCDs).  The stream is a blocking one (it orders against the null stream).
hipError_t svo_stream_create_masked(hipStream_t* st, int device, int percent) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return svo_stream_create(st, -1); }
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  const int keep = std::max(1, std::min(words, (words * percent + 50) / 100));
  std::vector<uint32_t> mask((size_t)words, 0u);
  for (int w = 0; w < keep; ++w) mask[w] = 0xffffffffu;
  if (hipExtStreamCreateWithCUMask(st, (uint32_t)words, mask.data()) != hipSuccess) { (void)hipGetLastError(); return svo_stream_create(st, -1); }
  return hipSuccess;
}

// The context's four main streams as four hardware queues of their own, created back to back.  A process's hardware queues sit on the
// command processor's four dispatch pipes in creation order (position in the list of its live queues, modulo 4; a destroyed
// queue's successors move up: tools/microbench/queue_block_probe, profiles/r05_queue_block.jsonl), and two queues on one pipe
// hold each other up whenever one has launches waiting (above).  Four queues made one after the other are on four different pipes
// whatever the process created before and whatever it creates or destroys elsewhere later - only a queue created or destroyed
// BETWEEN them could change that, and nothing is.  CU-masked streams are the one kind the runtime gives a queue of its own
// (pooled streams share four per priority, whose positions the process's history decides); they have no priority, so the tail's two
// streams keep off the CUs the batched front end is confined to (first mask word: four CUs of every XCD) instead:
//   stream           pose chain, and everything the host-buffer entries run      every CU but the first mask word's
//   stream_idx       index chain                                                 the same
//   stream_fe_batch  the batched tracker's front end                             "fe_cu_percent" of the CUs (first mask words)
//   stream_dense     dense stage in front of the tracker (depth_source 1, 2)     "dense_cu_percent" of the CUs
// An option that changes a mask destroys that one stream and makes a new one through svo_pick_stream (measured against the
// others, rejected candidates parked).  Per context (svo_create_ex flags; svo_stream_mode reports what is in effect):
// SVO_CREATE_POOLED_STREAMS - the runtime's pooled NON-BLOCKING streams as up to round 5 (main stream only; the others on demand
// through the picker; svo_create takes it from SVO_POOLED_QUEUES=1); SVO_CREATE_TAIL_ALL_CUS - the first two queues keep every CU
// (a context that never runs the batched tracker: its host-buffer / front-end / dense entries all run on `stream`).
int svo_stream_burst(svo_ctx* ctx) {
  const bool pooled = (ctx->create_flags & SVO_CREATE_POOLED_STREAMS) != 0;
  hipDeviceProp_t prop;
  if (pooled || hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) {
    (void)hipGetLastError();
    return svo_stream_create(&ctx->stream, +1) == hipSuccess ? SVO_OK : SVO_E_HIP;
  }
  const int words = (prop.multiProcessorCount + 31) / 32;
  auto first_words = [words](int percent) {
    std::vector<uint32_t> m((size_t)words, 0u);
    const int keep = percent >= 100 ? words : std::max(1, std::min(words, (words * percent + 50) / 100));
    for (int w = 0; w < keep; ++w) m[w] = 0xffffffffu;
    return m;
  };
  std::vector<uint32_t> tail((size_t)words, 0xffffffffu);
  if (words > 1 && !(ctx->create_flags & SVO_CREATE_TAIL_ALL_CUS)) tail[0] = 0;
  const std::vector<uint32_t> masks[4] = {tail, tail, first_words(ctx->opt_fe_cu_percent), first_words(ctx->opt_dense_cu_percent)};
  hipStream_t q[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < 4; ++k) {
    if (hipExtStreamCreateWithCUMask(&q[k], (uint32_t)words, masks[k].data()) != hipSuccess) {
      (void)hipGetLastError();
      for (int j = 0; j < k; ++j) hipStreamDestroy(q[j]);
      return svo_stream_create(&ctx->stream, +1) == hipSuccess ? SVO_OK : SVO_E_HIP;   // (no masked streams on this device: pooled ones)
    }
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, q[k], 100LL);   // (the queue exists from its first packet on at the latest)
  }
  for (int k = 0; k < 4; ++k) hipStreamSynchronize(q[k]);
  (void)hipGetLastError();
  ctx->stream = q[0]; ctx->stream_idx = q[1]; ctx->stream_fe_batch = q[2]; ctx->stream_dense = q[3];
  ctx->streams_burst = true;
  ctx->idx_probe_attempts = 1;                                 // what svo_debug_stream_probe reports: one candidate, and how the two chains
  ctx->idx_probe_spins = probe_pair_percent(q[0], q[1]);       // of the tail run side by side on these two (measured, ~1 ms)
  static const bool dbg = getenv("SVO_PICK_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[svo_stream_burst] ctx %p: pose %p, index %p, front end %p (%d %% of the CUs), dense %p (%d %%)\n", (void*)ctx, (void*)q[0], (void*)q[1],
                   (void*)q[2], ctx->opt_fe_cu_percent, (void*)q[3], ctx->opt_dense_cu_percent);
  return SVO_OK;
}

hipError_t svo_stream_create(hipStream_t* st, int role) {
  int least = 0, greatest = 0;   // numerically: greatest priority <= least priority
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
  const int prio = role > 0 ? greatest : (role < 0 ? least : (least + greatest) / 2);
  return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio);
}

int svo_host_cpus() {
  static const int cached = []() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us (-1 = none)
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else {
      FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
      FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
      if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1) {}
      if (fq) fclose(fq);
      if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1LL, (quota + period / 2) / period));
    if (const char* e = getenv("SVO_HOST_CPUS")) { if (atoi(e) > 0) n = atoi(e); }
    return n;
  }();
  return cached;
}
