// svo_internal.h - shared declarations of the HIP implementation behind include/svo.h.
// gfx950 (MI355X) only.  Everything here is compiled with -ffp-contract=off: the
// float/double stages must round once per operation to stay bit-identical with the
// parity oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <functional>
#include <initializer_list>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/svo.h"

#define SVO_NLEVELS 8
#define SVO_EDGE 31
#define SVO_FAST_THR 20
#define SVO_HALF_PATCH 15
#define SVO_CAP1 1024     // cap on Harris candidates per (image, level)
#define SVO_QMAX 128      // >= largest per-level quota (109 for 500 features)
#define SVO_DESC_BYTES 32

// FAST tile: 120 x 30 output pixels per 256-thread workgroup (see svo_orb.hip).
#define FAST_TW 120
#define FAST_TH 30

struct SvoGeom {
  int W, H;
  int w[SVO_NLEVELS], h[SVO_NLEVELS];
  int pitch[SVO_NLEVELS];          // bytes per row of the stored level (levels 1..7)
  int64_t loff[SVO_NLEVELS];       // byte offset of level l inside one image's pyramid slot
  int64_t pyr_bytes;               // bytes of one image's pyramid slot (levels 1..7)
  float scale[SVO_NLEVELS];
  int quota[SVO_NLEVELS];
  int cap[SVO_NLEVELS];            // raw-corner capacity per level
  int64_t coff[SVO_NLEVELS];       // entry offset of level l in one image's raw-corner slot
  int64_t corner_entries;          // entries of one image's raw-corner slot
  int tiles_x[SVO_NLEVELS], tiles_y[SVO_NLEVELS];
  int tile_base[SVO_NLEVELS + 1];  // prefix of tiles over levels (FAST kernel grid)
  // resize tables: offsets (in elements) into the device table arrays
  int xtab_off[SVO_NLEVELS], ytab_off[SVO_NLEVELS];
};

// Entry of the level-local selection (output of k_select).
struct SvoSel {
  int64_t R;     // exact Harris measure 25(ab-c^2)-(a+b)^2
  int16_t x, y;  // level coordinates
  int32_t pad;
};

// One launch of k_pyr_fused: levels l0 .. l0 + nl - 1 from level l0 - 1, strips of rt rows of the last level
struct SvoPyrGroup { int l0, nl, rt, cap0, cap1; size_t lds; };

struct SvoProfileEntry {
  std::string name;
  double total_ms = 0;
  int64_t launches = 0;
};

struct svo_ctx {
  int device = 0;
  int max_kp = 500, max_batch = 1, max_images = 2;
  SvoGeom g;
  hipStream_t stream = nullptr;
  std::string last_error;

  // device tables
  SvoGeom* d_geom = nullptr;
  int32_t* d_xofs = nullptr;  // per level: w[l] entries
  int32_t* d_xalpha = nullptr;  // packed (a0 | a1<<16)
  int32_t* d_yofs = nullptr;
  int32_t* d_ybeta = nullptr;

  // per-batch working set (HBM resident, sized for max_images)
  uint8_t* d_stage = nullptr;      // host-path image staging: max_images x H x stage_pitch
  int stage_pitch = 0;
  uint8_t* d_pyr = nullptr;        // max_images x pyr_bytes
  uint32_t* d_corners = nullptr;   // max_images x corner_entries, packed x|y<<12|score<<24
  int32_t* d_counters = nullptr;   // max_images x 8 raw-corner counts, then x 8*256 hist
  int32_t* d_hist = nullptr;
  SvoSel* d_sel = nullptr;         // max_images x 8 x SVO_QMAX
  int32_t* d_selcnt = nullptr;     // max_images x 8
  svo_kp* d_kp = nullptr;          // max_images x max_kp
  uint8_t* d_desc = nullptr;       // max_images x max_kp x 32
  int32_t* d_nkp = nullptr;        // max_images
  float* d_uR = nullptr;           // max_batch x max_kp
  float* d_depth = nullptr;
  int32_t* d_sad = nullptr;        // max_batch x max_kp best SAD (or -1)
  // generic scratch for the small host-path ops
  uint8_t* d_scratch = nullptr;
  size_t scratch_bytes = 0;
  uint8_t* h_pinned = nullptr;
  uint8_t* h_stage = nullptr;      // pinned image staging of the host path: 2 x H x stage_pitch
  size_t pinned_bytes = 0;

  // tracker state (svo_track.hip)
  void* d_track = nullptr;      // n_seq TrackState records
  int n_seq = 0;
  // svo_track_multi_step_dev with svo_set_option("multi_pipeline", 1): the front end of step t + 1 runs beside the tail of
  // step t; its outputs alternate between two private sets (ms_*[parity])
  int opt_multi_pipeline = 0, ms_parity = 0, ms_cap = 0;
  bool ms_tail_recorded[2] = {false, false};
  svo_kp* ms_kp[2] = {nullptr, nullptr};
  uint8_t* ms_desc[2] = {nullptr, nullptr};
  int32_t* ms_nkp[2] = {nullptr, nullptr};
  float* ms_depth[2] = {nullptr, nullptr};
  hipEvent_t ms_fe_done[2] = {nullptr, nullptr}, ms_tail_done[2] = {nullptr, nullptr};
  // svo_track_batch_dev: the front end's outputs exist twice and alternate between calls, so that the first sub-batches of call
  // c + 1 run while the tail of call c is still reading its set (set 0 = the context's own arrays, set 1 allocated on demand)
  svo_kp* tb_kp = nullptr; uint8_t* tb_desc = nullptr; int32_t* tb_nkp = nullptr;
  float *tb_uR = nullptr, *tb_depth = nullptr; int32_t* tb_sad = nullptr;
  hipEvent_t tb_done[2] = {nullptr, nullptr};   // the tail of the call that last used set p has finished
  bool tb_used[2] = {false, false};
  int tb_parity = 0;
  void* d_work = nullptr;       // TrackWork records (index chain -> pose chain), work_cap of them
  hipStream_t stream_elas_a = nullptr;   // svo_elas_batch_dev: phase A (descriptors, support candidates) of the next chunks, beside phase B of the earlier ones
  int stream_elas_a_pct = -1;            //   the share of the CUs it was made for (0: all; as the stream phase B runs on)
  hipEvent_t ev_elas_setup = nullptr;
  hipEvent_t shard_wait = nullptr;   // borrowed: the gather event of the sharded tracker call that last read this context's result arrays
  void* d_gate_pre = nullptr;   // GatePre records (brute-force matches + F solved ahead of the index chain), like d_work
  bool hyp_two_launch = false;  // set by an entry for the duration of its tail_enqueue calls: RANSAC samples as 16 + (those the bound can reach)
  int opt_dense_two_launch = 0; // depth_source = 1: the tail beside the dense stage uses the two-launch RANSAC (fewer CUs taken from ELAS)
  int idx_probe_attempts = -1;  // how many candidate streams the index chain's stream was chosen from (-1: not chosen yet, 0: probe off)
  int idx_probe_spins = 0;      // the chosen candidate's probe result: two chains together / one alone, per cent (~100 side by side, ~200 serialised)
  int opt_hyp_first = 8;        // many sequences: RANSAC samples per sequence in the first (and second) launch of a step
  int opt_gate_group = 1;       // 1: gated frames' F for a group of frames in one launch ahead of the index chain; 0: per frame, in the chain
  int work_cap = 0;             // records per half (two halves are allocated)
  int work_last_half = 0;       // the half the last tail call used (debug readers)
  hipStream_t stream_idx = nullptr;        // the pose-free index chain of the tracking tail runs here, ahead of the pose chain
  bool stream_diag_done = false;           // SVO_STREAM_DIAG=1 printed its table for this context
  bool streams_burst = false;              // stream, stream_idx, stream_fe_batch, stream_dense were made as four hardware queues back to back (svo_stream_burst)
  std::vector<hipStream_t> parked_streams; // candidates the stream picker rejected: kept until the context goes (destroying a queue moves every later one to another dispatch pipe)
  hipStream_t stream_fe = nullptr;         // the front end of the next step / chunk beside the tail (multi-sequence steps, MSA chunks)
  hipStream_t stream_fe_batch = nullptr;   // svo_track_batch_dev: the front end's sub-batches, confined to a share of the CUs
  hipStream_t stream_dense = nullptr;      // svo_track_batch_dev with depth_source 1: the dense front end's sub-batches
  int32_t* h_prod = nullptr;               //   pinned: the call's `produced` flags
  int h_prod_cap = 0;
  std::vector<hipEvent_t> ev_sub;          // front end of sub-batch j finished (recorded on `stream_fe`)
  hipEvent_t ev_frontend = nullptr;        // front end of a call finished (recorded on `stream`)
  std::vector<hipEvent_t> ev_frame;        // index chain of frame f finished (recorded on `stream_idx`)
  uint16_t* d_pnp_subsets = nullptr;   // [513][100][5]: RANSAC sample indices of cv::RNG((uint64)-1) for every point count
  int msa_lds_state = 0;         // k_msa_dp_bfs opted into > 64 KB of dynamic LDS: 0 not tried, 1 yes, -1 refused
  int pose_lds_state = 0;       // > 64 KB dynamic-LDS opt-in of the pose kernels: 0 untried, 1 granted, -1 refused
  int track_lds_state = 0;      // same for the tracker's kernels
  std::vector<SvoPyrGroup> pyr_plan;  // the pyramid as fused launches (svo_create); empty: one launch per level
  bool opt_pyr_fused = true;          // svo_set_option("pyr_fused")
  int opt_frontend_overlap = 2;      // svo_set_option("frontend_overlap"): slices of a batch svo_frontend_batch_dev runs side by side (0 / 1: none)
  std::vector<hipStream_t> fe_streams;   // their streams (slice 0 uses `stream`) ...
  std::vector<hipEvent_t> fe_events;     // ... and completion events
  int opt_track_group = 4;      // svo_set_option("track_group"): most frames the pose chain takes over per stream event (1..64)
  int opt_track_nblk = 3;       // svo_set_option("track_nblk"): runner-up blockers stored per packed entry (0..3)
  int opt_track_lcap = 8;       // svo_set_option("track_lcap"): packed entries a map point keeps before it goes "dense" (1..8)
  void* elas = nullptr;     // ElasState (svo_elas.hip), allocated on first svo_elas_process
  void* elas_batch = nullptr;   // ElasBatch: per-pair states of svo_elas_batch_dev
  int elas_strip_state = 0;     // k_cc_strip's > 64 KB dynamic-LDS opt-in on this ctx's device: 0 untried, 1 granted, -1 refused
  void* msa_arenas = nullptr;   // MsaArenas: device buffers of svo_msa_solve and of the tracker's MSA mode
  float* d_dense = nullptr;     // dense maps of svo_track_batch_dev with depth_source 1: 2 x dense_cap x W*H
  int dense_cap = 0;
  svo_camera cam{};
  int track_frame = 0;

  int opt_depth_source = 0; // svo_set_option("depth_source"): 0 sparse epipolar stereo, 1 dense ELAS map (svo_track_frame)
  int opt_fast_cand_cap = 2048;   // svo_set_option("fast_cand_cap"): entries of k_fast's candidate list (<= 2048)
  int opt_pose_mfma = 1;   // svo_set_option("pose_mfma"): the LM's sums over the edges - 1 on f64 MFMA (default), 2 one lane per quantity, 0 one lane for everything (svo_pose_dev.h)
  int opt_fe_cu_percent = 12;   // svo_set_option("fe_cu_percent"): share of the CUs the batched tracker's front-end stream may use
  int opt_tail_semi = 1;   // svo_set_option("tail_semi"): beside a dense stage (2: also many sequences) the first 8 samples, then the other samples + the frame part in one launch
  int opt_tail_fused = 1;  // svo_set_option("tail_fused"): one sequence, default solver: RANSAC samples + frame part in one launch (k_tp_tail_ord)
  int opt_pose_flag = 0;   // svo_set_option("pose_flag"): one sequence's pose kernels poll the index chain's per-frame tag instead of waiting on stream
                           // events.  OFF by default: the poll needs the index kernel to run CONCURRENTLY with the polling one, and a tool that
                           // serialises kernel dispatches (rocprofv3 --kernel-trace does) turns every frame into a timed-out poll
  int opt_epnp_exact = 2;  // svo_set_option("epnp_exact"): 2 = OpenCV's operations with their rounding, spread over a wave per sample (default); 1 = one lane per sample, loop by loop (the checker); 0 = the statistical wave solver
  int opt_dense_cu_percent = 88;   // svo_track_batch_dev with depth_source 1: share of the CUs the dense front end's stream may use (measured, 256 frames with boxes: 100 % 5.4 k, 88 % 5.7 k, 75 % 6.4 k, 62 % 6.0 k, 50 % 5.3 k frames/s - the tail's single-wave RANSAC workgroups need free CUs, ELAS needs most of the chip)
  int opt_shard_force_staged = 0;  // svo_track_sharded_dev gathers through pinned host memory even where a peer read exists (tests; env SVO_SHARD_FORCE_STAGED is read once, at svo_create)
  int opt_epnp_force_seq = 0;  // tests: mode 2 takes its sequential fallback for every sample
  int opt_debug_lose_sample = 0;   // tests: sample k - 1 of every fused pose launch never reports (0: off)
  bool timeout_reported = false;                  // svo_sync returned SVO_E_TIMEOUT for this context once (sticky flag 4, svo_track_check_timeout)
  uint32_t create_flags = 0;                      // svo_create_ex
  void* hostfeed = nullptr;                       // HostFeed (svo_hostfeed.hip): copy stream, image sets, pinned staging of the host-fed entries
  const hipEvent_t* feed_pair_event = nullptr;    // set by a host-fed entry for the duration of its inner call: pair i of the call (this context's
                                                  //   local index) is resident in HBM after feed_pair_event[i] (recorded on the feed's copy stream)
  bool profiling = false;
  std::vector<SvoProfileEntry> prof;
  void* prof_impl = nullptr;  // SvoProfState (svo_api.hip)
};

#define SVO_HIP(ctx, expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);      \
      return SVO_E_HIP;                                                           \
    }                                                                             \
  } while (0)

// A synchronous copy that stays on the context's own stream: hipMemcpy / hipMemset run on the NULL stream, which is a two-way barrier
// against every blocking stream of the process (the context's dedicated queues are blocking streams) and no barrier at all against
// non-blocking ones (round 4's hipMemset race) - the library makes no NULL-stream call.
static inline hipError_t svo_memcpy_sync(svo_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  return e;
}

// ---- stage launchers (each enqueues on ctx->stream, no sync) -------------------
// Images of a batch: index i < B is left image i, i >= B is right image i-B (nimg = B or 2B).
int svo_launch_orb(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR, int stride,
                   int B, int nimg);
int svo_launch_stereo(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR, int stride,
                      int B, const svo_camera* cam);
int svo_launch_descriptor_distance(svo_ctx* ctx, const uint8_t* a, const uint8_t* b, int count,
                                   int32_t* dist);
int svo_launch_hamming_argmin(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                              const uint8_t* mask, int32_t* idx, int32_t* best, int32_t* second);
int svo_launch_match_greedy(svo_ctx* ctx, const uint8_t* q, const uint8_t* q_skip, int M,
                            const uint8_t* t, int N, uint8_t* assigned, int max_dist, float ratio,
                            int32_t* idx, int32_t* best, int32_t* second, uint8_t* accepted,
                            const float* q_xy = nullptr, const float* t_xy = nullptr,
                            const int32_t* boxes = nullptr, int n_boxes = 0, const double* F = nullptr,
                            uint8_t* vetoed = nullptr);
int svo_launch_bf_match(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                        int32_t* train_idx, int32_t* dist, uint8_t* keep);
int svo_launch_bf_match_dev(svo_ctx* ctx, const uint8_t* q, const int* M_ptr, const uint8_t* t,
                            const int* N_ptr, int Mmax, int32_t* train_idx, int32_t* dist,
                            uint8_t* keep, int32_t* gmin);
int svo_launch_pose_opt(svo_ctx* ctx, const double* Xw, const double* obs, int n, const double* K,
                        double* T, svo_lm_stats* stats);
// cv::solvePnPRansac on device arrays: `subset` = 100 x 5 sample indices for this n, `hyp` = scratch of svo_pnp_hyp_bytes()
int svo_launch_pnp(svo_ctx* ctx, const double* Xw, const double* obs, int n, const double* K,
                   const double* Tfallback, const uint16_t* subset, void* hyp, double* T, uint8_t* mask,
                   svo_pnp_stats* stats);
size_t svo_pnp_hyp_bytes();
int svo_launch_epnp5_probe(svo_ctx* ctx, const double* X5, const double* u5, const double* K, double* Rt, int reps);
void svo_pnp_subsets(uint64_t state, int n, uint16_t* out /*[100 * 5]*/);
int svo_pose_lds_optin(svo_ctx* ctx);   // dynamic-LDS opt-in of the pose kernels (PoseLds > 64 KB)
template <typename T>
__host__ __device__ inline T* svo_byte_offset(T* p, size_t bytes) {
  return p ? reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + bytes) : p;
}
// Streams by role: +1 the ordered tail's chains (small dependent kernels whose workgroups must not queue behind the front end's
// thousands), -1 the batched front end running beside it, 0 everything else.  Maps onto the device's stream-priority range.
hipError_t svo_stream_create(hipStream_t* st, int role);
// svo_api.hip: the context's four main streams as hardware queues of their own, created back to back: four different dispatch pipes
int svo_stream_burst(svo_ctx* ctx);
// MSA (svo_msa.hip / svo_msa_graph.hip): one node of an aggregation tree by level-order position - first child's position |
// weight of the edge to the parent + (children << 8) | parent's position (-1: the root) | pixel
struct MsaBfsRec { int32_t cpos, meta, ppos, node; };
// svo_msa_graph.hip: svo_msa_tree's tree written directly as level-order records (rec: width * height entries), the level
// boundaries (levels + 1 entries) and the widest level.  SVO_E_CAPACITY: a node has more than 255 children (the records cannot
// hold it - the caller takes the child-list form).
int svo_msa_tree_rec(const uint8_t* m_img3, const double* r_gra, const double* c_gra, int width, int height, MsaBfsRec* rec,
                     std::vector<int32_t>* level_ptr, int* maxw, int32_t* root);
int svo_track_quiesce(svo_ctx* ctx, bool shard_too = true);   // waits for what overlapped tracker calls left in flight and may still read this context's arrays
int svo_shard_quiesce(svo_ctx* ctx);   // svo_track.hip: the sharded tracker's part of that
int svo_track_check_timeout(svo_ctx* ctx);   // svo_track.hip: sticky flag 4 -> SVO_E_TIMEOUT (once), fused pose launch off
int svo_hostfeed_flush(svo_ctx* ctx);  // svo_hostfeed.hip: records / front-end outputs waiting in pinned buffers -> the caller's arrays
void svo_hostfeed_release(svo_ctx* ctx);
// svo_api.hip: a new stream (made by `make`) that runs side by side with every non-null stream of `others`, chosen by measuring
// (up to six candidates); *attempts candidates tried, *percent the chosen one's worst "two chains together / one alone"
// `holds_up`: streams (latency chains, or big grids of their own) that a large grid waiting for CUs on the NEW stream must not hold up;
// `held_up_by`: streams whose large grids must not hold up the new one (see svo_api.hip: queues behind one dispatch pipe).
int svo_pick_stream(svo_ctx* ctx, const std::function<hipError_t(hipStream_t*)>& make, std::initializer_list<hipStream_t> others, hipStream_t* out,
                    int* attempts, int* percent, std::initializer_list<hipStream_t> holds_up = {}, std::initializer_list<hipStream_t> held_up_by = {});
// a large grid in dispatch on `blocker` beside a chain of short kernels on `victim`: the chain's time in per cent of its time alone
int svo_probe_block_percent(hipStream_t blocker, hipStream_t victim);
int svo_track_fe_batch_stream(svo_ctx* ctx);   // svo_track.hip: creates ctx->stream_fe_batch (a stream that runs beside the tail's two chains)
int svo_elas_batch_dev_hooked(svo_ctx* ctx, const uint8_t* d_L, const uint8_t* d_R, int stride, int W, int H, int B,
                              const svo_elas_params* params, float* d_D1, float* d_D2, int32_t* produced,
                              int (*hook)(void*, int, int), void* user);   // svo_elas.hip   // wait for the tails svo_track_batch_dev left in flight (svo_api.hip)
hipError_t svo_stream_create_masked(hipStream_t* st, int device, int percent);
int svo_frontend_nslices(const svo_ctx* ctx, int B);   // how svo_frontend_batch_dev slices a batch (svo_api.hip)
int svo_launch_disp2depth(svo_ctx* ctx, const float* disp, int count, float bf, float* depth);
int svo_launch_unproject(svo_ctx* ctx, const float* uvz, int n, const svo_camera* cam,
                         const float* Rwc, const float* twc, float* xyz);

extern "C" void svo_elas_release(svo_ctx* ctx);
void svo_msa_release(svo_ctx* ctx);
void svo_track_release(svo_ctx* ctx);   // tracker states, work records, second stream, events
int svo_upload_image(svo_ctx* ctx, const uint8_t* gray, int stride, int slot);
// dense ELAS stereo on images already in HBM; the two maps stay in HBM (valid until the next call)
// wall-clock profile entry for the host stages (reported next to the HIP-event kernel entries)
struct HostTimer {
  svo_ctx* ctx; const char* name; std::chrono::steady_clock::time_point t0;
  HostTimer(svo_ctx* c, const char* n) : ctx(c), name(n), t0(std::chrono::steady_clock::now()) {}
  ~HostTimer() {
    if (!ctx->profiling) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (auto& e : ctx->prof)
      if (e.name == name) { e.total_ms += ms; e.launches += 1; return; }
    SvoProfileEntry e; e.name = name; e.total_ms = ms; e.launches = 1;
    ctx->prof.push_back(e);
  }
};

int svo_msa_run_dev(svo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int pitch, int W, int H, int d, float* d_disp);
int svo_msa_run_many_dev(svo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int pitch, size_t frame_stride, int W, int H, int d,
                         int B, float* d_disp);
int svo_elas_run_dev(svo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int pitch, int W, int H,
                     const svo_elas_params* params, float** dD1, float** dD2, int* produced);

// profiling helper: time `fn` with HIP events on ctx->stream when profiling is on
struct SvoTimer {
  svo_ctx* ctx;
  const char* name;
  hipStream_t stream;
  SvoTimer(svo_ctx* c, const char* n, hipStream_t s = nullptr);   // s == nullptr: the ctx stream
  ~SvoTimer();
};
// Host CPUs this process may actually use at once: hardware threads, cut down to the cgroup CPU quota when there is one
// (a container with `cpu.max = 1600000 100000` gets 16 however many cores the machine shows; worker pools larger than
// that only buy throttling stalls - for every thread of the process, the one that feeds the GPU included).
int svo_host_cpus();

// Diagnostics of svo_elas_delaunay, summed over calls: microseconds spent sorting the points, building, emitting and
// ordering the triangles; points triangulated (printed and cleared by svo_elas_batch_dev under SVO_ELAS_BATCH_DEBUG=1)
extern std::atomic<long long> svo_delaunay_us[4];
extern std::atomic<long long> svo_delaunay_pts;

