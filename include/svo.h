/* svo.h - C-ABI of the MI355X-native stereo-VO tracking front end.
 *
 * This is the drop-in boundary.  The reference
 * (zssjh/stereo-semantic-vo) exposes no plugin/FFI seam around its hot path: the
 * seams are C++ member functions that write into public members of `frame`.
 * Every entry point below names the reference method it replaces (file:line under
 * the reference tree).  The one true C-ABI the reference owns - the YOLO dlopen
 * interface, include/YOLOv3SE.h:61-64,221-230 - is the model for the conventions:
 * opaque handle, caller-owned buffers, plain pointers and sizes, int return.
 *
 * Conventions
 *   - `svo_ctx` is one tracker context bound to ONE GPU (one HIP stream inside).
 *     Not thread-safe per context; use one context per host thread / per GPU.
 *   - All buffers are caller-owned.  Entry points ending in `_dev` take DEVICE
 *     pointers (HBM-resident, e.g. torch tensors' data_ptr()) and enqueue on the
 *     context stream without synchronising; all others take HOST pointers and
 *     return after the result is in the caller's memory.
 *   - Return 0 (SVO_OK) or a negative svo_status.  Nothing here falls back to a
 *     CPU path: if no HIP device is usable, svo_create fails with SVO_E_NODEVICE.
 *   - Images are 8-bit gray, row-major, `stride` bytes per row.
 *   - Keypoints use the 28-byte layout of cv::KeyPoint (pt.x, pt.y, size, angle,
 *     response, octave, class_id) so `frame::keypoints_l` can alias the buffer.
 *   - Descriptors are n x 32 bytes row-major, like `frame::f_descriptor`
 *     (CV_8U 500x32, reference src/frame.cc:78).
 */
#ifndef SVO_H
#define SVO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_ABI_VERSION 7   /* 7 (round 6): + svo_track_batch_host, svo_track_sharded_host, svo_frontend_batch_host (pipelined host-fed entries:
                              images start in host memory, uploads run on a copy stream ahead of the front end, records come back to
                              host memory; svo_boxes_host), svo_create_ex (per-context stream mode), svo_stream_mode; svo_sync also
                              completes the host-fed calls' outputs.
                              6 (round 5, late): + svo_debug_stream_pipes.  BEHAVIOUR: a context's four main streams are hardware queues of their own,
                              created back to back at svo_create (four dispatch pipes; see svo_debug_stream_pipes) - they are BLOCKING
                              streams (they order against the NULL stream, as every hipExtStreamCreateWithCUMask stream does) without a
                              priority, kept off each other by CU masks; SVO_POOLED_QUEUES=1 restores the pooled streams of ABI 5.
                              5 (round 5): option "tail_fused" (default 1: RANSAC samples + frame part of the default solver in one launch);
                              svo_debug_stream_probe's second value is now a ratio in per cent (see there); "epnp_exact" accepts every
                              non-zero value again (ABI 3's meaning); svo_destroy waits for batched / sharded work in flight.
                              4 (round 4): + svo_track_epnp_fallbacks, svo_debug_stream_probe; "epnp_exact" defaults to 2 (the
                              order-preserving solver); options gate_group, hyp_first, dense_cu_percent, dense_two_launch,
                              epnp_force_seq, shard_force_staged; svo_track_sharded_dev overlaps consecutive calls.
                              No signature of version 3 changed.  BEHAVIOUR changes a version-3 caller sees: (i) the tracker's
                              default RANSAC solver is the bit-comparable one - 8.8 k instead of 14.4 k frames/s on the headline
                              run, and a sample with a zero / repeated singular value is re-solved sequentially (counted by
                              svo_track_epnp_fallbacks; "epnp_exact" = 0 restores round 3's solver and rate); (ii) "epnp_exact"
                              keeps version 3's meaning for every value but 2: any other non-zero value selects the one-lane
                              checker (mode 1). */

typedef enum svo_status {
  SVO_OK = 0,
  SVO_E_INVALID = -1,   /* bad argument (null pointer, size out of range) */
  SVO_E_NODEVICE = -2,  /* no usable HIP device / kernels not loadable  */
  SVO_E_NOMEM = -3,     /* device or host allocation failed              */
  SVO_E_HIP = -4,       /* a HIP runtime call failed; see svo_last_error */
  SVO_E_CAPACITY = -5,  /* batch / keypoint capacity of the ctx exceeded */
  SVO_E_TIMEOUT = -6    /* svo_sync: a bounded wait inside the tracker's pose chain ran out (sticky flag 4, svo_track_overflowed): the frame
                           concerned was tracked as a PnP failure (its record has n_pnp_inliers = -1), nothing stale was used; returned
                           once, the context continues with the two-launch pose chain ("tail_fused" = 0) */
} svo_status;

typedef struct svo_ctx svo_ctx;

/* cv::KeyPoint-compatible record (reference: vector<cv::KeyPoint> keypoints_l,
 * include/frame.h:48). */
typedef struct svo_kp {
  float x, y;      /* pt, level-0 pixel coordinates                      */
  float size;      /* 31 * 1.2^octave                                     */
  float angle;     /* degrees [0,360), intensity-centroid orientation     */
  float response;  /* Harris response                                     */
  int32_t octave;  /* pyramid level 0..7                                  */
  int32_t class_id;/* -1                                                  */
} svo_kp;

/* Camera intrinsics the reference reads from the yaml (src/Tracking.cc:24-38). */
typedef struct svo_camera {
  float fx, fy, cx, cy, bf;
} svo_camera;

/* Statistics of one pose-only LM run (mirrors what g2o's verbose mode prints,
 * Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:395-409). */
typedef struct svo_lm_stats {
  int32_t n_edges;        /* edges used (return value of PoseOptimization)  */
  int32_t iterations;     /* outer iterations executed (<= 10)              */
  int32_t trials_total;   /* inner lambda trials over all iterations        */
  int32_t terminated;     /* 1 if an LM stop rule fired before 10 iters     */
  double chi2_initial;    /* robust chi2 before the first iteration         */
  double chi2_final;      /* robust chi2 at the accepted estimate           */
  double lambda_final;
} svo_lm_stats;

/* Result of cv::solvePnPRansac (src/pnpmatch.cc:227). */
typedef struct svo_pnp_stats {
  int32_t n_points;
  int32_t n_inliers;       /* consensus of the winning sample (rows of `inliers`, src/pnpmatch.cc:229) */
  int32_t best_hypothesis; /* index 0..99 of the winning minimal sample, -1 if none */
  int32_t ok;              /* 0 => OpenCV would return false (fewer than 5 points / no consensus of 5): pose = fallback */
  int32_t iterations;      /* samples the adaptive RANSAC loop visited (<= 100) */
} svo_pnp_stats;

/* Per-frame record produced by the tracker (what Tracking::Track leaves behind:
 * pose + counters, src/Tracking.cc:180-252). */
typedef struct svo_track_result {
  float Tcw[16];           /* row-major 4x4, CV_32F like frame::Tcw        */
  int32_t frame_id;
  int32_t n_kp;            /* keypoints extracted on the left image         */
  int32_t n_stereo;        /* keypoints with depth > 0                      */
  int32_t n_match_pass1;   /* last-frame map points matched (best < 15)     */
  int32_t n_match_pass2;   /* local-map points matched (best<30, ratio>2)   */
  int32_t n_pnp_inliers;
  int32_t n_lm_edges;
  int32_t n_new_mappoints;
  int32_t n_local_map;     /* local-map size after culling                  */
  int32_t lm_iterations;
  int32_t reserved[2];
} svo_track_result;

/* Offline detection boxes (main.cpp:59-97: one box per line, 4 ints `left right top bottom`) of a device-resident
 * call, as HBM arrays: frame (or sequence) f of the call has n[f] boxes (0 <= n[f] <= 64) of four int32 each at
 * boxes + 4 * stride * f.  All pointers are DEVICE pointers; a NULL svo_boxes_dev*, NULL members or n[f] = 0 mean "no boxes". */
typedef struct svo_boxes_dev {
  const int32_t* boxes;
  const int32_t* n;
  int32_t stride;          /* boxes reserved per frame (>= the largest n[f]) */
} svo_boxes_dev;

/* The same for the host-fed entries (svo_track_batch_host, svo_track_sharded_host): HOST pointers, frame f of the call has
 * n[f] boxes at boxes + 4 * stride * f - main.cpp:82-95 reads them from one text file per frame. */
typedef struct svo_boxes_host {
  const int32_t* boxes;
  const int32_t* n;
  int32_t stride;
} svo_boxes_host;

/* Parity probe record of one tracked frame (svo_debug_track_frames). */
typedef struct svo_track_debug {
  int32_t match_gid[512];  /* per keypoint: identity (creation sequence number) of the map point matched to it by
                            * Tracking::init / pass 1 / pass 2 (CurrentFrame->MapPoints[j]), -1 none */
  int32_t new_gid[512];    /* per keypoint: identity of the map point frame::createmappoint made from it, -1 none */
  int32_t frame_id;
  int32_t pnp_best, pnp_iterations, pnp_inliers, pnp_ok;   /* cv::solvePnPRansac: winning sample, samples visited, consensus */
  int32_t active_rows[2], rounds[2];   /* matching passes 1 / 2: rows that could match at all, resolution rounds */
  int32_t resolve_us;      /* duration of the frame's k_ti_resolve launch (in-kernel wall clock) */
  int64_t rt[6];           /* in-kernel wall clock (s_memrealtime, 10 ns ticks): k_ti_resolve start / end, first RANSAC-sample
                            * workgroup start, k_tp_frame end, k_tp_frame start, (unused) */
  double T_pnp[16];        /* the pose solvePnPRansac returned (row-major 4x4), before the CV_32F rounding and the LM */
} svo_track_debug;

/* ---- lifecycle ----------------------------------------------------------- */

int svo_abi_version(void);
const char* svo_strerror(int status);
/* Text of the last HIP/runtime failure on this ctx ("" if none). */
const char* svo_last_error(const svo_ctx* ctx);

/* One context per GPU.  W,H: image size; max_kp: keypoint budget per image (the
 * reference hard-codes N = 500, src/frame.cc:54); max_batch: the largest number of
 * stereo pairs one batched call may carry (>= 1). */
int svo_create(svo_ctx** out, int device, int W, int H, int max_kp, int max_batch);
/* svo_create with a per-context stream mode (flags; 0 = svo_create's default):
 *   default: the context's four main streams (pose chain = the stream every host-buffer entry runs on, index chain, batched front
 *     end, dense stage) are four hardware queues of their own made back to back - four dispatch pipes whatever the process created
 *     before (svo_debug_stream_pipes).  They are BLOCKING streams (HIP has no non-blocking CU-masked stream): work another component
 *     of the process puts on the NULL stream orders against them both ways.  The pose / index queues keep off the CUs the batched
 *     front end is confined to (the first mask word: four CUs of every XCD) - 7/8 of the chip for everything that runs on them.
 *   SVO_CREATE_TAIL_ALL_CUS: those two queues may use every CU - for a context that does not run the batched tracker
 *     (front-end-only, ELAS / MSA, host-buffer entries: they run on the first queue and get the whole chip).
 *   SVO_CREATE_POOLED_STREAMS: the runtime's pooled, prioritised NON-BLOCKING streams instead (ABI 5): no ordering against the
 *     NULL stream; every further stream is picked by measuring against the streams it must run beside; the rate depends on what
 *     the process created before.  svo_create sets it when the environment has SVO_POOLED_QUEUES=1 (read at
 *     every call).  Also what a device without CU-masked streams falls back to.
 * svo_stream_mode: what is in effect - bit 0: dedicated queues, bit 1: the tail's queues are confined off the front end's CUs. */
#define SVO_CREATE_POOLED_STREAMS 1u
#define SVO_CREATE_TAIL_ALL_CUS 2u
int svo_create_ex(svo_ctx** out, int device, int W, int H, int max_kp, int max_batch, uint32_t flags);
int svo_stream_mode(const svo_ctx* ctx);
void svo_destroy(svo_ctx* ctx);
/* Tuning switches.  "pose_mfma" (0..2, default 1): how svo_pose_opt and the tracker's pose optimisation add up the edges' terms of
 * H = J^T W J, b = -J^T W e and chi2 - always in the edges' insertion order, each addition rounded (g2o's loops,
 * base_unary_edge.hpp:43-72): 1 four edges per v_mfma_f64_4x4x4_4b_f64 (A = 1.0: an in-order sum of the four lane groups' values),
 * 2 one lane per quantity with plain additions, 0 one lane for everything (the CPU loop as it stands).  Identical bits in all
 * three, and identical to the CPU restatement.
 * "fast_cand_cap" (default 2048, the maximum): length of the per-tile list of scored pixels in the FAST kernel; tiles
 * with more fall back to scanning the score tile - same results, the switch exists so that tests can force that path.
 * "track_lcap" (default 8, the maximum): claimable keypoints (Hamming distance < 30) a map point keeps as packed
 * entries in the tracker's matching passes; rows with more are evaluated on their full distance row instead - same results, the switch exists so that tests
 * can force that path.
 * "track_nblk" (default 3, the maximum): runner-up blockers stored with each packed entry; with fewer the matching passes
 * consult the full distance row more often - same results, a test switch like "track_lcap".
 * "frontend_overlap" (default 2, 0..8): slices of a batch svo_frontend_batch_dev runs side by side on their own streams
 * (scheduling only; 0 / 1: one chain; also one chain while svo_profile_enable is on, so that per-kernel times are those of
 * kernels running alone).
 * "pyr_fused" (default 1): the pyramid as three fused launches (levels kept in LDS); 0: one launch per level - same levels.
 * "track_group" (default 4, 1..64): most frames the batched tracker's pose chain takes over from its index chain per
 * stream event (group sizes ramp up 1, 1, 2, 4, ...) - scheduling only, same records.
 * "multi_pipeline" (default 0): svo_track_multi_step_dev runs the front end of a step beside the tail of the previous
 * step (its own stream, two alternating private output sets) - same records.  Contract while it is on: between consecutive
 * steps nothing else is enqueued on the context (svo_sync and reading results are fine); svo_track_multi_reset restarts it.
 * "depth_source" (default 0): where svo_track_frame / svo_track_batch_dev take keypoint depth from - 0 the sparse
 * epipolar matcher (north star), 1 a dense ELAS map (svo_elas_*), 2 a dense MSA map (svo_msa_solve with d = 48: the
 * reference's live configuration, src/Tracking.cc:225-228 + src/frame.cc:82-91), both read per keypoint as
 * frame::computekeypoint_r / disp2Depth do.
 * "fe_cu_percent" (default 12 = 32 of the 256 CUs, 10..100): share of the compute units (whole 32-bit words of the CU mask; measured
 * with tools/microbench/cu_mask_probe: the first word is FOUR CUs ON EACH of the eight XCDs, not one XCD) the front-end stream of
 * svo_track_batch_dev may use.  Its kernels would otherwise fill every CU while the ordered tail runs beside them, and the
 * tail's small dependent kernels then queue for slots (7 us per frame on average); the front end needs a tenth of the tail's
 * time on the whole chip, so an eighth of it keeps up (measured: 100 % 13.2 k, 25 % 14.1 k, 12 % 14.4 k frames/s).  Scheduling only -
 * same records.
 * "dense_cu_percent" (default 88 = seven eighths of the CUs, 10..100): the same for the dense front end's stream of svo_track_batch_dev with
 * depth_source = 1 (ORB + ELAS maps + depth lookups, in chunks, beside the tail of the earlier chunks): ELAS needs most of the chip,
 * the tail's 100 single-wave RANSAC workgroups need free CUs (measured with boxes, 256 frames per call: 100 % 5.4 k, 75 % 6.4 k,
 * 50 % 5.3 k frames/s; with the later 16-pair chunks: 75 % 6.7-6.8 k, 88 % 6.9 k, 100 % 6.0 k).  Scheduling only - same records.
 * "hyp_first" (default 8; 4, 8, 12 or 16): many sequences per step (svo_track_multi_step_dev): RANSAC samples per sequence in the
 * step's first launch; the second launch holds as many again, the third the rest, and their workgroups leave at once when
 * cv::solvePnPRansac's adaptive iteration bound says the loop never reaches their sample (it visits a median of 4 samples on
 * synth-kitti, 8 or fewer on 96 % of the frames).  Same records (64 sequences: 16 -> 95 k, 8 -> 102 k, 4 -> 98 k frames/s).
 * "gate_group" (default 1): frames that carry detection boxes need the brute-force matches against the previous frame and the
 * 8-point F (src/pnpmatch.cc:253-337) before pass 1's epipolar veto.  1 = in the batched entries both are computed for a group of
 * consecutive frames in one launch each, ahead of the index chain - they depend on front-end results only, not on tracker state;
 * 0 = two launches per frame inside the chain (what the frame-by-frame entry always does).  Same records.
 * "pose_flag" (default 0): 1 = one sequence's pose kernels learn that their frame has been matched from a per-frame tag the
 * index chain publishes in HBM (agent-scope stores / polls, bounded) instead of waiting on one stream event per group of
 * frames - the pose chain then never stands still because a LATER frame of its group is slow to match (+0.6 % on the
 * headline run).  Same records.  Off by default because the poll relies on the index kernel running concurrently with the
 * polling kernel: under a tool that serialises dispatches (rocprofv3 --kernel-trace) every poll runs into its time-out.
 * "epnp_exact" (default 2): which EPnP solves the RANSAC samples of cv::solvePnPRansac (src/pnpmatch.cc:227).
 *   2 = order-preserving wave solver (csrc/svo_epnp_ord_dev.h): every IEEE operation of OpenCV 3.2's loops (epnp.cpp,
 *       JacobiSVDImpl_ / SVBkSbImpl_ of lapack.cpp: cyclic one-sided Jacobi SVDs, SVD / QR least squares, IEEE division and
 *       square root, no FMA contraction, every sum in its own k order) is kept; independent operations are spread over one
 *       wavefront per sample (Jacobi pairs on disjoint rows side by side, k-ordered sums on v_mfma_f64_4x4x4).  Bit-identical,
 *       sample by sample, to the CPU restatement the tests compare with; the rare branches it does not reproduce (a zero or
 *       repeated singular value, 25 sweeps without convergence) are handed to the sequential solver of mode 1.
 *   1 (and, as in ABI 3, every non-zero value other than 2) = the same operations, one lane per sample, loop by loop
 *       (csrc/svo_epnp_exact_dev.h): the checker of mode 2, 7.5x slower.
 *   0 = statistical wave solver (csrc/svo_epnp_dev.h; the default up to round 3): parallel-order two-sided Jacobi, normal
 *       equations, FMA - the same estimator with another rounding; 1.65x faster than mode 2, RANSAC's winner differs from a
 *       CPU run on ~3 % of the frames.
 * "tail_fused" (default 1): one sequence, default solver, no dense stage beside the tail: the frame's RANSAC samples and its frame
 *   part (RANSAC's rule, PoseOptimization, the new map points' positions, the record) run as ONE launch (k_tp_tail_ord: 100
 *   sample workgroups + the frame's workgroup, which waits for their agent-scope results) instead of two - the launch boundary
 *   and the frame part's start-up leave the pose chain's critical path (8.74 k -> 8.9-9.0 k frames/s; 9.1-9.2 k with the frame part waiting for the first 8 samples only and the
 *   context's streams on four dispatch pipes).  Same records.
 * "tail_semi" (default 1): beside a dense stage (depth_source 1 / 2, one sequence) the frame's first 8 RANSAC samples are one launch,
 *   the other 92 and the frame part a second one (k_tp_hyp_ord_first + k_tp_tail_ord): a sample workgroup of the second launch
 *   replays cv::solvePnPRansac's iteration bound over the first 8 (beyond sample 16: waits for samples 8..15, replays over 16) and
 *   leaves when the loop can never reach it - 8 CUs' float64 pipelines per ordinary frame instead of 100 taken from the dense
 *   kernels, still two launches per frame (configs[4]: 6.1-6.5 k -> 6.4-6.6 k frames/s at 256 frames per call).  2 = also with
 *   eight or more sequences per step (measured slower: 99.5 k instead of 100.9 k pairs/s; a switch).  0 = off.  Same records.
 * "epnp_force_seq" (default 0, tests): 1 = mode 2 takes its sequential fallback for every sample.
 * "dense_two_launch" (default 0): svo_track_batch_dev with depth_source = 1: the tail beside the dense stage launches its RANSAC
 *   samples as in the many-sequence mode (see "hyp_first") - fewer CUs taken from ELAS on ordinary frames.  Measured: no gain
 *   (6.4 k -> 6.2 k frames/s); kept as a switch.  Same records. */
int svo_set_option(svo_ctx* ctx, const char* key, int value);
/* Block until everything enqueued on the ctx stream has finished. */
int svo_sync(svo_ctx* ctx);
/* The ctx's hipStream_t as an opaque pointer (for event timing by the caller). */
void* svo_stream(svo_ctx* ctx);

/* ---- geometry the ctx derived from (W,H) ---------------------------------- */

/* Pyramid level sizes / scale factors / per-level keypoint quotas, 8 levels.
 * (cv::ORB defaults, src/frame.cc:77: nfeatures 500, scaleFactor 1.2, nlevels 8.) */
int svo_orb_geometry(const svo_ctx* ctx, int32_t w[8], int32_t h[8], float scale[8],
                     int32_t quota[8]);

/* ---- a-2: frame::featuredetect (src/frame.cc:75-79) ------------------------ */

/* ORB keypoints + descriptors of ONE image.  kp: capacity max_kp; desc: max_kp*32
 * bytes; *n receives the count.  Order: octave ascending, then Harris response
 * descending (ties: raster order). */
int svo_orb_extract(svo_ctx* ctx, const uint8_t* gray, int stride, svo_kp* kp,
                    uint8_t* desc, int32_t* n);

/* Stage probes used by the parity tests (same kernels as svo_orb_extract):
 * copy out pyramid level `level` of the last extracted image (tight rows). */
int svo_debug_pyramid_level(svo_ctx* ctx, int image_slot, int level, uint8_t* out);
/* FAST corners (after 3x3 NMS and the 31-px border filter) of `level` of the last
 * extracted image: xy_score receives n x 3 int32 {x, y, score}, UNORDERED. */
int svo_debug_fast_corners(svo_ctx* ctx, int image_slot, int level, int32_t* xy_score,
                           int capacity, int32_t* n);

/* ---- a-3/a-4: stereo association + depth ----------------------------------- */
/* Replaces frame::MB + frame::computekeypoint_r + frame::disp2Depth
 * (src/Tracking.cc:226-228, src/frame.cc:82-91,122-164) with the sparse epipolar
 * matcher the north star asks for.  Extracts ORB on both images, then for each
 * LEFT keypoint: row-band candidates on the right, Hamming argmin, 11x11 SAD
 * refinement +-5 px at the keypoint's pyramid level, parabola sub-pixel,
 * median-based outlier cut.  uR[i] / depth[i] = -1 where no match
 * (depth = bf / disparity, like disp2Depth). */
int svo_stereo_frame(svo_ctx* ctx, const uint8_t* grayL, int strideL, const uint8_t* grayR,
                     int strideR, const svo_camera* cam, svo_kp* kpL, uint8_t* descL,
                     int32_t* nL, float* uR, float* depth);
/* Also return the right image's keypoints (parity tests). kpR/descR may be NULL. */
int svo_stereo_frame_ex(svo_ctx* ctx, const uint8_t* grayL, int strideL, const uint8_t* grayR,
                        int strideR, const svo_camera* cam, svo_kp* kpL, uint8_t* descL,
                        int32_t* nL, float* uR, float* depth, svo_kp* kpR, uint8_t* descR,
                        int32_t* nR);

/* frame::disp2Depth (src/frame.cc:140-164): depth = bf/disp where disp != 0, else
 * -1, over a dense H x W float map. */
int svo_disp2depth(svo_ctx* ctx, const float* disp, int count, float bf, float* depth);
/* frame::UnprojectStereo (src/frame.cc:166-180) for n (u,v,z) triples with pose
 * Rwc (row-major 3x3) / twc: out n x 3 floats; rows with z <= 0 are left NaN. */
int svo_unproject(svo_ctx* ctx, const float* uvz, int n, const svo_camera* cam,
                  const float Rwc[9], const float twc[3], float* xyz);

/* ---- a-7/a-8/a-9: Hamming matching (src/pnpmatch.cc:14-30,61-199) ----------- */

/* pnpmatch::DescriptorDistance for `count` pairs: a,b are count x 32 bytes. */
int svo_descriptor_distance(svo_ctx* ctx, const uint8_t* a, const uint8_t* b, int count,
                            int32_t* dist);

/* Independent-row argmin: for each query row i (M x 32) scan train rows j (N x 32)
 * in index order skipping t_mask[j] != 0, with the reference's update rule
 * `if (d < best) { second = best; best = d; idx = j; }` (src/pnpmatch.cc:89-94):
 * best/second start at 256, idx at -1, ties go to the lowest j, and `second` is
 * the running best just before the last improvement - NOT the true runner-up. */
int svo_hamming_argmin(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                       const uint8_t* t_mask, int32_t* best_idx, int32_t* best,
                       int32_t* second);

/* The reference's greedy, order-dependent assignment (src/pnpmatch.cc:61-156 pass 1
 * with max_dist = 15, ratio = 0; :159-199 pass 2 with max_dist = 30, ratio = 2):
 * rows are visited in index order; row i is skipped if q_skip[i] != 0; an accepted
 * row claims train column best_idx (assigned[] is updated in place), so later rows
 * cannot take it.  Accept rule: best < max_dist && (ratio <= 0 ||
 * (float)second/(float)best > ratio).  Outputs per row: best_idx (or -1), best,
 * second, accepted (0/1).  `assigned` is N bytes in/out. */
int svo_match_greedy(svo_ctx* ctx, const uint8_t* q, const uint8_t* q_skip, int M,
                     const uint8_t* t, int N, uint8_t* assigned, int max_dist, float ratio,
                     int32_t* best_idx, int32_t* best, int32_t* second, uint8_t* accepted);

/* svo_match_greedy with the reference's semantic veto of pass 1 (src/pnpmatch.cc:101-144): a row
 * that would be accepted, whose matched train point t_xy[best_idx] lies inside a detection box
 * padded by 10 px and more than 0.1 px off the line F*[q_xy[i],1], is NOT accepted, claims no
 * column, and is reported in vetoed[i] (the caller marks its map point bad).  q_xy: M x 2 floats
 * (last-frame point of each row), t_xy: N x 2 floats, boxes: n_boxes x {left,right,top,bottom}. */
int svo_match_greedy_gated(svo_ctx* ctx, const uint8_t* q, const uint8_t* q_skip, int M,
                           const uint8_t* t, int N, uint8_t* assigned, int max_dist, float ratio,
                           const float* q_xy, const float* t_xy, const int32_t* boxes, int n_boxes,
                           const double F[9], int32_t* best_idx, int32_t* best, int32_t* second,
                           uint8_t* accepted, uint8_t* vetoed);

/* a-6: cv BruteForce-Hamming match() as used by find_feature_matches
 * (src/pnpmatch.cc:253-300): nearest train row per query row (ties: lowest j),
 * then keep[i] = dist[i] <= max(2*min_dist, 30). */
int svo_bf_match(svo_ctx* ctx, const uint8_t* q, int M, const uint8_t* t, int N,
                 int32_t* train_idx, int32_t* dist, uint8_t* keep);

/* a-6: cv::findFundamentalMat(cur_pts, last_pts, CV_FM_8POINT) (src/pnpmatch.cc:336): normalised
 * 8-point algorithm in float64 on the device (one wavefront: normal matrix by wave reductions, wave-parallel Jacobi
 * eigen-solver, rank-2 projection).  pts1 / pts2: n x 2 (HOST arrays, n <= 512); F row-major with p2^T F p1 = 0,
 * scaled to F[8] = 1.  n < 8 -> F = 0 (OpenCV returns an empty matrix). */
int svo_fundamental_8point(svo_ctx* ctx, const double* pts1, const double* pts2, int n, double F[9]);

/* ---- a-10: PnP-RANSAC initial pose (src/pnpmatch.cc:212-247) ---------------- */
/* cv::solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 8.0, 0.99, inliers) as OpenCV 3.2 runs it (the
 * version the reference links): NO extrinsic guess - every RANSAC sample is five correspondences
 * drawn by cv::RNG((uint64)-1) and solved by EPnP alone; squared reprojection error <= 8^2 (evaluated in float) counts
 * an inlier; a sample replaces the best one iff its consensus is larger, and the iteration bound then drops to
 * log(1 - 0.99) / log(1 - w^5) for inlier ratio w; the pose returned is the winning sample's (3.2 discards the refit it
 * computes on the inliers).  Xw n x 3, obs n x 2 (doubles holding the reference's float values), n <= 512;
 * K = {fx,fy,cx,cy}; T_cw row-major 4x4.  T_fallback_cw is what T_cw receives when OpenCV would return false (the
 * reference would carry on with unset rvec / tvec there).  rng_state: 0 = OpenCV's (uint64)-1.  inlier_mask: n bytes
 * (may be NULL).  All 100 samples are solved at once, one wavefront each (csrc/svo_epnp_dev.h). */
int svo_pnp_ransac(svo_ctx* ctx, const double* Xw, const double* obs, int n, const double K[4],
                   const double T_fallback_cw[16], uint64_t rng_state, double T_cw[16],
                   uint8_t* inlier_mask, svo_pnp_stats* stats);

/* Parity probe: the EPnP solve of ONE five-point sample (what every RANSAC sample of svo_pnp_ransac runs): R row-major,
 * t, and (optionally) the mean reprojection errors of the three beta candidates N = 1, 2, 3 (epnp.cpp compute_pose). */
int svo_debug_epnp5(svo_ctx* ctx, const double Xw5[15], const double uv5[10], const double K[4], double R[9],
                    double t[3], double rep_err[3]);

/* ---- a-5: Optimizer::PoseOptimization (src/Optimizer.cc:15-86) -------------- */
/* Pose-only Levenberg-Marquardt exactly as g2o runs it for this graph: one SE3
 * vertex, n unary EdgeSE3ProjectXYZOnlyPose edges, information I2, Huber
 * delta = (double)(float)sqrt(5.991), optimize(10).  float64 throughout.
 * T_cw is in/out (row-major 4x4). */
int svo_pose_opt(svo_ctx* ctx, const double* Xw, const double* obs, int n, const double K[4],
                 double T_cw[16], svo_lm_stats* stats);

/* ---- Tracking::Track (src/Tracking.cc:180-252), device-resident ------------- */
/* Reset the tracker state held in HBM (lastframe, LocalMapPoints, frame_num). */
int svo_track_reset(svo_ctx* ctx, const svo_camera* cam);
/* Track one stereo pair given as HOST gray images; fills *res. Boxes: n_boxes (<= 64) x 4
 * int32 {left,right,top,bottom} (offline detections, main.cpp:82-95), may be NULL.  With boxes
 * the frame is gated as in the reference: no map points are created inside a box padded by 5 px,
 * and a pass-1 match that lands in a box padded by 10 px and lies > 0.1 px off the epipolar line
 * marks its map point bad (src/pnpmatch.cc:101-144). */
int svo_track_frame(svo_ctx* ctx, const uint8_t* grayL, int strideL, const uint8_t* grayR,
                    int strideR, double timestamp, const int32_t* boxes, int n_boxes,
                    svo_track_result* res);

/* Parity probe: map-point pool row matched to each keypoint of the frame just tracked
 * (CurrentFrame->MapPoints after both passes, -1 = none). cur_mp: max_kp int32. */
int svo_debug_track_matches(svo_ctx* ctx, int32_t* cur_mp);
/* Parity probe: the F matrix and the number of epipolar vetoes of the frame just tracked. */
int svo_debug_track_gate(svo_ctx* ctx, double F[9], int32_t* n_vetoed);
/* Parity probe: map-point identities, RANSAC outcome and PnP pose of `n` frames of the LAST device-resident call
 * (svo_track_batch_dev / _tail_dev / _sharded_dev: frames first .. first + n - 1 of that call; svo_track_multi_step_dev:
 * sequences first ..; svo_track_frame: first = 0, n = 1).  Synchronises. */
int svo_debug_track_frames(svo_ctx* ctx, int first, int n, svo_track_debug* out);

/* Parity probe: cv::solvePnPRansac's outcome for the frame just tracked, and the pose (row-major 4x4, before the CV_32F
 * rounding of SetPose) it handed to PoseOptimization.  Either pointer may be NULL. */
int svo_debug_track_pnp(svo_ctx* ctx, svo_pnp_stats* stats, double T_pnp[16]);

/* ---- throughput mode: batched, device-resident ------------------------------ */
/* B stereo pairs already in HBM: d_grayL/d_grayR are B images of H rows x `stride`
 * bytes.  Runs extraction on all 2B images and the sparse stereo association for
 * the B pairs in a handful of launches on the ctx stream; does NOT synchronise.
 * Outputs (device pointers, capacity B*max_kp each): d_kpL, d_descL (x32), d_nL (B),
 * d_uR, d_depth.  Any output pointer may be NULL to keep results inside the ctx. */
int svo_frontend_batch_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR,
                           int stride, int B, const svo_camera* cam, svo_kp* d_kpL,
                           uint8_t* d_descL, int32_t* d_nL, float* d_uR, float* d_depth);
/* Batched front end followed by the ordered tracking tail for the same B pairs
 * (frames are consecutive frames of ONE sequence, in order).  boxes (may be NULL): the frames' offline detection
 * boxes in HBM, gating the chain exactly as svo_track_frame's host boxes do (creation gates, F from brute-force
 * matches by the 8-point algorithm, epipolar veto - all on the device).  d_results: B
 * svo_track_result records in HBM.  Does not synchronise.  Consecutive calls overlap: the front end's outputs and the
 * index chain's hand-over records exist twice and alternate, so the first sub-batches and the matching of call c + 1 run
 * while the pose chain of call c is still busy.  Contract: the calls that continue one sequence are issued back to back
 * on this context; svo_sync it before any OTHER entry point uses the context in between (svo_track_reset synchronises itself). */
int svo_track_batch_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR,
                        int stride, int B, const svo_boxes_dev* boxes, svo_track_result* d_results);

/* The ordered tracking tail ALONE, for front-end results that already lie in HBM - e.g. produced by
 * svo_frontend_batch_dev of OTHER contexts / GPUs and copied here (pair k -> GPU k mod G, then the
 * strict chain of src/Tracking.cc:231-250 in frame order on one GPU).  Frame f of the call: keypoints at
 * d_kp + f * kp_stride, descriptors at d_desc + f * kp_stride * 32, keypoint count d_n[f], per-keypoint depths
 * (<= 0: none) at d_depth + f * kp_stride.  boxes (may be NULL): frame f's detection boxes.  Records are identical to
 * svo_track_batch_dev on the same frames.  Does not synchronise. */
int svo_track_tail_dev(svo_ctx* ctx, const svo_kp* d_kp, const uint8_t* d_desc, const int32_t* d_n,
                       const float* d_depth, int kp_stride, int B, const svo_boxes_dev* boxes,
                       svo_track_result* d_results);
/* ONE sequence tracked with G contexts in one process, typically one per GPU (BASELINE configs[3]
 * "pair k -> GPU k mod 8 ... ordered tail"): the stateless front end of stereo pair k runs on ctxs[k mod G], the strict
 * temporal chain of src/Tracking.cc:231-250 runs in frame order on ctxs[0], which pulls each frame's keypoints /
 * descriptors / depths (~34 KB) on its own stream - direct peer reads (hipMemcpyPeerAsync over xGMI) where
 * hipDeviceCanAccessPeer says so, a bounce through pinned host memory otherwise; no collective.
 * d_grayL[g] / d_grayR[g]: the pairs of context g, resident on ITS device, k ascending (pair k at local index k / G),
 * `stride` bytes per row.  All contexts: same W, H, max_kp; ctxs[g] needs max_batch >= ceil(B / G).  svo_track_reset on
 * ctxs[0] first; B frames per call; d_results: B records on ctxs[0]'s device.  Records are identical to
 * svo_track_batch_dev on a single context.  boxes (may be NULL): frame k's detection boxes, on ctxs[0]'s device.
 * Does not synchronise (svo_sync(ctxs[0]) does), and consecutive calls overlap: a call's front ends run in sub-batches on
 * streams other than the pose chain's, a gather stream on ctxs[0]'s device collects them into one of two staging sets, and
 * the front ends of the next call run while this call's tail is in flight (two contexts on one GPU: the tail's own period,
 * 115 us per frame).  The images, boxes and result records of a call must stay valid until svo_sync(ctxs[0]); any other
 * entry point of one of the contexts first waits for what these calls left in flight. */
int svo_track_sharded_dev(svo_ctx* const* ctxs, int G, const uint8_t* const* d_grayL, const uint8_t* const* d_grayR,
                          int stride, int B, const svo_boxes_dev* boxes, svo_track_result* d_results);
/* ---- throughput mode, HOST-fed and pipelined (main.cpp:159-195: the reference reads one stereo pair from disk per
 * Tracking::Track; SURVEY.md section 8e: "H2D 2 P bytes, D2H ~30 KB" per pair, pair k uploaded to GPU k mod G) ------------
 * svo_track_batch_host = svo_track_batch_dev for B consecutive frames of ONE sequence that start in HOST memory: grayL / grayR
 * are B images of H rows x `stride` bytes each, back to back (frame f at grayL + f * H * stride); `results`: B records in host
 * memory.  The call does not synchronise: the images are uploaded on a copy stream of the context's own, eight pairs per
 * event, and every front-end sub-batch waits only for its own pairs; the records are copied back behind the pose chain.
 * Pinned source images (hipHostMalloc / hipHostRegister, e.g. torch's pin_memory) are copied where they lie and must stay
 * valid until svo_sync or until the second following host-fed call on this context returns; pageable images are first
 * gathered into a pinned ring by up to four worker threads inside the call (the call returns when they are staged - they
 * may be reused at once).  `results` is complete after svo_sync(ctx) (pageable: copied out of a pinned buffer there), or after the
 * second following host-fed call returned.  Consecutive calls overlap exactly like svo_track_batch_dev's (two image sets
 * alternate): the uploads of call c + 1 run while the tail of call c is busy.  Records are byte-identical to
 * svo_track_batch_dev's on the same frames.  boxes (may be NULL): the frames' offline detection boxes, host arrays.
 * Works with every "depth_source" (the dense stages wait for the call's last upload instead of sub-batch by sub-batch).
 * Contract as for svo_track_batch_dev: the calls that continue one sequence are issued back to back; svo_sync before any
 * other entry point uses the context. */
int svo_track_batch_host(svo_ctx* ctx, const uint8_t* grayL, const uint8_t* grayR, int stride, int B,
                         const svo_boxes_host* boxes, svo_track_result* results);
/* svo_track_sharded_dev for a sequence that starts in HOST memory - BASELINE configs[3] as SURVEY.md section 8e words it:
 * stereo pair k is uploaded to the GPU of ctxs[k mod G] on THAT context's copy stream (each GPU pulls only its own pairs
 * over its own PCIe link), its front end runs there, the ordered tail on ctxs[0].  grayL / grayR: the B frames of the call in
 * frame order (frame k at grayL + k * H * stride); results: B records in host memory, complete after svo_sync(ctxs[0]).
 * Everything else as svo_track_sharded_dev / svo_track_batch_host; records identical to both. */
int svo_track_sharded_host(svo_ctx* const* ctxs, int G, const uint8_t* grayL, const uint8_t* grayR, int stride, int B,
                           const svo_boxes_host* boxes, svo_track_result* results);
/* The stateless front end alone (svo_frontend_batch_dev), host to host and pipelined the same way.  Output arrays (host
 * pointers, capacity B * max_kp each, d_nL: B; any may be NULL) are complete after svo_sync(ctx) or after the second
 * following call returned.  Its rate is what the PCIe link delivers (0.93 MB per 1241x376 pair). */
int svo_frontend_batch_host(svo_ctx* ctx, const uint8_t* grayL, const uint8_t* grayR, int stride, int B,
                            const svo_camera* cam, svo_kp* kpL, uint8_t* descL, int32_t* nL, float* uR, float* depth);

/* Sticky flag of the device tracker (synchronises).  4: see SVO_E_TIMEOUT.  1: capacity - *flag != 0 once a frame needed more than the 4096 live
 * map points the pool holds, or a map point stayed alive for more than 2^20 creations (its slot in the position table
 * was about to be reused).  Neither can happen with the reference's 500 keypoints per frame on sequences of KITTI
 * length; results after the flag is set are not the reference's. */
int svo_track_overflowed(svo_ctx* ctx, int32_t* flag);
/* Diagnostics of the default RANSAC solver (svo_set_option "epnp_exact" = 2; cv::solvePnPRansac of src/pnpmatch.cc:227):
 * how many samples since svo_track_reset took its sequential fallback (a zero or repeated singular value in one of EPnP's
 * decompositions, or 25 Jacobi sweeps without convergence).  Results are the same either way; a frame with such a sample
 * takes about ten times as long in the pose chain.  Synchronises. */
int svo_track_epnp_fallbacks(svo_ctx* ctx, int64_t* count);

/* Diagnostics: how the stream of the tail's index chain was chosen.  The ordered tail (src/Tracking.cc:231-250 per frame) runs
 * as two chains on two streams that must overlap.  The HIP runtime maps a process's streams onto a few hardware queues per
 * priority (4 by default, least-used first), and two streams on ONE queue run one after the other: with three older
 * high-priority streams alive in the process the first two new ones share a queue, and the tracker ran at 5.9 k instead of
 * 8.8 k frames/s (tools/queue_history.py, profiles/r05_queue_pairs.jsonl).  Every stream of a context that has to run beside
 * another one is therefore PICKED BY MEASURING (svo_pick_stream): a chain of eight short dependent kernels on the existing
 * stream alone, then the same chain on both streams together; up to six candidates, the first whose ratio is below 150 % is
 * kept.  out[0] = candidates tried for the index chain's stream (0: probe off, SVO_NO_STREAM_PROBE=1; -1: no tracker call
 * yet), out[1] = "two chains together / one alone" of the chosen one, per cent (~105: side by side; ~200: serialised - then
 * svo_last_error says so). */
int svo_debug_stream_probe(svo_ctx* ctx, int32_t out[2]);

/* Diagnostics: do the large grids of the batched front end hold the ordered tail up?  gfx950 places a process's hardware queues on
 * the command processor's four dispatch pipes in creation order (position among the live queues modulo 4), and a queue whose
 * launches wait behind each other holds up the other queues of its pipe: with the front end's queue on the pose chain's pipe
 * the tracker ran at half its rate (tools/microbench/queue_block_probe, queue_prio_probe; profiles/r05_queue_block.jsonl).  A
 * context therefore makes its four main streams (pose chain, index chain, batched front end, dense stage) as four queues of
 * its own back to back - four different pipes, whatever the process did before or does elsewhere later.  This call MEASURES it
 * (about 10 ms, everything of the context synchronised first): a 3,072 x 64-thread grid on the pose stream (out[0]) and on the
 * index stream (out[1]) alone, then beside four queued launches of slot-filling grids on the front end's stream; reported as
 * 100 + 100 x (delay in rounds of the filling grids' workgroups): ~100 clear, ~180 sharing CUs only, >= 330 held up behind the pipe.
 * out[2], out[3]: the same with the dense stage's stream filling.  -1: that stream does not exist (yet). */
int svo_debug_stream_pipes(svo_ctx* ctx, int32_t out[4]);

/* Many independent sequences on one GPU ("G independent sequences" for pure
 * throughput; no counterpart in the reference, whose tracker is one static chain per process,
 * src/Tracking.cc:180-252).  svo_track_multi_reset allocates n_seq (<= max_batch) tracker states;
 * every svo_track_multi_step_dev advances ALL of them by one frame: stereo pair q (device-resident,
 * `stride` bytes per row, pairs back to back) is the next frame of sequence q, d_results[q] its record.
 * The front end runs batched over the n_seq pairs and every kernel of the temporal tail runs one
 * workgroup per sequence, so the strictly serial chain of one sequence overlaps with the others'.
 * Results are identical to n_seq separate single-sequence trackers.  boxes (may be NULL): entry q = the detection boxes
 * of sequence q's frame of this step. */
int svo_track_multi_reset(svo_ctx* ctx, int n_seq, const svo_camera* cam);
int svo_track_multi_step_dev(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR, int stride,
                             int n_seq, const svo_boxes_dev* boxes, svo_track_result* d_results);

/* Names + accumulated HIP-event time (ms) and launch count of the kernels the ctx
 * has timed since svo_profile_reset (only when svo_profile_enable(ctx,1)). */
int svo_profile_enable(svo_ctx* ctx, int on);
int svo_profile_reset(svo_ctx* ctx);
int svo_profile_get(svo_ctx* ctx, int index, char* name, int name_cap, double* total_ms,
                    int64_t* launches);

/* ------------------------------------------------------------------------------------------------
 * Dense ELAS stereo.  Replaces the vendored libelas,
 * Thirdparty/libelas/src/elas.h:40-165 + elas.cpp:32-150 (`Elas::process`).  Results are bit-identical
 * to the compiled reference stage by stage; see DESIGN.md section 8 for the two documented caveats
 * (duplicate right-image support points, memory libelas reads uninitialised is defined as 0).
 * ---------------------------------------------------------------------------------------------- */

/* Elas::parameters (elas.h:60-83): same fields, same order, bools as int32. */
typedef struct svo_elas_params {
  int32_t disp_min, disp_max;
  float support_threshold;
  int32_t support_texture, candidate_stepsize, incon_window_size, incon_threshold, incon_min_support;
  int32_t add_corners, grid_size;
  float beta, gamma, sigma, sradius;
  int32_t match_texture, lr_threshold;
  float speckle_sim_threshold;
  int32_t speckle_size, ipol_gap_width, filter_median, filter_adaptive_mean, postprocess_only_left;
  int32_t subsampling;              /* 1: every second pixel only; D1/D2 are then (width/2) x (height/2) */
} svo_elas_params;

/* Optional taps of every intermediate (any pointer may be NULL) and optional override of the two
 * triangle lists; used by the parity tests, which run the reference's stages on the same lists. */
typedef struct svo_elas_taps {
  uint8_t *desc1, *desc2;              /* W*H*16 each */
  int32_t* support;                    /* (u,v,d) triples */
  int32_t n_support, cap_support;
  int32_t *tri1, *tri2;                /* (c1,c2,c3) triples */
  float *planes1, *planes2;            /* (t1a,t1b,t1c,t2a,t2b,t2c) per triangle */
  int32_t n_tri1, n_tri2, cap_tri;
  int32_t *grid1, *grid2;              /* (disp_max+2)*grid_w*grid_h */
  float *D1_raw, *D2_raw, *D1_lr, *D2_lr, *D1_seg, *D2_seg, *D1_gap, *D2_gap, *D1_mean, *D2_mean;
  const int32_t *tri1_in, *tri2_in;
  int32_t n_tri1_in, n_tri2_in;
} svo_elas_taps;

/* `Elas::parameters(setting)` (elas.h:87-142): setting 0 = ROBOTICS, 1 = MIDDLEBURY. */
int svo_elas_default_params(int32_t setting, svo_elas_params* params);

/* `Elas::process(I1, I2, D1, D2, dims)` (elas.h:162, elas.cpp:32-150): dims = {width, height, bytes per
 * line}; D1/D2 are width*height floats (left / right reference), negative = invalid.  Host buffers in,
 * host buffers out (with subsampling = 1 the maps are (width/2) x (height/2), elas.h:157-160).  With fewer
 * than 3 support points it returns SVO_OK and leaves D1/D2 untouched, as the reference does (elas.cpp:70-75). */
int svo_elas_process(svo_ctx* ctx, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                     const int32_t* dims, const svo_elas_params* params);
int svo_elas_process_ex(svo_ctx* ctx, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                        const int32_t* dims, const svo_elas_params* params, svo_elas_taps* taps);

/* Throughput mode (no counterpart in the reference): B stereo pairs already in HBM (pair b at
 * d_L/d_R + b * stride * height, `stride` bytes per row), dense maps for all of them into d_D1/d_D2 (HBM,
 * pair b at + b * map size).  The small kernels of different pairs overlap on several HIP streams and the
 * two sequential host stages (support point clean-up, triangulation) run on a pool of host threads.  Results
 * are identical to B calls of svo_elas_process.  produced (host, B entries, may be NULL): 0 where a pair had
 * fewer than 3 support points and its maps were left untouched.  Synchronises before it returns. */
int svo_elas_batch_dev(svo_ctx* ctx, const uint8_t* d_L, const uint8_t* d_R, int stride, int width, int height,
                       int B, const svo_elas_params* params, float* d_D1, float* d_D2, int32_t* produced);

/* `Elas::computeDelaunayTriangulation` (elas.cpp:445-503 -> Triangle "zQB"): host-side, needs no GPU.
 * xy = n (x,y) int32 pairs; writes up to cap (c1,c2,c3) triples, counter-clockwise, in canonical order
 * (smallest index first, sorted); *n_tri = number of triangles. */
int svo_elas_delaunay(const int32_t* xy, int32_t n, int32_t* tri, int32_t cap, int32_t* n_tri);

/* ------------------------------------------------------------------------------------------------
 * MSA dense stereo: stages built so far.
 * ---------------------------------------------------------------------------------------------- */

/* `ctmf(src, dst, width, height, src_step_row, dst_step_row, r, channels, memsize)` (Thirdparty/MB/ctmf.h:7,
 * ctmf.c:222-433; called at Thirdparty/MB/MSA.cpp:58-59 with r = 1 on 3 channels and MSA.cpp:1006 with r = 2 on
 * 1 channel): per-channel (2r+1)^2 median of an 8-bit interleaved image, window clamped to the image.
 * Host buffers in and out; r in 1..3, channels in 1..4; `memsize` (a CPU cache hint) is dropped. */
int svo_ctmf(svo_ctx* ctx, const uint8_t* src, uint8_t* dst, int width, int height, int src_step_row,
             int dst_step_row, int r, int channels);

/* `MSA::init(l, r)` (Thirdparty/MB/MSA.cpp:22-63 with gradient :65-76, getCost :78-108, ctmf, gradient_after_ctmf
 * :110-139): from two 8UC3 BGR images (width x height, `step` bytes per row) the two matching-cost volumes
 * (height*width*disp floats each, disparity fastest; disp = MSA's `Disp` = d + 1, 49 in src/frame.cc:86), the
 * median-filtered colour images and the row / column gray gradients of those (float64, height*width each).
 * Host buffers in and out.  The later stages of MSA::solve are not built. */
int svo_msa_init(svo_ctx* ctx, const uint8_t* bgrL, const uint8_t* bgrR, int width, int height, int step, int disp,
                 float* costL, float* costR, uint8_t* m_img3L, uint8_t* m_img3R, double* r_graL, double* c_graL,
                 double* r_graR, double* c_graR);

/* The spanning tree MSA aggregates over, for one image: `build` -> `Tarjan` -> `getSeq0` -> `Kruskal1` -> `getSeq`
 * (MSA.cpp:152-373, 661-808, 898-926).  Host-side, needs no GPU.  m_img3 / r_gra / c_gra: outputs of svo_msa_init for that
 * image.  Outputs in the form svo_msa_tree_dp takes: seq[width*height], child_ptr[+1], child / child_c[-1], *root. */
int svo_msa_tree(const uint8_t* m_img3, const double* r_gra, const double* c_gra, int width, int height, int32_t* seq,
                 int32_t* child_ptr, int32_t* child, uint8_t* child_c, int32_t* root);

/* `MSA::TreeDp(cost)` (MSA.cpp:929-990) with `setExp(o)` (:1126-1130): two-pass aggregation of a cost volume (N nodes x
 * D disparities, float) over a spanning tree.  The tree is given as the reference holds it at that point: `seq` = its
 * BFS order from `root` (:898-926) and, per node, the children in the order its adjacency chain yields them (CSR:
 * child_ptr[N+1], child[N-1], child_c[N-1] = the edge's colour weight 0..255) - the order fixes the float rounding.
 * costA: N*D floats.  Host buffers in and out. */
int svo_msa_tree_dp(svo_ctx* ctx, const float* cost, int N, int D, const int32_t* seq, const int32_t* child_ptr,
                    const int32_t* child, const uint8_t* child_c, int root, double o, float* costA);
/* `MSA::WTA(disparity)` (MSA.cpp:992-1006): first minimum over the D disparities per pixel, then the 5x5 ctmf. */
int svo_msa_wta(svo_ctx* ctx, const float* costA, int width, int height, int D, uint8_t* disparity);
/* `MSA::LRcheck(d1, d2, cost)` (MSA.cpp:1027-1105): mask = pixels with d1 > 0 whose right-map disparity at x - d1 is
 * the same; their cost row becomes |d - d1|, every other row 0. */
int svo_msa_lrcheck(svo_ctx* ctx, const uint8_t* d1, const uint8_t* d2, int width, int height, int D, float* cost,
                    uint8_t* mask);

/* `MSA::solve(l, r, d, scale, Save)` (Thirdparty/MB/MSA.cpp:1132-1169), what `frame::MB` calls with d = 48, scale = 1
 * (src/frame.cc:82-91): dense left-reference disparity of two 8UC3 BGR images, one byte per pixel = disparity * scale.
 * The per-pixel and per-node stages run on the GPU with the cost volumes resident in HBM, the two aggregation trees are
 * built on two host threads.  No imshow / imwrite("test.png") (the reference does both on every call). */
int svo_msa_solve(svo_ctx* ctx, const uint8_t* bgrL, const uint8_t* bgrR, int width, int height, int step, int d, int scale,
                  uint8_t* disparity);

/* Throughput mode of the same solver (no counterpart in the reference): B gray stereo pairs already in HBM (pair b at
 * d_L / d_R + b * stride * height, `stride` bytes per row; a gray frame stands for the B = G = R colour image the
 * reference gets from a grayscale file), disparity maps as floats - `disp_img.convertTo(disp_32f, CV_32F, 1)` of
 * frame::MB - into d_disp (HBM, map b at + b * width * height).  Frames are solved 16 at a time: their aggregation
 * trees are built side by side on host threads and their level sweeps share launches.  Results are identical to B
 * calls of svo_msa_solve with scale 1.  Synchronises before it returns. */
int svo_msa_batch_dev(svo_ctx* ctx, const uint8_t* d_L, const uint8_t* d_R, int stride, int width, int height, int B, int d,
                      float* d_disp);

#ifdef __cplusplus
}
#endif
#endif /* SVO_H */
