/* orc_track.c - CPU restatement of the per-frame tracking loop.
 *
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 *
 * Follows Tracking::Track (reference src/Tracking.cc:180-252) and what it calls:
 *   frame ctor / SetPose            src/frame.cc:36-73
 *   Tracking::init                  src/Tracking.cc:42-97   (frame 0)
 *   pnpmatch::poseEstimationPnP     src/pnpmatch.cc:33-251  (passes 1 and 2, PnP)
 *   Optimizer::PoseOptimization     src/Optimizer.cc:15-86
 *   frame::createmappoint           src/frame.cc:182-238
 *   mappoint ctor / AddObservation  src/mappoint.cc:10-23
 *   local-map culling               src/Tracking.cc:239-250
 * with the north-star substitutions already made by the other oracle files: ORB
 * from orc_orb.c, per-keypoint sparse-stereo depth (orc_stereo.c) instead of the
 * dense depthimg lookup, cv::solvePnPRansac as restated in orc_pnp_cv.c (no prior pose; if it
 * fails the pose stays at the last frame's).  Deterministic orders replace the
 * reference's run-dependent ones: LocalMapPoints (a std::set<mappoint*> ordered by
 * heap address, include/Tracking.h:41) is iterated in creation order.
 * Offline detection boxes (semantic gating, SURVEY f-3; main.cpp:82-95 format) are honoured
 * exactly where the reference uses them: the +-5 px creation gate in Tracking::init (with its
 * never-reset `dynamic` flag, src/Tracking.cc:44,61-66,86) and frame::createmappoint
 * (src/frame.cc:198-203), and the +-10 px epipolar veto of pass 1 (src/pnpmatch.cc:101-144)
 * with F from pnpmatch::poseEstimation2D_2D (:302-337).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

#define TRK_MAXKP 512
#define TRK_CAP 4096

typedef struct {
  float pos[3];
  uint8_t desc[32];
  uint8_t bad, in_local;
  int32_t create_id;
  int32_t gid;       /* creation sequence number: the identity the device tracker's records are compared by */
  int32_t obs_frame; /* id of the last frame that observed it (observations.count(cur)) */
} mp_t;

struct orc_tracker {
  int W, H, nfeatures;
  float fx, fy, cx, cy, bf;
  int frame_num;
  int next_gid;
  mp_t* pool;
  int npool;
  int lastN;
  int32_t last_mp[TRK_MAXKP];
  float lastTcw[16];
  float last_xy[TRK_MAXKP][2];        /* LastFrame.keypoints_l[i].pt */
  uint8_t last_desc[TRK_MAXKP * 32];  /* LastFrame.f_descriptor      */
  int last_vetoes;                    /* epipolar vetoes in the frame just tracked */
  float force_Tcw[16];                /* orc_track_force_pose: teacher forcing for the next whole-frame call */
  int has_force;
};

orc_tracker* orc_track_create(int W, int H, int nfeatures, float fx, float fy, float cx, float cy,
                              float bf) {
  if (nfeatures > TRK_MAXKP) return NULL;
  orc_tracker* t = (orc_tracker*)calloc(1, sizeof(orc_tracker));
  t->W = W; t->H = H; t->nfeatures = nfeatures;
  t->fx = fx; t->fy = fy; t->cx = cx; t->cy = cy; t->bf = bf;
  t->pool = (mp_t*)calloc(TRK_CAP, sizeof(mp_t));
  for (int i = 0; i < 16; ++i) t->lastTcw[i] = (i % 5 == 0) ? 1.f : 0.f;
  return t;
}
int orc_track_last_vetoes(const orc_tracker* t) { return t->last_vetoes; }
/* Teacher forcing for the whole-frame entries (orc_track_frame / _boxes / _dense): the NEXT frame is tracked as always and
 * reports its own pose, but the tracker then continues from `Tcw` (see orc_track_tail's Tcw_force). */
void orc_track_force_pose(orc_tracker* t, const float Tcw[16]) {
  memcpy(t->force_Tcw, Tcw, sizeof t->force_Tcw);
  t->has_force = 1;
}
void orc_track_destroy(orc_tracker* t) {
  if (!t) return;
  free(t->pool);
  free(t);
}

/* frame::SetPose (src/frame.cc:66-73): Rwc = Rcw^T, twc = -Rwc*tcw (CV_32F; the
 * product accumulates in double and rounds once [upstream-memory]). */
static void pose_inverse_f(const float Tcw[16], float Rwc[9], float twc[3]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Rwc[3 * r + c] = Tcw[4 * c + r];
  for (int r = 0; r < 3; ++r) {
    double acc = (double)Rwc[3 * r] * (double)Tcw[3] + (double)Rwc[3 * r + 1] * (double)Tcw[7] +
                 (double)Rwc[3 * r + 2] * (double)Tcw[11];
    twc[r] = (float)(-acc);
  }
}

/* new mappoint(x3D, frame, i) + AddObservation + create_id (src/frame.cc:226-232) */
static int new_mappoint(orc_tracker* t, const float xyz[3], const uint8_t* desc, int frame_id) {
  if (t->npool >= TRK_CAP) return -1;
  mp_t* m = &t->pool[t->npool];
  memcpy(m->pos, xyz, sizeof m->pos);
  memcpy(m->desc, desc, 32);
  m->bad = 0; m->in_local = 1; m->create_id = frame_id; m->obs_frame = -1;
  m->gid = t->next_gid++;
  return t->npool++;
}

int orc_track_frame(orc_tracker* t, const uint8_t* grayL, int strideL, const uint8_t* grayR,
                    int strideR, orc_track_result* res, int32_t* cur_mp_out) {
  return orc_track_frame_boxes(t, grayL, strideL, grayR, strideR, NULL, 0, res, cur_mp_out, NULL);
}

int orc_track_frame_boxes(orc_tracker* t, const uint8_t* grayL, int strideL, const uint8_t* grayR,
                          int strideR, const int32_t* boxes, int n_boxes, orc_track_result* res,
                          int32_t* cur_mp_out, double F_out[9]) {
  return orc_track_frame_dense(t, grayL, strideL, grayR, strideR, NULL, boxes, n_boxes, res, cur_mp_out, F_out);
}

/* dense_disp != NULL: the reference's live data flow (src/Tracking.cc:226-228) - a dense W x H float
 * disparity map (there from MSA, here from libelas) is turned into depth by frame::disp2Depth
 * (src/frame.cc:140-164: depth = bf / disp wherever disp != 0, else -1) and read at the truncated keypoint
 * position as `depthimg.at<float>(y, x)` does; keypoints_r = x - disp unless disp == -1 (frame.cc:122-138,
 * without its carry-over of the previous keypoint's rx).  The right image is then not used at all. */
int orc_track_frame_dense(orc_tracker* t, const uint8_t* grayL, int strideL, const uint8_t* grayR,
                          int strideR, const float* dense_disp, const int32_t* boxes, int n_boxes,
                          orc_track_result* res, int32_t* cur_mp_out, double F_out[9]) {
  const int NF = t->nfeatures;
  orc_kp* kp = (orc_kp*)calloc((size_t)NF, sizeof(orc_kp));
  uint8_t* desc = (uint8_t*)calloc((size_t)NF, 32);
  float* uR = (float*)calloc((size_t)NF, sizeof(float));
  float* depth = (float*)calloc((size_t)NF, sizeof(float));
  int32_t nkp = 0;
  if (dense_disp) {
    nkp = orc_orb_extract(grayL, t->W, t->H, strideL, NF, kp, desc, NULL);
    for (int i = 0; i < nkp; ++i) {
      const float disp = dense_disp[(size_t)(int)kp[i].y * t->W + (int)kp[i].x];
      uR[i] = disp != -1.0f ? kp[i].x - disp : -1.0f;
      depth[i] = disp != 0.0f ? t->bf / disp : -1.0f;
    }
  } else {
    orc_stereo_frame(grayL, strideL, grayR, strideR, t->W, t->H, NF, t->bf, t->fx, kp, desc, &nkp, uR,
                     depth, NULL, NULL, NULL);
  }
  const int rc = orc_track_tail(t, kp, desc, nkp, depth, boxes, n_boxes, t->has_force ? t->force_Tcw : NULL, res, cur_mp_out, F_out, NULL);
  t->has_force = 0;
  free(kp); free(desc); free(uR); free(depth);
  return rc;
}

/* The ordered tail of Tracking::Track ALONE (src/Tracking.cc:231-250 and what it calls: matching passes, PnP, pose
 * optimisation, createmappoint, cull) for a frame whose front-end results are given: nkp keypoints (cv::KeyPoint layout),
 * their 32-byte descriptors and per-keypoint depths (<= 0: none).  This is what the full-length parity test feeds with
 * the device front end's outputs (which are bit-exact against orc_stereo_frame by their own tests), so that all 4,541
 * frames of a KITTI-00-sized run meet the restatement without 4,541 CPU ORB extractions.
 *   Tcw_force (nullable, row-major 4x4 float): "teacher forcing" - the frame's own pose is computed and reported in
 *     res->Tcw as always, but the tracker then CONTINUES from Tcw_force (the pose another implementation found for this
 *     frame): the positions of the map points created at the frame's end and the fallback pose of the next frame's PnP
 *     use it.  Each frame's PnP + LM is then compared on identical inputs instead of on a trajectory whose rounding
 *     differences accumulate.
 *   dbg (nullable): per keypoint the identity (creation sequence number) of the map point matched to it / created from
 *     it, cv::solvePnPRansac's outcome for this frame (zeros on frame 0) and the pose it returned, before the CV_32F
 *     rounding and the LM. */
int orc_track_tail(orc_tracker* t, const orc_kp* kp, const uint8_t* desc, int nkp, const float* depth,
                   const int32_t* boxes, int n_boxes, const float* Tcw_force, orc_track_result* res,
                   int32_t* cur_mp_out, double F_out[9], orc_tail_debug* dbg) {
  const int NF = t->nfeatures;
  double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  t->last_vetoes = 0;
  if (nkp > NF) nkp = NF;
  if (dbg) {
    memset(dbg, 0, sizeof *dbg);
    for (int i = 0; i < 16; ++i) dbg->T_pnp[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < TRK_MAXKP; ++i) dbg->match_gid[i] = dbg->new_gid[i] = -1;
  }
  const int id = t->frame_num;
  int32_t cur_mp[TRK_MAXKP];
  for (int i = 0; i < TRK_MAXKP; ++i) cur_mp[i] = -1;
  float Tcw[16];
  for (int i = 0; i < 16; ++i) Tcw[i] = (i % 5 == 0) ? 1.f : 0.f; /* frame ctor: SetPose(I) */
  memset(res, 0, sizeof *res);
  res->frame_id = id; res->n_kp = nkp;
  for (int i = 0; i < nkp; ++i) res->n_stereo += depth[i] > 0;
  const double K[4] = {t->fx, t->fy, t->cx, t->cy};

  if (id == 0) {
    /* Tracking::init (src/Tracking.cc:42-97): pose I, map points for every keypoint with depth */
    const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
    int dynamic = 0; /* declared outside the loop and never reset (src/Tracking.cc:44) */
    for (int i = 0; i < nkp; ++i) {
      if (n_boxes > 0 && orc_point_in_boxes(kp[i].x, kp[i].y, boxes, n_boxes, 5)) dynamic = 1;
      if (depth[i] > 0 && !dynamic) {
        float uvz[3] = {kp[i].x, kp[i].y, depth[i]}, xyz[3];
        orc_unproject(uvz, 1, t->fx, t->fy, t->cx, t->cy, I3, z3, xyz);
        cur_mp[i] = new_mappoint(t, xyz, desc + 32 * (size_t)i, id);
        res->n_new_mappoints++;
      }
    }
  } else {
    /* pnpmatch::poseEstimation2D_2D (src/pnpmatch.cc:302-337): F from background matches.
     * Only its use inside boxes matters, so it is computed when the frame has boxes. */
    if (n_boxes > 0) {
      int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * TRK_MAXKP);
      int32_t* td = (int32_t*)malloc(sizeof(int32_t) * TRK_MAXKP);
      uint8_t* keep = (uint8_t*)malloc(TRK_MAXKP);
      double* p1 = (double*)malloc(sizeof(double) * 2 * TRK_MAXKP);
      double* p2 = (double*)malloc(sizeof(double) * 2 * TRK_MAXKP);
      orc_bf_match(desc, nkp, t->last_desc, t->lastN, ti, td, keep);
      int np = 0;
      for (int i = 0; i < nkp; ++i) {
        if (!keep[i]) continue;
        if (orc_point_in_boxes(kp[i].x, kp[i].y, boxes, n_boxes, 10)) continue;
        p1[2 * np] = kp[i].x; p1[2 * np + 1] = kp[i].y;
        p2[2 * np] = t->last_xy[ti[i]][0]; p2[2 * np + 1] = t->last_xy[ti[i]][1];
        ++np;
      }
      orc_fundamental_8point(p1, p2, np, F);
      free(ti); free(td); free(keep); free(p1); free(p2);
    }
    /* pass 1 (src/pnpmatch.cc:61-156): last frame's map points -> current keypoints */
    uint8_t* assigned = (uint8_t*)calloc(TRK_MAXKP, 1);
    for (int i = 0; i < t->lastN; ++i) {
      const int m = t->last_mp[i];
      if (m < 0 || t->pool[m].bad) continue;
      int32_t bi, bd, sd;
      orc_hamming_argmin(t->pool[m].desc, 1, desc, nkp, assigned, &bi, &bd, &sd);
      if (bi >= 0 && bd < 15) {
        /* epipolar veto for matches that land inside a padded box (:103-144) */
        if (n_boxes > 0 && orc_point_in_boxes(kp[bi].x, kp[bi].y, boxes, n_boxes, 10) &&
            orc_epipolar_distance(F, t->last_xy[i][0], t->last_xy[i][1], kp[bi].x, kp[bi].y) > 0.1) {
          t->pool[m].bad = 1;
          t->last_vetoes++;
          continue;
        }
        assigned[bi] = 1;
        cur_mp[bi] = m;
        t->pool[m].obs_frame = id;
        res->n_match_pass1++;
      }
    }
    /* pass 2 (src/pnpmatch.cc:159-199): local map points not yet observed by this frame */
    for (int m = 0; m < t->npool; ++m) {
      mp_t* p = &t->pool[m];
      if (!p->in_local || p->bad || p->obs_frame == id) continue;
      int32_t bi, bd, sd;
      uint8_t acc;
      orc_match_greedy(p->desc, NULL, 1, desc, nkp, assigned, 30, 2.f, &bi, &bd, &sd, &acc);
      if (acc) {
        cur_mp[bi] = m;
        p->obs_frame = id;
        res->n_match_pass2++;
      }
    }
    free(assigned);
    /* PnP (src/pnpmatch.cc:212-247): pose = PnP result (Tcl * I), stored CV_32F */
    double* Xw = (double*)malloc(sizeof(double) * 3 * TRK_MAXKP);
    double* ob = (double*)malloc(sizeof(double) * 2 * TRK_MAXKP);
    int n = 0;
    for (int j = 0; j < nkp; ++j)
      if (cur_mp[j] >= 0) {
        const mp_t* p = &t->pool[cur_mp[j]];
        Xw[3 * n] = p->pos[0]; Xw[3 * n + 1] = p->pos[1]; Xw[3 * n + 2] = p->pos[2];
        ob[2 * n] = kp[j].x; ob[2 * n + 1] = kp[j].y;
        ++n;
      }
    double Tp[16], Td[16];
    for (int i = 0; i < 16; ++i) Tp[i] = t->lastTcw[i];
    orc_pnp_stats ps;
    orc_pnp_ransac(Xw, ob, n, K, Tp, 0, Td, NULL, &ps);   /* fails -> the pose stays at the last frame's */
    res->n_pnp_inliers = ps.n_inliers;
    if (dbg) { dbg->pnp = ps; memcpy(dbg->T_pnp, Td, sizeof Td); }
    for (int i = 0; i < 16; ++i) Tcw[i] = (float)Td[i];
    free(Xw); free(ob);
  }
  /* Optimizer::PoseOptimization on every frame, frame 0 included (src/Tracking.cc:107-121) */
  {
    double* Xw = (double*)malloc(sizeof(double) * 3 * TRK_MAXKP);
    double* ob = (double*)malloc(sizeof(double) * 2 * TRK_MAXKP);
    int n = 0;
    for (int j = 0; j < nkp; ++j)
      if (cur_mp[j] >= 0) {
        const mp_t* p = &t->pool[cur_mp[j]];
        Xw[3 * n] = p->pos[0]; Xw[3 * n + 1] = p->pos[1]; Xw[3 * n + 2] = p->pos[2];
        ob[2 * n] = kp[j].x; ob[2 * n + 1] = kp[j].y;
        ++n;
      }
    double Td[16];
    for (int i = 0; i < 16; ++i) Td[i] = Tcw[i];
    orc_lm_stats ls;
    memset(&ls, 0, sizeof ls);
    orc_pose_opt(Xw, ob, n, K, Td, &ls, NULL, 0);
    res->n_lm_edges = n;
    res->lm_iterations = ls.iterations;
    for (int i = 0; i < 16; ++i) Tcw[i] = (float)Td[i];
    free(Xw); free(ob);
  }
  memcpy(res->Tcw, Tcw, sizeof Tcw);
  if (dbg)
    for (int j = 0; j < nkp; ++j)
      if (cur_mp[j] >= 0) dbg->match_gid[j] = t->pool[cur_mp[j]].gid;
  if (Tcw_force) memcpy(Tcw, Tcw_force, sizeof Tcw);   /* teacher forcing: carry on from the other implementation's pose */
  if (cur_mp_out) memcpy(cur_mp_out, cur_mp, sizeof(int32_t) * (size_t)NF);
  /* lastframe = frame(currentframe); lastframe.createmappoint (src/frame.cc:182-238) */
  {
    float Rwc[9], twc[3];
    pose_inverse_f(Tcw, Rwc, twc);
    for (int i = 0; i < nkp; ++i) {
      if (cur_mp[i] >= 0) continue;
      if (n_boxes > 0 && orc_point_in_boxes(kp[i].x, kp[i].y, boxes, n_boxes, 5)) continue;
      if (depth[i] > 0) {
        float uvz[3] = {kp[i].x, kp[i].y, depth[i]}, xyz[3];
        orc_unproject(uvz, 1, t->fx, t->fy, t->cx, t->cy, Rwc, twc, xyz);
        cur_mp[i] = new_mappoint(t, xyz, desc + 32 * (size_t)i, id);
        if (dbg && cur_mp[i] >= 0) dbg->new_gid[i] = t->pool[cur_mp[i]].gid;
        res->n_new_mappoints++;
      }
    }
  }
  t->lastN = nkp;
  for (int i = 0; i < nkp; ++i) { t->last_xy[i][0] = kp[i].x; t->last_xy[i][1] = kp[i].y; }
  memcpy(t->last_desc, desc, 32 * (size_t)nkp);
  if (F_out) memcpy(F_out, F, sizeof F);
  memcpy(t->last_mp, cur_mp, sizeof cur_mp);
  memcpy(t->lastTcw, Tcw, sizeof Tcw);
  /* cull (src/Tracking.cc:239-250) */
  if (t->frame_num >= 4)
    for (int m = 0; m < t->npool; ++m)
      if (t->pool[m].in_local && t->pool[m].create_id <= t->frame_num - 4) t->pool[m].in_local = 0;
  /* stable compaction of the pool: keep local-map members and points the last frame references */
  {
    uint8_t* live = (uint8_t*)calloc(TRK_CAP, 1);
    int32_t* remap = (int32_t*)malloc(sizeof(int32_t) * TRK_CAP);
    for (int m = 0; m < t->npool; ++m) live[m] = t->pool[m].in_local;
    for (int i = 0; i < t->lastN; ++i)
      if (t->last_mp[i] >= 0) live[t->last_mp[i]] = 1;
    int k = 0;
    for (int m = 0; m < t->npool; ++m) {
      remap[m] = live[m] ? k : -1;
      if (live[m]) { if (k != m) t->pool[k] = t->pool[m]; ++k; }
    }
    t->npool = k;
    for (int i = 0; i < t->lastN; ++i)
      if (t->last_mp[i] >= 0) t->last_mp[i] = remap[t->last_mp[i]];
    int nl = 0;
    for (int m = 0; m < t->npool; ++m) nl += t->pool[m].in_local;
    res->n_local_map = nl;
    free(live); free(remap);
  }
  t->frame_num++;
  return 0;
}
