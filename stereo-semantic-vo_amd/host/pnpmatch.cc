#include <cstdio>
#include <cstdlib>
#include "pnpmatch.h"

#include <vector>

using namespace svo_host;

int pnpmatch::DescriptorDistance(svo_ctx* ctx, const uint8_t* a, const uint8_t* b) {
  int32_t d = -1;
  svo_descriptor_distance(ctx, a, b, 1, &d);
  return d;
}

void pnpmatch::find_feature_matches(frame* Cur, frame& Last, std::vector<int>& matches) {
  const int M = (int)Cur->keypoints_l.size(), N = (int)Last.keypoints_l.size();
  matches.assign(M, -1);
  if (M == 0 || N == 0) return;
  std::vector<int32_t> ti(M), td(M);
  std::vector<uint8_t> keep(M);
  svo_bf_match(Cur->ctx, Cur->f_descriptor.data(), M, Last.f_descriptor.data(), N, ti.data(), td.data(), keep.data());
  for (int i = 0; i < M; ++i)
    if (keep[i] && ti[i] >= 0) matches[i] = ti[i];
}

int pnpmatch::poseEstimation2D_2D(frame* Cur, frame& Last, double F[9]) {
  std::vector<int> matches;
  find_feature_matches(Cur, Last, matches);
  std::vector<double> p1, p2;
  for (size_t i = 0; i < matches.size(); ++i) {
    if (matches[i] < 0) continue;
    const float cx = Cur->keypoints_l[i].x, cy = Cur->keypoints_l[i].y;
    bool dynamic = false;
    for (const auto& b : Cur->offline_box)
      if (cx > b[0] - 10 && cx < b[1] + 10 && cy > b[2] - 10 && cy < b[3] + 10) { dynamic = true; break; }
    if (dynamic) continue;
    p1.push_back(cx); p1.push_back(cy);
    p2.push_back(Last.keypoints_l[matches[i]].x); p2.push_back(Last.keypoints_l[matches[i]].y);
  }
  return svo_fundamental_8point(Cur->ctx, p1.data(), p2.data(), (int)p1.size() / 2, F);
}

int pnpmatch::poseEstimationPnP(frame* Cur, frame& Last, std::set<mappoint*, mappoint_by_creation>& localmappoints,
                                const Mat44f& /*mVelocity*/, const svo_camera& K) {
  svo_ctx* ctx = Cur->ctx;
  const int nkp = (int)Cur->keypoints_l.size();
  std::vector<uint8_t> assigned(nkp > 0 ? nkp : 1, 0);
  // the reference always runs poseEstimation2D_2D first (:36); F is only consumed inside boxes
  double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int n_boxes = (int)Cur->offline_box.size();
  std::vector<int32_t> boxes;
  if (n_boxes > 0) {
    poseEstimation2D_2D(Cur, Last, F);
    for (const auto& b : Cur->offline_box) for (int k = 0; k < 4; ++k) boxes.push_back(b[k]);
  }
  // ---- pass 1: last frame's map points (src/pnpmatch.cc:61-156), threshold 15 --------------------
  {
    const int M = (int)Last.keypoints_l.size();
    std::vector<uint8_t> q((size_t)(M > 0 ? M : 1) * 32, 0), skip(M > 0 ? M : 1, 1), acc(M > 0 ? M : 1, 0);
    std::vector<int32_t> bi(M > 0 ? M : 1), bd(M > 0 ? M : 1), sd(M > 0 ? M : 1);
    for (int i = 0; i < M; ++i) {
      mappoint* mp = Last.MapPoints[i];
      if (mp && !mp->bad) { memcpy(&q[32 * (size_t)i], mp->m_descriptor, 32); skip[i] = 0; }
    }
    std::vector<uint8_t> vetoed(M > 0 ? M : 1, 0);
    if (M > 0) {
      std::vector<float> qxy(2 * (size_t)M), txy(2 * (size_t)(nkp > 0 ? nkp : 1));
      for (int i = 0; i < M; ++i) { qxy[2 * i] = Last.keypoints_l[i].x; qxy[2 * i + 1] = Last.keypoints_l[i].y; }
      for (int j = 0; j < nkp; ++j) { txy[2 * j] = Cur->keypoints_l[j].x; txy[2 * j + 1] = Cur->keypoints_l[j].y; }
      svo_match_greedy_gated(ctx, q.data(), skip.data(), M, Cur->f_descriptor.data(), nkp, assigned.data(), 15,
                             0.f, qxy.data(), txy.data(), boxes.data(), n_boxes, F, bi.data(), bd.data(),
                             sd.data(), acc.data(), vetoed.data());
    }
    for (int i = 0; i < M; ++i) {
      if (skip[i]) continue;
      if (vetoed[i]) { Last.MapPoints[i]->bad = true; continue; }   // :138-143
      if (i < (int)Cur->match_score.size()) Cur->match_score[i] = (float)sd[i] / (float)bd[i];  // :99
      if (acc[i]) {
        mappoint* mp = Last.MapPoints[i];
        Cur->MapPoints[bi[i]] = mp;
        mp->AddObservation(Cur, bi[i]);
      }
    }
  }
  // ---- pass 2: local map points not yet observed by this frame (:159-199), 30 / ratio 2 -----------
  {
    std::vector<mappoint*> rows;
    for (mappoint* mp : localmappoints)
      if (mp && !mp->bad && !mp->observations.count(Cur)) rows.push_back(mp);
    const int M = (int)rows.size();
    if (M > 0) {
      std::vector<uint8_t> q((size_t)M * 32), acc(M, 0);
      std::vector<int32_t> bi(M), bd(M), sd(M);
      for (int i = 0; i < M; ++i) memcpy(&q[32 * (size_t)i], rows[i]->m_descriptor, 32);
      svo_match_greedy(ctx, q.data(), nullptr, M, Cur->f_descriptor.data(), nkp, assigned.data(), 30, 2.f,
                       bi.data(), bd.data(), sd.data(), acc.data());
      for (int i = 0; i < M; ++i)
        if (acc[i]) {
          Cur->MapPoints[bi[i]] = rows[i];
          rows[i]->AddObservation(Cur, bi[i]);
        }
    }
  }
  // ---- PnP (:212-247) ------------------------------------------------------------------------------
  std::vector<double> pts3d, pts2d;
  for (int j = 0; j < nkp; ++j) {
    mappoint* mp = Cur->MapPoints[j];
    if (!mp) continue;
    pts2d.push_back(Cur->keypoints_l[j].x); pts2d.push_back(Cur->keypoints_l[j].y);
    for (int r = 0; r < 3; ++r) pts3d.push_back(mp->worldpos.at(r));
  }
  const int n = (int)pts2d.size() / 2;
  const double Kd[4] = {K.fx, K.fy, K.cx, K.cy};
  double Tp[16], T[16];
  for (int i = 0; i < 16; ++i) Tp[i] = Last.Tcw.m[i];
  svo_pnp_stats st{};
  svo_pnp_ransac(ctx, pts3d.data(), pts2d.data(), n, Kd, Tp, 0, T, nullptr, &st);   // cv::solvePnPRansac(..., false, 100, 8.0, 0.99)
  if (getenv("SVO_HOST_DEBUG")) fprintf(stderr, "host pnp: n %d best %d inliers %d iters %d t %.9g %.9g %.9g\n", n, st.best_hypothesis, st.n_inliers, st.iterations, T[3], T[7], T[11]);
  Mat44f Tcl;
  for (int i = 0; i < 16; ++i) Tcl.m[i] = (float)T[i];
  Cur->SetPose(Tcl);   // Tcl * I
  return n > 0 ? st.n_inliers / n : 0;
}
