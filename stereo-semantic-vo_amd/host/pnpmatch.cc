#include "pnpmatch.h"

#include <vector>

using namespace svo_host;

int pnpmatch::DescriptorDistance(svo_ctx* ctx, const uint8_t* a, const uint8_t* b) {
  int32_t d = -1;
  svo_descriptor_distance(ctx, a, b, 1, &d);
  return d;
}

int pnpmatch::poseEstimationPnP(frame* Cur, frame& Last, std::set<mappoint*, mappoint_by_creation>& localmappoints,
                                const Mat44f& /*mVelocity*/, const svo_camera& K) {
  svo_ctx* ctx = Cur->ctx;
  const int nkp = (int)Cur->keypoints_l.size();
  std::vector<uint8_t> assigned(nkp > 0 ? nkp : 1, 0);
  // ---- pass 1: last frame's map points (src/pnpmatch.cc:61-156), threshold 15 --------------------
  {
    const int M = (int)Last.keypoints_l.size();
    std::vector<uint8_t> q((size_t)(M > 0 ? M : 1) * 32, 0), skip(M > 0 ? M : 1, 1), acc(M > 0 ? M : 1, 0);
    std::vector<int32_t> bi(M > 0 ? M : 1), bd(M > 0 ? M : 1), sd(M > 0 ? M : 1);
    for (int i = 0; i < M; ++i) {
      mappoint* mp = Last.MapPoints[i];
      if (mp && !mp->bad) { memcpy(&q[32 * (size_t)i], mp->m_descriptor, 32); skip[i] = 0; }
    }
    if (M > 0)
      svo_match_greedy(ctx, q.data(), skip.data(), M, Cur->f_descriptor.data(), nkp, assigned.data(), 15, 0.f,
                       bi.data(), bd.data(), sd.data(), acc.data());
    for (int i = 0; i < M; ++i) {
      if (skip[i]) continue;
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
  svo_pnp_ransac(ctx, pts3d.data(), pts2d.data(), n, Kd, Tp, 0x5EED0000ULL + (uint64_t)Cur->id, T, nullptr, &st);
  Mat44f Tcl;
  for (int i = 0; i < 16; ++i) Tcl.m[i] = (float)T[i];
  Cur->SetPose(Tcl);   // Tcl * I
  return n > 0 ? st.n_inliers / n : 0;
}
