#include "Optimizer.h"

#include <vector>

int Optimizer::PoseOptimization(frame* pFrame) {
  std::vector<double> Xw, obs;
  const int n = (int)pFrame->keypoints_l.size();
  for (int i = 0; i < n && i < pFrame->N; ++i) {
    mappoint* pMP = pFrame->MapPoints[i];
    if (!pMP) continue;
    obs.push_back(pFrame->keypoints_l[i].x); obs.push_back(pFrame->keypoints_l[i].y);
    for (int r = 0; r < 3; ++r) Xw.push_back(pMP->worldpos.at(r));
  }
  const int m = (int)obs.size() / 2;
  const double K[4] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy};
  double T[16];
  for (int i = 0; i < 16; ++i) T[i] = pFrame->Tcw.m[i];   // convert::toSE3Quat reads CV_32F
  svo_lm_stats st{};
  svo_pose_opt(pFrame->ctx, Xw.data(), obs.data(), m, K, T, &st);
  svo_host::Mat44f pose;
  for (int i = 0; i < 16; ++i) pose.m[i] = (float)T[i];   // convert::toCvMat -> CV_32F
  pFrame->SetPose(pose);
  return m;
}
