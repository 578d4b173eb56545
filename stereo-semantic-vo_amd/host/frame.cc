#include "frame.h"

#include <cmath>

using namespace svo_host;

static long g_mappoint_seq = 0;

mappoint::mappoint(const Vec3f& pos, frame* pFrame, int id) : worldpos(pos), bad(false) {
  memcpy(m_descriptor, pFrame->f_descriptor.data() + 32 * (size_t)id, 32);
  observation_num = 0;
  create_id = -1;
  seq = g_mappoint_seq++;
}
void mappoint::AddObservation(frame* FM, size_t idx) {
  if (observations.count(FM)) return;
  observations[FM] = (int)idx;
  observation_num++;
}

frame::frame() { Tcw = eye4(); }

frame::frame(frame* o)
    : ctx(o->ctx), N(o->N), timestamp(o->timestamp), id(o->id), leftimg(o->leftimg),
      rightimg(o->rightimg), keypoints_l(o->keypoints_l), keypoints_r(o->keypoints_r),
      kp_disp(o->kp_disp), kp_depth(o->kp_depth), f_descriptor(o->f_descriptor),
      MapPoints(o->MapPoints), match_score(o->match_score), inlier(o->inlier),
      offline_box(o->offline_box), width(o->width), height(o->height), fx(o->fx), fy(o->fy),
      cx(o->cx), cy(o->cy), bf(o->bf) {
  SetPose(o->Tcw);
}

frame::frame(svo_ctx* c, const GrayImage& imLeft, const GrayImage& imRight, double t,
             const svo_camera& K, const std::vector<std::vector<int>>& detection_box)
    : ctx(c), timestamp(t), leftimg(imLeft), rightimg(imRight), offline_box(detection_box) {
  fx = K.fx; fy = K.fy; cx = K.cx; cy = K.cy; bf = K.bf;
  width = (float)imLeft.cols; height = (float)imLeft.rows;
  N = 500;
  MapPoints.assign(N, nullptr);
  inlier.assign(N, false);
  match_score.assign(N, -1.f);
  SetPose(eye4());
}

// src/frame.cc:66-73
void frame::SetPose(const Mat44f& mTcw) {
  Tcw = mTcw;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { Rcw.at(r, c) = Tcw.at(r, c); Rwc.at(c, r) = Tcw.at(r, c); }
  for (int r = 0; r < 3; ++r) tcw.at(r) = Tcw.at(r, 3);
  for (int r = 0; r < 3; ++r) {
    const double acc = (double)Rwc.at(r, 0) * (double)tcw.at(0) + (double)Rwc.at(r, 1) * (double)tcw.at(1) +
                       (double)Rwc.at(r, 2) * (double)tcw.at(2);
    twc.at(r) = (float)(-acc);
  }
}

void frame::featuredetect(const GrayImage& img) {
  keypoints_l.assign(N, svo_kp{});
  f_descriptor.assign((size_t)N * 32, 0);
  int32_t n = 0;
  const int rc = svo_orb_extract(ctx, img.ptr(), img.cols, keypoints_l.data(), f_descriptor.data(), &n);
  if (rc != SVO_OK) n = 0;
  keypoints_l.resize(n);
}

// Stereo association for the keypoints of the LEFT image (fills keypoints_l / f_descriptor too, with
// the same content featuredetect produced).  Returns the number of keypoints with depth.
int frame::MB(const GrayImage& left, const GrayImage& right) {
  std::vector<svo_kp> kp(N);
  std::vector<uint8_t> desc((size_t)N * 32);
  std::vector<float> uR(N, -1.f), depth(N, -1.f);
  int32_t n = 0;
  svo_camera cam{fx, fy, cx, cy, bf};
  if (svo_stereo_frame(ctx, left.ptr(), left.cols, right.ptr(), right.cols, &cam, kp.data(), desc.data(),
                       &n, uR.data(), depth.data()) != SVO_OK)
    n = 0;
  kp.resize(n);
  keypoints_l = kp;
  f_descriptor = desc;
  kp_disp.assign(n, -1.f);
  kp_depth.assign(depth.begin(), depth.begin() + n);
  keypoints_r.assign(uR.begin(), uR.begin() + n);
  int valid = 0;
  for (int i = 0; i < n; ++i)
    if (depth[i] > 0) { kp_disp[i] = kp[i].x - uR[i]; ++valid; }
  return valid;
}

int frame::ElasMatch(const GrayImage& left, const GrayImage& right) {
  svo_elas_params ep;
  svo_elas_default_params(0, &ep);   // Elas::parameters(ROBOTICS), the library's default
  const size_t n = (size_t)left.cols * left.rows;
  dispimg.assign(n, -10.f);
  std::vector<float> D2(n, -10.f);
  const int32_t dims[3] = {left.cols, left.rows, left.cols};
  if (svo_elas_process(ctx, left.ptr(), right.ptr(), dispimg.data(), D2.data(), dims, &ep) != SVO_OK) return 0;
  int valid = 0;
  for (float d : dispimg) valid += d >= 0;
  return valid;
}

int frame::MBdense(const GrayImage& left, const GrayImage& right) {
  const size_t n = (size_t)left.cols * left.rows;
  std::vector<uint8_t> l3(3 * n), r3(3 * n), disp(n, 0);
  for (int i = 0; i < left.rows; ++i)
    for (int j = 0; j < left.cols; ++j) {
      const size_t t = (size_t)i * left.cols + j;
      l3[3 * t] = l3[3 * t + 1] = l3[3 * t + 2] = left.ptr()[(size_t)i * left.cols + j];
      r3[3 * t] = r3[3 * t + 1] = r3[3 * t + 2] = right.ptr()[(size_t)i * right.cols + j];
    }
  dispimg.assign(n, -1.f);
  if (svo_msa_solve(ctx, l3.data(), r3.data(), left.cols, left.rows, 3 * left.cols, 48, 1, disp.data()) != SVO_OK) return 0;
  int valid = 0;
  for (size_t t = 0; t < n; ++t) { dispimg[t] = (float)disp[t]; valid += disp[t] != 0; }
  return valid;
}

// src/frame.cc:122-138.  After MB() keypoints_r already holds the sub-pixel right x; after ElasMatch() it is read
// off the dense map at the truncated keypoint position, as `dispimg.at<float>(ly, lx)` does.
void frame::computekeypoint_r() {
  if (dispimg.empty()) return;
  const int n = (int)keypoints_l.size(), W = (int)width;
  keypoints_r.assign(n, -1.f);
  kp_disp.assign(n, -1.f);
  for (int i = 0; i < n; ++i) {
    const float d = dispimg[(size_t)(int)keypoints_l[i].y * W + (int)keypoints_l[i].x];
    kp_disp[i] = d;
    if (d != -1.f) keypoints_r[i] = keypoints_l[i].x - d;
  }
}
// src/frame.cc:140-164: depth = bf / disp wherever disp != 0, else -1 (per keypoint here)
void frame::disp2Depth(float bf_) {
  if (dispimg.empty()) return;   // MB(): kp_depth was produced on the device
  const int n = (int)keypoints_l.size();
  kp_depth.assign(n, -1.f);
  for (int i = 0; i < n; ++i)
    if (kp_disp[i] != 0.f) kp_depth[i] = bf_ / kp_disp[i];
}

// src/frame.cc:166-180
bool frame::UnprojectStereo(float u, float v, float z, Vec3f& x3D) const {
  if (!(z > 0)) return false;
  const float x = (u - cx) * z * (1 / fx);
  const float y = (v - cy) * z * (1 / fy);
  for (int r = 0; r < 3; ++r) {
    const double acc = (double)Rwc.at(r, 0) * (double)x + (double)Rwc.at(r, 1) * (double)y +
                       (double)Rwc.at(r, 2) * (double)z;
    x3D.at(r) = (float)(acc + (double)twc.at(r));
  }
  return true;
}

// src/frame.cc:182-238
void frame::createmappoint(std::set<mappoint*, mappoint_by_creation>& localmap) {
  const int n = (int)keypoints_l.size();
  for (int i = 0; i < n && i < N; ++i) {
    if (MapPoints[i]) continue;
    bool dynamic = false;
    const float u = keypoints_l[i].x, v = keypoints_l[i].y;
    const float z = kp_depth[i];
    for (const auto& b : offline_box)
      if (u > b[0] - 5 && u < b[1] + 5 && v > b[2] - 5 && v < b[3] + 5) { dynamic = true; break; }
    if (dynamic) continue;
    Vec3f x3D;
    if (UnprojectStereo(u, v, z, x3D)) {
      mappoint* newmp = new mappoint(x3D, this, i);
      newmp->AddObservation(this, i);
      newmp->create_id = (int)id;
      MapPoints[i] = newmp;
      localmap.insert(newmp);
    }
  }
}
