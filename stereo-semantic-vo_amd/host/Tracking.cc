#include "Tracking.h"

#include <iomanip>
#include <sstream>
#include <stdexcept>

#include "Optimizer.h"
#include "convert.h"
#include "pnpmatch.h"

using namespace svo_host;

// cv::FileStorage stand-in: "Camera.fx: 718.856" style keys of Stereo/KITTI*.yaml
bool read_camera_yaml(const std::string& path, svo_camera& cam, int* width, int* height) {
  std::ifstream in(path);
  if (!in) return false;
  std::string line;
  int found = 0;
  while (std::getline(in, line)) {
    const size_t c = line.find(':');
    if (c == std::string::npos || line[0] == '#' || line[0] == '%') continue;
    const std::string key = line.substr(0, c);
    const double v = atof(line.c_str() + c + 1);
    if (key == "Camera.fx") { cam.fx = (float)v; ++found; }
    else if (key == "Camera.fy") { cam.fy = (float)v; ++found; }
    else if (key == "Camera.cx") { cam.cx = (float)v; ++found; }
    else if (key == "Camera.cy") { cam.cy = (float)v; ++found; }
    else if (key == "Camera.bf") { cam.bf = (float)v; ++found; }
    else if (key == "Camera.width" && width) *width = (int)v;
    else if (key == "Camera.height" && height) *height = (int)v;
  }
  return found == 5;
}

Tracking::Tracking(const std::string& strSettingPath, int dev) : device(dev) {
  int w = 1241, h = 376;
  if (!read_camera_yaml(strSettingPath, K, &w, &h)) throw std::runtime_error("bad settings file " + strSettingPath);
  bf = K.bf;
  if (svo_create(&ctx, device, w, h, 500, 1) != SVO_OK) throw std::runtime_error("svo_create failed (no GPU?)");
  width = w; height = h;
  Velocity = eye4();
}
Tracking::Tracking(const svo_camera& cam, int w, int h, int dev) : device(dev), K(cam), bf(cam.bf) {
  if (svo_create(&ctx, device, w, h, 500, 1) != SVO_OK) throw std::runtime_error("svo_create failed (no GPU?)");
  width = w; height = h;
  Velocity = eye4();
}
Tracking::~Tracking() {
  if (ctx_batch) svo_destroy(ctx_batch);
  svo_destroy(ctx);
}

void Tracking::TrackBatch(const uint8_t* L, const uint8_t* R, int stride, int n, const double* timestamps,
                          const std::vector<std::vector<std::vector<int>>>& detection_box) {
  if (n < 1) return;
  if (n > batch_capacity) throw std::runtime_error("TrackBatch: more frames than batch_capacity");
  if (!ctx_batch) {
    if (svo_create(&ctx_batch, device, width, height, 500, batch_capacity) != SVO_OK) throw std::runtime_error("svo_create failed");
    if (svo_set_option(ctx_batch, "depth_source", depth_source) != SVO_OK || svo_track_reset(ctx_batch, &K) != SVO_OK)
      throw std::runtime_error(std::string("TrackBatch: ") + svo_last_error(ctx_batch));
    batch_results.reserve(1 << 16);   // (the library writes into this array until FinishBatches: it must not move)
  }
  const size_t first = batch_results.size();
  if (first + (size_t)n > batch_results.capacity()) {   // a longer sequence: complete what is in flight before the array moves
    if (svo_sync(ctx_batch) != SVO_OK) throw std::runtime_error(std::string("TrackBatch: ") + svo_last_error(ctx_batch));
    batch_results.reserve(2 * batch_results.capacity() + (size_t)n);
  }
  batch_results.resize(first + (size_t)n);
  for (int k = 0; k < n; ++k) batch_timestamps.push_back(timestamps ? timestamps[k] : 0.0);
  // the frames' boxes as the flat arrays svo_boxes_host wants (main.cpp:82-95: 4 ints per line, left right top bottom)
  int most = 0;
  for (int k = 0; k < n && k < (int)detection_box.size(); ++k) most = std::max(most, (int)detection_box[k].size());
  std::vector<int32_t> flat((size_t)n * std::max(most, 1) * 4, 0), cnt((size_t)n, 0);
  for (int k = 0; k < n && k < (int)detection_box.size(); ++k) {
    cnt[k] = (int32_t)detection_box[k].size();
    for (size_t b = 0; b < detection_box[k].size(); ++b)
      for (int j = 0; j < 4; ++j) flat[((size_t)k * most + b) * 4 + j] = detection_box[k][b][j];
  }
  const svo_boxes_host bx{flat.data(), cnt.data(), std::max(most, 1)};
  const int rc = svo_track_batch_host(ctx_batch, L, R, stride, n, most > 0 ? &bx : nullptr, batch_results.data() + first);
  if (rc != SVO_OK) throw std::runtime_error(std::string("svo_track_batch_host: ") + svo_last_error(ctx_batch));
  frame_num += n;
}

void Tracking::FinishBatches(std::ofstream& f, std::ofstream& f2) {
  if (!ctx_batch) return;
  if (svo_sync(ctx_batch) != SVO_OK) throw std::runtime_error(std::string("FinishBatches: ") + svo_last_error(ctx_batch));
  frame fr;   // SetPose's Rwc / twc arithmetic (src/frame.cc:66-73), SaveTrajectoryAndDraw's formats
  for (size_t k = 0; k < batch_results.size(); ++k) {
    Mat44f T;
    for (int i = 0; i < 16; ++i) T.m[i] = batch_results[k].Tcw[i];
    fr.SetPose(T);
    fr.timestamp = batch_timestamps[k];
    currentframe = &fr;
    SaveTrajectoryAndDraw(f, f2);
  }
  currentframe = nullptr;
}

void Tracking::init() {
  bool dynamic = false;   // declared outside the loop and never reset, exactly as src/Tracking.cc:44
  currentframe->SetPose(eye4());
  Velocity = eye4();
  const int n = (int)currentframe->keypoints_l.size();
  for (int i = 0; i < n && i < currentframe->N; ++i) {
    const float u = currentframe->keypoints_l[i].x, v = currentframe->keypoints_l[i].y;
    const float z = currentframe->kp_depth[i];
    for (const auto& b : currentframe->offline_box)
      if (u > b[0] - 5 && u < b[1] + 5 && v > b[2] - 5 && v < b[3] + 5) { dynamic = true; break; }
    Vec3f x3D;
    if (z > 0 && !dynamic && currentframe->UnprojectStereo(u, v, z, x3D)) {
      mappoint* newmp = new mappoint(x3D, currentframe, i);
      newmp->AddObservation(currentframe, i);
      newmp->create_id = (int)currentframe->id;
      LocalMapPoints.insert(newmp);
      currentframe->MapPoints[i] = newmp;
    }
  }
}

void Tracking::GetVelocity() {
  // Velocity = Tcw * LastTwc (computed, never consumed - src/Tracking.cc:99-106)
  const Mat44f LastTwc = convert::R_t_to_Twc(lastframe.Rcw, lastframe.tcw);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double acc = 0;
      for (int k = 0; k < 4; ++k) acc += (double)currentframe->Tcw.at(r, k) * (double)LastTwc.at(k, c);
      Velocity.at(r, c) = (float)acc;
    }
}

void Tracking::Tracklastframe() {
  if (frame_num == 0) init();
  else pnpmatch::poseEstimationPnP(currentframe, lastframe, LocalMapPoints, Velocity, K);
  Optimizer::PoseOptimization(currentframe);
}

void Tracking::SaveTrajectoryAndDraw(std::ofstream& f, std::ofstream& f2) {
  const Mat33f& R = currentframe->Rwc;
  const std::vector<float> q = convert::toQuaternion(R);
  const Vec3f& t = currentframe->twc;
  if (f2.is_open())
    f2 << std::setprecision(6) << currentframe->timestamp << std::setprecision(7) << " " << t.at(0) << " "
       << t.at(1) << " " << t.at(2) << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << std::endl;
  if (f.is_open())
    f << std::setprecision(9) << R.at(0, 0) << " " << R.at(0, 1) << " " << R.at(0, 2) << " " << t.at(0) << " "
      << R.at(1, 0) << " " << R.at(1, 1) << " " << R.at(1, 2) << " " << t.at(1) << " " << R.at(2, 0) << " "
      << R.at(2, 1) << " " << R.at(2, 2) << " " << t.at(2) << std::endl;
}

void Tracking::Track(const GrayImage& imLeft, const GrayImage& imRight, double timestamp, std::ofstream& f,
                     std::ofstream& f2, const std::vector<std::vector<int>>& detection_box) {
  currentframe = new frame(ctx, imLeft, imRight, timestamp, K, detection_box);
  if (depth_source == 1) {                // src/Tracking.cc:225-228 literally: features, dense map, lookups
    currentframe->featuredetect(imLeft);
    currentframe->ElasMatch(imLeft, imRight);
    currentframe->computekeypoint_r();
    currentframe->disp2Depth(bf);
  } else if (depth_source == 2) {         // the same four calls with the reference's own MB body (MSA)
    currentframe->featuredetect(imLeft);
    currentframe->MBdense(imLeft, imRight);
    currentframe->computekeypoint_r();
    currentframe->disp2Depth(bf);
  } else {
    currentframe->MB(imLeft, imRight);    // featuredetect + stereo association in one device pass
    currentframe->computekeypoint_r();
    currentframe->disp2Depth(bf);
  }
  currentframe->id = frame_num;
  Tracklastframe();
  SaveTrajectoryAndDraw(f, f2);
  if (frame_num > 0) GetVelocity();
  frame* prev = currentframe;
  lastframe = frame(currentframe);
  lastframe.createmappoint(LocalMapPoints);
  if (frame_num >= 4) {
    for (auto it = LocalMapPoints.begin(); it != LocalMapPoints.end();) {
      if ((*it)->create_id <= frame_num - 4) it = LocalMapPoints.erase(it);
      else ++it;
    }
  }
  // The reference leaks every frame (`new frame`, never deleted).  Map points key their
  // observations by the frame's ADDRESS, so the object must stay allocated for addresses to
  // remain unique; its payload (images, keypoints, descriptors) is released instead.
  prev->leftimg = GrayImage(); prev->rightimg = GrayImage();
  std::vector<svo_kp>().swap(prev->keypoints_l);
  std::vector<uint8_t>().swap(prev->f_descriptor);
  std::vector<float>().swap(prev->keypoints_r);
  std::vector<float>().swap(prev->kp_disp);
  std::vector<float>().swap(prev->kp_depth);
  std::vector<mappoint*>().swap(prev->MapPoints);
  std::vector<float>().swap(prev->match_score);
  std::vector<bool>().swap(prev->inlier);
  currentframe = nullptr;
  frame_num++;
}
