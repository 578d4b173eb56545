// Tracking.h - per-frame orchestrator, mirrors the reference's include/Tracking.h.
#pragma once
#include <fstream>
#include <set>
#include <string>

#include "frame.h"

class Tracking {
 public:
  // Reads Camera.fx/fy/cx/cy/bf from an ORB-SLAM2-format yaml (the only five keys the reference
  // consumes, src/Tracking.cc:22-40) and creates the GPU context.
  Tracking(const std::string& strSettingPath, int device = 0);
  // 0: sparse epipolar stereo (default); 1: the reference's live flow, a dense disparity map (libelas here,
  // MSA there) -> disp2Depth -> per-keypoint lookups (src/Tracking.cc:226-228)
  int depth_source = 0;
  Tracking(const svo_camera& cam, int width, int height, int device = 0);
  ~Tracking();
  void init();                                                          // src/Tracking.cc:42-97
  // src/Tracking.cc:180-252 (imdepth / img_detect / Pangolin matrix arguments dropped: GUI only)
  void Track(const svo_host::GrayImage& imLeft, const svo_host::GrayImage& imRight, double timestamp,
             std::ofstream& f, std::ofstream& f2, const std::vector<std::vector<int>>& detection_box);
  // The same loop PIPELINED (svo_track_batch_host): the next n stereo pairs of the sequence at once, images in host memory
  // (n x rows x stride bytes each side, frame k at L + k * rows * stride); returns at once - uploads, front end and the ordered
  // tail run behind the caller's back while it decodes the next pairs.  detection_box[k]: frame k's offline boxes (may be
  // empty).  FinishBatches() waits for everything and writes the trajectory rows of all batched frames in order
  // (SaveTrajectoryAndDraw's two formats).  Not to be mixed with Track() on one sequence.
  void TrackBatch(const uint8_t* L, const uint8_t* R, int stride, int n, const double* timestamps,
                  const std::vector<std::vector<std::vector<int>>>& detection_box);
  void FinishBatches(std::ofstream& f, std::ofstream& f2);
  void GetVelocity();                                                   // :99-106
  void Tracklastframe();                                                // :107-121
  void SaveTrajectoryAndDraw(std::ofstream& f, std::ofstream& f2);      // :124-144

 public:
  svo_ctx* ctx = nullptr;
  svo_ctx* ctx_batch = nullptr;           // TrackBatch's context (max_batch = batch_capacity), made on first use
  int batch_capacity = 64;
  std::vector<svo_track_result> batch_results;   // one record per batched frame (stable storage: reserved for the sequence)
  std::vector<double> batch_timestamps;
  int width = 0, height = 0;
  int device;
  frame lastframe;
  frame* currentframe = nullptr;
  int frame_num = 0;                      // static in the reference (one tracker per process)
  svo_camera K{};
  float bf = 0;
  std::set<mappoint*, mappoint_by_creation> LocalMapPoints;
  svo_host::Mat44f Velocity;
};

bool read_camera_yaml(const std::string& path, svo_camera& cam, int* width, int* height);
