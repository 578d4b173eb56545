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
  void GetVelocity();                                                   // :99-106
  void Tracklastframe();                                                // :107-121
  void SaveTrajectoryAndDraw(std::ofstream& f, std::ofstream& f2);      // :124-144

 public:
  svo_ctx* ctx = nullptr;
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
