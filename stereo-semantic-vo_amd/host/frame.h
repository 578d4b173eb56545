// frame.h - per-frame container, mirrors the reference's include/frame.h (members keep their names;
// cv::Mat members become PODs).  The OpenCV / MSA calls of src/frame.cc are replaced by the C-ABI:
//   featuredetect      -> svo_orb_extract          (reference src/frame.cc:75-79)
//   MB                 -> svo_stereo_frame         (src/frame.cc:82-91; sparse matcher, north star)
//   computekeypoint_r  -> copies uR                (src/frame.cc:122-138)
//   disp2Depth         -> depth = bf / disparity   (src/frame.cc:140-164; per keypoint)
//   UnprojectStereo    -> same float arithmetic    (src/frame.cc:166-180)
//   MBdense            -> svo_msa_solve            (src/frame.cc:82-91 as the reference has it: MSA::solve(l, r, 48, 1))
//   ElasMatch          -> svo_elas_process         (src/frame.cc:93-120 dense disparity; the reference's body is
//                         OpenCV SGBM under that name, the vendored solver it names is libelas: include/frame.h:15)
#pragma once
#include <set>
#include <vector>

#include "../../include/svo.h"
#include "image.h"
#include "mappoint.h"

class frame {
 public:
  frame();
  frame(frame* other);   // the reference's copy used for `lastframe = frame(currentframe)`
  frame(svo_ctx* ctx, const svo_host::GrayImage& imLeft, const svo_host::GrayImage& imRight,
        double timestamp, const svo_camera& K, const std::vector<std::vector<int>>& detection_box);

  void SetPose(const svo_host::Mat44f& mTcw);
  void featuredetect(const svo_host::GrayImage& img);
  int MB(const svo_host::GrayImage& left, const svo_host::GrayImage& right);
  // dense left-reference disparity map (float, width x height, negative = invalid) into `dispimg`;
  // returns the number of valid pixels
  int ElasMatch(const svo_host::GrayImage& left, const svo_host::GrayImage& right);
  // the reference's own MB body: MSA dense disparity (0 = none) of the two images as B = G = R colour images
  int MBdense(const svo_host::GrayImage& left, const svo_host::GrayImage& right);
  void disp2Depth(float bf);
  bool UnprojectStereo(float u, float v, float z, svo_host::Vec3f& x3D) const;
  void createmappoint(std::set<mappoint*, mappoint_by_creation>& localmap);
  void computekeypoint_r();

 public:
  svo_ctx* ctx = nullptr;
  int N = 0;                              // reference hard-codes 500 (src/frame.cc:54)
  double timestamp = 0;
  long id = 0;
  svo_host::GrayImage leftimg, rightimg;
  std::vector<svo_kp> keypoints_l;        // cv::KeyPoint layout
  std::vector<float> keypoints_r;         // right-image x per keypoint (-1: none)
  std::vector<float> kp_disp, kp_depth;   // per-keypoint stand-ins for dispimg / depthimg
  std::vector<float> dispimg;             // dense disparity of ElasMatch (empty until called)
  std::vector<uint8_t> f_descriptor;      // N x 32
  std::vector<mappoint*> MapPoints;
  std::vector<float> match_score;
  std::vector<bool> inlier;
  std::vector<std::vector<int>> offline_box;
  float width = 0, height = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
  svo_host::Mat44f Tcw;
  svo_host::Mat33f Rcw, Rwc;
  svo_host::Vec3f tcw, twc;
};
