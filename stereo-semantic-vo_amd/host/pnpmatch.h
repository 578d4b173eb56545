// pnpmatch.h - mirrors the reference's include/pnpmatch.h; the brute-force loops and the OpenCV
// calls of src/pnpmatch.cc run on the GPU through the C-ABI.
#pragma once
#include <set>

#include "frame.h"

class pnpmatch {
 public:
  // src/pnpmatch.cc:14-30 (a, b: 32-byte descriptors)
  static int DescriptorDistance(svo_ctx* ctx, const uint8_t* a, const uint8_t* b);
  // src/pnpmatch.cc:33-251: passes 1 and 2 + PnP-RANSAC; sets CurrentFrame's pose
  static int poseEstimationPnP(frame* cframe, frame& lastframe,
                               std::set<mappoint*, mappoint_by_creation>& localmappoints,
                               const svo_host::Mat44f& mVelocity, const svo_camera& K);
  // src/pnpmatch.cc:253-300: BruteForce-Hamming matches cur -> last kept when
  // distance <= max(2*min_dist, 30); matches[i] = index into the last frame or -1
  static void find_feature_matches(frame* CurrentFrame, frame& LastFrame, std::vector<int>& matches);
  // src/pnpmatch.cc:302-337: 8-point F from the matches whose current point is outside every
  // detection box padded by 10 px (row-major, p_last^T F p_cur = 0)
  static int poseEstimation2D_2D(frame* CurrentFrame, frame& LastFrame, double fundamental_matrix[9]);
};
