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
};
