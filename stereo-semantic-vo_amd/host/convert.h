// convert.h - mirrors the reference's include/convert.h (pose packing helpers, src/convert.cc).
#pragma once
#include <vector>

#include "image.h"

class convert {
 public:
  static svo_host::Mat44f R_t_to_Tcw(const svo_host::Mat33f& R, const svo_host::Vec3f& t);
  static svo_host::Mat44f R_t_to_Twc(const svo_host::Mat33f& Rcw, const svo_host::Vec3f& tcw);
  // Eigen::Quaterniond(R) as convert::toQuaternion (src/convert.cc:76-88): {x, y, z, w}
  static std::vector<float> toQuaternion(const svo_host::Mat33f& M);
};
