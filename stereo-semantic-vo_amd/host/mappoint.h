// mappoint.h - 3-D landmark, mirrors the reference's include/mappoint.h / src/mappoint.cc.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>

#include "image.h"

class frame;

class mappoint {
 public:
  // mappoint(cv::Mat& pos, frame* pFrame, int id): copies descriptor row `id` of the creating frame
  // (reference src/mappoint.cc:10-15)
  mappoint(const svo_host::Vec3f& pos, frame* pFrame, int id);
  svo_host::Vec3f GetWorldPos() const { return worldpos; }
  void AddObservation(frame* fm, size_t idx);   // src/mappoint.cc:17-23

 public:
  svo_host::Vec3f worldpos;        // CV_32F 3x1 in the reference
  uint8_t m_descriptor[32];
  bool bad;
  int observation_num;
  int create_id;
  long seq;                        // creation order: the deterministic LocalMapPoints order
  std::map<frame*, int> observations;
};

struct mappoint_by_creation {
  bool operator()(const mappoint* a, const mappoint* b) const { return a->seq < b->seq; }
};
