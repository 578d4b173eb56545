// svo_gate.h - semantic gating helpers shared by the matching kernels.
// Reference: the padded-box tests of src/Tracking.cc:61-66, src/frame.cc:198-203 (pad 5) and
// src/pnpmatch.cc:103-121 (pad 10), and the point-to-epipolar-line distance of :110-114, which the
// reference evaluates with F applied to the LAST frame's point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SVO_MAX_BOXES 64

__host__ __device__ inline bool svo_in_boxes(float x, float y, const int32_t* boxes, int n_boxes, int pad) {
  for (int k = 0; k < n_boxes; ++k) {
    const int left = boxes[4 * k], right = boxes[4 * k + 1], top = boxes[4 * k + 2], bottom = boxes[4 * k + 3];
    if (x > left - pad && x < right + pad && y > top - pad && y < bottom + pad) return true;
  }
  return false;
}

__host__ __device__ inline double svo_epipolar_distance(const double* F, float last_x, float last_y,
                                                        float cur_x, float cur_y) {
  const double A = F[0] * last_x + F[1] * last_y + F[2];
  const double B = F[3] * last_x + F[4] * last_y + F[5];
  const double C = F[6] * last_x + F[7] * last_y + F[8];
  return fabs(A * cur_x + B * cur_y + C) / sqrt(A * A + B * B);
}
