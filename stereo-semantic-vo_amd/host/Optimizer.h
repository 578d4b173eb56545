// Optimizer.h - mirrors the reference's include/Optimizer.h.
#pragma once
#include "frame.h"

class Optimizer {
 public:
  // src/Optimizer.cc:15-86: pose-only LM over the frame's map-point observations; returns the
  // number of edges (nInitialCorrespondences).
  static int PoseOptimization(frame* pFrame);
};
