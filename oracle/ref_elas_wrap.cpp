// ref_elas_wrap.cpp - thin C wrapper (this repo's own code) around the REAL libelas of the
// reference (Thirdparty/libelas/src/*.cpp, compiled where it lies by oracle/Makefile.ref).
// TEST INFRASTRUCTURE ONLY: it is the compiled-reference oracle for SURVEY.md section 8 row f-2
// (dense ELAS stereo on the GPU - a "next" row, not started).  libelas is vendored by the
// reference but never called by its own code (include/frame.h:15 includes the header only).
#include <stdint.h>

#include "elas.h"

extern "C" int ref_elas_process(const uint8_t* left, const uint8_t* right, int width, int height,
                                int bytes_per_line, int robotics, float* D1, float* D2) {
  Elas::parameters param(robotics ? Elas::ROBOTICS : Elas::MIDDLEBURY);
  param.postprocess_only_left = false;
  Elas elas(param);
  const int32_t dims[3] = {width, height, bytes_per_line};
  elas.process(const_cast<uint8_t*>(left), const_cast<uint8_t*>(right), D1, D2, dims);
  return 0;
}
