// elas_main.cc - the libelas demo program (reference Thirdparty/libelas/src/main.cpp:31-134) on the
// MI355X path: same command line (`elas demo` | `elas left.pgm right.pgm`), same outputs
// (<name>_disp.pgm next to each input, both maps scaled so that the largest disparity is 255), with
// Elas::process replaced by svo_elas_process.  tests/test_elas_tool.py compares its output files byte for
// byte with those of the reference's own program.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/svo.h"
#include "image.h"

using svo_host::GrayImage;

static bool save_pgm(const std::string& path, int w, int h, const std::vector<uint8_t>& px) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fprintf(f, "P5\n%d %d\n255\n", w, h);
  const bool ok = fwrite(px.data(), 1, px.size(), f) == px.size();
  fclose(f);
  return ok;
}

static svo_ctx* g_ctx = nullptr;

static void process(const char* file_1, const char* file_2) {
  std::cout << "Processing: " << file_1 << ", " << file_2 << std::endl;
  GrayImage I1, I2;
  if (!svo_host::read_pgm(file_1, I1) || !svo_host::read_pgm(file_2, I2)) {
    std::cout << "ERROR: Could not read file " << file_1 << " / " << file_2 << std::endl;
    return;
  }
  if (I1.cols <= 0 || I1.rows <= 0 || I1.cols != I2.cols || I1.rows != I2.rows) {
    std::cout << "ERROR: Images must be of same size, but" << std::endl;
    std::cout << "       I1: " << I1.cols << " x " << I1.rows << ", I2: " << I2.cols << " x " << I2.rows << std::endl;
    return;
  }
  const int width = I1.cols, height = I1.rows;
  const int32_t dims[3] = {width, height, width};
  const size_t n = (size_t)width * height;
  std::vector<float> D1(n), D2(n);
  if (!g_ctx && svo_create(&g_ctx, 0, std::max(width, 64), std::max(height, 64), 500, 1) != SVO_OK) {
    std::cout << "ERROR: no MI355X context (svo_create failed)" << std::endl;
    return;
  }
  svo_elas_params param;
  svo_elas_default_params(0, &param);
  param.postprocess_only_left = 0;   // main.cpp:62
  if (svo_elas_process(g_ctx, I1.ptr(), I2.ptr(), D1.data(), D2.data(), dims, &param) != SVO_OK) {
    std::cout << "ERROR: " << svo_last_error(g_ctx) << std::endl;
    return;
  }
  float disp_max = 0;   // main.cpp:67-71
  for (size_t i = 0; i < n; ++i) {
    if (D1[i] > disp_max) disp_max = D1[i];
    if (D2[i] > disp_max) disp_max = D2[i];
  }
  std::vector<uint8_t> o1(n), o2(n);
  for (size_t i = 0; i < n; ++i) {   // main.cpp:76-79
    o1[i] = (uint8_t)std::max(255.0 * D1[i] / disp_max, 0.0);
    o2[i] = (uint8_t)std::max(255.0 * D2[i] / disp_max, 0.0);
  }
  const std::string a(file_1), b(file_2);
  save_pgm(a.substr(0, a.size() - 4) + "_disp.pgm", width, height, o1);
  save_pgm(b.substr(0, b.size() - 4) + "_disp.pgm", width, height, o2);
}

int main(int argc, char** argv) {
  if (argc == 2 && !strcmp(argv[1], "demo")) {
    for (const char* name : {"cones", "aloe", "raindeer", "urban1", "urban2", "urban3", "urban4"}) {
      const std::string l = std::string("img/") + name + "_left.pgm", r = std::string("img/") + name + "_right.pgm";
      process(l.c_str(), r.c_str());
    }
    std::cout << "... done!" << std::endl;
  } else if (argc == 3) {
    process(argv[1], argv[2]);
    std::cout << "... done!" << std::endl;
  } else {
    std::cout << std::endl << "ELAS demo program usage: " << std::endl
              << "./elas demo ................ process all test images (image dir)" << std::endl
              << "./elas left.pgm right.pgm .. process a single stereo pair" << std::endl
              << "./elas -h .................. shows this help" << std::endl << std::endl
              << "Note: All images must be pgm greylevel images. All output" << std::endl
              << "      disparities will be scaled such that disp_max = 255." << std::endl << std::endl;
  }
  if (g_ctx) svo_destroy(g_ctx);
  return 0;
}
