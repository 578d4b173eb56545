// image.h - minimal image / matrix views standing in for cv::Mat on the host side.
// OpenCV is on neither the build box nor the GPU box, so the host
// classes keep the reference's member NAMES but hold these PODs instead of cv::Mat.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace svo_host {

struct GrayImage {   // 8-bit gray, row-major, tight rows (what main.cpp's imread + cvtColor yields)
  int cols = 0, rows = 0;
  std::vector<uint8_t> data;
  bool empty() const { return data.empty(); }
  const uint8_t* ptr() const { return data.data(); }
};

// binary PGM (P5) reader/writer - the only image codec the harness needs offline
inline bool read_pgm(const std::string& path, GrayImage& img) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0};
  int w = 0, h = 0, mx = 0;
  if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P5") != 0) { fclose(f); return false; }
  int c = fgetc(f);
  while (c == ' ' || c == '\n' || c == '\r' || c == '\t' || c == '#') {
    if (c == '#') while (c != '\n' && c != EOF) c = fgetc(f);
    c = fgetc(f);
  }
  ungetc(c, f);
  // sizes are bounded like svo_create's (a header is untrusted input: no negative or huge allocation)
  if (fscanf(f, "%d %d %d", &w, &h, &mx) != 3 || mx > 255 || mx < 1 || w <= 0 || h <= 0 || w > 4095 || h > 4095) { fclose(f); return false; }
  fgetc(f);
  img.cols = w; img.rows = h;
  img.data.resize((size_t)w * h);
  const bool ok = fread(img.data.data(), 1, img.data.size(), f) == img.data.size();
  fclose(f);
  return ok;
}

// 4x4 / 3x3 / 3x1 float matrices (CV_32F in the reference), row-major
struct Mat44f { float m[16]; float& at(int r, int c) { return m[4 * r + c]; } float at(int r, int c) const { return m[4 * r + c]; } };
struct Mat33f { float m[9]; float& at(int r, int c) { return m[3 * r + c]; } float at(int r, int c) const { return m[3 * r + c]; } };
struct Vec3f { float v[3]; float& at(int i) { return v[i]; } float at(int i) const { return v[i]; } };

inline Mat44f eye4() { Mat44f T{}; for (int i = 0; i < 4; ++i) T.m[5 * i] = 1.f; return T; }

}  // namespace svo_host
