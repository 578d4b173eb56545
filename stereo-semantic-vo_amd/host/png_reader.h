// png_reader.h - minimal PNG decoder (zlib only) for the KITTI image files the reference loads
// with cv::imread (main.cpp:160-162: image_2/NNNNNN.png, image_3/NNNNNN.png).  Supports 8-bit
// gray (colour type 0) and 8-bit RGB / RGBA (types 2, 6), non-interlaced; colour is reduced to
// gray with cv::cvtColor's fixed-point weights (R*4899 + G*9617 + B*1868 + 8192) >> 14, which is
// what cv::ORB does internally when handed the reference's 8UC3 images.
#pragma once
#include <zlib.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "image.h"

namespace svo_host {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

inline bool read_png(const std::string& path, GrayImage& img) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
  fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return false;
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = (const char*)&file[pos + 4];
    const uint8_t* d = &file[pos + 8];
    if (pos + 12 + len > file.size()) return false;
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return false;   // a short IHDR would be read past its end
      w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
    }
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + len;
  }
  if (!w || !h || w > 4095 || h > 4095 || depth != 8 || interlace || (ctype != 0 && ctype != 2 && ctype != 6)) return false;
  const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : 4;
  const size_t stride = (size_t)w * ch;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawlen = raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) return false;
  std::vector<uint8_t> cur(stride), prev(stride, 0);
  img.cols = (int)w; img.rows = (int)h;
  img.data.resize((size_t)w * h);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* row = &raw[(stride + 1) * y];
    const int ft = row[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
      int pr = 0;
      switch (ft) {
        case 0: pr = 0; break;
        case 1: pr = a; break;
        case 2: pr = b; break;
        case 3: pr = (a + b) >> 1; break;
        case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                  pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
        default: return false;
      }
      cur[i] = (uint8_t)(row[1 + i] + pr);
    }
    uint8_t* out = &img.data[(size_t)y * w];
    if (ch == 1) memcpy(out, cur.data(), w);
    else
      for (uint32_t x = 0; x < w; ++x) {
        const int R = cur[x * ch], G = cur[x * ch + 1], B = cur[x * ch + 2];
        out[x] = (uint8_t)((R * 4899 + G * 9617 + B * 1868 + 8192) >> 14);
      }
    prev.swap(cur);
  }
  return true;
}

inline bool read_image(const std::string& path, GrayImage& img) {
  if (path.size() > 4 && path.substr(path.size() - 4) == ".pgm") return read_pgm(path, img);
  return read_png(path, img);
}

}  // namespace svo_host
