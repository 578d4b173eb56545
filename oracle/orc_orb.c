/* orc_orb.c - CPU restatement of ORB extraction (reference a-2:
 * frame::featuredetect, src/frame.cc:75-79 -> cv::ORB::create() defaults,
 * detectAndCompute on one gray image).
 *
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).  PARITY UNPINNED: OpenCV 3.2 is
 * not in /root/reference and not installed; the pipeline below follows the
 * OpenCV 3.2 ORB stages as recalled in SURVEY.md section 8 a-notes, with these
 * documented choices that make every stage exactly reproducible on a GPU:
 *   - rBRIEF pattern: include/svo_brief_pattern.h (seeded), not bit_pattern_31_;
 *   - Harris response is ordered by the exact int64 25(ab-c^2)-(a+b)^2;
 *   - retainBest by FAST score keeps ties (threshold semantics); retainBest by
 *     Harris takes exactly `quota` by (response desc, raster asc);
 *   - output order: octave asc, then Harris rank;
 *   - sin/cos of the orientation come from a fixed double polynomial.
 * Build with -ffp-contract=off: float/double expressions below must round once
 * per operation, exactly like the HIP kernels.
 */
#include "svo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/svo_brief_pattern.h"

static const int8_t kPattern[SVO_BRIEF_NTESTS][4] = SVO_BRIEF_PATTERN_INIT;

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

/* ---- geometry --------------------------------------------------------------- */

/* cv::ORB getScale + per-level sizes + feature quotas (SURVEY 8 a-notes). */
int orc_geometry(int W, int H, int nfeatures, int32_t w[8], int32_t h[8], float scale[8],
                 int32_t quota[8]) {
  for (int l = 0; l < ORC_NLEVELS; ++l) {
    scale[l] = (float)pow(1.2000000476837158 /* (double)1.2f */, (double)l);
    float inv = 1.0f / scale[l];
    w[l] = cv_round_f((float)W * inv);
    h[l] = cv_round_f((float)H * inv);
  }
  float factor = (float)(1.0 / 1.2000000476837158);
  float ndesired =
      (float)nfeatures * (1.0f - factor) / (1.0f - (float)pow((double)factor, (double)ORC_NLEVELS));
  int sum = 0;
  for (int l = 0; l < ORC_NLEVELS - 1; ++l) {
    quota[l] = cv_round_f(ndesired);
    sum += quota[l];
    ndesired *= factor;
  }
  quota[ORC_NLEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  return 0;
}

/* umax table of the radius-15 disc (OpenCV / ORB-SLAM2 construction). */
void orc_umax(int32_t umax[16]) {
  const int hp = ORC_HALF_PATCH;
  int vmax = (int)floor(hp * sqrt(2.0) / 2 + 1);
  int vmin = (int)ceil(hp * sqrt(2.0) / 2);
  for (int v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)hp * hp - (double)v * v));
  for (int v = hp, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

int64_t orc_level_offset(int W, int H, int level) {
  int32_t w[8], h[8], q[8];
  float s[8];
  orc_geometry(W, H, 500, w, h, s, q);
  int64_t off = 0;
  for (int l = 0; l < level; ++l) off += (int64_t)w[l] * h[l];
  return off;
}
int64_t orc_pyramid_size(int W, int H) { return orc_level_offset(W, H, ORC_NLEVELS); }

/* ---- pyramid ---------------------------------------------------------------- */

/* cv::resize INTER_LINEAR for CV_8UC1 (fixed point, 11-bit coefficients). */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst,
                          int dw, int dh, int dstride) {
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
  int* yofs = (int*)malloc(sizeof(int) * dh);
  short* ibeta = (short*)malloc(sizeof(short) * 2 * dh);
  double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = (short)cv_round_f((1.f - fx) * 2048.f);
    ialpha[2 * dx + 1] = (short)cv_round_f(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
    yofs[dy] = sy;
    ibeta[2 * dy] = (short)cv_round_f((1.f - fy) * 2048.f);
    ibeta[2 * dy + 1] = (short)cv_round_f(fy * 2048.f);
  }
  for (int dy = 0; dy < dh; ++dy) {
    int sy0 = yofs[dy], sy1 = sy0 + 1 < sh ? sy0 + 1 : sh - 1;
    const uint8_t* r0 = src + (size_t)sy0 * sstride;
    const uint8_t* r1 = src + (size_t)sy1 * sstride;
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    for (int dx = 0; dx < dw; ++dx) {
      int sx0 = xofs[dx], sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
      int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
      int S0 = r0[sx0] * a0 + r0[sx1] * a1;
      int S1 = r1[sx0] * a0 + r1[sx1] * a1;
      int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      dst[(size_t)dy * dstride + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
  free(xofs); free(ialpha); free(yofs); free(ibeta);
}

/* 8 levels, level l resized from level l-1, packed tight one after another. */
void orc_build_pyramid(const uint8_t* gray, int W, int H, int stride, uint8_t* pyr) {
  int32_t w[8], h[8], q[8];
  float s[8];
  orc_geometry(W, H, 500, w, h, s, q);
  for (int y = 0; y < H; ++y) memcpy(pyr + (size_t)y * W, gray + (size_t)y * stride, W);
  int64_t off = 0;
  for (int l = 1; l < ORC_NLEVELS; ++l) {
    int64_t noff = off + (int64_t)w[l - 1] * h[l - 1];
    orc_resize_linear_u8(pyr + off, w[l - 1], h[l - 1], w[l - 1], pyr + noff, w[l], h[l], w[l]);
    off = noff;
  }
}

/* ---- FAST-9/16 -------------------------------------------------------------- */

static const int kRingX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kRingY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cv cornerScore<16> closed form: max over the 16 arcs of 9 contiguous ring
 * pixels of min(center - ring) (dark arc) and of min(ring - center) (bright
 * arc), minus 1.  The pixel is a FAST-9 corner at threshold t iff that maximum
 * exceeds t.  Returns 0 for "not a corner at ORC_FAST_THR". */
int orc_fast_score_at(const uint8_t* img, int w, int h, int stride, int x, int y) {
  if (x < 3 || y < 3 || x >= w - 3 || y >= h - 3) return 0;
  int v = img[(size_t)y * stride + x];
  int d[16];
  for (int i = 0; i < 16; ++i) d[i] = v - img[(size_t)(y + kRingY[i]) * stride + x + kRingX[i]];
  int best = -256;
  for (int s = 0; s < 16; ++s) {
    int mn = 255, mx = -255;
    for (int i = 0; i < 9; ++i) {
      int di = d[(s + i) & 15];
      if (di < mn) mn = di;
      if (di > mx) mx = di;
    }
    if (mn > best) best = mn;      /* all ring pixels darker than centre by >= mn */
    if (-mx > best) best = -mx;    /* all ring pixels brighter by >= -mx          */
  }
  return best > ORC_FAST_THR ? best - 1 : 0;
}

/* FAST + strict 3x3 non-max suppression + runByImageBorder(border).  Output:
 * {x, y, score} triples in raster order.  Returns the count (<= cap). */
int orc_fast_corners(const uint8_t* img, int w, int h, int stride, int thr, int border,
                     int32_t* xys, int cap) {
  (void)thr;
  int n = 0;
  if (w <= 2 * border || h <= 2 * border) return 0;
  int* sc = (int*)calloc((size_t)w * h, sizeof(int));
  for (int y = border - 1; y <= h - border; ++y)
    for (int x = border - 1; x <= w - border; ++x)
      sc[(size_t)y * w + x] = orc_fast_score_at(img, w, h, stride, x, y);
  for (int y = border; y < h - border; ++y)
    for (int x = border; x < w - border; ++x) {
      int s = sc[(size_t)y * w + x];
      if (!s) continue;
      int ok = 1;
      for (int dy = -1; dy <= 1 && ok; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          if (!dx && !dy) continue;
          if (sc[(size_t)(y + dy) * w + x + dx] >= s) { ok = 0; break; }
        }
      if (ok && n < cap) {
        xys[3 * n] = x; xys[3 * n + 1] = y; xys[3 * n + 2] = s;
        ++n;
      }
    }
  free(sc);
  return n;
}

/* ---- Harris, orientation ---------------------------------------------------- */

/* cv HarrisResponses, blockSize 7: a = sum Ix^2, b = sum Iy^2, c = sum IxIy over the
 * 7x7 block with the 3x3 Sobel-like integer gradients; returns the exact
 * 25(ab - c^2) - (a+b)^2  ( = 25 x the k=0.04 Harris measure, unscaled). */
int64_t orc_harris_at(const uint8_t* img, int stride, int x, int y) {
  int64_t a = 0, b = 0, c = 0;
  for (int dy = -3; dy <= 3; ++dy)
    for (int dx = -3; dx <= 3; ++dx) {
      const uint8_t* p = img + (size_t)(y + dy) * stride + x + dx;
      int Ix = (p[1] - p[-1]) * 2 + (p[-stride + 1] - p[-stride - 1]) + (p[stride + 1] - p[stride - 1]);
      int Iy = (p[stride] - p[-stride]) * 2 + (p[stride - 1] - p[-stride - 1]) + (p[stride + 1] - p[-stride + 1]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
  return 25 * (a * b - c * c) - (a + b) * (a + b);
}

/* response as cv would scale it: (ab - c^2 - 0.04 (a+b)^2) * (1/(4*7*255))^4. */
float orc_harris_to_float(int64_t R) {
  const double scale = 1.0 / (4.0 * 7.0 * 255.0);
  const double k = scale * scale * scale * scale / 25.0;
  return (float)((double)R * k);
}

/* cv::fastAtan2 (degrees, 7th-order odd polynomial). */
float orc_fast_atan2(float y, float x) {
  const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
  float ax = fabsf(x), ay = fabsf(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + 2.220446e-16f);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + 2.220446e-16f);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* cv ICAngles: intensity centroid over the radius-15 disc. */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y) {
  int32_t umax[16];
  orc_umax(umax);
  const uint8_t* c = img + (size_t)y * stride + x;
  int m01 = 0, m10 = 0;
  for (int u = -ORC_HALF_PATCH; u <= ORC_HALF_PATCH; ++u) m10 += u * c[u];
  for (int v = 1; v <= ORC_HALF_PATCH; ++v) {
    int vsum = 0, d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int vp = c[u + v * stride], vm = c[u - v * stride];
      vsum += vp - vm;
      m10 += u * (vp + vm);
    }
    m01 += v * vsum;
  }
  return orc_fast_atan2((float)m01, (float)m10);
}

/* Deterministic sin/cos (double polynomial, one rounding per op) of a float
 * angle in radians, |angle| <= 2*pi + eps; results rounded to float. */
void orc_sincos(float angle_rad, float* s, float* c) {
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double PIO2_HI = 1.57079632679489655800e+00;
  const double PIO2_LO = 6.12323399573676603587e-17;
  double x = (double)angle_rad;
  double kq = floor(x * TWO_OVER_PI + 0.5);
  double r = (x - kq * PIO2_HI) - kq * PIO2_LO;
  double r2 = r * r;
  double ps = -1.0 / 1307674368000.0;             /* r^15 */
  ps = ps * r2 + 1.0 / 6227020800.0;              /* r^13 */
  ps = ps * r2 - 1.0 / 39916800.0;                /* r^11 */
  ps = ps * r2 + 1.0 / 362880.0;                  /* r^9  */
  ps = ps * r2 - 1.0 / 5040.0;                    /* r^7  */
  ps = ps * r2 + 1.0 / 120.0;                     /* r^5  */
  ps = ps * r2 - 1.0 / 6.0;                       /* r^3  */
  double sr = r + r * (r2 * ps);
  double pc = 1.0 / 20922789888000.0;             /* r^16 */
  pc = pc * r2 - 1.0 / 87178291200.0;             /* r^14 */
  pc = pc * r2 + 1.0 / 479001600.0;               /* r^12 */
  pc = pc * r2 - 1.0 / 3628800.0;                 /* r^10 */
  pc = pc * r2 + 1.0 / 40320.0;                   /* r^8  */
  pc = pc * r2 - 1.0 / 720.0;                     /* r^6  */
  pc = pc * r2 + 1.0 / 24.0;                      /* r^4  */
  pc = pc * r2 - 0.5;                             /* r^2  */
  double cr = 1.0 + r2 * pc;
  int q = (int)((long long)kq & 3);
  double sv, cvv;
  switch (q) {
    case 0: sv = sr; cvv = cr; break;
    case 1: sv = cr; cvv = -sr; break;
    case 2: sv = -sr; cvv = -cr; break;
    default: sv = -cr; cvv = sr; break;
  }
  *s = (float)sv;
  *c = (float)cvv;
}

/* ---- blur + descriptor ------------------------------------------------------ */

static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on CV_8U: 8-bit fixed-point
 * separable kernel round(g*256) = {18,34,49,55,49,34,18}; the column pass
 * rounds with (sum + 2^15) >> 16 and saturates. */
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  static const int k[7] = {18, 34, 49, 55, 49, 34, 18};
  int* tmp = (int*)malloc(sizeof(int) * (size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = -3; i <= 3; ++i) s += k[i + 3] * src[(size_t)y * sstride + reflect101(x + i, w)];
      tmp[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = -3; i <= 3; ++i) s += k[i + 3] * tmp[(size_t)reflect101(y + i, h) * w + x];
      int v = (s + 32768) >> 16;
      dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
    }
  free(tmp);
}

/* cv computeOrbDescriptors (WTA_K 2): rotate the pattern by the keypoint angle,
 * round to the pixel grid, compare blurred intensities, 8 tests per byte. */
void orc_describe(const uint8_t* blurred, int stride, int x, int y, float angle_deg,
                  uint8_t desc[32]) {
  float angle = angle_deg * 0.017453292f; /* (float)(CV_PI/180) */
  float a, b;
  orc_sincos(angle, &b, &a); /* a = cos, b = sin */
  const uint8_t* center = blurred + (size_t)y * stride + x;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int k = 0; k < 8; ++k) {
      const int8_t* t = kPattern[8 * i + k];
      float x0 = (float)t[0] * a - (float)t[1] * b, y0 = (float)t[0] * b + (float)t[1] * a;
      float x1 = (float)t[2] * a - (float)t[3] * b, y1 = (float)t[2] * b + (float)t[3] * a;
      int v0 = center[cv_round_f(y0) * stride + cv_round_f(x0)];
      int v1 = center[cv_round_f(y1) * stride + cv_round_f(x1)];
      val |= (v0 < v1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

/* ---- full extraction --------------------------------------------------------- */

typedef struct { int64_t R; int32_t key, x, y; } cand_t;
static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa;
  const cand_t* b = (const cand_t*)pb;
  if (a->R != b->R) return a->R > b->R ? -1 : 1;
  return a->key < b->key ? -1 : a->key > b->key ? 1 : 0;
}

/* cv::ORB computeKeyPoints + computeDescriptors over a prebuilt pyramid. */
int orc_orb_from_pyramid(const uint8_t* pyr, int W, int H, int nfeatures, orc_kp* kp,
                         uint8_t* desc) {
  int32_t w[8], h[8], quota[8];
  float scale[8];
  orc_geometry(W, H, nfeatures, w, h, scale, quota);
  int n_out = 0;
  int64_t off = 0;
  for (int l = 0; l < ORC_NLEVELS; ++l) {
    const uint8_t* img = pyr + off;
    const int lw = w[l], lh = h[l];
    off += (int64_t)lw * lh;
    int cap = (lw / 2 + 1) * (lh / 2 + 1);
    int32_t* xys = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)cap);
    int n = orc_fast_corners(img, lw, lh, lw, ORC_FAST_THR, ORC_EDGE, xys, cap);
    /* retainBest(2*quota) by FAST score, ties kept; then the CAP1 rule */
    int hist[256];
    memset(hist, 0, sizeof hist);
    for (int i = 0; i < n; ++i) hist[xys[3 * i + 2]]++;
    int T = 0, target = 2 * quota[l];
    if (n > target) {
      int cum = 0;
      for (int s = 255; s >= 0; --s) {
        cum += hist[s];
        if (cum >= target) { T = s; break; }
      }
    }
    for (;;) {
      int cnt = 0;
      for (int s = T; s < 256; ++s) cnt += hist[s];
      if (cnt <= ORC_CAP1) break;
      ++T;
    }
    cand_t* cands = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    int nc = 0;
    for (int i = 0; i < n; ++i) {
      if (xys[3 * i + 2] < T) continue;
      int x = xys[3 * i], y = xys[3 * i + 1];
      cands[nc].x = x; cands[nc].y = y; cands[nc].key = y * lw + x;
      cands[nc].R = orc_harris_at(img, lw, x, y);
      ++nc;
    }
    qsort(cands, (size_t)nc, sizeof(cand_t), cand_cmp);
    int take = nc < quota[l] ? nc : quota[l];
    uint8_t* blurred = NULL;
    if (take > 0) {
      blurred = (uint8_t*)malloc((size_t)lw * lh);
      orc_gaussian_blur7(img, lw, lh, lw, blurred, lw);
    }
    for (int i = 0; i < take; ++i) {
      orc_kp* k = &kp[n_out];
      int x = cands[i].x, y = cands[i].y;
      k->angle = orc_ic_angle(img, lw, x, y);
      k->x = (float)x * scale[l];
      k->y = (float)y * scale[l];
      k->size = 31.0f * scale[l];
      k->response = orc_harris_to_float(cands[i].R);
      k->octave = l;
      k->class_id = -1;
      orc_describe(blurred, lw, x, y, k->angle, desc + 32 * (size_t)n_out);
      ++n_out;
    }
    free(blurred); free(cands); free(xys);
  }
  return n_out;
}

int orc_orb_extract(const uint8_t* gray, int W, int H, int stride, int nfeatures, orc_kp* kp,
                    uint8_t* desc, uint8_t* pyr_out) {
  uint8_t* pyr = pyr_out ? pyr_out : (uint8_t*)malloc((size_t)orc_pyramid_size(W, H));
  orc_build_pyramid(gray, W, H, stride, pyr);
  int n = orc_orb_from_pyramid(pyr, W, H, nfeatures, kp, desc);
  if (!pyr_out) free(pyr);
  return n;
}
