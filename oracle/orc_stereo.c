/* orc_stereo.c - CPU restatement of the stereo association + depth stage.
 *
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 *
 * The reference's live stereo path is dense (frame::MB -> MSA::solve,
 * src/frame.cc:82-91, then computekeypoint_r :122-138 and disp2Depth :140-164).
 * BASELINE.json's north_star replaces it with sparse epipolar block matching
 * ("Frame::ComputeStereoMatches"), which the reference does not contain
 * (SURVEY.md section 0 item 2): the behaviour below is the published ORB-SLAM2
 * algorithm that name refers to - row-band candidates, Hamming argmin with
 * octave and disparity gates, 11x11 SAD refinement over +-5 px at the keypoint's
 * pyramid level, parabola sub-pixel fit, median-based outlier cut - with the
 * window bounds checked correctly.  PARITY UNPINNED (no reference source).
 * orc_disp2depth / orc_unproject restate the reference's own code.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

#define TH_HIGH 100
#define TH_LOW 50
#define SAD_W 5
#define SAD_L 5

static int pair_cmp(const void* a, const void* b) {
  const int32_t* pa = (const int32_t*)a;
  const int32_t* pb = (const int32_t*)b;
  if (pa[0] != pb[0]) return pa[0] < pb[0] ? -1 : 1;
  return pa[1] < pb[1] ? -1 : pa[1] > pb[1] ? 1 : 0;
}

int orc_stereo_match(const uint8_t* pyrL, const uint8_t* pyrR, int W, int H, const orc_kp* kpL,
                     const uint8_t* dL, int nL, const orc_kp* kpR, const uint8_t* dR, int nR,
                     float bf, float fx, float* uR, float* depth) {
  int32_t w[8], h[8], quota[8];
  float scale[8];
  orc_geometry(W, H, 500, w, h, scale, quota);
  int64_t off[8];
  off[0] = 0;
  for (int l = 1; l < 8; ++l) off[l] = off[l - 1] + (int64_t)w[l - 1] * h[l - 1];

  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const float minD = 0.f, maxD = fx; /* mbf / mb with minZ = mb */
  /* right-keypoint row bands: r = 2 * scaleFactor[octave] */
  int32_t* minr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nR > 0 ? nR : 1));
  int32_t* maxr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nR > 0 ? nR : 1));
  for (int i = 0; i < nR; ++i) {
    float r = 2.0f * scale[kpR[i].octave];
    maxr[i] = (int)ceilf(kpR[i].y + r);
    minr[i] = (int)floorf(kpR[i].y - r);
  }
  int32_t* distidx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(nL > 0 ? nL : 1));
  int nd = 0;
  for (int iL = 0; iL < nL; ++iL) {
    uR[iL] = -1.f;
    depth[iL] = -1.f;
    const int levelL = kpL[iL].octave;
    const float vL = kpL[iL].y, uL = kpL[iL].x;
    const int row = (int)vL;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH, bestIdxR = -1;
    for (int iR = 0; iR < nR; ++iR) {
      if (row < minr[iR] || row > maxr[iR]) continue;
      if (kpR[iR].octave < levelL - 1 || kpR[iR].octave > levelL + 1) continue;
      const float u = kpR[iR].x;
      if (u >= minU && u <= maxU) {
        int dist = orc_descriptor_distance(dL + 32 * (size_t)iL, dR + 32 * (size_t)iR);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist >= thOrbDist || bestIdxR < 0) continue;
    /* sub-pixel refinement by SAD at the keypoint's level */
    const float uR0 = kpR[bestIdxR].x;
    const float inv = 1.0f / scale[levelL];
    const int su = (int)roundf(uL * inv), sv = (int)roundf(vL * inv), sr0 = (int)roundf(uR0 * inv);
    const int lw = w[levelL], lh = h[levelL];
    const uint8_t* IL = pyrL + off[levelL];
    const uint8_t* IR = pyrR + off[levelL];
    if (sv - SAD_W < 0 || sv + SAD_W >= lh || su - SAD_W < 0 || su + SAD_W >= lw) continue;
    if (sr0 - SAD_L - SAD_W < 0 || sr0 + SAD_L + SAD_W >= lw) continue;
    int best = 0x7fffffff, bestinc = 0, dists[2 * SAD_L + 1];
    const int cL = IL[(size_t)sv * lw + su];
    for (int inc = -SAD_L; inc <= SAD_L; ++inc) {
      const int cR = IR[(size_t)sv * lw + sr0 + inc];
      int sad = 0;
      for (int dy = -SAD_W; dy <= SAD_W; ++dy)
        for (int dx = -SAD_W; dx <= SAD_W; ++dx) {
          int a = IL[(size_t)(sv + dy) * lw + su + dx] - cL;
          int b = IR[(size_t)(sv + dy) * lw + sr0 + inc + dx] - cR;
          sad += abs(a - b);
        }
      if (sad < best) { best = sad; bestinc = inc; }
      dists[SAD_L + inc] = sad;
    }
    if (bestinc == -SAD_L || bestinc == SAD_L) continue;
    const float d1 = (float)dists[SAD_L + bestinc - 1], d2 = (float)dists[SAD_L + bestinc],
                d3 = (float)dists[SAD_L + bestinc + 1];
    const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
    if (deltaR < -1 || deltaR > 1) continue;
    float bestuR = scale[levelL] * ((float)sr0 + (float)bestinc + deltaR);
    float disparity = uL - bestuR;
    if (disparity >= minD && disparity < maxD) {
      if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
      depth[iL] = bf / disparity;
      uR[iL] = bestuR;
      distidx[2 * nd] = best;
      distidx[2 * nd + 1] = iL;
      ++nd;
    }
  }
  int nvalid = nd;
  if (nd > 0) {
    qsort(distidx, (size_t)nd, 2 * sizeof(int32_t), pair_cmp);
    const float median = (float)distidx[2 * (nd / 2)];
    const float thDist = 2.1f /* 1.5f*1.4f */ * median;
    for (int i = nd - 1; i >= 0; --i) {
      if ((float)distidx[2 * i] < thDist) break;
      uR[distidx[2 * i + 1]] = -1.f;
      depth[distidx[2 * i + 1]] = -1.f;
      --nvalid;
    }
  }
  free(minr); free(maxr); free(distidx);
  return nvalid;
}

int orc_stereo_frame(const uint8_t* grayL, int strideL, const uint8_t* grayR, int strideR, int W,
                     int H, int nfeatures, float bf, float fx, orc_kp* kpL, uint8_t* dL,
                     int32_t* nL, float* uR, float* depth, orc_kp* kpR, uint8_t* dR,
                     int32_t* nR) {
  size_t psz = (size_t)orc_pyramid_size(W, H);
  uint8_t* pL = (uint8_t*)malloc(psz);
  uint8_t* pR = (uint8_t*)malloc(psz);
  orc_kp* kr = kpR ? kpR : (orc_kp*)malloc(sizeof(orc_kp) * (size_t)nfeatures);
  uint8_t* dr = dR ? dR : (uint8_t*)malloc(32 * (size_t)nfeatures);
  int nl = orc_orb_extract(grayL, W, H, strideL, nfeatures, kpL, dL, pL);
  int nr = orc_orb_extract(grayR, W, H, strideR, nfeatures, kr, dr, pR);
  int nv = orc_stereo_match(pL, pR, W, H, kpL, dL, nl, kr, dr, nr, bf, fx, uR, depth);
  *nL = nl;
  if (nR) *nR = nr;
  if (!kpR) free(kr);
  if (!dR) free(dr);
  free(pL); free(pR);
  return nv;
}

/* frame::disp2Depth, reference src/frame.cc:140-164: depth starts at -1 and is
 * bf/disp wherever disp != 0 (note: the -1 "no data" fill of dispimg is nonzero,
 * so it maps to depth = -bf, exactly as the reference does). */
void orc_disp2depth(const float* disp, int count, float bf, float* depth) {
  for (int i = 0; i < count; ++i) {
    depth[i] = -1.f;
    if (!disp[i]) continue;
    depth[i] = bf / disp[i];
  }
}

/* frame::UnprojectStereo, reference src/frame.cc:166-180 (float32 throughout;
 * x = (u-cx)*z*(1/fx); x3D = Rwc*x3Dc + twc).  Rows with z <= 0 -> NaN (the
 * reference returns an empty cv::Mat). */
void orc_unproject(const float* uvz, int n, float fx, float fy, float cx, float cy,
                   const float Rwc[9], const float twc[3], float* xyz) {
  for (int i = 0; i < n; ++i) {
    const float u = uvz[3 * i], v = uvz[3 * i + 1], z = uvz[3 * i + 2];
    if (z > 0) {
      const float x = (u - cx) * z * (1 / fx);
      const float y = (v - cy) * z * (1 / fy);
      for (int r = 0; r < 3; ++r) {
        /* cv::gemm(Rwc, x3Dc, 1, twc, 1) on CV_32F accumulates in double and
         * rounds once [upstream-memory] */
        double acc = (double)Rwc[3 * r] * (double)x + (double)Rwc[3 * r + 1] * (double)y +
                     (double)Rwc[3 * r + 2] * (double)z;
        xyz[3 * i + r] = (float)(acc + (double)twc[r]);
      }
    } else {
      xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = NAN;
    }
  }
}
