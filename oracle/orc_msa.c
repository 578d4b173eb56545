/* orc_msa.c - CPU restatement of the FIRST stage of the reference's MSA dense stereo, `MSA::init`
 * (Thirdparty/MB/MSA.cpp:22-63) with the helpers it calls: `gradient` (:65-76), `getCost` (:78-108),
 * `ctmf` r = 1 on the colour images (:58-59, Thirdparty/MB/ctmf.c) and `gradient_after_ctmf` (:110-139).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED except for the median filter,
 * which tests/test_msa.py pins on the reference's own compiled ctmf.c: MSA.cpp itself needs OpenCV and
 * cannot be built here.  All arithmetic is double, as in MSA.h:59-66; costs are stored as float. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "svo_oracle.h"

static uint8_t gray_of(const uint8_t* bgr) { /* :44-45 */
  return (uint8_t)(int)(0.299 * bgr[2] + 0.587 * bgr[1] + 0.114 * bgr[0] + 0.5);
}

/* :65-76 - central differences + 127.5, one-sided at the two borders */
static void msa_gradient(const uint8_t* img, int n, int m, double* gra) {
  for (int i = 0; i < n; ++i) {
    double plus = img[i * m + 1], minus = img[i * m];
    gra[i * m] = plus - minus + 127.5;
    for (int j = 1; j < m - 1; ++j) {
      plus = img[i * m + j + 1];
      gra[i * m + j] = (plus - minus) * 0.5 + 127.5;
      minus = img[i * m + j];
    }
    gra[i * m + m - 1] = (plus - minus) + 127.5;
  }
}

/* ctmf with r = 1: per channel the median of the 3x3 window clamped to the image (ctmf.c:222-320) */
static void median3x3(const uint8_t* src, uint8_t* dst, int n, int m, int cn) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j)
      for (int c = 0; c < cn; ++c) {
        uint8_t v[9];
        int k = 0;
        for (int di = -1; di <= 1; ++di)
          for (int dj = -1; dj <= 1; ++dj) {
            int ii = i + di, jj = j + dj;
            ii = ii < 0 ? 0 : (ii > n - 1 ? n - 1 : ii);
            jj = jj < 0 ? 0 : (jj > m - 1 ? m - 1 : jj);
            v[k++] = src[(ii * m + jj) * cn + c];
          }
        for (int a = 1; a < 9; ++a) { /* insertion sort */
          const uint8_t x = v[a];
          int b = a - 1;
          while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
          v[b + 1] = x;
        }
        dst[(i * m + j) * cn + c] = v[4];
      }
}

/* bgrL/bgrR: n rows x m columns x 3, tight.  disp = Disp = d + 1 (49 in the reference, src/frame.cc:86).
 * costL/costR: n*m*disp floats; m3L/m3R: median-filtered colour images; the four gradient maps: n*m doubles. */
int orc_msa_init(const uint8_t* bgrL, const uint8_t* bgrR, int n, int m, int disp, float* costL, float* costR,
                 uint8_t* m3L, uint8_t* m3R, double* r_graL, double* c_graL, double* r_graR, double* c_graR) {
  if (n < 2 || m < 2 || disp < 1) return -1;
  const double max_dif_gra = 2.0, max_dif_col = 7.0, weight_col = 0.11; /* :29-31 */
  const size_t N = (size_t)n * m;
  uint8_t* imgL = (uint8_t*)malloc(N); uint8_t* imgR = (uint8_t*)malloc(N);
  double* graL = (double*)malloc(N * sizeof(double)); double* graR = (double*)malloc(N * sizeof(double));
  for (size_t t = 0; t < N; ++t) { imgL[t] = gray_of(bgrL + 3 * t); imgR[t] = gray_of(bgrR + 3 * t); }
  msa_gradient(imgL, n, m, graL);
  msa_gradient(imgR, n, m, graR);
  /* getCost :78-108 */
  for (int i = 0; i < n; ++i) {
    const int occ = i * m;
    for (int j = 0; j < m; ++j)
      for (int d = 0; d < disp; ++d) {
        const int o = (j - d >= 0) ? i * m + j - d : occ;
        double dif_gra = fabs(graL[i * m + j] - graR[o]);
        if (dif_gra > max_dif_gra) dif_gra = max_dif_gra;
        double dif_col = 0.0;
        for (int k = 0; k < 3; ++k) dif_col += abs((int)bgrL[(i * m + j) * 3 + k] - (int)bgrR[o * 3 + k]);
        dif_col = dif_col / 3;
        if (dif_col > max_dif_col) dif_col = max_dif_col;
        costL[(size_t)(i * m + j) * disp + d] = (float)(weight_col * dif_col + (1 - weight_col) * dif_gra);
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j)
      for (int d = 0; d < disp; ++d) {
        const size_t a = (size_t)(i * m + j) * disp + d;
        if (j + d < m) costR[a] = costL[(size_t)(i * m + j + d) * disp + d];
        else costR[a] = costR[a - 1];
      }
  median3x3(bgrL, m3L, n, m, 3);
  median3x3(bgrR, m3R, n, m, 3);
  /* gradient_after_ctmf :110-139 */
  for (int s = 0; s < 2; ++s) {
    const uint8_t* img3 = s ? m3R : m3L;
    double* r_gra = s ? r_graR : r_graL;
    double* c_gra = s ? c_graR : c_graL;
    uint8_t* img = imgL; /* reuse */
    for (size_t t = 0; t < N; ++t) img[t] = gray_of(img3 + 3 * t);
    for (int i = 0; i < n; ++i) {
      double plus = img[i * m + 1], minus = img[i * m];
      r_gra[i * m] = plus - minus;
      for (int j = 1; j < m - 1; ++j) {
        plus = img[i * m + j + 1];
        r_gra[i * m + j] = (plus - minus) * 0.5;
        minus = img[i * m + j];
      }
      r_gra[i * m + m - 1] = (plus - minus);
    }
    for (int j = 0; j < m; ++j) {
      double plus = img[m + j], minus = img[j];
      c_gra[j] = plus - minus;
      for (int i = 1; i < n - 1; ++i) {
        plus = img[(i + 1) * m + j];
        c_gra[i * m + j] = (plus - minus) * 0.5;
        minus = img[i * m + j];
      }
      c_gra[(n - 1) * m + j] = (plus - minus);
    }
  }
  free(imgL); free(imgR); free(graL); free(graR);
  return 0;
}

/* ---- the stages of MSA::solve that consume the tree (MSA.cpp:929-1105) ------------------------------------
 * The tree arrives as the reference holds it when TreeDp runs: BFS order `seq` (:898-926), parent `fa`, and for
 * every node its CHILDREN in the order the adjacency chain `lk/branch` yields them (the chain minus the parent),
 * here as CSR (child_ptr, child, child_c = the edge's colour weight 0..255).  The float accumulation order over
 * the children is that order, so it is part of the input. */

/* MSA::setExp (:1126-1130) */
void orc_msa_exp_table(double o, double Exp[256]) {
  for (int i = 0; i <= 255; ++i) Exp[i] = exp(-i * 1.0 / o / 255);
}

/* MSA::TreeDp (:929-990): cost -> costA (both N*D floats); costUp is scratch of the same size */
int orc_msa_tree_dp(const float* cost, int N, int D, const int32_t* seq, const int32_t* child_ptr,
                    const int32_t* child, const uint8_t* child_c, int root, const double Exp[256], float* costUp,
                    float* costA) {
  for (size_t i = 0; i < (size_t)N * D; ++i) costUp[i] = cost[i];
  for (int k = N - 1; k >= 0; --k) {
    const int u = seq[k];
    for (int e = child_ptr[u]; e < child_ptr[u + 1]; ++e) {
      const int v = child[e];
      const double w = Exp[child_c[e]];
      for (int d = 0; d < D; ++d) costUp[(size_t)u * D + d] += w * costUp[(size_t)v * D + d];
    }
  }
  for (int d = 0; d < D; ++d) costA[(size_t)root * D + d] = costUp[(size_t)root * D + d];
  for (int k = 0; k < N; ++k) {
    const int u = seq[k];
    for (int e = child_ptr[u]; e < child_ptr[u + 1]; ++e) {
      const int v = child[e];
      const double w = Exp[child_c[e]];
      for (int d = 0; d < D; ++d)
        costA[(size_t)v * D + d] = w * costA[(size_t)u * D + d] + (1 - w * w) * costUp[(size_t)v * D + d];
    }
  }
  return 0;
}

static void median_r2(const uint8_t* src, uint8_t* dst, int n, int m) { /* ctmf r = 2, 1 channel, clamped window */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      uint8_t v[25];
      int k = 0;
      for (int di = -2; di <= 2; ++di)
        for (int dj = -2; dj <= 2; ++dj) {
          int ii = i + di, jj = j + dj;
          ii = ii < 0 ? 0 : (ii > n - 1 ? n - 1 : ii);
          jj = jj < 0 ? 0 : (jj > m - 1 ? m - 1 : jj);
          v[k++] = src[ii * m + jj];
        }
      for (int a = 1; a < 25; ++a) {
        const uint8_t x = v[a];
        int b = a - 1;
        while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
        v[b + 1] = x;
      }
      dst[i * m + j] = v[12];
    }
}

/* MSA::WTA (:992-1006): first minimum over the disparities, then the 5x5 median */
int orc_msa_wta(const float* costA, int n, int m, int D, uint8_t* disparity) {
  uint8_t* tmp = (uint8_t*)malloc((size_t)n * m);
  for (int i = 0; i < n * m; ++i) {
    int k = 0;
    for (int j = 1; j < D; ++j)
      if (costA[(size_t)i * D + j] < costA[(size_t)i * D + k]) k = j;
    tmp[i] = (uint8_t)k;
  }
  median_r2(tmp, disparity, n, m);
  free(tmp);
  return 0;
}

/* MSA::LRcheck (:1027-1105): stable pixels (d > 0 and the right map agrees exactly at j - d) get the cost
 * |d - d1|, every other pixel a zero cost row; mask = the stable pixels */
int orc_msa_lrcheck(const uint8_t* d1, const uint8_t* d2, int n, int m, int D, float* cost, uint8_t* mask) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const int t = i * m + j, d = d1[t];
      mask[t] = (j - d >= 0 && d > 0 && abs(d - (int)d2[t - d]) == 0) ? 1 : 0;
    }
  for (int t = 0; t < n * m; ++t)
    for (int d = 0; d < D; ++d) cost[(size_t)t * D + d] = mask[t] ? (float)abs(d - (int)d1[t]) : 0.0f;
  return 0;
}
