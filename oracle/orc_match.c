/* orc_match.c - CPU restatement of the Hamming matching stage.
 *
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).  These functions follow the
 * reference's own code (not OpenCV), so they are exact restatements:
 *   orc_descriptor_distance  <- pnpmatch::DescriptorDistance, src/pnpmatch.cc:14-30
 *   orc_hamming_argmin       <- the inner j-loop of poseEstimationPnP, :75-94
 *   orc_match_greedy         <- pass 1 (:61-156) and pass 2 (:159-199) control flow
 *   orc_bf_match             <- find_feature_matches, :253-300 (cv BFMatcher::match
 *                               = nearest train row, first minimum [upstream-memory])
 */
#include <stddef.h>
#include "svo_oracle.h"

/* src/pnpmatch.cc:14-30 - 8 x int32 SWAR popcount, verbatim arithmetic. */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t wa = (uint32_t)a[4 * i] | ((uint32_t)a[4 * i + 1] << 8) |
                  ((uint32_t)a[4 * i + 2] << 16) | ((uint32_t)a[4 * i + 3] << 24);
    uint32_t wb = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) |
                  ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    uint32_t v = wa ^ wb;
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}

/* One row of src/pnpmatch.cc:75-94: scan j in index order, skip masked columns,
 * `if (dist < best) { second = best; best = dist; idx = j; }`. */
static void scan_row(const uint8_t* q, const uint8_t* t, int N, const uint8_t* mask, int32_t* idx,
                     int32_t* best, int32_t* second) {
  int bestDist = 256, secondBest = 256, bestIdx = -1;
  for (int j = 0; j < N; ++j) {
    if (mask && mask[j]) continue;
    int dist = orc_descriptor_distance(q, t + 32 * (size_t)j);
    if (dist < bestDist) {
      secondBest = bestDist;
      bestDist = dist;
      bestIdx = j;
    }
  }
  *idx = bestIdx; *best = bestDist; *second = secondBest;
}

void orc_hamming_argmin(const uint8_t* q, int M, const uint8_t* t, int N, const uint8_t* t_mask,
                        int32_t* best_idx, int32_t* best, int32_t* second) {
  for (int i = 0; i < M; ++i)
    scan_row(q + 32 * (size_t)i, t, N, t_mask, &best_idx[i], &best[i], &second[i]);
}

/* src/pnpmatch.cc:61-156 (max_dist 15, ratio 0) and :159-199 (max_dist 30,
 * ratio 2): greedy in row order; an accepted row claims its column. */
void orc_match_greedy(const uint8_t* q, const uint8_t* q_skip, int M, const uint8_t* t, int N,
                      uint8_t* assigned, int max_dist, float ratio, int32_t* best_idx,
                      int32_t* best, int32_t* second, uint8_t* accepted) {
  for (int i = 0; i < M; ++i) {
    best_idx[i] = -1; best[i] = 256; second[i] = 256; accepted[i] = 0;
    if (q_skip && q_skip[i]) continue;
    scan_row(q + 32 * (size_t)i, t, N, assigned, &best_idx[i], &best[i], &second[i]);
    int ok = best[i] < max_dist;
    if (ok && ratio > 0) ok = (float)second[i] / (float)best[i] > ratio;
    if (ok && best_idx[i] >= 0) {
      accepted[i] = 1;
      assigned[best_idx[i]] = 1;
    }
  }
}

/* src/pnpmatch.cc:277-299: nearest train row per query; min over all matches;
 * keep those with distance <= max(2*min_dist, 30). */
void orc_bf_match(const uint8_t* q, int M, const uint8_t* t, int N, int32_t* train_idx,
                  int32_t* dist, uint8_t* keep) {
  double min_dist = 10000;
  for (int i = 0; i < M; ++i) {
    int bi = -1, bd = 0x7fffffff;
    for (int j = 0; j < N; ++j) {
      int d = orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
      if (d < bd) { bd = d; bi = j; }
    }
    train_idx[i] = bi;
    dist[i] = bi >= 0 ? bd : -1;
    if (bi >= 0 && bd < min_dist) min_dist = bd;
  }
  double thr = 2 * min_dist > 30.0 ? 2 * min_dist : 30.0;
  for (int i = 0; i < M; ++i) keep[i] = (train_idx[i] >= 0 && dist[i] <= thr) ? 1 : 0;
}
