// svo_elas_filter.cc - host stage 1 of the dense ELAS path: the order-dependent clean-up of the lattice of support
// point candidates (reference Thirdparty/libelas/src/elas.cpp:445-537: removeInconsistentSupportPoints, then
// removeRedundantSupportPoints along the columns and along the rows).
//
// The three passes visit the lattice column by column (u outer, v inner) and delete in place, so later decisions see
// earlier deletions; that order is kept.  What changes is the work per candidate: the window / the ten neighbours of
// a candidate are examined 8 lattice cells per instruction (SSE2, 16-bit lanes) on padded copies of the lattice - a
// transposed one for the two passes whose neighbours run along a column, a row-major one for the pass along a row -
// instead of one cell per loop iteration.  Padding cells are -1 (invalid), which is what the reference's border
// clipping amounts to.  ~0.5 ms -> ~0.1 ms per stereo pair; with 16 host cores per GPU this stage and the two
// triangulations bound svo_elas_batch_dev.
//
// Host-only code (x86-64, SSE2 is part of the baseline), built without any device pass.
#include <emmintrin.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace {

constexpr int PAD = 16;   // cells of -1 around the lattice: 16-lane loads starting 5 cells before a candidate stay inside

// lanes i = 0 .. 15 <-> offsets -5 + i from the candidate.  bit i of the result: cell valid and |cell - d| <= thr
inline unsigned match16(const int16_t* p, int d, int thr) {
  const __m128i vd = _mm_set1_epi16((short)d), vt = _mm_set1_epi16((short)(thr + 1)), neg1 = _mm_set1_epi16(-1);
  unsigned bits = 0;
  for (int h = 0; h < 2; ++h) {
    const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 8 * h));
    const __m128i diff = _mm_sub_epi16(x, vd);
    const __m128i ad = _mm_max_epi16(diff, _mm_sub_epi16(_mm_setzero_si128(), diff));
    const __m128i ok = _mm_and_si128(_mm_cmpgt_epi16(x, neg1), _mm_cmpgt_epi16(vt, ad));
    const unsigned m = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(ok, _mm_setzero_si128())) & 0xffu;   // one bit per lane
    bits |= m << (8 * h);
  }
  return bits;
}

}  // namespace

// D: Hc x Wc row-major lattice of disparities (-1 = none), cleaned in place.  Returns 0, or -1 when the parameters are
// outside what the vector path covers (window / distance > 5, disparities that do not fit 16-bit differences): the
// caller then runs its scalar loops.
extern "C" int svo_elas_filter_lattice(int16_t* D, int Wc, int Hc, int incon_window, int incon_threshold, int incon_min_support,
                                       int red_max_dist, int red_threshold) {
  if (incon_window < 0 || incon_window > 5 || red_max_dist < 0 || red_max_dist > 5 || incon_threshold < 0 ||
      incon_threshold > 4096 || red_threshold < 0 || red_threshold > 4096 || Wc < 1 || Hc < 1)
    return -1;
  static thread_local std::vector<int16_t> tbuf, rbuf;
  const int ST = Hc + 2 * PAD, SR = Wc + 2 * PAD;
  tbuf.assign((size_t)(Wc + 2 * PAD) * ST, (int16_t)-1);
  int16_t* T = tbuf.data() + (size_t)PAD * ST + PAD;   // T[u * ST + v]
  for (int v = 0; v < Hc; ++v)
    for (int u = 0; u < Wc; ++u) T[u * ST + v] = D[v * Wc + u];
  // ---- removeInconsistentSupportPoints: fewer than min_support cells of the (2w + 1)^2 window within the threshold ----
  {
    const int w = incon_window;
    const unsigned lanes = ((1u << (2 * w + 1)) - 1u) << (5 - w);   // offsets -w .. +w
    for (int u = 0; u < Wc; ++u)
      for (int v = 0; v < Hc; ++v) {
        const int d = T[u * ST + v];
        if (d < 0) continue;
        // (only "at least min_support" matters, not which cells are counted first: the candidate's own column, then outwards)
        int support = __builtin_popcount(match16(&T[u * ST + v - 5], d, incon_threshold) & lanes);
        for (int k = 1; k <= w && support < incon_min_support; ++k) {
          support += __builtin_popcount(match16(&T[(u - k) * ST + v - 5], d, incon_threshold) & lanes);
          if (support < incon_min_support) support += __builtin_popcount(match16(&T[(u + k) * ST + v - 5], d, incon_threshold) & lanes);
        }
        if (support < incon_min_support) T[u * ST + v] = -1;
      }
  }
  // ---- removeRedundantSupportPoints along the column: a supporter within max_dist above AND one below ----------------
  const unsigned before = ((1u << red_max_dist) - 1u) << (5 - red_max_dist), after = ((1u << red_max_dist) - 1u) << 6;
  for (int u = 0; u < Wc; ++u)
    for (int v = 0; v < Hc; ++v) {
      const int d = T[u * ST + v];
      if (d < 0) continue;
      const unsigned m = match16(&T[u * ST + v - 5], d, red_threshold);
      if ((m & before) && (m & after)) T[u * ST + v] = -1;
    }
  // ---- ... and along the row (same visiting order, on a row-major copy) --------------------------------------------
  rbuf.assign((size_t)(Hc + 2 * PAD) * SR, (int16_t)-1);
  int16_t* R = rbuf.data() + (size_t)PAD * SR + PAD;   // R[v * SR + u]
  for (int u = 0; u < Wc; ++u)
    for (int v = 0; v < Hc; ++v) R[v * SR + u] = T[u * ST + v];
  for (int u = 0; u < Wc; ++u)
    for (int v = 0; v < Hc; ++v) {
      const int d = R[v * SR + u];
      if (d < 0) continue;
      const unsigned m = match16(&R[v * SR + u - 5], d, red_threshold);
      if ((m & before) && (m & after)) R[v * SR + u] = -1;
    }
  for (int v = 0; v < Hc; ++v) memcpy(D + (size_t)v * Wc, R + (size_t)v * SR, sizeof(int16_t) * (size_t)Wc);
  return 0;
}
