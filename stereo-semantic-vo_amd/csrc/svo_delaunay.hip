// svo_delaunay.hip - host-side Delaunay triangulation of the ELAS support points.
//
// Replaces the call `triangulate("zQB", ...)` of Elas::computeDelaunayTriangulation
// (Thirdparty/libelas/src/elas.cpp:445-503), i.e. J. R. Shewchuk's Triangle run with its default
// algorithm: divide and conquer (Guibas & Stolfi 1985, "Primitives for the manipulation of general
// subdivisions and the computation of Voronoi diagrams") with Dwyer's alternating cuts.  The support
// points sit on a 5-pixel lattice, so co-circular quadruples are the rule, not the exception, and
// which diagonal is chosen changes the plane prior of the pixels underneath.  To reproduce the
// reference's choice this is the same published algorithm with the same decisions:
//   * vertices sorted by (x, y), duplicates dropped, halves split by alternating x / y medians,
//     subsets of <= 3 vertices sorted by x (triangle.cpp:5582-5608, 6160-6220);
//   * strict tests everywhere: lower tangent by ccw > 0, candidate valid iff ccw > 0 (evaluated once
//     per knitting step), edge deleted iff incircle > 0 and only while a real triangle is exposed,
//     right candidate taken iff left is finished or incircle(ul, ll, lr, ur) > 0 (triangle.cpp:5700-5940);
//   * horizontal cuts run the same merge after moving the four hull handles from the x-extreme to the
//     y-extreme vertices and back (triangle.cpp:5666-5700, 5795-5820).
// Own data structure (array quad-edge instead of Triangle's ghost-triangle mesh) and exact int64
// predicates (coordinates are pixel integers < 2^15, so the 4th-degree incircle determinant fits).
// Output is canonical: every triangle counter-clockwise in (u, v), smallest vertex index first,
// triangles sorted lexicographically - Triangle's own output order is an artefact of its memory pool.
// Of several input points with identical coordinates the lowest index is used (Triangle keeps
// whichever its randomised quicksort happens to place first).
#include <algorithm>
#include <cstdint>
#include <vector>
#include <atomic>
#include <chrono>

#include "svo_internal.h"

namespace {

struct Pt { int32_t x, y, id; };

struct QuadEdge {
  // one record per quad (an undirected edge with its dual): Onext of its 4 directed edges, the two end points of the primal
  // pair (directed edge 4q: org[0] -> org[1], 4q + 2 the other way), deleted or not - 32 bytes, what a step of the merge reads
  // of an edge sits in one cache line (three parallel arrays before)
  struct Quad { int32_t nxt[4]; int32_t org[2]; int32_t dead, pad; };
  std::vector<Quad> quad;
  const Pt* p = nullptr;

  static int rot(int e) { return (e & ~3) | ((e + 1) & 3); }
  static int sym(int e) { return e ^ 2; }
  static int irot(int e) { return (e & ~3) | ((e + 3) & 3); }
  int& N(int e) { return quad[e >> 2].nxt[e & 3]; }
  int N(int e) const { return quad[e >> 2].nxt[e & 3]; }
  int edges() const { return 4 * (int)quad.size(); }
  bool dead(int e) const { return quad[e >> 2].dead != 0; }
  int onext(int e) const { return N(e); }
  int oprev(int e) const { return rot(N(rot(e))); }
  int lnext(int e) const { return rot(N(irot(e))); }
  int lprev(int e) const { return sym(N(e)); }
  int rnext(int e) const { return irot(N(rot(e))); }
  int rprev(int e) const { return N(sym(e)); }
  // (primal directed edges only: e & 1 == 0)
  int o(int e) const { return quad[e >> 2].org[(e >> 1) & 1]; }
  int d(int e) const { return quad[e >> 2].org[((e >> 1) & 1) ^ 1]; }

  int make_edge(int a, int b) {
    const int q = edges();
    quad.push_back(Quad{{q, q + 3, q + 2, q + 1}, {a, b}, 0, 0});
    return q;
  }
  void splice(int a, int b) {
    const int alpha = rot(N(a)), beta = rot(N(b));
    std::swap(N(a), N(b));
    std::swap(N(alpha), N(beta));
  }
  int connect(int a, int b) {
    const int e = make_edge(d(a), o(b));
    splice(e, lnext(a));
    splice(sym(e), b);
    return e;
  }
  void remove(int e) {
    splice(e, oprev(e));
    splice(sym(e), oprev(sym(e)));
    quad[e >> 2].dead = 1;
  }

  // > 0 iff a, b, c make a left turn
  int64_t ccw(int a, int b, int c) const {
    return (int64_t)(p[b].x - p[a].x) * (p[c].y - p[a].y) - (int64_t)(p[b].y - p[a].y) * (p[c].x - p[a].x);
  }
  // > 0 iff d lies strictly inside the circle through a, b, c (a, b, c counter-clockwise)
  int64_t incircle(int a, int b, int c, int dd) const {
    const int64_t ax = p[a].x - p[dd].x, ay = p[a].y - p[dd].y;
    const int64_t bx = p[b].x - p[dd].x, by = p[b].y - p[dd].y;
    const int64_t cx = p[c].x - p[dd].x, cy = p[c].y - p[dd].y;
    const int64_t al = ax * ax + ay * ay, bl = bx * bx + by * by, cl = cx * cx + cy * cy;
    return al * (bx * cy - by * cx) + bl * (cx * ay - cy * ax) + cl * (ax * by - ay * bx);
  }
};

struct Hull { int lo, ro; };   // lo: ccw hull edge out of the "leftmost" vertex, ro: cw hull edge out of the "rightmost"

struct Builder {
  QuadEdge q;
  std::vector<Pt> pts;

  // The alternating-cut order of the vertices (what build() recursed over with std::nth_element on the points themselves: a third
  // of a triangulation's time went into that comparator): the subsets are a function of the point SET - all coordinates differ -
  // so they are cut on packed integer keys (x << 16 | y for a vertical cut, y << 16 | x for a horizontal one: one unsigned
  // compare instead of two signed ones and a branch), before any edge exists; pts is then put into that order once.
  struct Key { uint32_t kx, ky; int32_t src; };
  std::vector<Key> keys;
  std::vector<Pt> tmp;
  void cut(Key* s, int n, int axis) {
    if (n <= 3) {
      std::sort(s, s + n, [](const Key& a, const Key& b) { return a.kx < b.kx; });
      return;
    }
    const int nl = n >> 1;
    if (axis) std::nth_element(s, s + nl, s + n, [](const Key& a, const Key& b) { return a.ky < b.ky; });
    else std::nth_element(s, s + nl, s + n, [](const Key& a, const Key& b) { return a.kx < b.kx; });
    cut(s, nl, 1 - axis);
    cut(s + nl, n - nl, 1 - axis);
  }
  void order_points() {
    const int n = (int)pts.size();
    keys.resize(n);
    for (int i = 0; i < n; ++i) {
      const uint32_t X = (uint32_t)(pts[i].x + 32768), Y = (uint32_t)(pts[i].y + 32768);
      keys[i] = Key{X << 16 | Y, Y << 16 | X, i};
    }
    // (pts arrives sorted by x, then y: the first, vertical cut is the middle of the array as it stands)
    if (n <= 3) cut(keys.data(), n, 0);
    else { cut(keys.data(), n >> 1, 1); cut(keys.data() + (n >> 1), n - (n >> 1), 1); }
    tmp.resize(n);
    for (int i = 0; i < n; ++i) tmp[i] = pts[keys[i].src];
    pts.swap(tmp);
  }

  Hull build(int lo, int n, int axis) {   // (pts in order_points()'s order)
    if (n == 2) {
      const int a = q.make_edge(lo, lo + 1);
      return {a, QuadEdge::sym(a)};
    }
    if (n == 3) {
      const int a = q.make_edge(lo, lo + 1), b = q.make_edge(lo + 1, lo + 2);
      q.splice(QuadEdge::sym(a), b);
      const int64_t area = q.ccw(lo, lo + 1, lo + 2);
      if (area > 0) { q.connect(b, a); return {a, QuadEdge::sym(b)}; }
      if (area < 0) { const int c = q.connect(b, a); return {QuadEdge::sym(c), c}; }
      return {a, QuadEdge::sym(b)};
    }
    const int nl = n >> 1;
    const Hull L = build(lo, nl, 1 - axis);
    const Hull R = build(lo + nl, n - nl, 1 - axis);
    return merge(L, R, axis);
  }

  Hull merge(const Hull& L, const Hull& R, int axis) {
    int ldo = L.lo, ldi = L.ro, rdi = R.lo, rdo = R.ro;
    const Pt* p = pts.data();
    if (axis == 1) {   // horizontal cut: handles move to the bottom-/top-most vertices
      while (p[q.d(ldo)].y < p[q.o(ldo)].y) ldo = q.rprev(ldo);
      while (p[q.o(q.lprev(ldi))].y > p[q.o(ldi)].y) ldi = q.lprev(ldi);
      while (p[q.d(rdi)].y < p[q.o(rdi)].y) rdi = q.rprev(rdi);
      while (p[q.o(q.lprev(rdo))].y > p[q.o(rdo)].y) rdo = q.lprev(rdo);
    }
    // lower common tangent
    bool changed;
    do {
      changed = false;
      if (q.ccw(q.o(ldi), q.d(ldi), q.o(rdi)) > 0) { ldi = q.lnext(ldi); changed = true; }
      if (q.ccw(q.d(rdi), q.o(rdi), q.o(ldi)) > 0) { rdi = q.rprev(rdi); changed = true; }
    } while (changed);
    int basel = q.connect(QuadEdge::sym(rdi), ldi);   // lower-right -> lower-left
    if (q.o(ldi) == q.o(ldo)) ldo = QuadEdge::sym(basel);
    if (q.o(rdi) == q.o(rdo)) rdo = basel;
    for (;;) {
      const int ll = q.d(basel), lr = q.o(basel);
      int lcand = q.onext(QuadEdge::sym(basel));
      int rcand = q.oprev(basel);
      const bool left_finished = q.ccw(q.d(lcand), ll, lr) <= 0;
      const bool right_finished = q.ccw(q.d(rcand), ll, lr) <= 0;
      if (left_finished && right_finished) break;
      if (!left_finished) {
        for (;;) {   // delete left edges that fail the circle test while a real triangle is exposed
          const int ul = q.d(lcand), x = q.d(q.onext(lcand));
          if (!(q.d(q.lnext(lcand)) == x && q.ccw(ll, ul, x) > 0)) break;
          if (!(q.incircle(ll, lr, ul, x) > 0)) break;
          const int t = q.onext(lcand);
          q.remove(lcand);
          lcand = t;
        }
      }
      if (!right_finished) {
        for (;;) {
          const int ur = q.d(rcand), x = q.d(q.oprev(rcand));
          if (!(q.d(q.lnext(QuadEdge::sym(rcand))) == x && q.ccw(ur, lr, x) > 0)) break;
          if (!(q.incircle(ll, lr, ur, x) > 0)) break;
          const int t = q.oprev(rcand);
          q.remove(rcand);
          rcand = t;
        }
      }
      if (left_finished || (!right_finished && q.incircle(q.d(lcand), ll, lr, q.d(rcand)) > 0))
        basel = q.connect(rcand, QuadEdge::sym(basel));
      else
        basel = q.connect(QuadEdge::sym(basel), QuadEdge::sym(lcand));
    }
    if (axis == 1) {   // back to the left-/right-most vertices
      while (p[q.o(q.rnext(ldo))].x < p[q.o(ldo)].x) ldo = q.rnext(ldo);
      while (p[q.d(rdo)].x > p[q.o(rdo)].x) rdo = q.lnext(rdo);
    }
    return {ldo, rdo};
  }
};

}  // namespace

// xy: n points as (x, y) int32 pairs.  tri: up to cap (c1, c2, c3) index triples into xy.
// Returns the number of triangles (which may exceed cap: nothing is written beyond cap), < 0 on error.
std::atomic<long long> svo_delaunay_us[4];   // diagnostics (svo_internal.h): sort points, build, emit, sort triangles
std::atomic<long long> svo_delaunay_pts;
extern "C" int svo_elas_delaunay(const int32_t* xy, int32_t n, int32_t* tri, int32_t cap, int32_t* n_tri) {
  auto T0 = std::chrono::steady_clock::now();
  auto lap = [&](int k) { auto t = std::chrono::steady_clock::now(); svo_delaunay_us[k] += (long long)std::chrono::duration<double, std::micro>(t - T0).count(); T0 = t; };
  if (!xy || !n_tri || n < 0 || cap < 0 || (cap > 0 && !tri)) return SVO_E_INVALID;
  *n_tri = 0;
  // one builder per host thread, its arrays keep their capacity: fresh 100 KB+ vectors per call are mmap / munmap pairs,
  // and those serialise the worker threads of svo_elas_batch_dev on the process's address-space lock
  static thread_local Builder b;
  b.q.quad.clear();
  // sorted by (x, y, index) as ONE integer key per point; of several points with the same coordinates the lowest index stays
  static thread_local std::vector<uint64_t> skey;
  skey.resize(n);
  for (int i = 0; i < n; ++i) {
    if (xy[2 * i] < -32768 || xy[2 * i] > 32767 || xy[2 * i + 1] < -32768 || xy[2 * i + 1] > 32767)
      return SVO_E_INVALID;
    skey[i] = (uint64_t)(uint32_t)(xy[2 * i] + 32768) << 48 | (uint64_t)(uint32_t)(xy[2 * i + 1] + 32768) << 32 | (uint32_t)i;
  }
  std::sort(skey.begin(), skey.end());
  b.pts.clear();
  for (int i = 0; i < n; ++i) {
    if (i && (skey[i] >> 32) == (skey[i - 1] >> 32)) continue;
    const int id = (int)(uint32_t)skey[i];
    b.pts.push_back(Pt{xy[2 * id], xy[2 * id + 1], id});
  }
  const int m = (int)b.pts.size();
  if (m < 3) return SVO_OK;
  lap(0); svo_delaunay_pts += m;
  b.q.quad.reserve(4 * (size_t)m);
  b.order_points();
  b.q.p = b.pts.data();
  b.build(0, m, 0);
  lap(1);
  b.q.p = b.pts.data();
  // canonical order = lexicographic (a, b, c); with fewer than 2^21 points the triple packs into one 64-bit
  // key, and sorting plain integers is several times cheaper than sorting structs with a comparator
  const bool packed = n < (1 << 21);
  struct T { int32_t a, b, c; };
  std::vector<T> out;
  static thread_local std::vector<uint64_t> keys;
  keys.clear();
  if (packed) keys.reserve(2 * m); else out.reserve(2 * m);
  const QuadEdge& q = b.q;
  for (int e = 0; e < q.edges(); e += 2) {   // primal directed edges
    if (q.dead(e)) continue;
    const int A = q.o(e), B = q.d(e);                // (both in the edge's own record: the next edge of the face is only
    const int32_t ia = b.pts[A].id, ib = b.pts[B].id;  // fetched for the half of the edges that pass)
    if (ia >= ib) continue;                          // emitted from its smallest corner only
    const int e2 = q.lnext(e);
    const int e3 = q.lnext(e2);
    if (q.lnext(e3) != e) continue;
    const int C = q.o(e3);
    const int32_t ic = b.pts[C].id;
    if (ia >= ic || q.ccw(A, B, C) <= 0) continue;
    if (packed) keys.push_back(((uint64_t)ia << 42) | ((uint64_t)ib << 21) | (uint64_t)ic);
    else out.push_back({ia, ib, ic});
  }
  if (packed) {
    lap(2);
    {
      // keys ascending = (a, b, c) lexicographic.  A triangle is listed under its smallest corner a, a vertex has ~2 of
      // those: counting sort on a, then an insertion sort inside each (tiny) bucket - linear, against ~13 compare
      // levels of std::sort
      static thread_local std::vector<uint64_t> sorted;
      static thread_local std::vector<int32_t> start, cursor;
      sorted.resize(keys.size());
      start.assign((size_t)n + 1, 0);
      for (uint64_t k : keys) ++start[(size_t)(k >> 42) + 1];
      for (int a = 0; a < n; ++a) start[a + 1] += start[a];      // start[a] .. start[a + 1]: bucket a
      cursor.assign(start.begin(), start.end() - 1);
      for (uint64_t k : keys) sorted[cursor[(size_t)(k >> 42)]++] = k;
      for (int a = 0; a < n; ++a)
        for (int i = start[a] + 1; i < start[a + 1]; ++i) {
          const uint64_t k = sorted[i];
          int j = i - 1;
          while (j >= start[a] && sorted[j] > k) { sorted[j + 1] = sorted[j]; --j; }
          sorted[j + 1] = k;
        }
      keys.swap(sorted);
    }
    lap(3);
    *n_tri = (int32_t)keys.size();
    for (int i = 0; i < std::min<int>((int)keys.size(), cap); ++i) {
      tri[3 * i] = (int32_t)(keys[i] >> 42); tri[3 * i + 1] = (int32_t)((keys[i] >> 21) & 0x1fffff);
      tri[3 * i + 2] = (int32_t)(keys[i] & 0x1fffff);
    }
    return SVO_OK;
  }
  std::sort(out.begin(), out.end(), [](const T& x, const T& y) {
    if (x.a != y.a) return x.a < y.a;
    if (x.b != y.b) return x.b < y.b;
    return x.c < y.c;
  });
  *n_tri = (int32_t)out.size();
  for (int i = 0; i < std::min<int>((int)out.size(), cap); ++i) {
    tri[3 * i] = out[i].a; tri[3 * i + 1] = out[i].b; tri[3 * i + 2] = out[i].c;
  }
  return SVO_OK;
}
