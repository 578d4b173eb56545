// svo_msa_graph.hip - host side of MSA dense stereo: the spanning tree the cost
// aggregation runs over.  Replaces, for one image, the reference's call sequence
//     build -> Tarjan -> getSeq0 -> Kruskal1 -> getSeq        (Thirdparty/MB/MSA.cpp:1141-1147 / 1153-1158)
// i.e. (1) a directed 4-neighbour graph over the median-filtered image, edges pointing from the flatter pixel to the
// steeper one (both ways when their gradient magnitudes differ by less than 1), weighted by the largest channel
// difference, plus a super-root connected to every pixel at cost 1e9 (:152-192); (2) its minimum spanning
// arborescence by the Chu-Liu/Edmonds contraction algorithm in Tarjan's formulation with mergeable (leftist) heaps of
// incoming edges (:200-346, :1207-1290) - pixels hanging off the super-root are the roots of a forest of flat regions;
// (3) region merging: small regions are joined over the cheapest cross-region edges under the reference's size rules,
// then whatever is left is connected with weight 255 (:661-808); (4) a breadth-first order from the first root
// (:898-926).  These stages are sequential and their results depend on tie order (heap merges, the unstable
// std::sort), so they stay on the host and follow the reference decision for decision; the two images' trees are
// independent and are built on two threads by the caller.  Output is what svo_msa_tree_dp consumes.
#include <math.h>

#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "svo_internal.h"

namespace {

std::atomic<int> g_active_builders{0};

// SVO_MSA_STAGE_STATS=1: host time per stage summed over all builders (under whatever load they run), printed at exit
struct StageStats {
  std::atomic<long long> ns[8];
  std::atomic<long long> trees{0};
  const bool on = getenv("SVO_MSA_STAGE_STATS") != nullptr;
  StageStats() { for (auto& x : ns) x = 0; }
  ~StageStats() {
    if (!on || trees == 0) return;
    static const char* name[8] = {"heaps per pixel", "contraction", "expansion", "links + roots", "label_regions", "merge_regions", "bfs records", ""};
    long long tot = 0;
    for (int k = 0; k < 7; ++k) tot += ns[k];
    fprintf(stderr, "[msa stages] %lld trees, %.2f ms each:", (long long)trees, tot * 1e-6 / trees);
    for (int k = 0; k < 7; ++k) fprintf(stderr, "  %s %.2f", name[k], ns[k] * 1e-6 / trees);
    fprintf(stderr, "\n");
  }
};
StageStats g_stats;
struct StageClock {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(int k) {
    if (!g_stats.on) return;
    const auto now = std::chrono::steady_clock::now();
    g_stats.ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(now - t).count();
    t = now;
  }
};

// The directed graph is implicit: pixel p owns five arc slots e = 5p + k for the arcs INTO p - k = 0 from the
// super-root (weight 1e9), 1 from its left neighbour, 2 right, 3 up, 4 down (present or not) - which is also the
// order in which the reference inserts them into p's heap (its horizontal loop visits the pair (p-1, p) before
// (p, p+1), its vertical loop (p-m, p) before (p, p+m); MSA.cpp:152-192).  Heap node e + 1 carries arc e.
//
// Leftist heaps over the arcs, all in one node pool; node 0 is "empty".  Rank of an empty child counts as 0, like a
// leaf's (the reference's convention, :1232-1234) - it decides which child ends up left, hence later merges.
class EdgeHeaps {
 public:
  // rank: the node's own; lrank: its LEFT child's, kept here so that the way back up a merge path never has to touch the left
  // children (they are never on a merge path - only right spines are walked - so the copy stays true; one cache miss per step less)
  struct Node { int left, right, key; uint16_t rank, lrank; };
  void reset(size_t arcs) { node_.resize(arcs + 1); node_[0] = Node{0, 0, 0, 0, 0}; }   // nodes are init()-ed before use
  void init(int h, int key) { node_[h] = Node{0, 0, key, 0, 0}; }
  int key(int h) const { return node_[h].key; }
  static int arc(int h) { return h - 1; }
  // `spine` is scratch of the caller (the per-pixel construction runs on several threads)
  int meld(int a, int b, std::vector<int>&) {
    if (!a || !b) return a + b;
    // walk down the right spines, always continuing below the smaller key (ties: the first heap stays on top).  The merge path is
    // at most the two right spines: <= 2 log2(nodes + 1) < 64 + 64 entries for any pool that fits an int
    int spine[128];
    size_t ns = 0;
    for (;;) {
      if (!a || !b) { a += b; break; }
      {   // (as selects: which of the two keys is smaller is a coin toss to the branch predictor)
        const bool sw = node_[a].key > node_[b].key;
        const int na = sw ? b : a, nb = sw ? a : b;
        a = na; b = nb;
      }
      if (ns == 128) { broken = true; a += b; break; }   // (cannot happen while the leftist invariant holds; never write past the array)
      spine[ns++] = a;
      a = node_[a].right;
    }
    int sub = a;
    unsigned sub_rank = node_[sub].rank;   // (sub != 0 here: one of the two heaps outlasts the other)
    for (size_t i = ns; i-- > 0;) {
      Node& t = node_[spine[i]];
      // right child := sub; the child of the larger rank goes left (an empty child counts as rank 0, like a leaf)
      if (t.lrank < sub_rank) { t.right = t.left; t.left = sub; const unsigned r = t.lrank; t.lrank = (uint16_t)sub_rank; sub_rank = r; }
      else t.right = sub;
      t.rank = (uint16_t)(t.right ? sub_rank + 1 : 0);
      sub = spine[i];
      sub_rank = t.rank;
    }
    return sub;
  }
  bool broken = false;   // a merge path longer than any leftist heap can have: the structure is corrupt, the caller gives up
  // A pixel's own heap - the super-root's arc, then whichever of the four neighbour arcs exist, inserted in slot order - has one of
  // a few thousand shapes: the merges compare keys with `>` only, so the shape is a function of which arcs exist (4 bits) and of
  // how their weights rank among each other (rank = how many of the others are smaller: 2 bits each; the super-root's 1e9 is
  // above them all).  All shapes are built once with the merges themselves and kept as 63-bit templates - per node (slot + 1):
  // left, right, rank, left's rank in 3 bits each, then the root - and a pixel's heap is written from its template instead of
  // being merged together (12 % of a tree's host time were those four merges per pixel).
  static const uint64_t* pixel_templates() {
    static std::vector<uint64_t> tab;
    static std::once_flag once;
    std::call_once(once, [] {
      tab.assign(16 * 256, 0);
      EdgeHeaps tmp;
      tmp.reset(5);
      std::vector<int> spine;
      for (int mask = 0; mask < 16; ++mask)
        for (int code = 0; code < 256; ++code) {
          tmp.init(1, 1000);
          int h = 1;
          for (int slot = 1; slot <= 4; ++slot)
            if (mask >> (slot - 1) & 1) { tmp.init(slot + 1, (code >> (2 * (slot - 1))) & 3); h = tmp.meld(h, slot + 1, spine); }
          uint64_t t = (uint64_t)h << 60;
          for (int k = 0; k < 5; ++k) {
            const Node& nd = tmp.node_[k + 1];
            t |= (uint64_t)((unsigned)nd.left | (unsigned)nd.right << 3 | (unsigned)nd.rank << 6 | (unsigned)nd.lrank << 9) << (12 * k);
          }
          tab[mask << 8 | code] = t;
        }
    });
    return tab.data();
  }
  // writes the nodes base + 1 (+ slot) of one pixel from template t (keys: SR_W for slot 0, w[slot - 1] for the others); returns the root
  int place_pixel(int base, uint64_t t, int mask, int sr_w, const int* w) {
    auto put = [&](int k, int key) {
      const unsigned f = (unsigned)(t >> (12 * k)) & 0xfffu, l = f & 7u, r = (f >> 3) & 7u;
      node_[base + 1 + k] = Node{l ? base + (int)l : 0, r ? base + (int)r : 0, key, (uint16_t)((f >> 6) & 7u), (uint16_t)((f >> 9) & 7u)};
    };
    put(0, sr_w);
    for (int k = 1; k <= 4; ++k)
      if (mask >> (k - 1) & 1) put(k, w[k - 1]);
    return base + (int)(t >> 60);
  }
  int meld(int a, int b) { return meld(a, b, spine_); }
  int drop_min(int h) { return meld(node_[h].left, node_[h].right); }
  void subtract(int h, int delta) {   // from every key of heap h
    if (!h) return;
    walk_.assign(1, h);
    for (size_t i = 0; i < walk_.size(); ++i) {
      Node& t = node_[walk_[i]];
      t.key -= delta;
      if (t.left) walk_.push_back(t.left);
      if (t.right) walk_.push_back(t.right);
    }
  }

 private:
  std::vector<Node> node_;
  std::vector<int> spine_, walk_;
};

struct Sets {   // union-find with full path compression; unite(a, b) makes a's root the root
  std::vector<int> up;
  void reset(int n) { up.resize(n + 1); for (int i = 0; i <= n; ++i) up[i] = i; }
  int find(int x) {
    int r = x;
    while (up[r] != r) r = up[r];
    while (x != r) { const int nx = up[x]; up[x] = r; x = nx; }
    return r;
  }
  void unite(int a, int b) { a = find(a); b = find(b); up[b] = a; }
};

struct Link { int32_t to, w, next; };   // adjacency chains, newest first, as TreeDp walks them

class TreeBuilder {
 public:
  // one builder serves many images: every array below keeps its capacity between runs (fresh 100 MB allocations cost
  // ~10 % of a run in page faults)
  void reshape(int rows, int cols) { n_ = rows; m_ = cols; N_ = rows * cols; }

  // returns the root pixel, or < 0
  int run(const uint8_t* img3, const double* gx, const double* gy, int32_t* seq, int32_t* child_ptr, int32_t* child,
          uint8_t* child_w) {
    img3_ = img3;
    static const bool dbg = getenv("SVO_MSA_TREE_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
      if (!dbg) return;
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[msa tree] %-24s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    dbg_mark_ = dbg ? &t_last : nullptr;
    arborescence(gx, gy);
    mark("arborescence (rest)");
    if (roots_.empty()) return -1;
    if (heaps_.broken || !label_regions()) return -2;        // not a forest over the roots
    mark("label_regions");
    merge_regions();
    mark("merge_regions");
    const int rt = finish_csr(seq, child_ptr, child, child_w);
    mark("bfs order + child lists");
    return rt;
  }
  // the same tree as level-order records, written while it is walked breadth-first (see HostTree::build in svo_msa.hip)
  int run_rec(const uint8_t* img3, const double* gx, const double* gy, MsaBfsRec* rec, std::vector<int32_t>& level_ptr, int* maxw) {
    img3_ = img3;
    dbg_mark_ = nullptr;
    StageClock clk;
    stage_clock_ = &clk;
    arborescence(gx, gy);
    stage_clock_ = nullptr;
    clk.lap(3);
    if (roots_.empty()) return -1;
    if (heaps_.broken || !label_regions()) return -2;        // not a forest over the roots
    clk.lap(4);
    merge_regions();
    clk.lap(5);
    struct Done { StageClock& c; ~Done() { c.lap(6); if (g_stats.on) ++g_stats.trees; } } done{clk};
    // the parent of the node at level-order position t, kept BY POSITION (written and read in sequence: the walk's random
    // accesses are the adjacency chains alone)
    std::vector<int32_t>& parent = parent_;
    parent.resize(N_);
    parent[0] = -1;
    level_ptr.clear(); level_ptr.push_back(0);
    int n_seq = 1, level_end = 1, w = 0;
    bool fits = true;
    rec[0] = MsaBfsRec{1, 0, -1, roots_[0]};
    for (int t = 0; t < n_seq; ++t) {
      if (t == level_end) {                       // first node of the next level: everything enqueued so far belongs to it
        w = std::max(w, t - level_ptr.back());
        level_ptr.push_back(t);
        level_end = n_seq;
      }
      const int u = rec[t].node, first = n_seq, pu = parent[t];
      bool overrun = false;
      neighbours(u, [&](int v, int wv) {
        if (v == pu || overrun) return;
        if (n_seq == N_) { overrun = true; return; }   // not a tree (a cycle: more arrivals than pixels)
        parent[n_seq] = u;
        rec[n_seq++] = MsaBfsRec{0, wv, t, v};
      });
      if (overrun) return -2;
      const int nch = n_seq - first;
      if (nch > 255) fits = false;
      rec[t].cpos = first;
      rec[t].meta |= nch << 8;
    }
    if (n_seq != N_) return -2;
    w = std::max(w, N_ - level_ptr.back());
    level_ptr.push_back(N_);
    *maxw = w;
    return fits ? roots_[0] : -3;
  }

 private:
  int finish_csr(int32_t* seq, int32_t* child_ptr, int32_t* child, uint8_t* child_w) {
    // breadth-first order from the first root; parent[] doubles as the visited mark
    std::vector<int32_t>& parent = parent_;
    parent.assign(N_, -1);
    int n_seq = 0;
    seq[n_seq++] = roots_[0];
    for (int t = 0; t < n_seq; ++t) {
      const int u = seq[t];
      bool overrun = false;
      neighbours(u, [&](int v, int) {
        if (v == parent[u] || overrun) return;
        if (n_seq == N_) { overrun = true; return; }   // not a tree
        seq[n_seq++] = v;
        parent[v] = u;
      });
      if (overrun) return -2;
    }
    if (n_seq != N_) return -2;
    int n_child = 0;
    child_ptr[0] = 0;
    for (int u = 0; u < N_; ++u) {
      neighbours(u, [&](int v, int wv) {
        if (v != parent[u]) { child[n_child] = v; child_w[n_child] = (uint8_t)wv; ++n_child; }
      });
      child_ptr[u + 1] = n_child;
    }
    return roots_[0];
  }
  int n_ = 0, m_ = 0, N_ = 0;
  EdgeHeaps heaps_;
  Sets merged_;
  struct Vtx { int strong, weak, incoming, from, key, edge; };
  struct Cnode { int up, grp, entry; };   // `up` links a (super-)node to the cycle node that absorbed it, `grp` remembers it for the expansion, `entry`: the arc chosen into it
  std::vector<Vtx> vtx_;
  std::vector<Cnode> con_;
  std::vector<int> chosen_from_, chosen_key_, todo_, size_, sum_;
  std::vector<uint8_t> arc_w_, done_;
  std::vector<int32_t> parent_;
  struct Cand { int a, b, w; double key; };
  std::vector<Cand> cand_;
  const uint8_t* img3_ = nullptr;
  std::chrono::steady_clock::time_point* dbg_mark_ = nullptr;
  StageClock* stage_clock_ = nullptr;
  void sub_mark(const char* what) {
    if (stage_clock_) stage_clock_->lap(what[0] == 'h' ? 0 : what[0] == 'c' ? 1 : 2);
    if (!dbg_mark_) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[msa tree]   %-22s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - *dbg_mark_).count());
    *dbg_mark_ = now;
  }
  // The tree's adjacency.  TreeDp walks a pixel's edges newest first (adjacency chains filled by push-front): the edges of the
  // region merging, made last, come first (head_ / link_: chains, a few thousand edges), then the arborescence's - made in pixel
  // order, one per non-root pixel p between p and its parent, so that a pixel's chain reads: its children below and right of it
  // (p + m, p + 1), its own parent, its children left of and above it (p - 1, p - m).  Every arborescence edge joins
  // 4-neighbours, so they are kept as ONE byte per pixel (adj_: bits 0-3 child at +m / +1 / -1 / -m, bits 4-6 where the parent is:
  // 1 +m, 2 +1, 3 -1, 4 -m, bit 7: the pixel has merge edges) and the weight of a pixel's parent edge (wpar_) - 0.9 MB that stay in
  // the caches instead of 11 MB of chain links the breadth-first walk met at random (17 % of a tree's host time).
  std::vector<int32_t> head_;
  std::vector<Link> link_;
  std::vector<uint8_t> adj_, wpar_;
  int n_arb_ = 0;   // arborescence edges
  size_t edge_links() const { return 2 * (size_t)n_arb_ + link_.size(); }   // what link_.size() was when every edge was a pair of links
  template <class F>
  void neighbours(int u, F&& f) const {
    const unsigned a = adj_[u];
    if (a & 0x80u) for (int i = head_[u]; i >= 0; i = link_[i].next) f(link_[i].to, link_[i].w);
    if (a & 1u) f(u + m_, (int)wpar_[u + m_]);
    if (a & 2u) f(u + 1, (int)wpar_[u + 1]);
    switch ((a >> 4) & 7u) {
      case 1: f(u + m_, (int)wpar_[u]); break;
      case 2: f(u + 1, (int)wpar_[u]); break;
      case 3: f(u - 1, (int)wpar_[u]); break;
      case 4: f(u - m_, (int)wpar_[u]); break;
      default: break;
    }
    if (a & 4u) f(u - 1, (int)wpar_[u - 1]);
    if (a & 8u) f(u - m_, (int)wpar_[u - m_]);
  }
  std::vector<int32_t> roots_, region_, region_max_, order_;

  void connect(int u, int v, int w) {
    link_.push_back({v, w, head_[u]}); head_[u] = (int)link_.size() - 1;
    link_.push_back({u, w, head_[v]}); head_[v] = (int)link_.size() - 1;
    adj_[u] |= 0x80u; adj_[v] |= 0x80u;
  }
  int colour_gap(int a, int b) const {
    int g = 0;
    for (int k = 0; k < 3; ++k) g = std::max(g, abs((int)img3_[a * 3 + k] - (int)img3_[b * 3 + k]));
    return g;
  }

  void arborescence(const double* gx, const double* gy) {
    const int SR = N_;   // the super-root
    const int SR_W = 1000000000;
    EdgeHeaps& heaps = heaps_;
    heaps.reset((size_t)N_ * 5);
    // per vertex (pixel or super-root), in ONE record: what the contraction reads and writes together for a vertex - the two
    // union-finds' links, its heap, the arc chosen for it (six arrays before: a vertex met at random cost six cache misses)
    std::vector<Vtx>& vtx = vtx_;
    vtx.resize(N_ + 1);
    for (int i = 0; i <= N_; ++i) vtx[i] = Vtx{i, i, 0, -1, 0, -1};
    std::vector<uint8_t>& arc_w = arc_w_;   // weights of the neighbour arcs (slot 0 is SR_W)
    arc_w.resize((size_t)N_ * 5);
    const int arc_step[5] = {0, -1, +1, -m_, +m_};
    auto arc_from = [&](int e) {   // (a table and one select instead of a chain of data-dependent branches)
      const int p = e / 5, k = e - 5 * p;
      const int q = p + arc_step[k];
      return k == 0 ? SR : q;
    };
    auto arc_weight = [&](int e) { return e % 5 == 0 ? SR_W : (int)arc_w[e]; };
    {
      // every pixel's heap of incoming arcs depends on that pixel alone: built in row bands on a few threads
      const uint64_t* templates = EdgeHeaps::pixel_templates();
      auto band = [&](int row0, int row1) {
        for (int i = row0; i < row1; ++i)
          for (int j = 0; j < m_; ++j) {
            const int p = i * m_ + j;
            // an arc runs from the flatter pixel to the steeper one, both ways when the gradient magnitudes differ
            // by less than 1 (the reference truncates the difference to int); a = left / upper pixel of the pair
            int mask = 0, w[4] = {0, 0, 0, 0};
            if (j > 0 && (int)(fabs(gx[p - 1]) - fabs(gx[p])) <= 0) { mask |= 1; w[0] = colour_gap(p, p - 1); }
            if (j + 1 < m_ && (int)(fabs(gx[p]) - fabs(gx[p + 1])) >= 0) { mask |= 2; w[1] = colour_gap(p, p + 1); }
            if (i > 0 && (int)(fabs(gy[p - m_]) - fabs(gy[p])) <= 0) { mask |= 4; w[2] = colour_gap(p, p - m_); }
            if (i + 1 < n_ && (int)(fabs(gy[p]) - fabs(gy[p + m_])) >= 0) { mask |= 8; w[3] = colour_gap(p, p + m_); }
            // rank of each existing arc's weight among the others (how many of them are smaller)
            int code = 0;
            for (int a = 0; a < 4; ++a) {
              if (!(mask >> a & 1)) continue;
              int r = 0;
              for (int b = 0; b < 4; ++b) r += (mask >> b & 1) && w[b] < w[a];
              code |= r << (2 * a);
              arc_w[5 * p + 1 + a] = (uint8_t)w[a];
            }
            vtx[p].incoming = heaps.place_pixel(5 * p, templates[mask << 8 | code], mask, SR_W, w);
          }
      };
      // (alone: four threads; with many builders running side by side the cores are taken already)
      const int nthr = std::max(1, std::min(g_active_builders.load() > 4 ? 1 : 4, n_ / 64));
      std::vector<std::thread> pool;
      for (int t = 1; t < nthr; ++t) pool.emplace_back(band, n_ * t / nthr, n_ * (t + 1) / nthr);
      band(0, n_ / nthr);
      for (auto& th : pool) th.join();
    }

    sub_mark("heaps per pixel");
    // contraction phase
    std::vector<int>& todo = todo_;
    todo.resize(N_ + 1);
    for (int i = 0; i <= N_; ++i) todo[i] = i;
    // the contraction forest, one record per (super-)node: see Contraction
    std::vector<Cnode>& con = con_;
    con.resize(2 * N_ + 2);
    for (int i = 0; i < 2 * N_ + 2; ++i) con[i] = Cnode{i, i, -1};
    auto strong_find = [&](int x) {   // union-find with full path compression
      int r = x;
      while (vtx[r].strong != r) r = vtx[r].strong;
      while (x != r) { const int nx = vtx[x].strong; vtx[x].strong = r; x = nx; }
      return r;
    };
    auto weak_find = [&](int x) {
      int r = x;
      while (vtx[r].weak != r) r = vtx[r].weak;
      while (x != r) { const int nx = vtx[x].weak; vtx[x].weak = r; x = nx; }
      return r;
    };
    auto strong_unite = [&](int a, int b) { a = strong_find(a); b = strong_find(b); vtx[b].strong = a; };   // a's root stays the root
    // stops one step BELOW the top (the reference's DFU2, :1299-1313) - the caller reads the top as con[.].up
    auto con_find = [&](int x) {
      const int x0 = x;
      int below = x;
      while (x != con[x].up) { below = x; x = con[x].up; }
      if (con[x0].up == x || con[x0].up == x0) return x0;
      for (int q = x0; q != below;) { const int nx = con[q].up; con[q].up = below; q = nx; }
      return below;
    };
    int n_super = N_ + 1;
    while (!todo.empty()) {
      const int v = todo.back(); todo.pop_back();
      if (v != strong_find(v)) continue;
      Vtx& V = vtx[v];
      int u = -1, w = 0, pick = -1;
      bool found = false;
      while (V.incoming) {
        w = heaps.key(V.incoming); pick = EdgeHeaps::arc(V.incoming);
        V.incoming = heaps.drop_min(V.incoming);
        u = strong_find(arc_from(pick));
        if (u != v) { found = true; break; }
      }
      if (!found) continue;   // the super-root
      V.from = u; V.key = w; V.edge = pick;
      const int cv = con_find(pick / 5), top_v = con[cv].up;
      con[top_v].entry = pick;
      { const int wu = weak_find(u), wv = weak_find(v); if (wu != wv) { vtx[wu].weak = wv; continue; } }   // (= unite(v, u))
      // u already reaches v: the chosen edges close a cycle -> contract it into a new super-node
      const int super = n_super++;
      con[top_v].up = super;
      con[top_v].grp = super;
      auto settle = [&](int x) { if (vtx[x].key > 0) { heaps.subtract(vtx[x].incoming, vtx[x].key); vtx[x].key = 0; } };
      settle(v);
      for (int ek = vtx[u].edge, k = strong_find(vtx[u].from); k != v; ek = vtx[k].edge, k = strong_find(vtx[k].from)) {
        const int ck = con_find(arc_from(ek)), top_k = con[ck].up;
        con[top_k].up = super;
        con[top_k].grp = super;
        strong_unite(v, k);
        settle(k);
        V.incoming = heaps.meld(V.incoming, vtx[k].incoming);
      }
      settle(u);
      const int cu = con_find(arc_from(pick)), top_u = con[cu].up;
      con[top_u].up = super;
      con[top_u].grp = super;
      strong_unite(v, u);
      V.incoming = heaps.meld(V.incoming, vtx[u].incoming);
      todo.push_back(v);
    }
    sub_mark("contraction");
    // expansion phase: newest super-nodes first, each keeps the entering edge that was chosen for it
    // (the reference compresses every contraction path once more here, :1266-1270 - DFU2's result is not used and the walk below
    // follows grp, not up: dropped)
    std::vector<int>&chosen_from = chosen_from_, &chosen_key = chosen_key_;
    chosen_from.assign(N_ + 1, -1); chosen_key.assign(N_ + 1, 0);
    std::vector<uint8_t>& done = done_;
    done.assign(n_super, 0);
    for (int i = n_super - 1; i >= 0; --i) {
      if (i == SR || done[i]) continue;
      done[i] = 1;
      const int k = con[i].entry;
      if (k < 0) continue;
      int u = k / 5;
      while (u != i) {
        done[u] = 1;
        u = con[u].grp;
        if (u == con[u].grp) break;
      }
      if (u == i) { chosen_from[k / 5] = arc_from(k); chosen_key[k / 5] = arc_weight(k); }
    }
    sub_mark("expansion");
    head_.assign(N_ + 1, -1);
    link_.clear();
    adj_.assign(N_, 0); wpar_.assign(N_, 0);
    n_arb_ = 0;
    roots_.clear();
    bool odd_edge = false;
    for (int p = 0; p < N_; ++p) {
      const int q = chosen_from[p];
      if (q < 0) continue;
      if (q >= N_) { roots_.push_back(p); continue; }
      // the arborescence edge q -> p: between 4-neighbours (the arcs are)
      const int dir = q == p + m_ ? 1 : q == p + 1 ? 2 : q == p - 1 ? 3 : q == p - m_ ? 4 : 0;
      if (!dir) { odd_edge = true; continue; }
      adj_[p] |= (uint8_t)(dir << 4);
      adj_[q] |= (uint8_t)(dir == 1 ? 8 : dir == 2 ? 4 : dir == 3 ? 2 : 1);   // seen from q, p is the opposite way
      wpar_[p] = (uint8_t)chosen_key[p];
      ++n_arb_;
    }
    if (odd_edge) roots_.clear();   // (cannot happen: run() then reports the failure)
  }

  // region id and largest edge per tree of the forest (the reference's getSeq0 walks the forest breadth-first from all roots for
  // this; the results - which root a pixel hangs off, the maximum over a tree's edge weights - do not depend on the order, so they
  // are read off the arborescence's parent pointers directly: a pixel's region is its nearest labelled ancestor's, paths are
  // labelled as they are walked (round 5: 8 ms -> ~2 ms per tree; the breadth-first walk touched every adjacency chain at random)
  // Returns false when the parent pointers do not form a forest over the roots (a cycle, or a pixel whose chain ends nowhere):
  // the walk is bounded by the pixel count, and an unlabelled pixel is not quietly counted into region 0.
  bool label_regions() {
    region_.assign(N_, -1); region_max_.assign(roots_.size(), 0);
    const std::vector<int>& up = chosen_from_;
    for (size_t i = 0; i < roots_.size(); ++i) region_[roots_[i]] = (int)i;
    std::vector<int32_t>& path = order_;
    for (int p = 0; p < N_; ++p) {
      if (region_[p] >= 0) continue;
      path.clear();
      int q = p;
      while (q >= 0 && q < N_ && region_[q] < 0) {
        if ((int)path.size() >= N_) return false;          // longer than the image: a cycle
        path.push_back(q); q = up[q];
      }
      if (q < 0 || q >= N_) return false;                   // the chain left the image without meeting a root
      const int lab = region_[q];
      for (int x : path) region_[x] = lab;
    }
    for (int p = 0; p < N_; ++p)
      if (up[p] >= 0 && up[p] < N_) region_max_[region_[p]] = std::max(region_max_[region_[p]], chosen_key_[p]);
    return true;
  }

  void merge_regions() {
    // pixel count and colour sums per region, kept at the region's root pixel
    std::vector<int>&size = size_, &sum = sum_;
    size.assign(N_, 0); sum.assign((size_t)N_ * 3, 0);
    for (int u = 0; u < N_; ++u) {
      const int rt = roots_[region_[u]];
      ++size[rt];
      for (int k = 0; k < 3; ++k) sum[rt * 3 + k] += img3_[u * 3 + k];
    }
    std::vector<Cand>& cand = cand_;
    cand.clear();
    auto candidates = [&](int u, int v) {
      const int fu = region_[u], fv = region_[v];
      if (fu == fv) return;
      const int ru = roots_[fu], rv = roots_[fv];
      const int px = colour_gap(u, v);
      cand.push_back({u, v, px, (double)px});
      int mean_gap = 0;
      for (int k = 0; k < 3; ++k) mean_gap = std::max(mean_gap, abs(sum[ru * 3 + k] / size[ru] - sum[rv * 3 + k] / size[rv]));
      cand.push_back({ru, rv, mean_gap, mean_gap * 0.2});
    };
    for (int i = 0; i < n_; ++i) for (int j = 0; j + 1 < m_; ++j) candidates(i * m_ + j, i * m_ + j + 1);
    // the reference's second loop pairs (i, j) with (i, j + 1) again - flat index + 1, wrapping into the next row at
    // the last column - instead of the pixel below (:713-714); reproduced
    for (int j = 0; j < m_; ++j) for (int i = 0; i + 1 < n_; ++i) candidates(i * m_ + j, i * m_ + j + 1);
    std::sort(cand.begin(), cand.end(), [](const Cand& x, const Cand& y) { return x.key < y.key; });

    Sets& merged = merged_;
    merged.reset((int)roots_.size());
    const size_t full = (size_t)(N_ - 1) * 2;
    for (size_t i = 0; i < cand.size() && edge_links() < full; ++i) {
      const int u = cand[i].a, v = cand[i].b, c = cand[i].w;
      const int fu = merged.find(region_[u]), fv = merged.find(region_[v]);
      const int s_u = size[roots_[region_[u]]], s_v = size[roots_[region_[v]]];
      const int tu = (int)(region_max_[region_[u]] + sqrt((double)N_) * 150 / 128 * s_u);
      const int tv = (int)(region_max_[region_[v]] + sqrt((double)N_) * 150 / 128 * s_v);
      if (fu != fv && c < std::min(tu, tv) && abs(s_u - s_v) <= 50 && (s_u <= 50 || s_v <= 50)) {
        connect(u, v, c);
        merged.up[fu] = fv;
        size[roots_[region_[u]]] += size[roots_[region_[v]]];
        size[roots_[region_[v]]] = size[roots_[region_[u]]];
        region_max_[region_[u]] = std::max(region_max_[region_[u]], region_max_[region_[v]]);
        region_[v] = region_[u];
      }
    }
    for (size_t i = 0; i < cand.size() && edge_links() < full; ++i) {
      const int u = cand[i].a, v = cand[i].b;
      const int fu = merged.find(region_[u]), fv = merged.find(region_[v]);
      if (fu != fv) { connect(u, v, 255); merged.up[fu] = fv; }
    }
  }
};

}  // namespace

static std::unique_ptr<TreeBuilder> builder_take();
static void builder_give(std::unique_ptr<TreeBuilder> tb);

int svo_msa_tree_rec(const uint8_t* m_img3, const double* r_gra, const double* c_gra, int width, int height, MsaBfsRec* rec,
                     std::vector<int32_t>* level_ptr, int* maxw, int32_t* root) {
  if (!m_img3 || !r_gra || !c_gra || !rec || !level_ptr || !maxw || !root || width < 2 || height < 2 || (int64_t)width * height > (1 << 24))
    return SVO_E_INVALID;
  std::unique_ptr<TreeBuilder> tb = builder_take();
  tb->reshape(height, width);
  g_active_builders.fetch_add(1);
  const int rt = tb->run_rec(m_img3, r_gra, c_gra, rec, *level_ptr, maxw);
  g_active_builders.fetch_sub(1);
  builder_give(std::move(tb));
  if (rt == -3) return SVO_E_CAPACITY;
  if (rt < 0) return SVO_E_INVALID;
  *root = rt;
  return SVO_OK;
}

// One image's aggregation tree (host-side, needs no GPU).  m_img3: median-filtered colour image (height*width*3), r_gra /
// c_gra: its gradients (svo_msa_init).  seq: width*height, child_ptr: +1, child / child_w: -1 entries.
extern "C" int svo_msa_tree(const uint8_t* m_img3, const double* r_gra, const double* c_gra, int width, int height,
                            int32_t* seq, int32_t* child_ptr, int32_t* child, uint8_t* child_w, int32_t* root) {
  if (!m_img3 || !r_gra || !c_gra || !seq || !child_ptr || !child || !child_w || !root || width < 2 || height < 2 ||
      (int64_t)width * height > (1 << 24))
    return SVO_E_INVALID;
  std::unique_ptr<TreeBuilder> tb = builder_take();
  tb->reshape(height, width);
  g_active_builders.fetch_add(1);
  const int rt = tb->run(m_img3, r_gra, c_gra, seq, child_ptr, child, child_w);
  g_active_builders.fetch_sub(1);
  builder_give(std::move(tb));
  if (rt < 0) return SVO_E_INVALID;
  *root = rt;
  return SVO_OK;
}

// builders are kept (at most 32) for their allocations; concurrent callers each get their own
static std::mutex g_builder_pool_mutex;
static std::vector<std::unique_ptr<TreeBuilder>> g_builder_pool;
static std::unique_ptr<TreeBuilder> builder_take() {
  std::unique_ptr<TreeBuilder> tb;
  {
    std::lock_guard<std::mutex> lock(g_builder_pool_mutex);
    if (!g_builder_pool.empty()) { tb = std::move(g_builder_pool.back()); g_builder_pool.pop_back(); }
  }
  if (!tb) tb.reset(new TreeBuilder());
  return tb;
}
static void builder_give(std::unique_ptr<TreeBuilder> tb) {
  std::lock_guard<std::mutex> lock(g_builder_pool_mutex);
  if (g_builder_pool.size() < 32) g_builder_pool.push_back(std::move(tb));
}
