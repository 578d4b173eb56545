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

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

#include "svo_internal.h"

namespace {

struct DirEdge { int32_t from, to, w; };

// Leftist heaps over edge indices, all in one node pool; node 0 is "empty".  Rank of an empty child counts as 0,
// like a leaf's (the reference's convention, :1232-1234) - it decides which child ends up left, hence later merges.
class EdgeHeaps {
 public:
  explicit EdgeHeaps(size_t cap) : node_(cap + 1) { node_[0] = Node{0, 0, 0, 0, 0}; used_ = 0; }
  int make(int edge, int key) { node_[++used_] = Node{0, 0, 0, edge, key}; return used_; }
  int key(int h) const { return node_[h].key; }
  int edge(int h) const { return node_[h].edge; }
  int meld(int a, int b) {
    if (!a || !b) return a + b;
    spine_.clear();
    // walk down the right spines, always continuing below the smaller key (ties: the first heap stays on top)
    for (;;) {
      if (!a || !b) { a += b; break; }
      if (node_[a].key > node_[b].key) std::swap(a, b);
      spine_.push_back(a);
      a = node_[a].right;
    }
    int sub = a;
    for (size_t i = spine_.size(); i-- > 0;) {
      Node& t = node_[spine_[i]];
      t.right = sub;
      if (node_[t.left].rank < node_[t.right].rank) std::swap(t.left, t.right);
      t.rank = t.right ? node_[t.right].rank + 1 : 0;
      sub = spine_[i];
    }
    return sub;
  }
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
  struct Node { int left, right, rank, edge, key; };
  std::vector<Node> node_;
  std::vector<int> spine_, walk_;
  int used_;
};

struct Sets {   // union-find with full path compression; unite(a, b) makes a's root the root
  std::vector<int> up;
  explicit Sets(int n) : up(n + 1) { for (int i = 0; i <= n; ++i) up[i] = i; }
  int find(int x) {
    int r = x;
    while (up[r] != r) r = up[r];
    while (x != r) { const int nx = up[x]; up[x] = r; x = nx; }
    return r;
  }
  void unite(int a, int b) { a = find(a); b = find(b); up[b] = a; }
};

// The contraction forest of the arborescence algorithm: `up` links a (super-)node to the cycle node that absorbed
// it, `grp` remembers that cycle node for the expansion.  find() stops one step BELOW the top (the reference's DFU2,
// :1299-1313) - the caller reads the top as up[find(x)].
struct Contraction {
  std::vector<int> up, grp;
  explicit Contraction(int n) : up(n + 1), grp(n + 1) { for (int i = 0; i <= n; ++i) { up[i] = i; grp[i] = i; } }
  int find(int x) {
    path_.clear();
    const int x0 = x;
    while (x != up[x]) { path_.push_back(x); x = up[x]; }
    if (up[x0] == x || up[x0] == x0) return x0;
    const int below_top = path_.back();
    path_.pop_back();
    for (int p : path_) up[p] = below_top;
    return below_top;
  }
 private:
  std::vector<int> path_;
};

struct Link { int32_t to, w, next; };   // adjacency chains, newest first, as TreeDp walks them

class TreeBuilder {
 public:
  TreeBuilder(int rows, int cols) : n_(rows), m_(cols), N_(rows * cols) {}

  // returns the root pixel, or < 0
  int run(const uint8_t* img3, const double* gx, const double* gy, std::vector<int32_t>& seq, std::vector<int32_t>& child_ptr,
          std::vector<int32_t>& child, std::vector<uint8_t>& child_w) {
    img3_ = img3;
    arborescence(gx, gy);
    if (roots_.empty()) return -1;
    label_regions();
    merge_regions();
    // breadth-first order from the first root; parent[] doubles as the visited mark
    std::vector<int32_t> parent(N_, -1);
    seq.clear(); seq.reserve(N_);
    seq.push_back(roots_[0]);
    for (size_t t = 0; t < seq.size(); ++t) {
      const int u = seq[t];
      for (int i = head_[u]; i >= 0; i = link_[i].next) {
        const int v = link_[i].to;
        if (v == parent[u]) continue;
        seq.push_back(v);
        parent[v] = u;
      }
    }
    if ((int)seq.size() != N_) return -2;
    child_ptr.assign(N_ + 1, 0); child.clear(); child_w.clear();
    child.reserve(N_); child_w.reserve(N_);
    for (int u = 0; u < N_; ++u) {
      for (int i = head_[u]; i >= 0; i = link_[i].next)
        if (link_[i].to != parent[u]) { child.push_back(link_[i].to); child_w.push_back((uint8_t)link_[i].w); }
      child_ptr[u + 1] = (int32_t)child.size();
    }
    return roots_[0];
  }

 private:
  int n_, m_, N_;
  const uint8_t* img3_ = nullptr;
  std::vector<int32_t> head_;
  std::vector<Link> link_;
  std::vector<int32_t> roots_, region_, region_max_, order_;

  void connect(int u, int v, int w) {
    link_.push_back({v, w, head_[u]}); head_[u] = (int)link_.size() - 1;
    link_.push_back({u, w, head_[v]}); head_[v] = (int)link_.size() - 1;
  }
  int colour_gap(int a, int b) const {
    int g = 0;
    for (int k = 0; k < 3; ++k) g = std::max(g, abs((int)img3_[a * 3 + k] - (int)img3_[b * 3 + k]));
    return g;
  }

  void arborescence(const double* gx, const double* gy) {
    const int SR = N_;   // the super-root
    std::vector<DirEdge> edge;
    edge.reserve((size_t)N_ * 5);
    EdgeHeaps heaps((size_t)N_ * 5 + 8);
    std::vector<int> incoming(N_ + 1, 0);
    auto arc = [&](int from, int to, int w) {
      edge.push_back({from, to, w});
      incoming[to] = heaps.meld(incoming[to], heaps.make((int)edge.size() - 1, w));
    };
    for (int p = 0; p < N_; ++p) arc(SR, p, 1000000000);
    auto neighbours = [&](int a, int b, const double* g) {
      const int w = colour_gap(a, b);
      const int steeper = (int)(fabs(g[a]) - fabs(g[b]));   // the reference truncates the difference to int
      if (steeper == 0) { arc(b, a, w); arc(a, b, w); }
      else if (steeper < 0) arc(a, b, w);                  // a is flatter: a -> b
      else arc(b, a, w);
    };
    for (int i = 0; i < n_; ++i) for (int j = 0; j + 1 < m_; ++j) neighbours(i * m_ + j, i * m_ + j + 1, gx);
    for (int j = 0; j < m_; ++j) for (int i = 0; i + 1 < n_; ++i) neighbours(i * m_ + j, (i + 1) * m_ + j, gy);

    // contraction phase
    std::vector<int> chosen_from(2 * N_ + 2, -1), chosen_key(2 * N_ + 2, 0), chosen_edge(N_ + 1, -1), entry(2 * N_ + 2, -1);
    std::vector<int> todo(N_ + 1);
    for (int i = 0; i <= N_; ++i) todo[i] = i;
    Sets strong(N_), weak(N_);
    Contraction con(2 * N_);
    int n_super = N_ + 1;
    while (!todo.empty()) {
      const int v = todo.back(); todo.pop_back();
      if (v != strong.find(v)) continue;
      int u = -1, w = 0, pick = -1;
      bool found = false;
      while (incoming[v]) {
        w = heaps.key(incoming[v]); pick = heaps.edge(incoming[v]);
        incoming[v] = heaps.drop_min(incoming[v]);
        u = strong.find(edge[pick].from);
        if (u != v) { found = true; break; }
      }
      if (!found) continue;   // the super-root
      chosen_from[v] = u; chosen_key[v] = w; chosen_edge[v] = pick;
      const int cv = con.find(edge[pick].to), top_v = con.up[cv];
      entry[top_v] = pick;
      if (weak.find(u) != weak.find(v)) { weak.unite(v, u); continue; }
      // u already reaches v: the chosen edges close a cycle -> contract it into a new super-node
      con.up[top_v] = n_super++;
      con.grp[top_v] = con.up[top_v];
      auto settle = [&](int x) { if (chosen_key[x] > 0) { heaps.subtract(incoming[x], chosen_key[x]); chosen_key[x] = 0; } };
      settle(v);
      for (int ek = chosen_edge[u], k = strong.find(chosen_from[u]); k != v; ek = chosen_edge[k], k = strong.find(chosen_from[k])) {
        const int ck = con.find(edge[ek].from), top_k = con.up[ck];
        con.up[top_k] = con.up[top_v];
        con.grp[top_k] = con.grp[top_v];
        strong.unite(v, k);
        settle(k);
        incoming[v] = heaps.meld(incoming[v], incoming[k]);
      }
      settle(u);
      const int cu = con.find(edge[pick].from), top_u = con.up[cu];
      con.up[top_u] = con.up[top_v];
      con.grp[top_u] = con.grp[top_v];
      strong.unite(v, u);
      incoming[v] = heaps.meld(incoming[v], incoming[u]);
      todo.push_back(v);
    }
    // expansion phase: newest super-nodes first, each keeps the entering edge that was chosen for it
    for (int i = 0; i < n_super; ++i) { con.find(i); chosen_from[i] = -1; }
    std::vector<uint8_t> done(n_super, 0);
    for (int i = n_super - 1; i >= 0; --i) {
      if (i == SR || done[i]) continue;
      done[i] = 1;
      const int k = entry[i];
      if (k < 0) continue;
      int u = edge[k].to;
      while (u != i) {
        done[u] = 1;
        u = con.grp[u];
        if (u == con.grp[u]) break;
      }
      if (u == i) { chosen_from[edge[k].to] = edge[k].from; chosen_key[edge[k].to] = edge[k].w; }
    }
    head_.assign(N_ + 1, -1);
    link_.clear(); link_.reserve((size_t)N_ * 2 + 4);
    roots_.clear();
    for (int p = 0; p < N_; ++p) {
      if (chosen_from[p] < 0) continue;
      if (chosen_from[p] < N_) connect(chosen_from[p], p, chosen_key[p]);
      else roots_.push_back(p);
    }
  }

  // region id and largest edge per tree of the forest, breadth-first from all roots (the reference's getSeq0)
  void label_regions() {
    region_.assign(N_, 0); region_max_.assign(roots_.size(), 0);
    std::vector<int32_t> parent(N_, 0);
    order_.clear(); order_.reserve(N_);
    for (size_t i = 0; i < roots_.size(); ++i) { order_.push_back(roots_[i]); parent[roots_[i]] = -1; region_[roots_[i]] = (int)i; }
    for (size_t t = 0; t < order_.size(); ++t) {
      const int u = order_[t];
      for (int i = head_[u]; i >= 0; i = link_[i].next) {
        const int v = link_[i].to;
        if (v == parent[u]) continue;
        order_.push_back(v);
        parent[v] = u; region_[v] = region_[u];
        region_max_[region_[u]] = std::max(region_max_[region_[u]], link_[i].w);
      }
    }
  }

  void merge_regions() {
    // pixel count and colour sums per region, kept at the region's root pixel
    std::vector<int> size(N_, 0), sum((size_t)N_ * 3, 0);
    for (int u : order_) {
      const int rt = roots_[region_[u]];
      ++size[rt];
      for (int k = 0; k < 3; ++k) sum[rt * 3 + k] += img3_[u * 3 + k];
    }
    struct Cand { int a, b, w; double key; };
    std::vector<Cand> cand;
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

    Sets merged((int)roots_.size());
    const size_t full = (size_t)(N_ - 1) * 2;
    for (size_t i = 0; i < cand.size() && link_.size() < full; ++i) {
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
    for (size_t i = 0; i < cand.size() && link_.size() < full; ++i) {
      const int u = cand[i].a, v = cand[i].b;
      const int fu = merged.find(region_[u]), fv = merged.find(region_[v]);
      if (fu != fv) { connect(u, v, 255); merged.up[fu] = fv; }
    }
  }
};

}  // namespace

// One image's aggregation tree (host-side, needs no GPU).  m_img3: median-filtered colour image (height*width*3), r_gra /
// c_gra: its gradients (svo_msa_init).  seq: width*height, child_ptr: +1, child / child_w: -1 entries.
extern "C" int svo_msa_tree(const uint8_t* m_img3, const double* r_gra, const double* c_gra, int width, int height,
                            int32_t* seq, int32_t* child_ptr, int32_t* child, uint8_t* child_w, int32_t* root) {
  if (!m_img3 || !r_gra || !c_gra || !seq || !child_ptr || !child || !child_w || !root || width < 2 || height < 2 ||
      (int64_t)width * height > (1 << 24))
    return SVO_E_INVALID;
  TreeBuilder tb(height, width);
  std::vector<int32_t> s, p, c;
  std::vector<uint8_t> w;
  const int rt = tb.run(m_img3, r_gra, c_gra, s, p, c, w);
  if (rt < 0) return SVO_E_INVALID;
  std::copy(s.begin(), s.end(), seq);
  std::copy(p.begin(), p.end(), child_ptr);
  std::copy(c.begin(), c.end(), child);
  std::copy(w.begin(), w.end(), child_w);
  *root = rt;
  return SVO_OK;
}
