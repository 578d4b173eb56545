// orc_msa_graph.cpp - literal CPU restatement of the graph stages of the reference's MSA dense stereo and of the
// `MSA::solve` sequence around them (Thirdparty/MB/MSA.cpp): build :152-192, Tarjan :200-346 with the leftist heap
// LTREE :1207-1290 and the union-finds DFU :1172-1193 / DFU2 :1293-1313, getSeq0 :348-373, getSize :854-877,
// baseSort1 :661-762, Kruskal1 :764-808, getSeq :898-926, solve :1132-1169.
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: MSA.cpp needs OpenCV and cannot be built here, so this file is a
// reading of it, quirks included (they are marked "sic").  C++ because baseSort1 sorts with the unstable std::sort and
// the order of equal keys is whatever libstdc++'s introsort leaves.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

extern "C" {
#include "svo_oracle.h"
}

namespace {

struct Edge { int x, y, c; };                       // x -> y
struct Edge2 { int u, v, c; double w; bool operator<(const Edge2& o) const { return w < o.w; } };
struct Branch { int u, v, c, next; };

struct LTree {                                      // :1207-1290, node 0 = null (dis[0] = 0, sic)
  std::vector<int> l, r, dis, id, val;
  int tot = 0;
  const std::vector<Edge>* e = nullptr;
  void reset(size_t cap) { l.assign(cap, 0); r.assign(cap, 0); dis.assign(cap, 0); id.assign(cap, 0); val.assign(cap, 0); tot = 0; }
  int merge(int a, int b) {
    if (!a || !b) return a + b;
    std::vector<std::pair<int, int>> st;
    st.push_back({a, b});
    for (;;) {
      int u = st.back().first, v = st.back().second;
      if (!u || !v) { st.back().first = u + v; break; }
      if (val[u] > val[v]) { std::swap(st.back().first, st.back().second); std::swap(u, v); }
      st.push_back({r[u], v});
    }
    for (int i = (int)st.size() - 2; i >= 0; --i) {
      const int u = st[i].first;
      r[u] = st[i + 1].first;
      if (dis[l[u]] < dis[r[u]]) std::swap(l[u], r[u]);
      dis[u] = (!r[u]) ? 0 : (dis[r[u]] + 1);
    }
    return st[0].first;
  }
  void pop(int& a) { a = merge(l[a], r[a]); }
  int new_node(int k) { id[++tot] = k; val[tot] = (*e)[k].c; l[tot] = r[tot] = dis[tot] = 0; return tot; }
  void update(int a, int w) {                        // subtract w from every key of the heap
    if (!a) return;
    std::vector<int> q(1, a);
    for (size_t h = 0; h < q.size(); ++h) {
      const int u = q[h];
      val[u] -= w;
      if (l[u]) q.push_back(l[u]);
      if (r[u]) q.push_back(r[u]);
    }
  }
};

struct Dfu {                                        // :1172-1193
  std::vector<int> f;
  void init(int n) { f.resize(n + 1); for (int i = 0; i <= n; ++i) f[i] = i; }
  int find(int u) {
    int root = u;
    while (root != f[root]) root = f[root];
    while (u != root) { const int nx = f[u]; f[u] = root; u = nx; }
    return root;
  }
  void unite(int u, int v) { u = find(u); v = find(v); f[v] = u; }
};

struct Dfu2 {                                       // :1293-1313: returns the node just below the root (sic)
  std::vector<int> f, g;
  void init(int n) { f.resize(n + 1); g.resize(n + 1); for (int i = 0; i <= n; ++i) { f[i] = i; g[i] = i; } }
  int find(int u) {
    std::vector<int> st;
    const int k = u;
    while (u != f[u]) { st.push_back(u); u = f[u]; }
    if (f[k] == u || f[k] == k) return k;
    u = st.back(); st.pop_back();
    while (!st.empty()) { f[st.back()] = u; st.pop_back(); }
    return u;
  }
};

struct Msa {
  int n = 0, m = 0, Disp = 0;
  std::vector<Edge> e;
  std::vector<int> in, lk, root, anc, m_edg, fa, seq, size, sum_col, key, pre;
  std::vector<Branch> branch;
  std::vector<Edge2> sortE;
  LTree heap;
  int topR = 0, max_edg = 0;
  int T(int u, int v) const { return u * m + v; }

  void insert(int v, int u, int c) {                // :141-145: edge u -> v into the heap of v
    e.push_back({u, v, c});
    in[v] = heap.merge(in[v], heap.new_node((int)e.size() - 1));
  }
  void add(int u, int v, int c) { branch.push_back({u, v, c, lk[u]}); lk[u] = (int)branch.size() - 1; }   // :194-197

  void build(const double* r_gra, const double* c_gra, const uint8_t* img3) {   // :152-192
    const int nn = n * m, rt = nn;
    e.clear(); e.reserve((size_t)nn * 5 + 8);
    heap.e = &e; heap.reset((size_t)nn * 5 + 8);
    in.assign(nn + 1, 0);
    for (int i = 0; i < nn; ++i) insert(i, rt, (int)1e9);
    auto link = [&](int t1, int t2, const double* gra) {
      int dif_col = 0;
      for (int k = 0; k < 3; ++k) dif_col = std::max(dif_col, abs((int)img3[t1 * 3 + k] - (int)img3[t2 * 3 + k]));
      const int dif_gra = (int)(fabs(gra[t1]) - fabs(gra[t2]));    // double difference truncated to int (sic)
      if (abs(dif_gra) <= 0) { insert(t1, t2, dif_col); insert(t2, t1, dif_col); }
      else if (dif_gra < 0) insert(t2, t1, dif_col);
      else insert(t1, t2, dif_col);
    };
    for (int i = 0; i < n; ++i) for (int j = 0; j < m - 1; ++j) link(T(i, j), T(i, j + 1), r_gra);
    for (int j = 0; j < m; ++j) for (int i = 0; i < n - 1; ++i) link(T(i, j), T(i + 1, j), c_gra);
  }

  void tarjan() {                                    // :200-346
    const int nn = n * m;
    key.assign(2 * nn + 2, 0); pre.assign(2 * nn + 2, -1);
    std::vector<int> Pre(2 * nn + 2, -1), inEdg(nn + 1, -1), ie(2 * nn + 2, -1), st;
    lk.assign(nn + 1, -1);
    for (int i = 0; i <= nn; ++i) st.push_back(i);
    Dfu S, W; Dfu2 bel;
    S.init(nn); W.init(nn); bel.init(2 * nn);
    int cnt = nn + 1;
    while (!st.empty()) {
      const int v = st.back(); st.pop_back();
      if (v != S.find(v)) continue;
      int u = -1, w = 0, cur = -1;
      bool hasIn = false;
      while (in[v]) {
        w = heap.val[in[v]]; cur = heap.id[in[v]];
        heap.pop(in[v]);
        u = S.find(e[cur].x);
        if (u != v) { hasIn = true; break; }
      }
      if (!hasIn) continue;
      pre[v] = u; key[v] = w; inEdg[v] = cur;
      const int fv = bel.find(e[cur].y), ffv = bel.f[fv];
      ie[ffv] = cur; Pre[ffv] = fv;
      if (W.find(u) == W.find(v)) {
        bel.f[ffv] = cnt++;
        bel.g[ffv] = bel.f[ffv];
        if (key[v] > 0) { heap.update(in[v], key[v]); key[v] = 0; }
        for (int ek = inEdg[u], k = S.find(pre[u]); k != v; ek = inEdg[k], k = S.find(pre[k])) {
          const int kk = e[ek].x, fk = bel.find(kk), ffk = bel.f[fk];
          bel.f[ffk] = bel.f[ffv];
          bel.g[ffk] = bel.g[ffv];
          S.unite(v, k);
          if (key[k] > 0) { heap.update(in[k], key[k]); key[k] = 0; }
          in[v] = heap.merge(in[v], in[k]);
        }
        if (key[u] > 0) { heap.update(in[u], key[u]); key[u] = 0; }
        const int fu = bel.find(e[cur].x), ffu = bel.f[fu];
        bel.f[ffu] = bel.f[ffv];
        bel.g[ffu] = bel.g[ffv];
        S.unite(v, u);
        in[v] = heap.merge(in[v], in[u]);
        st.push_back(v);
      } else {
        W.unite(v, u);
      }
    }
    for (int i = 0; i < cnt; ++i) { bel.find(i); pre[i] = -1; }
    std::vector<uint8_t> mark(cnt, 0);
    for (int i = cnt - 1; i >= 0; --i) {
      if (i == nn || mark[i]) continue;
      mark[i] = 1;
      const int k = ie[i];
      if (k < 0) continue;                           // the reference would read e[-1] here; never happens on its inputs
      int u = e[k].y;
      while (u != i) {
        mark[u] = 1;
        u = bel.g[u];
        if (u == bel.g[u]) break;
      }
      if (u == i) { pre[e[k].y] = e[k].x; key[e[k].y] = e[k].c; }
    }
    max_edg = 0; branch.clear(); branch.reserve((size_t)nn * 2 + 4); root.clear(); topR = 0;
    for (int i = 0; i < nn; ++i) {
      if (pre[i] == -1) continue;                    // "Sth wrong!"
      if (pre[i] < nn) { add(pre[i], i, key[i]); add(i, pre[i], key[i]); max_edg = std::max(max_edg, key[i]); }
      else { root.push_back(i); ++topR; }
    }
  }

  void get_seq0() {                                  // :348-373
    const int nn = n * m;
    fa.assign(nn, 0); anc.assign(nn, 0); m_edg.assign(std::max(topR, 1), 0); seq.clear(); seq.reserve(nn);
    std::vector<int> queue;
    queue.reserve(nn);
    for (int i = 0; i < topR; ++i) { queue.push_back(root[i]); fa[root[i]] = -1; anc[root[i]] = i; m_edg[i] = 0; }
    for (size_t t = 0; t < queue.size(); ++t) {
      const int u = queue[t];
      seq.push_back(u);
      for (int i = lk[u]; i > -1; i = branch[i].next) {
        const int v = branch[i].v;
        if (v == fa[u]) continue;
        queue.push_back(v);
        fa[v] = u; anc[v] = anc[u];
        m_edg[anc[u]] = std::max(m_edg[anc[u]], branch[i].c);
      }
    }
  }

  void get_size(const uint8_t* img3) {               // :854-877
    const int nn = n * m;
    size.assign(nn, 0); sum_col.assign((size_t)nn * 3, 0);
    for (int k = 0; k < (int)seq.size(); ++k) {
      const int u = seq[k], rt = root[anc[u]];
      ++size[rt];
      for (int i = 0; i < 3; ++i) sum_col[rt * 3 + i] += img3[u * 3 + i];
    }
  }

  void base_sort1(const uint8_t* img3) {             // :661-762
    std::vector<Edge2> oriE;
    auto pair_edges = [&](int u, int v) {
      const int fu = anc[u], fv = anc[v];
      if (fu == fv) return;
      const int ru = root[fu], rv = root[fv];
      int dif_col = 0;
      for (int k = 0; k < 3; ++k) dif_col = std::max(dif_col, abs((int)img3[u * 3 + k] - (int)img3[v * 3 + k]));
      oriE.push_back({u, v, dif_col, (double)dif_col});
      dif_col = 0;
      for (int k = 0; k < 3; ++k)
        dif_col = std::max(dif_col, abs(sum_col[ru * 3 + k] / size[ru] - sum_col[rv * 3 + k] / size[rv]));
      oriE.push_back({ru, rv, dif_col, dif_col * 0.2});
    };
    for (int i = 0; i < n; ++i) for (int j = 0; j < m - 1; ++j) pair_edges(T(i, j), T(i, j + 1));
    for (int j = 0; j < m; ++j) for (int i = 0; i < n - 1; ++i) pair_edges(T(i, j), T(i, j + 1));   // T(i, j + 1), not T(i + 1, j) (sic)
    std::sort(oriE.begin(), oriE.end());
    sortE = oriE;
  }

  void kruskal1(const uint8_t* img3) {               // :764-808
    const int nn = n * m;
    get_size(img3);
    base_sort1(img3);
    Dfu S;
    S.init(topR);
    const int totE = (int)sortE.size();
    for (int i = 0; i < totE && (int)branch.size() < (nn - 1) * 2; ++i) {
      const int u = sortE[i].u, v = sortE[i].v, c = sortE[i].c;
      const int fu = S.find(anc[u]), fv = S.find(anc[v]);
      const int s_u = size[root[anc[u]]], s_v = size[root[anc[v]]];
      const int tu = (int)(m_edg[anc[u]] + sqrt((double)nn) * 150 / 128 * s_u);
      const int tv = (int)(m_edg[anc[v]] + sqrt((double)nn) * 150 / 128 * s_v);
      const int lim = 50, area = 50;
      if (fu != fv && c < std::min(tu, tv) && (abs(s_u - s_v) <= lim) && (s_u <= area || s_v <= area)) {
        add(u, v, c); add(v, u, c);
        S.f[fu] = fv;
        size[root[anc[u]]] += size[root[anc[v]]];
        size[root[anc[v]]] = size[root[anc[u]]];
        m_edg[anc[u]] = std::max(m_edg[anc[u]], m_edg[anc[v]]);
        anc[v] = anc[u];
      }
    }
    for (int i = 0; i < totE && (int)branch.size() < (nn - 1) * 2; ++i) {
      const int u = sortE[i].u, v = sortE[i].v;
      const int fu = S.find(anc[u]), fv = S.find(anc[v]);
      if (fu != fv) { add(u, v, 255); add(v, u, 255); S.f[fu] = fv; }
    }
  }

  void get_seq() {                                   // :898-926
    const int nn = n * m;
    fa.assign(nn, -1);
    seq.clear();
    std::vector<int> queue(1, root[0]);
    queue.reserve(nn);
    for (size_t t = 0; t < queue.size(); ++t) {
      const int u = queue[t];
      seq.push_back(u);
      for (int i = lk[u]; i > -1; i = branch[i].next) {
        const int v = branch[i].v;
        if (v == fa[u]) continue;
        queue.push_back(v);
        fa[v] = u;
      }
    }
  }

  // the tree of one image as TreeDp walks it: BFS order, children per node in chain order
  int tree(const uint8_t* m_img3, const double* r_gra, const double* c_gra, std::vector<int32_t>& oseq, std::vector<int32_t>& cptr,
           std::vector<int32_t>& child, std::vector<uint8_t>& cc) {
    build(r_gra, c_gra, m_img3);
    tarjan();
    if (topR < 1) return -1;
    get_seq0();
    kruskal1(m_img3);
    get_seq();
    const int nn = n * m;
    if ((int)seq.size() != nn) return -2;
    oseq.assign(seq.begin(), seq.end());
    cptr.assign(nn + 1, 0); child.clear(); cc.clear();
    for (int u = 0; u < nn; ++u) {
      for (int i = lk[u]; i > -1; i = branch[i].next)
        if (branch[i].v != fa[u]) { child.push_back(branch[i].v); cc.push_back((uint8_t)branch[i].c); }
      cptr[u + 1] = (int32_t)child.size();
    }
    return root[0];
  }
};

}  // namespace

extern "C" {

/* The spanning tree MSA aggregates over, for ONE image: m_img3 = its median-filtered colour image (n*m*3), r_gra / c_gra its
 * gradients (orc_msa_init).  seq: n*m, child_ptr: n*m+1, child / child_c: n*m-1.  Returns the root pixel (>= 0) or < 0. */
int orc_msa_tree(const uint8_t* m_img3, const double* r_gra, const double* c_gra, int n, int m, int32_t* seq,
                 int32_t* child_ptr, int32_t* child, uint8_t* child_c) {
  Msa M;
  M.n = n; M.m = m;
  std::vector<int32_t> s, p, c; std::vector<uint8_t> w;
  const int rt = M.tree(m_img3, r_gra, c_gra, s, p, c, w);
  if (rt < 0) return rt;
  memcpy(seq, s.data(), s.size() * sizeof(int32_t));
  memcpy(child_ptr, p.data(), p.size() * sizeof(int32_t));
  memcpy(child, c.data(), c.size() * sizeof(int32_t));
  memcpy(child_c, w.data(), w.size());
  return rt;
}

/* MSA::solve(l, r, d, scale, Save) (:1132-1169): disparity image (n*m bytes, value = disparity * scale), left reference. */
int orc_msa_solve(const uint8_t* bgrL, const uint8_t* bgrR, int n, int m, int d, int scale, uint8_t* out) {
  const int D = d + 1, N = n * m;
  const double o = 0.1;
  std::vector<float> costL((size_t)N * D), costR((size_t)N * D), up((size_t)N * D), A((size_t)N * D);
  std::vector<uint8_t> m3L((size_t)N * 3), m3R((size_t)N * 3), d0(N), d1(N), mask(N);
  std::vector<double> rgL(N), cgL(N), rgR(N), cgR(N);
  if (orc_msa_init(bgrL, bgrR, n, m, D, costL.data(), costR.data(), m3L.data(), m3R.data(), rgL.data(), cgL.data(), rgR.data(),
                   cgR.data())) return -1;
  double Exp[256];
  orc_msa_exp_table(o, Exp);
  std::vector<int32_t> seq(N), cptr(N + 1), child(N), tseq, tptr, tchild;
  std::vector<uint8_t> cc(N), tcc;
  /* right image as base image */
  int root = orc_msa_tree(m3R.data(), rgR.data(), cgR.data(), n, m, seq.data(), cptr.data(), child.data(), cc.data());
  if (root < 0) return -2;
  orc_msa_tree_dp(costR.data(), N, D, seq.data(), cptr.data(), child.data(), cc.data(), root, Exp, up.data(), A.data());
  orc_msa_wta(A.data(), n, m, D, d1.data());
  /* left image as base image */
  root = orc_msa_tree(m3L.data(), rgL.data(), cgL.data(), n, m, seq.data(), cptr.data(), child.data(), cc.data());
  if (root < 0) return -3;
  orc_msa_tree_dp(costL.data(), N, D, seq.data(), cptr.data(), child.data(), cc.data(), root, Exp, up.data(), A.data());
  orc_msa_wta(A.data(), n, m, D, d0.data());
  /* refine */
  orc_msa_lrcheck(d0.data(), d1.data(), n, m, D, costL.data(), mask.data());
  orc_msa_exp_table(o / 2, Exp);
  orc_msa_tree_dp(costL.data(), N, D, seq.data(), cptr.data(), child.data(), cc.data(), root, Exp, up.data(), A.data());
  orc_msa_wta(A.data(), n, m, D, d0.data());
  for (int i = 0; i < N; ++i) out[i] = (uint8_t)(d0[i] * scale);
  return 0;
}

}  // extern "C"
