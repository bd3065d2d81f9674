#include "spd.h"

#include <omp.h>

#include "graph.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <queue>

#ifndef DPGO_NO_DEVICE   /* (tools/cpu_baseline builds this file for the host alone) */
#include <hip/hip_runtime.h>
#endif

namespace dpgo {
int spd_factor_numeric_device(const CsrMatrix &A, SpdFactor &F, const std::vector<std::vector<int>> &children,
                              double *flops_out, double *mfma_ms_out);
#ifdef DPGO_NO_DEVICE
void spd_release_device(SpdFactor &) {}
void spd_release_numeric(SpdFactor &) {}
#endif
namespace {

struct TreeNode {
  std::vector<int> verts;
  int parent = -1;
  std::vector<int> children;
};

// Nested dissection by BFS level structures (George's automatic ND): the middle
// level of a breadth-first search from a pseudo-peripheral vertex is a vertex
// separator, because edges only join equal or adjacent levels.
struct Dissector {
  const CsrMatrix &A;
  int leaf;
  std::vector<TreeNode> nodes;
  std::vector<int> stamp, level, queue_;
  int cur = 0;

  Dissector(const CsrMatrix &A_, int leaf_) : A(A_), leaf(leaf_), stamp(A_.n, -1), level(A_.n, 0) {}

  int make(std::vector<int> verts, int parent) {
    TreeNode t;
    t.verts = std::move(verts);
    t.parent = parent;
    nodes.push_back(std::move(t));
    int id = (int)nodes.size() - 1;
    if (parent >= 0) nodes[parent].children.push_back(id);
    return id;
  }

  // BFS restricted to vertices with stamp == cur; returns number of levels and fills `order`.
  int bfs(int root, std::vector<int> &order) {
    order.clear();
    order.push_back(root);
    level[root] = 0;
    stamp[root] = cur + 1;  // visited marker
    size_t head = 0;
    int maxlev = 0;
    while (head < order.size()) {
      int v = order[head++];
      for (int k = A.ptr[v]; k < A.ptr[v + 1]; k++) {
        int w = A.col[k];
        if (stamp[w] != cur) continue;
        stamp[w] = cur + 1;
        level[w] = level[v] + 1;
        maxlev = std::max(maxlev, level[w]);
        order.push_back(w);
      }
    }
    for (int v : order) stamp[v] = cur;  // unmark
    return maxlev + 1;
  }

  // Minimum vertex cover of the bipartite graph between X = order[x0, x1) and Y = order[x1, y1) (edges of A among
  // the current vertex set; X and Y are consecutive BFS levels).  Hopcroft-Karp matching, then Koenig's construction.
  std::vector<int> lidx_, matchx_, matchy_, dist_;
  void min_cover(const std::vector<int> &order, int x0, int x1, int y1, std::vector<int> &cover) {
    const int nx = x1 - x0, ny = y1 - x1;
    if ((int)lidx_.size() < A.n) lidx_.assign(A.n, -1);
    for (int i = 0; i < ny; i++) lidx_[order[x1 + i]] = i;
    const int ylev = level[order[x1]];
    matchx_.assign(nx, -1);
    matchy_.assign(ny, -1);
    dist_.assign(nx, 0);
    auto ynbr = [&](int xv, int k) {   // local index of neighbour k of X vertex xv if it lies in Y, else -1
      const int w = A.col[k];
      return (stamp[w] == cur && level[w] == ylev) ? lidx_[w] : -1;
    };
    std::vector<int> q, it(nx);
    for (;;) {
      // BFS from the free X vertices: layers of shortest alternating paths
      q.clear();
      for (int i = 0; i < nx; i++) {
        dist_[i] = matchx_[i] < 0 ? 0 : -1;
        if (matchx_[i] < 0) q.push_back(i);
      }
      bool found = false;
      for (size_t h = 0; h < q.size(); h++) {
        const int i = q[h], xv = order[x0 + i];
        for (int k = A.ptr[xv]; k < A.ptr[xv + 1]; k++) {
          const int j = ynbr(xv, k);
          if (j < 0) continue;
          const int i2 = matchy_[j];
          if (i2 < 0) found = true;
          else if (dist_[i2] < 0) { dist_[i2] = dist_[i] + 1; q.push_back(i2); }
        }
      }
      if (!found) break;
      // DFS along the layers (iterative)
      for (int i = 0; i < nx; i++) it[i] = A.ptr[order[x0 + i]];
      for (int s = 0; s < nx; s++) {
        if (matchx_[s] >= 0) continue;
        std::vector<int> path(1, s);
        while (!path.empty()) {
          const int i = path.back(), xv = order[x0 + i];
          bool advanced = false;
          while (it[i] < A.ptr[xv + 1]) {
            const int j = ynbr(xv, it[i]++);
            if (j < 0) continue;
            const int i2 = matchy_[j];
            if (i2 < 0) {   // augment along the path
              int jj = j;
              for (int p = (int)path.size() - 1; p >= 0; p--) {
                const int pi = path[p], prev = matchx_[pi];
                matchx_[pi] = jj;
                matchy_[jj] = pi;
                jj = prev;
              }
              path.clear();
              advanced = true;
              break;
            }
            if (dist_[i2] == dist_[i] + 1) { path.push_back(i2); advanced = true; break; }
          }
          if (!advanced) { dist_[i] = -2; path.pop_back(); }   // dead end
        }
      }
    }
    // Koenig: Z = vertices reachable from the free X vertices by alternating paths; cover = (X \ Z) + (Y & Z)
    std::vector<char> zx(nx, 0), zy(ny, 0);
    q.clear();
    for (int i = 0; i < nx; i++)
      if (matchx_[i] < 0) { zx[i] = 1; q.push_back(i); }
    for (size_t h = 0; h < q.size(); h++) {
      const int i = q[h], xv = order[x0 + i];
      for (int k = A.ptr[xv]; k < A.ptr[xv + 1]; k++) {
        const int j = ynbr(xv, k);
        if (j < 0 || zy[j]) continue;
        zy[j] = 1;
        const int i2 = matchy_[j];
        if (i2 >= 0 && !zx[i2]) { zx[i2] = 1; q.push_back(i2); }
      }
    }
    cover.clear();
    for (int i = 0; i < nx; i++)
      if (!zx[i]) cover.push_back(order[x0 + i]);
    for (int j = 0; j < ny; j++)
      if (zy[j]) cover.push_back(order[x1 + j]);
    for (int i = 0; i < ny; i++) lidx_[order[x1 + i]] = -1;
  }

  // dot product with a summation order that does not depend on the thread count: fixed chunks, then their sums in order
  static double pdot(const double *a, const double *b, int m) {
    constexpr int CH = 4096;
    const int nch = (m + CH - 1) / CH;
    std::vector<double> part(nch);
#pragma omp parallel for if (nch > 1)
    for (int c = 0; c < nch; c++) {
      double s = 0;
      for (int i = c * CH, e = std::min(m, (c + 1) * CH); i < e; i++) s += a[i] * b[i];
      part[c] = s;
    }
    double s = 0;
    for (double v : part) s += v;
    return s;
  }

  // Approximate Fiedler vector of the subgraph induced by `verts` (all with stamp == cur): Lanczos on its Laplacian in
  // the complement of the constant vector, full reorthogonalisation, Ritz vector of the smallest Ritz value.
  // ys: one vector per depth in `depths` (Ritz vectors of the same Lanczos run: different approximations of the
  // Fiedler vector cut the graph differently, and the smallest separator among them is kept).
  void fiedler(const std::vector<int> &verts, const std::vector<int> &depths, std::vector<std::vector<double>> &ys) {
    const int lanczos_steps = *std::max_element(depths.begin(), depths.end());
    const int m = (int)verts.size(), kmax = std::min(m - 1, lanczos_steps);
    if ((int)lidx_.size() < A.n) lidx_.assign(A.n, -1);
    for (int i = 0; i < m; i++) lidx_[verts[i]] = i;
    auto apply = [&](const std::vector<double> &x, std::vector<double> &out) {   // out = L x
#pragma omp parallel for if (m > 4096)
      for (int i = 0; i < m; i++) {
        const int v = verts[i];
        double s = 0;
        int deg = 0;
        for (int k = A.ptr[v]; k < A.ptr[v + 1]; k++) {
          const int w = A.col[k];
          if (stamp[w] != cur) continue;
          s += x[lidx_[w]];
          deg++;
        }
        out[i] = deg * x[i] - s;
      }
    };
    auto deflate = [&](std::vector<double> &x) {
      double mean = 0;
      for (double v : x) mean += v;
      mean /= m;
      for (double &v : x) v -= mean;
    };
    // (the long loops run on the host threads: at the top of the tree m is the whole graph; sums are taken over
    // fixed chunks so that the result does not depend on the number of threads)
    const bool par = m > 4096;
    (void)par;
    std::vector<double> Q((size_t)kmax * m);   // Lanczos vectors, row k = q_k
    std::vector<double> alpha, beta, q(m), w(m), coef(kmax);
    for (int i = 0; i < m; i++) q[i] = std::sin(0.7 * i + 0.3) + 1e-3 * (i % 7);
    deflate(q);
    double nrm = 0;
    for (double v : q) nrm += v * v;
    nrm = std::sqrt(nrm);
    for (double &v : q) v /= nrm;
    for (int k = 0; k < kmax; k++) {
      std::copy(q.begin(), q.end(), Q.begin() + (size_t)k * m);
      apply(q, w);
      alpha.push_back(pdot(w.data(), q.data(), m));
      deflate(w);
      for (int pass = 0; pass < 2; pass++) {   // classical Gram-Schmidt, twice
        const int nk = k + 1;
#pragma omp parallel for if (par)
        for (int j = 0; j < nk; j++) {
          const double *qq = &Q[(size_t)j * m];
          double c = 0;
          for (int i = 0; i < m; i++) c += w[i] * qq[i];
          coef[j] = c;
        }
#pragma omp parallel for if (par)
        for (int i = 0; i < m; i++) {
          double acc = w[i];
          for (int j = 0; j < nk; j++) acc -= coef[j] * Q[(size_t)j * m + i];
          w[i] = acc;
        }
      }
      const double b = std::sqrt(pdot(w.data(), w.data(), m));
      if (b < 1e-10) break;
      beta.push_back(b);
      for (int i = 0; i < m; i++) q[i] = w[i] / b;
    }
    ys.clear();
    for (int depth : depths) {
    const int kk = std::min((int)alpha.size(), depth);
    if (!ys.empty() && kk == (int)alpha.size() && depth > (int)alpha.size()) break;   // Lanczos ended early: nothing new
    // smallest eigenvalue of the tridiagonal (alpha, beta) by bisection on the Sturm count
    double lo = 0, hi = 0;
    for (int i = 0; i < kk; i++) hi = std::max(hi, alpha[i] + (i > 0 ? beta[i - 1] : 0) + (i < kk - 1 && i < (int)beta.size() ? beta[i] : 0));
    lo = -1e-9;
    for (int it = 0; it < 100; it++) {
      const double mid = 0.5 * (lo + hi);
      int neg = 0;
      double dd = 1.0;
      for (int i = 0; i < kk; i++) {
        dd = alpha[i] - mid - (i > 0 ? beta[i - 1] * beta[i - 1] / dd : 0.0);
        if (dd == 0.0) dd = 1e-300;
        if (dd < 0) neg++;
      }
      if (neg >= 1) hi = mid; else lo = mid;
    }
    const double theta = 0.5 * (lo + hi);
    // its eigenvector by inverse iteration on the tridiagonal (Thomas algorithm, slightly shifted)
    std::vector<double> s(kk, 1.0), c(kk), dprime(kk);
    for (int rep = 0; rep < 3; rep++) {
      const double sh = theta - 1e-7 * (std::fabs(theta) + 1e-3);
      for (int i = 0; i < kk; i++) {
        const double diag = alpha[i] - sh, sub = i > 0 ? beta[i - 1] : 0.0;
        const double denom = i > 0 ? diag - sub * c[i - 1] : diag;
        c[i] = (i < kk - 1 ? beta[i] : 0.0) / denom;
        dprime[i] = (s[i] - (i > 0 ? sub * dprime[i - 1] : 0.0)) / denom;
      }
      s[kk - 1] = dprime[kk - 1];
      for (int i = kk - 2; i >= 0; i--) s[i] = dprime[i] - c[i] * s[i + 1];
      double n2 = 0;
      for (double v : s) n2 += v * v;
      n2 = std::sqrt(n2);
      for (double &v : s) v /= n2;
    }
    std::vector<double> y(m, 0.0);
    for (int k = 0; k < kk; k++)
      for (int i = 0; i < m; i++) y[i] += s[k] * Q[(size_t)k * m + i];
    ys.push_back(std::move(y));
    }
    for (int i = 0; i < m; i++) lidx_[verts[i]] = -1;
  }

  void dissect(std::vector<int> verts, int parent) {
    // split into connected components
    cur += 2;
    for (int v : verts) stamp[v] = cur;
    std::vector<std::vector<int>> comps;
    std::vector<int> order;
    const int mark = cur;
    for (int v : verts) {
      if (stamp[v] != mark) continue;
      bfs(v, order);
      for (int w : order) stamp[w] = -1;  // remove from the set
      comps.push_back(order);
    }
    for (auto &c : comps) dissect_connected(std::move(c), parent);
  }

  void dissect_connected(std::vector<int> verts, int parent) {
    if ((int)verts.size() <= leaf) {
      make(std::move(verts), parent);
      return;
    }
    cur += 2;
    for (int v : verts) stamp[v] = cur;
    std::vector<int> order;
    int root = verts[0];
    int nlev = bfs(root, order);
    for (int it = 0; it < 3; it++) {  // pseudo-peripheral vertex
      int far = order.back();
      int nl2 = bfs(far, order);
      if (nl2 <= nlev) { nlev = nl2; break; }
      nlev = nl2;
    }
    if (nlev < 3) {
      make(std::move(verts), parent);
      return;
    }
    // A whole level separates, but usually a smaller set does: the cut between levels l and l+1 is a bipartite
    // graph, and a minimum vertex cover of it (Koenig: from a maximum matching) separates {levels < l} + (level l
    // outside the cover) from (level l+1 outside the cover) + {levels > l+1}.  It is never larger than either level.
    // Every cut that leaves both sides at least `window` of the vertices is tried, in the level structures of both
    // ends of the pseudo-diameter; the smallest cover wins.  window = 0.45 (measured: 0.3 gives 2 % fewer entries but
    // 20 % more tree levels, i.e. slower solves; more than two roots change nothing).
    static const double window = [] { const char *e = getenv("DPGO_ND_WINDOW"); return e ? atof(e) : 0.45; }();
    static const int nroots = [] { const char *e = getenv("DPGO_ND_ROOTS"); return e ? atoi(e) : 2; }();
    const double total = (double)order.size();
    std::vector<int> sep, lo, hi, best_cover, lp;
    size_t best_size = (size_t)-1;
    for (int trial = 0; trial < nroots; trial++) {
      if (trial > 0) nlev = bfs(order.back(), order);   // the level structure seen from the other end
      // BFS order is sorted by level: level l is order[lp[l] .. lp[l + 1])
      lp.assign(nlev + 1, 0);
      for (int v : order) lp[level[v] + 1]++;
      for (int l = 0; l < nlev; l++) lp[l + 1] += lp[l];
      int best_l = -1;
      for (int l = 0; l + 1 < nlev; l++) {
        const double below = lp[l + 1] / total, above = (total - lp[l + 1]) / total;   // if the cover were empty
        if (below < window || above < window) continue;
        if (best_size != (size_t)-1 && (size_t)std::min(lp[l + 1] - lp[l], lp[l + 2] - lp[l + 1]) >= 2 * best_size) continue;
        std::vector<int> cover;
        min_cover(order, lp[l], lp[l + 1], lp[l + 2], cover);
        if (cover.size() < best_size) { best_size = cover.size(); best_l = l; best_cover.swap(cover); }
      }
      if (best_l >= 0) {   // this level structure gave the best separator so far: materialise the three sets
        cur += 2;
        for (int v : best_cover) stamp[v] = cur;
        sep.clear(); lo.clear(); hi.clear();
        for (int v : order) {
          if (stamp[v] == cur) sep.push_back(v);
          else if (level[v] <= best_l) lo.push_back(v);
          else hi.push_back(v);
        }
        for (int v : order) stamp[v] = cur;   // the whole vertex set is current again for the next BFS
      }
    }
    // Spectral candidate (Pothen, Simon, Liou 1990): split along the Fiedler vector of the subgraph, again with a
    // minimum vertex cover of the cut as the separator.  On lattice-like graphs it finds the flat cross-sections
    // that level sets grown from a corner (diagonal planes) miss.
    static const int spectral = [] { const char *e = getenv("DPGO_ND_SPECTRAL"); return e ? atoi(e) : 1; }();
    if (spectral && order.size() >= 64) {
      static const std::vector<int> depths = [] {
        std::vector<int> d;
        const char *e = getenv("DPGO_ND_LANCZOS");
        std::string str = e ? e : "40,60,90,120";
        for (size_t p = 0; p < str.size();) {
          size_t c = str.find(',', p);
          if (c == std::string::npos) c = str.size();
          d.push_back(atoi(str.substr(p, c - p).c_str()));
          p = c + 1;
        }
        return d;
      }();
      std::vector<std::vector<double>> fvs;
      fiedler(order, depths, fvs);
      std::vector<int> perm(order.size());
      for (const std::vector<double> &fv : fvs)
      for (double q : {0.5, 0.49, 0.51, 0.48, 0.52, 0.47, 0.53, 0.46, 0.54, 0.45, 0.55}) {
        if (q == 0.5) {
          for (size_t i = 0; i < perm.size(); i++) perm[i] = (int)i;
          std::sort(perm.begin(), perm.end(), [&](int a, int b) { return fv[a] < fv[b] || (fv[a] == fv[b] && a < b); });
        }
        if (q < window || 1.0 - q < window) continue;
        const size_t cut = (size_t)(q * perm.size());
        // side 0 / 1 kept in level[] (the BFS levels are not needed any more)
        for (size_t i = 0; i < perm.size(); i++) level[order[perm[i]]] = i < cut ? 0 : 1;
        std::vector<int> bd, by;
        for (int v : order) {
          bool boundary = false;
          for (int k = A.ptr[v]; k < A.ptr[v + 1] && !boundary; k++) {
            const int w = A.col[k];
            boundary = stamp[w] == cur && level[w] != level[v];
          }
          if (boundary) (level[v] == 0 ? bd : by).push_back(v);
        }
        if (bd.empty() || by.empty()) continue;
        const int nx = (int)bd.size();
        bd.insert(bd.end(), by.begin(), by.end());
        std::vector<int> cover;
        min_cover(bd, 0, nx, (int)bd.size(), cover);
        if (cover.size() < best_size) {
          best_size = cover.size();
          cur += 2;
          for (int v : cover) stamp[v] = cur;
          sep.clear(); lo.clear(); hi.clear();
          for (int v : order) {
            if (stamp[v] == cur) sep.push_back(v);
            else if (level[v] == 0) lo.push_back(v);
            else hi.push_back(v);
          }
          for (int v : order) stamp[v] = cur;
        }
      }
    }
    if (best_size == (size_t)-1) {
      // no balanced cut between two levels (few, fat levels): fall back to the level nearest the middle
      int half = 1;
      for (int l = 0; l < nlev; l++)
        if (lp[l] <= total / 2 && lp[l + 1] >= total / 2) half = l;
      const int best = std::min(std::max(half, 1), nlev - 2);
      for (int v : order) {
        if (level[v] == best) sep.push_back(v);
        else if (level[v] < best) lo.push_back(v);
        else hi.push_back(v);
      }
    }
    int id = make(std::move(sep), parent);
    dissect(std::move(lo), id);
    dissect(std::move(hi), id);
  }
};

}  // namespace

// tree: a nested-dissection tree of A computed earlier (the dissection does not depend on `collapse`); if *tree is
// empty it is computed here and stored there.
static double setup_lap(double &t0, const char *what) {   // DPGO_SETUP_TIMING=1: phase times on stderr
  const double t = omp_get_wtime();
  if (getenv("DPGO_SETUP_TIMING")) fprintf(stderr, "[setup]   spd: %-37s %8.3f s\n", what, t - t0);
  t0 = t;
  return t;
}

static int spd_factor_impl(const CsrMatrix &A, SpdFactor &F, int leaf, int collapse, bool symbolic_only,
                           std::vector<TreeNode> *tree, int block, bool keep_device) {
  const int n = A.n;
  double t_lap = omp_get_wtime();
  omp_set_num_threads(host_threads());
  spd_release_device(F);
  const bool keep_numeric = F.keep_numeric;
  spd_release_numeric(F);   // (a kept numeric context belongs to the pattern that is about to be replaced)
  F = SpdFactor();
  F.n = n;
  F.keep_device = keep_device;
  F.keep_numeric = keep_numeric;
  // adjacency without the diagonal
  CsrMatrix adj;
  adj.n = n;
  adj.ptr.assign(n + 1, 0);
  for (int i = 0; i < n; i++)
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; k++)
      if (A.col[k] != i) adj.ptr[i + 1]++;
  for (int i = 0; i < n; i++) adj.ptr[i + 1] += adj.ptr[i];
  adj.col.resize(adj.ptr[n]);
  {
    std::vector<int> pos(adj.ptr.begin(), adj.ptr.end() - 1);
    for (int i = 0; i < n; i++)
      for (int k = A.ptr[i]; k < A.ptr[i + 1]; k++)
        if (A.col[k] != i) adj.col[pos[i]++] = A.col[k];
  }
  Dissector D(adj, leaf);
  if (tree && !tree->empty()) D.nodes = *tree;
  else {
    // block > 1 (G_RR: the d rows of a pose have the same neighbours): the quotient graph -- one vertex per block of
    // `block` consecutive unknowns -- is dissected instead and every tree node expanded afterwards.  Same separators
    // as on the scalar graph with indistinguishable vertices kept together, at 1 / block^2 of the work.
    CsrMatrix quot;
    const bool use_q = block > 1 && n % block == 0;
    if (use_q) {
      const int nq = n / block;
      quot.n = nq;
      quot.ptr.assign(nq + 1, 0);
      std::vector<int> mark(nq, -1);
      for (int q = 0; q < nq; q++) {
        for (int i = q * block; i < (q + 1) * block; i++)
          for (int k = adj.ptr[i]; k < adj.ptr[i + 1]; k++) {
            const int w = adj.col[k] / block;
            if (w != q && mark[w] != q) { mark[w] = q; quot.col.push_back(w); }
          }
        std::sort(quot.col.begin() + quot.ptr[q], quot.col.end());
        quot.ptr[q + 1] = (int)quot.col.size();
      }
    }
    const CsrMatrix &gq = use_q ? quot : adj;
    const int gn = gq.n, gleaf = use_q ? std::max(1, leaf / block) : leaf;
    Dissector Dq(gq, gleaf);
    {
      Dissector &D = Dq;   // (the code below fills D.nodes)
      // connected components (the nodes of a group are disconnected from each other) are dissected independently,
      // one host thread each; the result does not depend on the number of threads
      std::vector<std::vector<int>> comps;
      {
        std::vector<int> comp(gn, -1), stack;
        for (int r = 0; r < gn; r++) {
          if (comp[r] >= 0) continue;
          const int c = (int)comps.size();
          comps.emplace_back();
          comp[r] = c;
          stack.assign(1, r);
          while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            comps[c].push_back(v);
            for (int k = gq.ptr[v]; k < gq.ptr[v + 1]; k++)
              if (comp[gq.col[k]] < 0) { comp[gq.col[k]] = c; stack.push_back(gq.col[k]); }
          }
          std::sort(comps[c].begin(), comps[c].end());
        }
      }
      // (components are started largest first; tiny ones share a dissector)
      std::vector<int> big;
      std::vector<int> small_verts;
      for (int c = 0; c < (int)comps.size(); c++) {
        if ((int)comps[c].size() >= 2048) big.push_back(c);
        else small_verts.insert(small_verts.end(), comps[c].begin(), comps[c].end());
      }
      std::vector<std::vector<TreeNode>> parts(big.size());
#pragma omp parallel for schedule(dynamic, 1) if (big.size() > 1)
      for (int b = 0; b < (int)big.size(); b++) {
        Dissector Dc(gq, gleaf);
        Dc.dissect(comps[big[b]], -1);
        parts[b] = std::move(Dc.nodes);
      }
      for (auto &p : parts) {
        const int off = (int)D.nodes.size();
        for (TreeNode &t : p) {
          if (t.parent >= 0) t.parent += off;
          for (int &ch : t.children) ch += off;
          D.nodes.push_back(std::move(t));
        }
      }
      if (!small_verts.empty()) {
        std::sort(small_verts.begin(), small_verts.end());
        Dissector Ds(gq, gleaf);
        Ds.dissect(std::move(small_verts), -1);
        const int off = (int)D.nodes.size();
        for (TreeNode &t : Ds.nodes) {
          if (t.parent >= 0) t.parent += off;
          for (int &ch : t.children) ch += off;
          D.nodes.push_back(std::move(t));
        }
      }
    }
    D.nodes = std::move(Dq.nodes);
    if (use_q)
      for (TreeNode &t : D.nodes) {
        std::vector<int> vs;
        vs.reserve(t.verts.size() * block);
        for (int q : t.verts)
          for (int b = 0; b < block; b++) vs.push_back(q * block + b);
        t.verts.swap(vs);
      }
    if (tree) *tree = D.nodes;
    setup_lap(t_lap, "nested dissection");
  }
  // Level collapsing: absorb every tree node whose depth is not a multiple of `collapse` into its
  // nearest ancestor whose depth is.  The merged front factors the absorbed separators together as one
  // dense block (a few structural zeros become explicit), which divides the number of tree levels --
  // i.e. of dependent kernel launches per solve -- by `collapse`.  On the GPU a level costs ~10 us of
  // latency regardless of its size, so fewer and fatter levels win until the extra bytes dominate.
  if (collapse > 1) {
    const int n0 = (int)D.nodes.size();
    std::vector<int> depth(n0, 0), keeper(n0, 0);
    // parents are created before their children, so a forward pass sees parents first
    for (int t = 0; t < n0; t++) {
      const int p = D.nodes[t].parent;
      depth[t] = p < 0 ? 0 : depth[p] + 1;
      keeper[t] = (depth[t] % collapse == 0) ? t : keeper[p];
    }
    std::vector<TreeNode> merged;
    std::vector<int> new_id(n0, -1);
    for (int t = 0; t < n0; t++)
      if (keeper[t] == t) {
        new_id[t] = (int)merged.size();
        merged.push_back(TreeNode());
      }
    for (int t = 0; t < n0; t++) {
      TreeNode &dst = merged[new_id[keeper[t]]];
      dst.verts.insert(dst.verts.end(), D.nodes[t].verts.begin(), D.nodes[t].verts.end());
      if (keeper[t] == t) {
        const int p = D.nodes[t].parent;
        dst.parent = p < 0 ? -1 : new_id[keeper[p]];
      }
    }
    for (int k = 0; k < (int)merged.size(); k++)
      if (merged[k].parent >= 0) merged[merged[k].parent].children.push_back(k);
    D.nodes.swap(merged);
  }
  const int nt = (int)D.nodes.size();
  // post-order
  std::vector<int> post;  // tree node ids in post-order
  post.reserve(nt);
  {
    std::vector<std::pair<int, int>> st;
    for (int r = 0; r < nt; r++) {
      if (D.nodes[r].parent != -1) continue;
      st.push_back({r, 0});
      while (!st.empty()) {
        auto &top = st.back();
        if (top.second < (int)D.nodes[top.first].children.size()) {
          int c = D.nodes[top.first].children[top.second++];
          st.push_back({c, 0});
        } else {
          post.push_back(top.first);
          st.pop_back();
        }
      }
    }
  }
  std::vector<int> fid_of_tree(nt);
  for (int f = 0; f < nt; f++) fid_of_tree[post[f]] = f;
  F.nfronts = nt;
  F.w.resize(nt);
  F.u.assign(nt, 0);
  F.parent.assign(nt, -1);
  std::vector<std::vector<int>> children(nt);
  std::vector<int> front_of(n), elim_pos(n);
  F.piv_ptr.assign(nt + 1, 0);
  for (int f = 0; f < nt; f++) {
    const TreeNode &t = D.nodes[post[f]];
    F.w[f] = (int)t.verts.size();
    F.piv_ptr[f + 1] = F.piv_ptr[f] + F.w[f];
    if (t.parent >= 0) {
      F.parent[f] = fid_of_tree[t.parent];
      children[F.parent[f]].push_back(f);
    }
  }
  F.piv_idx.resize(n);
  for (int f = 0; f < nt; f++) {
    std::vector<int> vs = D.nodes[post[f]].verts;
    std::sort(vs.begin(), vs.end());
    for (int k = 0; k < (int)vs.size(); k++) {
      F.piv_idx[F.piv_ptr[f] + k] = vs[k];
      front_of[vs[k]] = f;
      elim_pos[vs[k]] = F.piv_ptr[f] + k;
    }
  }
  // symbolic: update rows of each front, sorted by elimination position
  std::vector<std::vector<int>> upd(nt);
  {
    std::vector<int> mark(n, -1);
    for (int f = 0; f < nt; f++) {
      std::vector<int> &us = upd[f];
      for (int k = F.piv_ptr[f]; k < F.piv_ptr[f + 1]; k++) {
        int v = F.piv_idx[k];
        for (int e = adj.ptr[v]; e < adj.ptr[v + 1]; e++) {
          int w = adj.col[e];
          if (front_of[w] > f && mark[w] != f) { mark[w] = f; us.push_back(w); }
        }
      }
      for (int c : children[f])
        for (int w : upd[c])
          if (front_of[w] > f && mark[w] != f) { mark[w] = f; us.push_back(w); }
      std::sort(us.begin(), us.end(), [&](int a, int b) { return elim_pos[a] < elim_pos[b]; });
      F.u[f] = (int)us.size();
    }
  }
  F.upd_ptr.assign(nt + 1, 0);
  F.w_off.assign(nt + 1, 0);
  F.wt_off.assign(nt + 1, 0);
  F.ldw.assign(nt, 0);
  F.ldm.assign(nt, 0);
  F.entries = 0;
  F.pos_off.assign(nt + 1, 0);
  F.ubuf_off.assign(nt + 1, 0);
  for (int f = 0; f < nt; f++) {
    F.upd_ptr[f + 1] = F.upd_ptr[f] + F.u[f];
    // leading dimensions padded to 16 doubles (128 B): every 64-lane tile of a row starts on a cache line
    F.ldw[f] = (F.w[f] + 15) & ~15;
    F.ldm[f] = (F.w[f] + F.u[f] + 15) & ~15;
    F.w_off[f + 1] = F.w_off[f] + (int64_t)(F.w[f] + F.u[f]) * F.ldw[f];
    F.wt_off[f + 1] = F.wt_off[f] + (int64_t)F.w[f] * F.ldm[f];
    F.entries += (int64_t)(F.w[f] + F.u[f]) * F.w[f];
    F.pos_off[f + 1] = F.pos_off[f] + F.w[f] + F.u[f];
    F.ubuf_off[f + 1] = F.ubuf_off[f] + F.u[f];
    F.max_front = std::max(F.max_front, F.w[f] + F.u[f]);
  }
  F.total_pos = F.pos_off[nt];
  F.total_upd = F.ubuf_off[nt];
  F.upd_idx.resize(F.upd_ptr[nt]);
  for (int f = 0; f < nt; f++) std::copy(upd[f].begin(), upd[f].end(), F.upd_idx.begin() + F.upd_ptr[f]);
  if (symbolic_only) {
    F.height.assign(nt, 0);
    for (int f = 0; f < nt; f++)
      if (F.parent[f] >= 0) F.height[F.parent[f]] = std::max(F.height[F.parent[f]], F.height[f] + 1);
    int mh = 0;
    for (int f = 0; f < nt; f++) mh = std::max(mh, F.height[f]);
    F.by_height.assign(mh + 1, {});
    return 0;
  }
  setup_lap(t_lap, "symbolic");
  bool device_numeric = false;
#ifndef DPGO_NO_DEVICE
  {
    int ndev = 0;
    device_numeric = !getenv("DPGO_SPD_HOST_FACTOR") && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
  }
#endif
  if (!(device_numeric && F.keep_device)) {   // (the device numeric phase allocates the host copies it needs itself)
    F.W.assign(F.w_off[nt], 0.0);
    F.WT.assign(F.wt_off[nt], 0.0);
  }
  setup_lap(t_lap, "host W / WT allocation");

  // numeric multifrontal factorisation, level by level: fronts of one tree height are independent.
  // Small fronts are spread over threads; big fronts are factored one at a time with threads inside.
  std::vector<int> height(nt, 0);
  for (int f = 0; f < nt; f++)
    if (F.parent[f] >= 0) height[F.parent[f]] = std::max(height[F.parent[f]], height[f] + 1);
  int maxh_n = 0;
  for (int f = 0; f < nt; f++) maxh_n = std::max(maxh_n, height[f]);
  std::vector<std::vector<int>> lvl_fronts(maxh_n + 1);
  for (int f = 0; f < nt; f++) lvl_fronts[height[f]].push_back(f);
  std::vector<std::vector<double>> Umat(nt);
  std::vector<std::vector<int>> asm_lists(F.total_pos);
  int fail = 0;
  const int BIG = 384;

  auto factor_front = [&](int f, std::vector<int> &loc, std::vector<double> &Fm, std::vector<double> &Linv,
                          bool inner_par) {
    const int w = F.w[f], u = F.u[f], m = w + u;
    if (w == 0) {
      if (u > 0) fail = 1;
      return;
    }
    const int *piv = &F.piv_idx[F.piv_ptr[f]];
    const int *up = u ? &F.upd_idx[F.upd_ptr[f]] : nullptr;
    for (int k = 0; k < w; k++) loc[piv[k]] = k;
    for (int k = 0; k < u; k++) loc[up[k]] = w + k;
    Fm.assign((size_t)m * m, 0.0);
    for (int k = 0; k < w; k++) {
      int v = piv[k];
      for (int e = A.ptr[v]; e < A.ptr[v + 1]; e++) {
        int l = loc[A.col[e]];
        if (l >= k) Fm[(size_t)l * m + k] += A.val[e];
      }
    }
    for (int c : children[f]) {
      const int uc = F.u[c];
      const int *upc = uc ? &F.upd_idx[F.upd_ptr[c]] : nullptr;
      const std::vector<double> &Uc = Umat[c];
      for (int a = 0; a < uc; a++) {
        int la = loc[upc[a]];
        for (int b = 0; b <= a; b++) {
          int lb = loc[upc[b]];
          int r = std::max(la, lb), cc = std::min(la, lb);
          Fm[(size_t)r * m + cc] += Uc[(size_t)a * uc + b];
        }
      }
      std::vector<double>().swap(Umat[c]);
    }
    // blocked right-looking partial Cholesky of the first w columns (lower triangle, row-major)
    const int NB = 48;
    for (int kb = 0; kb < w && !fail; kb += NB) {
      const int ke = std::min(kb + NB, w);
      for (int k = kb; k < ke; k++) {   // diagonal block, unblocked
        double dkk = Fm[(size_t)k * m + k];
        if (!(dkk > 0.0)) {
          fprintf(stderr, "[dpgo_amd] ERROR: spd_factor: non-positive pivot %g (front %d, col %d)\n", dkk, f, k);
          fail = 1;
          break;
        }
        const double lkk = std::sqrt(dkk), inv = 1.0 / lkk;
        Fm[(size_t)k * m + k] = lkk;
        for (int i = k + 1; i < ke; i++) Fm[(size_t)i * m + k] *= inv;
        for (int i = k + 1; i < ke; i++) {
          const double lik = Fm[(size_t)i * m + k];
          for (int j = k + 1; j <= i; j++) Fm[(size_t)i * m + j] -= lik * Fm[(size_t)j * m + k];
        }
      }
      if (fail) break;
      // panel: rows below the block, L[i, kb:ke] = F[i, kb:ke] L_kk^-T
#pragma omp parallel for schedule(static) if (inner_par)
      for (int i = ke; i < m; i++) {
        double *row = &Fm[(size_t)i * m];
        for (int k = kb; k < ke; k++) {
          double s = row[k];
          const double *lk = &Fm[(size_t)k * m];
          for (int q = kb; q < k; q++) s -= row[q] * lk[q];
          row[k] = s / lk[k];
        }
      }
      // trailing update: F[i, j] -= L[i, kb:ke] . L[j, kb:ke] for ke <= j <= i
      const int nbk = ke - kb;
#pragma omp parallel for schedule(dynamic, 16) if (inner_par)
      for (int i = ke; i < m; i++) {
        double *row = &Fm[(size_t)i * m];
        const double *li = row + kb;
        for (int j = ke; j <= i; j++) {
          const double *lj = &Fm[(size_t)j * m + kb];
          double s = 0;
          for (int q = 0; q < nbk; q++) s += li[q] * lj[q];
          row[j] -= s;
        }
      }
    }
    if (fail) return;
    if (u) {   // Schur complement for the parent
      Umat[f].resize((size_t)u * u);
      for (int a = 0; a < u; a++)
        for (int b = 0; b <= a; b++) Umat[f][(size_t)a * u + b] = Fm[(size_t)(w + a) * m + (w + b)];
    }
    // Linv = L11^-1 by forward substitution on the identity, one row at a time:
    //   Linv[i, :] = (e_i - sum_{k<i} L[i,k] Linv[k, :]) / L[i,i]; columns are independent
    Linv.assign((size_t)w * w, 0.0);
    const int CB = 64;
    const int ncb = (w + CB - 1) / CB;
#pragma omp parallel for schedule(dynamic, 1) if (inner_par)
    for (int cb = 0; cb < ncb; cb++) {
      const int j0 = cb * CB, j1 = std::min(j0 + CB, w);
      for (int i = j0; i < w; i++) {
        double *ri = &Linv[(size_t)i * w];
        const double *Li = &Fm[(size_t)i * m];
        const int je = std::min(j1, i + 1);
        if (i < j1) ri[i] = 1.0;
        for (int k = j0; k < i; k++) {
          const double lik = Li[k];
          if (lik == 0.0) continue;
          const double *rk = &Linv[(size_t)k * w];
          const int jk = std::min(je, k + 1);
          for (int j = j0; j < jk; j++) ri[j] -= lik * rk[j];
        }
        const double inv = 1.0 / Li[i];
        for (int j = j0; j < je; j++) ri[j] *= inv;
      }
    }
    double *Wf = &F.W[F.w_off[f]];
    double *WTf = &F.WT[F.wt_off[f]];
    const int ldw = F.ldw[f], ldm = F.ldm[f];
    for (int i = 0; i < w; i++)
      for (int j = 0; j <= i; j++) Wf[(size_t)i * ldw + j] = Linv[(size_t)i * w + j];
    // W bottom = -L21 Linv, row-oriented: out[a, :] = -sum_k L21[a,k] Linv[k, 0..k]
#pragma omp parallel for schedule(dynamic, 8) if (inner_par)
    for (int a = 0; a < u; a++) {
      const double *l21 = &Fm[(size_t)(w + a) * m];
      double *out = &Wf[(size_t)(w + a) * ldw];
      for (int j = 0; j < w; j++) out[j] = 0.0;
      for (int k = 0; k < w; k++) {
        const double l = l21[k];
        if (l == 0.0) continue;
        const double *rk = &Linv[(size_t)k * w];
        for (int j = 0; j <= k; j++) out[j] -= l * rk[j];
      }
    }
#pragma omp parallel for schedule(static) if (inner_par)
    for (int k = 0; k < w; k++)
      for (int p = 0; p < m; p++) WTf[(size_t)k * ldm + p] = Wf[(size_t)p * ldw + k];
    for (int k = 0; k < w; k++) loc[piv[k]] = -1;
    for (int k = 0; k < u; k++) loc[up[k]] = -1;
  };

  // assembly lists of the solve (which update-buffer rows feed every front position): symbolic
  {
    std::vector<int> loc(n, -1);
    for (int f = 0; f < nt; f++) {
      const int w = F.w[f], u = F.u[f];
      const int *piv = &F.piv_idx[F.piv_ptr[f]];
      const int *up = u ? &F.upd_idx[F.upd_ptr[f]] : nullptr;
      for (int k = 0; k < w; k++) loc[piv[k]] = k;
      for (int k = 0; k < u; k++) loc[up[k]] = w + k;
      for (int c : children[f]) {
        const int uc = F.u[c];
        const int *upc = uc ? &F.upd_idx[F.upd_ptr[c]] : nullptr;
        for (int a = 0; a < uc; a++) asm_lists[F.pos_off[f] + loc[upc[a]]].push_back(F.ubuf_off[c] + a);
      }
      for (int k = 0; k < w; k++) loc[piv[k]] = -1;
      for (int k = 0; k < u; k++) loc[up[k]] = -1;
    }
  }
  setup_lap(t_lap, "assembly lists");
  // numeric phase: on the GPU when there is one (spd_dev.hip: front elimination with v_mfma_f64_16x16x4_f64), else --
  // or with DPGO_SPD_HOST_FACTOR=1 -- the host loop below
  bool on_device = false;
#ifndef DPGO_NO_DEVICE
  {
    int ndev = 0;
    if (!getenv("DPGO_SPD_HOST_FACTOR") && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
      double flops = 0, ms = 0;
      const bool report = getenv("DPGO_SPD_DUMP") != nullptr;
      if (spd_factor_numeric_device(A, F, children, report ? &flops : nullptr, report ? &ms : nullptr) != 0) return -1;
      if (report)
        fprintf(stderr, "[spd] device factorisation: %.2f GFLOP in the MFMA tile kernel, %.2f ms there = %.1f TFLOP/s fp64\n",
                flops * 1e-9, ms, flops / (ms * 1e-3) * 1e-12);
      on_device = true;
    }
  }
#endif
  if (on_device) setup_lap(t_lap, "numeric (device, incl. copies)");
  for (int h = 0; h <= maxh_n && !fail && !on_device; h++) {
    std::vector<int> small, big;
    for (int f : lvl_fronts[h]) (F.w[f] + F.u[f] >= BIG ? big : small).push_back(f);
#pragma omp parallel
    {
      std::vector<int> loc(n, -1);
      std::vector<double> Fm, Linv;
#pragma omp for schedule(dynamic, 1)
      for (int i = 0; i < (int)small.size(); i++) factor_front(small[i], loc, Fm, Linv, false);
    }
    if (!big.empty()) {
      std::vector<int> loc(n, -1);
      std::vector<double> Fm, Linv;
      for (int f : big) factor_front(f, loc, Fm, Linv, true);
    }
  }
  if (fail) return -1;
  F.children = children;
  F.asm_ptr.assign(F.total_pos + 1, 0);
  for (int p = 0; p < F.total_pos; p++) F.asm_ptr[p + 1] = F.asm_ptr[p] + (int)asm_lists[p].size();
  F.asm_src.resize(F.asm_ptr[F.total_pos]);
  for (int p = 0; p < F.total_pos; p++)
    std::copy(asm_lists[p].begin(), asm_lists[p].end(), F.asm_src.begin() + F.asm_ptr[p]);
  // levels
  F.height.assign(nt, 0);
  F.depth.assign(nt, 0);
  for (int f = 0; f < nt; f++)
    if (F.parent[f] >= 0) F.height[F.parent[f]] = std::max(F.height[F.parent[f]], F.height[f] + 1);
  for (int f = nt - 1; f >= 0; f--)
    if (F.parent[f] >= 0) F.depth[f] = F.depth[F.parent[f]] + 1;
  int maxh = 0, maxd = 0;
  for (int f = 0; f < nt; f++) { maxh = std::max(maxh, F.height[f]); maxd = std::max(maxd, F.depth[f]); }
  F.by_height.assign(maxh + 1, {});
  F.by_depth.assign(maxd + 1, {});
  for (int f = 0; f < nt; f++) {
    if (F.w[f] == 0 && F.u[f] == 0) continue;
    F.by_height[F.height[f]].push_back(f);
    F.by_depth[F.depth[f]].push_back(f);
  }
  return 0;
}

int spd_refactor(const CsrMatrix &A, SpdFactor &F) {
#ifndef DPGO_NO_DEVICE
  int ndev = 0;
  if (F.n == A.n && (int)F.children.size() == F.nfronts && !getenv("DPGO_SPD_HOST_FACTOR") &&
      hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0)
    return spd_factor_numeric_device(A, F, F.children, nullptr, nullptr);
#endif
  // host path: the same ordering parameters as the first factorisation, so that the switch does not change the bits
  const int leaf = F.leaf, collapse = F.collapse, block = F.block;
  return spd_factor(A, F, leaf, collapse, block, F.keep_device);
}

int spd_factor(const CsrMatrix &A, SpdFactor &F, int leaf, int collapse, int block, bool keep_device) {
  std::vector<TreeNode> tree;   // dissected once, reused for every merge depth tried below
  if (const char *e = getenv("DPGO_SPD_COLLAPSE")) collapse = atoi(e);
  if (collapse <= 0) {
    // choose the number of merged levels from the measured cost of one sweep on MI355X: a level costs
    // ~12 us whatever it holds (launch, gather chain, drain) plus its bytes at ~7 TB/s.  (With the 3 TB/s of a whole
    // solve in place of the streaming rate the model undervalues a level: the deeper merge it now picks for G_RR + lambda I
    // is equal in the early regime and 4 % faster to the reference objective, where CG steps run on the few nodes that
    // still iterate and the per-level cost is all that is left.)
    double best = 1e300;
    int best_c = 1;
    // (up to five merged levels: a factor this small -- one node per GPU -- is all latency, and its G_tt is best served by
    // two levels, 0.419 -> 0.401 ms per iteration; the eight-node factors stay at three)
    for (int c = 1; c <= 5; c++) {
      SpdFactor S;
      if (spd_factor_impl(A, S, leaf, c, true, &tree, block, false) != 0) continue;
      const double t = 12e-6 * (double)S.by_height.size() + 8.0 * (double)S.entries / 7.0e12;
      if (t < best) { best = t; best_c = c; }
    }
    collapse = best_c;
  }
  const int rc = spd_factor_impl(A, F, leaf, collapse, false, &tree, block, keep_device);
  F.leaf = leaf;
  F.collapse = collapse;   // (the merge depth actually used)
  F.block = block;
  return rc;
}

void spd_solve_host(const SpdFactor &F, double *X, int nc) {
  std::vector<double> ubuf((size_t)F.total_upd * nc, 0.0), f, out;
  for (int s = 0; s < F.nfronts; s++) {  // post-order == forward order
    const int w = F.w[s], u = F.u[s], m = w + u;
    const int *piv = &F.piv_idx[F.piv_ptr[s]];
    f.assign((size_t)m * nc, 0.0);
    for (int p = 0; p < m; p++) {
      for (int c = 0; c < nc; c++) f[(size_t)p * nc + c] = p < w ? X[(size_t)piv[p] * nc + c] : 0.0;
      for (int a = F.asm_ptr[F.pos_off[s] + p]; a < F.asm_ptr[F.pos_off[s] + p + 1]; a++)
        for (int c = 0; c < nc; c++) f[(size_t)p * nc + c] += ubuf[(size_t)F.asm_src[a] * nc + c];
    }
    const double *Wf = &F.W[F.w_off[s]];
    for (int p = 0; p < m; p++)
      for (int c = 0; c < nc; c++) {
        double acc = p < w ? 0.0 : f[(size_t)p * nc + c];
        for (int k = 0; k < w; k++) acc += Wf[(size_t)p * F.ldw[s] + k] * f[(size_t)k * nc + c];
        if (p < w) X[(size_t)piv[p] * nc + c] = acc;
        else ubuf[(size_t)(F.ubuf_off[s] + p - w) * nc + c] = acc;
      }
  }
  for (int s = F.nfronts - 1; s >= 0; s--) {
    const int w = F.w[s], u = F.u[s], m = w + u;
    const int *piv = &F.piv_idx[F.piv_ptr[s]];
    const int *up = u ? &F.upd_idx[F.upd_ptr[s]] : nullptr;
    f.assign((size_t)m * nc, 0.0);
    for (int p = 0; p < m; p++)
      for (int c = 0; c < nc; c++) f[(size_t)p * nc + c] = X[(size_t)(p < w ? piv[p] : up[p - w]) * nc + c];
    const double *Wf = &F.W[F.w_off[s]];
    out.assign((size_t)w * nc, 0.0);
    for (int p = 0; p < m; p++)
      for (int k = 0; k < w; k++)
        for (int c = 0; c < nc; c++) out[(size_t)k * nc + c] += Wf[(size_t)p * F.ldw[s] + k] * f[(size_t)p * nc + c];
    for (int k = 0; k < w; k++)
      for (int c = 0; c < nc; c++) X[(size_t)piv[k] * nc + c] = out[(size_t)k * nc + c];
  }
}

}  // namespace dpgo
