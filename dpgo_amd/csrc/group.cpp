#include "group.h"

#include <limits>
#include "graph.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <set>
#include <thread>

namespace dpgo {

// A failed HIP call never aborts the host process (this is a shared library): it is logged and thrown as
// DeviceError, which every entry point of the C ABI (capi.cpp) turns into the reference's `return -1`.
#define HIP_CHECK(x)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "[dpgo_amd] ERROR: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      throw DeviceError(hipGetErrorString(e_));                                                   \
    }                                                                                             \
  } while (0)

static bool g_leak_device_buffers = false;
void dev_leak_buffers(bool on) { g_leak_device_buffers = on; }
template <class T>
void DevBuf<T>::release() {
  if (g_leak_device_buffers) { p = nullptr; n = 0; return; }
  if (p) (void)hipFree(p);
  p = nullptr;
  n = 0;
}
template <class T>
void DevBuf<T>::alloc(size_t count, bool zero) {
  release();
  n = count;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIP_CHECK(hipMalloc((void **)&p, bytes));
  if (zero) HIP_CHECK(hipMemset(p, 0, bytes));
}
template <class T>
void DevBuf<T>::upload(const std::vector<T> &h) {
  alloc(h.size(), h.empty());
  if (!h.empty()) HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
}
template <class T>
void DevBuf<T>::download(std::vector<T> &h) const {
  h.resize(n);
  if (n) HIP_CHECK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
}
template struct DevBuf<int>;
template struct DevBuf<double>;
template struct DevBuf<int64_t>;
template struct DevBuf<int4>;
template struct DevBuf<Seg>;
template struct DevBuf<SpdItem>;
template struct DevBuf<unsigned>;
template struct DevBuf<CgNode>;
template struct DevBuf<NodeBits>;
template struct DevBuf<PanelSrc>;
template struct DevBuf<InterInc>;
template struct DevBuf<RootDesc>;
template struct DevBuf<RootRow>;
template struct DevBuf<float>;

// tuning hooks (tools/env_ab.sh): an integer from the environment, or the default
static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

// The factor stores explicit inverses of the pivot blocks: a pivot range beyond 1e13 (a badly scaled dataset: information
// matrices that differ by many orders of magnitude, or a regulariser far below the weights) leaves few correct digits.
static void warn_conditioning(const char *what, const SpdFactor &F) {
  if (F.pivot_min > 0.0 && F.pivot_max / F.pivot_min > 1e13)
    fprintf(stderr, "[dpgo_amd] WARNING: %s is badly conditioned (pivots %.3e .. %.3e, ratio %.1e): expect about %d correct "
                    "digits from its solves.\n", what, F.pivot_min, F.pivot_max, F.pivot_max / F.pivot_min,
            std::max(0, 16 - (int)std::ceil(std::log10(F.pivot_max / F.pivot_min))));
}

// Device layout of a factor.  The solve streams every W_s once per sweep, so the matrices are re-packed
// into PANELS: the entries one tile reads, contiguous, in the order it reads them.
//   forward tile (rows p0 .. p0+count of [y ; dupd], all columns k < kend):  panel[k][r] = WT_s[k][p0 + r]
//   backward tile (pivot columns k0 .. k0+count, rows p >= k0):              panel[p - k0][r] = W_s[p][k0 + r]
// (rows of a panel are ld = count rounded up to 16 doubles apart).  The zero triangle of L11^-1 is simply not
// stored, a wave's consecutive loads are consecutive in memory, and every tile is a pure sequential stream.
// Within a launch the tiles are ordered by decreasing length, so the long ones start first and the short
// ones fill the tail.
void SpdSolverDev::upload(int dcols, const std::vector<int> &node_of_unknown) {
  piv_idx.upload(F.piv_idx);
  upd_idx.upload(F.upd_idx);
  asm_ptr.upload(F.asm_ptr);
  {
    // pull-ordered update buffer: the a-th entry of the assembly lists is row a; the child row that feeds it
    // (F.asm_src[a]) is told where to write.  Every update row has exactly one reader (its parent).
    std::vector<int> dst(std::max(F.total_upd, 1), 0);
    for (size_t a = 0; a < F.asm_src.size(); a++) dst[F.asm_src[a]] = (int)a;
    ubuf_dst.upload(dst);
  }
  ubuf.alloc((size_t)std::max(F.total_upd, 1) * dcols);
  ytmp.alloc((size_t)std::max(F.n, 1) * dcols);
  fwd_level_bytes.clear();
  bwd_level_bytes.clear();
  // algorithmic bytes of one level: every front entry once (8 B) + its in/out vector entries
  auto lvl_bytes = [&](const std::vector<int> &lvl) {
    double b = 0;
    // (the pivot block L11^-1 is triangular: w(w+1)/2 entries)
    for (int f : lvl) b += 8.0 * ((double)F.u[f] * F.w[f] + 0.5 * (double)F.w[f] * (F.w[f] + 1)) + 2.0 * 8.0 * dcols * (F.w[f] + F.u[f]);
    return b;
  };
  for (const auto &lvl : F.by_height) fwd_level_bytes.push_back(lvl_bytes(lvl));
  for (const auto &lvl : F.by_depth) bwd_level_bytes.push_back(lvl_bytes(lvl));
  // two classes of tiles: small fronts (one wave per tile) and wide fronts (8 waves per tile, the columns /
  // rows of the reduction split between the waves).
  // The reduction length decides: columns (w) in the forward sweep, rows (w+u) in the backward sweep.
  const int wide_above = 96;   // (<= 128: a narrow tile's reduction is one LDS chunk)
  auto wide = [&](int f, bool fwd) { return (fwd ? F.w[f] : F.w[f] + F.u[f]) > wide_above; };
  // a small narrow class joins the wide class (whole workgroups are cheap when there are few of them)
  auto tiles64 = [&](const std::vector<int> &lvl, bool fwd, bool want_wide) {
    int cnt = 0;
    for (int f : lvl)
      if (wide(f, fwd) == want_wide) cnt += ((fwd ? F.w[f] + F.u[f] : F.w[f]) + 63) / 64;
    return cnt;
  };
  const int MERGE_BELOW = 1024;
  // The roots are applied through the EXPLICIT inverse of their Schur complement (one launch instead of two, below),
  // which is only as good as that complement is conditioned: cancellation costs kappa * eps in every component, where
  // the two triangular sweeps confine the damage to the near-null direction.  So the pivots of every root are looked
  // at first (d_kk = 1 / Linv_kk^2, within the spectrum of the complement), and a factor with a root whose pivots span
  // more than 1e9 -- G_tt of a graph hosted by ONE node is the Laplacian + 1e-11 I, singular along the gauge -- keeps
  // the two sweeps.
  fused_root = env_int("DPGO_SPD_FUSE_ROOT", 1) != 0;
  // a factor that is re-done every iteration or so (Rescale::Dynamic, G_tt): forming the roots' products again costs 0.5 ms
  // per refactorisation at the headline size, the launch it saves 10 us per solve
  if (F.keep_numeric && env_int("DPGO_SPD_FUSE_ROOT_DYNAMIC", 0) == 0) fused_root = false;
  if (fused_root) {
    double worst = 1.0;
    std::vector<double> diag;
    for (int f = 0; f < F.nfronts && fused_root; f++) {
      if (!(F.parent[f] < 0 && F.u[f] == 0 && F.w[f] > 0)) continue;
      const int w = F.w[f], ld = F.ldw[f];
      diag.assign(w, 0.0);
      if (F.dev_W) HIP_CHECK(hipMemcpy2D(diag.data(), sizeof(double), F.dev_W + F.w_off[f], sizeof(double) * (ld + 1), sizeof(double), w, hipMemcpyDeviceToHost));
      else if (!F.W.empty()) for (int k = 0; k < w; k++) diag[k] = F.W[F.w_off[f] + (size_t)k * ld + k];
      else { fused_root = false; break; }
      double lo = 1e300, hi = 0.0;
      for (int k = 0; k < w; k++) { lo = std::min(lo, std::fabs(diag[k])); hi = std::max(hi, std::fabs(diag[k])); }
      if (!(lo > 0.0) || !std::isfinite(hi)) { fused_root = false; break; }
      worst = std::max(worst, (hi / lo) * (hi / lo));
    }
    if (worst > std::pow(10.0, env_int("DPGO_SPD_FUSE_ROOT_MAXLOG", 9))) fused_root = false;
    if (getenv("DPGO_SPD_DUMP")) fprintf(stderr, "[spd] dof %d roots: pivot range %.2e -> %s\n", dof, worst, fused_root ? "fused" : "two sweeps");
  }
  auto is_root = [&](int f) { return fused_root && F.parent[f] < 0 && F.u[f] == 0 && F.w[f] > 0; };
  int nnodes = 1;
  for (int a : node_of_unknown) nnodes = std::max(nnodes, a + 1);
  auto node_of_front = [&](int f) { return node_of_unknown[F.piv_idx[F.piv_ptr[f]]]; };   // a front never spans two nodes (they are disconnected)
  struct Tile { int f, first, count, rows; int64_t len; };   // rows: tile height of its class; len: panel rows
  auto sweep = [&](bool fwd, std::vector<Level> &out_levels, DevBuf<SpdItem> &items_dev, DevBuf<double> &panels_dev) {
    const auto &levels = fwd ? F.by_height : F.by_depth;
    std::vector<Tile> tiles;
    out_levels.clear();
    for (const auto &lvl_all : levels) {
      std::vector<int> lvl;   // (the roots have a launch of their own)
      for (int f : lvl_all)
        if (!is_root(f)) lvl.push_back(f);
      const bool merge = tiles64(lvl, fwd, true) > 0 && tiles64(lvl, fwd, false) < MERGE_BELOW;
      // Few wide tiles at this level: 16-row tiles put 4x more workgroups (CUs) on them.  A workgroup streams
      // ~30-45 GB/s, so a level is as slow as its longest tile whenever it has fewer tiles than the chip has
      // workgroup slots.  Forward tiles re-assemble the front's input vector once per tile, which makes small
      // tiles expensive: 16 rows only below 192 tiles; backward tiles pay off up to 800 tiles when the
      // reduction (front height) is long.  Thresholds from the per-launch table (DPGO_SPD_DUMP) of the
      // headline instance at 8 and at 1 node per GPU.
      const int wide_tiles = tiles64(lvl, fwd, true) + (merge ? tiles64(lvl, fwd, false) : 0);
      int longest = 0;
      for (int f : lvl) longest = std::max(longest, F.w[f] + F.u[f]);
      const bool fine = fwd ? wide_tiles < env_int("DPGO_SPD_FINE_FWD", 192)
                            : (wide_tiles < env_int("DPGO_SPD_FINE_BWD", 256) || (wide_tiles < env_int("DPGO_SPD_FINE_BWD_TALL", 800) && longest >= 1000));
      const int rows = (wide_tiles > 0 && fine) ? 16 : 64;
      // node by node; within a node the wide tiles first (one workgroup each, longest first), then the narrow ones
      // (one wave each): a launch picks the ranges of the nodes that are still live (Level::map)
      Level lev{(int)tiles.size(), 0, 0, rows, std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0),
                std::vector<int>(nnodes, 0), std::vector<double>(nnodes, 0.0)};
      for (int f : lvl) lev.node_bytes[node_of_front(f)] += lvl_bytes(std::vector<int>{f});
      for (int a = 0; a < nnodes; a++)
        for (int pass = 1; pass >= 0; pass--) {
          const size_t begin = tiles.size();
          const int th = pass == 1 ? rows : 64;
          for (int f : lvl) {
            if (node_of_front(f) != a || (int)(wide(f, fwd) || merge) != pass) continue;
            const int w = F.w[f], m = w + F.u[f], ext = fwd ? m : w;
            for (int r = 0; r < ext; r += th) {
              const int cnt = std::min(th, ext - r);
              // forward: columns k < kend of rows r..; backward: rows p >= r of columns r..
              const int64_t len = fwd ? ((r + th <= w) ? r + th : w) : (m - r);
              tiles.push_back({f, r, cnt, th, len});
            }
          }
          std::stable_sort(tiles.begin() + begin, tiles.end(), [](const Tile &x, const Tile &y) { return x.len * x.count > y.len * y.count; });
          (pass == 1 ? lev.wstart : lev.nstart)[a] = (int)begin;
          (pass == 1 ? lev.wcount : lev.ncount)[a] = (int)(tiles.size() - begin);
          (pass == 1 ? lev.nwide : lev.nnarrow) += (int)(tiles.size() - begin);
        }
      out_levels.push_back(lev);
    }
    // panel offsets
    std::vector<SpdItem> items(tiles.size());
    int64_t total = 0;
    for (size_t i = 0; i < tiles.size(); i++) {
      const Tile &t = tiles[i];
      const int f = t.f, ld = (t.count + 15) / 16 * 16;
      SpdItem it;
      it.front = f; it.first = t.first; it.count = t.count; it.w = F.w[f];
      it.u = F.u[f]; it.ld = ld; it.piv_ptr = F.piv_ptr[f]; it.upd_ptr = F.upd_ptr[f];
      it.pos_off = F.pos_off[f]; it.ubuf_off = F.ubuf_off[f];
      it.node = node_of_front(f);
      it.wait_ctr = -1;
      it.mat_off = total;
      it.wait_need = 0; it.signal_ctr = -1;
      items[i] = it;
      total += t.len * ld;
    }
    if (getenv("DPGO_SPD_DUMP")) {
      int64_t used = 0;
      for (const Tile &t : tiles) used += t.len * t.count;
      fprintf(stderr, "[spd] dof %d %s panels: %.1f MB stored, %.1f MB of entries (padding %.1f %%), %zu tiles\n", dof, fwd ? "fwd" : "bwd",
              total * 8e-6, used * 8e-6, 100.0 * (total - used) / std::max<int64_t>(used, 1), tiles.size());
    }
    items_dev.upload(items);
    if (F.dev_W && F.dev_WT) {
      // the factor is still on the device: the panels are cut out of it there (no trip through the host)
      std::vector<PanelSrc> srcs(tiles.size());
      for (size_t i = 0; i < tiles.size(); i++) {
        const Tile &t = tiles[i];
        if (fwd) srcs[i] = PanelSrc{(long long)(F.wt_off[t.f] + t.first), F.ldm[t.f], (int)t.len};
        else srcs[i] = PanelSrc{(long long)(F.w_off[t.f] + (int64_t)t.first * F.ldw[t.f] + t.first), F.ldw[t.f], (int)t.len};
      }
      DevBuf<PanelSrc> &srcs_dev = fwd ? fwd_srcs : bwd_srcs;   // kept: repack() cuts the panels again after a refactorisation
      srcs_dev.upload(srcs);
      panels_dev.alloc((size_t)std::max<int64_t>(total, 1));   // (zero-filled: the padding of a panel row stays zero)
      launch_pack_panels(nullptr, items_dev.p, srcs_dev.p, (int)tiles.size(), fwd ? F.dev_WT : F.dev_W, panels_dev.p);
      HIP_CHECK(hipDeviceSynchronize());
      return;
    }
    std::vector<double> panels((size_t)std::max<int64_t>(total, 1), 0.0);
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < tiles.size(); i++) {
      const Tile &t = tiles[i];
      const SpdItem &it = items[i];
      double *dst = panels.data() + it.mat_off;
      if (fwd) {
        const double *src = F.WT.data() + F.wt_off[t.f] + t.first;
        const int ldm = F.ldm[t.f];
        for (int64_t k = 0; k < t.len; k++)
          for (int r = 0; r < t.count; r++) dst[k * it.ld + r] = src[(size_t)k * ldm + r];
      } else {
        const double *src = F.W.data() + F.w_off[t.f] + (size_t)t.first * F.ldw[t.f] + t.first;
        const int ldw = F.ldw[t.f];
        for (int64_t p = 0; p < t.len; p++)
          for (int r = 0; r < t.count; r++) dst[p * it.ld + r] = src[(size_t)p * ldw + r];
      }
    }
    panels_dev.upload(panels);
  };
  sweep(true, fwd_levels, fwd_items, WT);
  sweep(false, bwd_levels, bwd_items, W);
  // ---- the roots: tiles of the full w x w product L11^-T L11^-1, forward-style (rows of the front, all columns)
  root_level = Level{0, 0, 0, 64, std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0),
                     std::vector<int>(nnodes, 0), std::vector<double>(nnodes, 0.0)};
  root_items.release();
  Wroot.release();
  root_sym = false;
  root_rows.release();
  root_part.release();
  root_pack.release();
  if (fused_root) {
    // one triangle instead of the full product when the roots are big enough for the second (combine) launch to pay
    // (measured, DESIGN 3.4: an item pays three dependent loads for its inputs before 32 KB of stream, and the second
    // launch costs 5 us -- the form wins where ONE root holds tens of megabytes (a single node per GPU: 19.5 -> 15.8 us
    // for G_tt's 44 MB root) and is a wash on eight roots of 8-23 MB each, which keep the full product)
    std::vector<int> roots;
    double largest_mb = 0;
    for (int f = 0; f < F.nfronts; f++)
      if (is_root(f)) { roots.push_back(f); largest_mb = std::max(largest_mb, 8e-6 * (double)F.w[f] * F.w[f]); }
    const int force = env_int("DPGO_SPD_ROOT_SYM", -1);
    root_sym = !roots.empty() && nnodes <= MAX_LOCAL_NODES &&
               (force == 1 || (force != 0 && largest_mb >= env_int("DPGO_SPD_ROOT_SYM_MB", 32)));
  }
  if (fused_root && root_sym) {
    std::vector<int> roots;
    for (int f = 0; f < F.nfronts; f++)
      if (is_root(f)) roots.push_back(f);
    long long nblocks = 0;
    for (int f : roots) { const long long nb = (F.w[f] + 63) / 64; nblocks += nb * (nb + 1) / 2; }
    // blocks per item (a workgroup each, one dependent gather per item): as many as still leave the chip three workgroups
    // per CU, at most ROOT_SYM_MAXJ
    const int S = (int)std::min<long long>(ROOT_SYM_MAXJ, std::max<long long>(1, env_int("DPGO_SPD_ROOT_SYM_BLOCKS", (int)(nblocks / 768))));
    root_sym_level = Level{0, 0, 0, 64, std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0),
                           std::vector<int>(nnodes, 0), std::vector<double>(nnodes, 0.0)};
    root_rows_level = root_sym_level;
    std::vector<SpdItem> items, pack;
    std::vector<PanelSrc> srcs;
    std::vector<RootRow> rows;
    std::vector<RootDesc> rdesc;
    std::vector<double> host_src;
    std::vector<int64_t> p_off(F.nfronts, 0);
    int64_t ptotal = 0, total = 0;
    int nslots = 0;
    root_max_w = 0;
    for (int f : roots) {
      RootDesc rd;
      rd.src_off = F.dev_W ? F.w_off[f] : (long long)host_src.size();
      if (!F.dev_W) host_src.insert(host_src.end(), F.W.begin() + F.w_off[f], F.W.begin() + F.w_off[f] + (size_t)F.w[f] * F.ldw[f]);
      rd.dst_off = ptotal; rd.ld = F.ldw[f]; rd.w = F.w[f];
      p_off[f] = ptotal;
      ptotal += (int64_t)F.w[f] * F.w[f];
      root_max_w = std::max(root_max_w, F.w[f]);
      rdesc.push_back(rd);
    }
    for (int a = 0; a < nnodes; a++) {
      root_sym_level.nstart[a] = (int)items.size();
      root_rows_level.wstart[a] = (int)rows.size();
      for (int f : roots) {
        if (node_of_front(f) != a) continue;
        const int w = F.w[f], nb = (w + 63) / 64;
        const int tbase = nslots;               // transposed slot of block (I, J), J < I: tbase + I (I - 1) / 2 + J
        nslots += nb * (nb - 1) / 2;
        for (int I = 0; I < nb; I++) {
          RootRow rr;
          rr.piv_ptr = F.piv_ptr[f]; rr.first = I * 64; rr.count = std::min(64, w - I * 64); rr.node = a;
          rr.dslot = nslots; rr.ndslots = 0; rr.tbase = tbase; rr.nb = nb; rr.R = I; rr.pad0 = rr.pad1 = rr.pad2 = 0;
          for (int J0 = 0; J0 <= I; J0 += S) {
            const int nJ = std::min(S, I + 1 - J0);
            SpdItem it;
            it.front = f; it.first = I * 64; it.count = rr.count; it.w = w;
            it.u = J0; it.ld = nJ; it.piv_ptr = F.piv_ptr[f]; it.upd_ptr = nslots++;   // its direct slot
            it.pos_off = F.pos_off[f]; it.ubuf_off = tbase + I * (I - 1) / 2 + J0;     // its first transposed slot
            it.node = a; it.wait_ctr = -1; it.mat_off = total; it.wait_need = 0; it.signal_ctr = -1;
            items.push_back(it);
            rr.ndslots++;
            for (int J = J0; J < J0 + nJ; J++) {
              // block (I, J) k-major: row k of the panel = entries (I*64 .. , J*64 + k) of P = row J*64 + k of the symmetric P
              SpdItem pk = it;
              pk.mat_off = total; pk.ld = 64; pk.count = rr.count;
              pack.push_back(pk);
              srcs.push_back(PanelSrc{(long long)(p_off[f] + (int64_t)(J * 64) * w + I * 64), w, std::min(64, w - J * 64)});
              total += 4096;
            }
          }
          rows.push_back(rr);
          root_sym_level.node_bytes[a] += 8.0 * 4096.0 * (I + 1);
        }
        root_sym_level.node_bytes[a] += 2.0 * 8.0 * dcols * w;
      }
      root_sym_level.ncount[a] = (int)items.size() - root_sym_level.nstart[a];
      root_sym_level.nnarrow += root_sym_level.ncount[a];
      root_rows_level.wcount[a] = (int)rows.size() - root_rows_level.wstart[a];
      root_rows_level.nwide += root_rows_level.wcount[a];
    }
    root_level = root_sym_level;   // (what the dumps and the byte counts look at)
    root_items.upload(items);
    root_pack.upload(pack);
    root_srcs.upload(srcs);
    root_rows.upload(rows);
    root_desc.upload(rdesc);
    root_part.alloc((size_t)std::max(nslots, 1) * 64 * dcols);
    DevBuf<double> src_dev;
    if (!F.dev_W) src_dev.upload(host_src);
    Proot.alloc((size_t)ptotal, false);
    Wroot.alloc((size_t)total);   // (zero-filled: rows and columns of a block past w stay zero)
    launch_root_syrk(nullptr, root_desc.p, (int)rdesc.size(), root_max_w, F.dev_W ? F.dev_W : src_dev.p, Proot.p);
    launch_pack_panels(nullptr, root_pack.p, root_srcs.p, (int)pack.size(), Proot.p, Wroot.p);
    HIP_CHECK(hipDeviceSynchronize());
    if (!F.keep_numeric) Proot.release();
    if (getenv("DPGO_SPD_DUMP"))
      fprintf(stderr, "[spd] dof %d fused roots as one triangle: %zu fronts, %zu items of <= %d blocks, %zu block rows, %d slots, %.1f MB of panels\n",
              dof, roots.size(), items.size(), S, rows.size(), nslots, total * 8e-6);
  } else if (fused_root) {
    std::vector<int> roots;
    for (int f = 0; f < F.nfronts; f++)
      if (is_root(f)) roots.push_back(f);
    int t64 = 0;
    for (int f : roots) t64 += (F.w[f] + 63) / 64;
    // few tiles: 16-row tiles reach 4x more CUs; a single root per GPU (one node per GPU): 8-row tiles, 8x
    const int rows = (t64 < env_int("DPGO_SPD_FINE_ROOT8", 64)) ? 8 : (t64 < env_int("DPGO_SPD_FINE_ROOT", 192) ? 16 : 64);
    root_level.rows = rows;
    std::vector<Tile> tiles;
    for (int a = 0; a < nnodes; a++) {
      const size_t begin = tiles.size();
      for (int f : roots) {
        if (node_of_front(f) != a) continue;
        for (int r = 0; r < F.w[f]; r += rows) tiles.push_back({f, r, std::min(rows, F.w[f] - r), rows, (int64_t)F.w[f]});
        root_level.node_bytes[a] += 8.0 * (double)F.w[f] * F.w[f] + 2.0 * 8.0 * dcols * F.w[f];
      }
      root_level.wstart[a] = (int)begin;
      root_level.wcount[a] = (int)(tiles.size() - begin);
      root_level.nwide += (int)(tiles.size() - begin);
    }
    std::vector<SpdItem> items(tiles.size());
    std::vector<PanelSrc> srcs(tiles.size());
    // where the roots' W_s = L11^-1 are: still on the device after a device factorisation, else uploaded here; the dense
    // products P_f = L11^-T L11^-1 go to a buffer of their own (Proot), the tiles' panels are cut out of it
    std::vector<double> host_src;
    std::vector<RootDesc> rdesc;
    std::vector<int64_t> p_off(F.nfronts, 0);
    int64_t ptotal = 0;
    root_max_w = 0;
    for (int f : roots) {
      RootDesc rd;
      rd.src_off = F.dev_W ? F.w_off[f] : (long long)host_src.size();
      if (!F.dev_W) host_src.insert(host_src.end(), F.W.begin() + F.w_off[f], F.W.begin() + F.w_off[f] + (size_t)F.w[f] * F.ldw[f]);
      rd.dst_off = ptotal; rd.ld = F.ldw[f]; rd.w = F.w[f];
      p_off[f] = ptotal;
      ptotal += (int64_t)F.w[f] * F.w[f];
      root_max_w = std::max(root_max_w, F.w[f]);
      rdesc.push_back(rd);
    }
    int64_t total = 0;
    for (size_t i = 0; i < tiles.size(); i++) {
      const Tile &t = tiles[i];
      const int f = t.f, ld = rows == 8 ? 8 : (t.count + 15) / 16 * 16;   // (8-row tiles: a wave's load spans 8 consecutive 64-byte rows)
      SpdItem it;
      it.front = f; it.first = t.first; it.count = t.count; it.w = F.w[f];
      it.u = 0; it.ld = ld; it.piv_ptr = F.piv_ptr[f]; it.upd_ptr = F.upd_ptr[f];
      it.pos_off = F.pos_off[f]; it.ubuf_off = F.ubuf_off[f];
      it.node = node_of_front(f);
      it.wait_ctr = -1;
      it.mat_off = total;
      it.wait_need = 0; it.signal_ctr = -1;
      items[i] = it;
      srcs[i] = PanelSrc{(long long)(p_off[f] + t.first), F.w[f], F.w[f]};   // columns first.. of the dense w x w product
      total += (int64_t)F.w[f] * ld;
    }
    if (!tiles.empty()) {
      root_items.upload(items);
      root_srcs.upload(srcs);
      root_desc.upload(rdesc);
      DevBuf<double> src_dev;
      if (!F.dev_W) src_dev.upload(host_src);
      Proot.alloc((size_t)ptotal, false);
      Wroot.alloc((size_t)total);   // (zero-filled: the padding of a panel row stays zero)
      launch_root_syrk(nullptr, root_desc.p, (int)rdesc.size(), root_max_w, F.dev_W ? F.dev_W : src_dev.p, Proot.p);
      launch_pack_panels(nullptr, root_items.p, root_srcs.p, (int)tiles.size(), Proot.p, Wroot.p);
      // The tile height above fits ALL roots of the group.  The late steps of the truncated CG run on one to four of the
      // group's nodes: a launch over so few roots gets the next finer class (64 -> 16, 16 -> 8 rows: four / two times the
      // workgroups on the same bytes), cut from the same products -- spd_run picks by the number of live tiles.  Not for
      // a factor that is re-done (its panels would have to be cut twice) nor for the fp32 experiment.
      root_fine_rows = 0;
      root_fine_items.release();
      Wroot_fine.release();
      const int fine = rows == 64 ? 16 : (rows == 16 ? 8 : 0);
      if (fine && nnodes > 1 && !F.keep_numeric) {
        std::vector<Tile> ft;
        root_fine_level = Level{0, 0, 0, fine, std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0), std::vector<int>(nnodes, 0),
                                std::vector<int>(nnodes, 0), root_level.node_bytes};
        for (int a = 0; a < nnodes; a++) {
          const size_t begin = ft.size();
          for (int f : roots) {
            if (node_of_front(f) != a) continue;
            for (int r = 0; r < F.w[f]; r += fine) ft.push_back({f, r, std::min(fine, F.w[f] - r), fine, (int64_t)F.w[f]});
          }
          root_fine_level.wstart[a] = (int)begin;
          root_fine_level.wcount[a] = (int)(ft.size() - begin);
          root_fine_level.nwide += (int)(ft.size() - begin);
        }
        std::vector<SpdItem> fitems(ft.size());
        std::vector<PanelSrc> fsrcs(ft.size());
        int64_t ftotal = 0;
        for (size_t i = 0; i < ft.size(); i++) {
          const Tile &t = ft[i];
          const int f = t.f, ld = fine == 8 ? 8 : (t.count + 15) / 16 * 16;
          SpdItem it;
          it.front = f; it.first = t.first; it.count = t.count; it.w = F.w[f];
          it.u = 0; it.ld = ld; it.piv_ptr = F.piv_ptr[f]; it.upd_ptr = F.upd_ptr[f];
          it.pos_off = F.pos_off[f]; it.ubuf_off = F.ubuf_off[f];
          it.node = node_of_front(f);
          it.wait_ctr = -1; it.mat_off = ftotal; it.wait_need = 0; it.signal_ctr = -1;
          fitems[i] = it;
          fsrcs[i] = PanelSrc{(long long)(p_off[f] + t.first), F.w[f], F.w[f]};
          ftotal += (int64_t)F.w[f] * ld;
        }
        DevBuf<PanelSrc> fsrcs_dev;
        root_fine_items.upload(fitems);
        fsrcs_dev.upload(fsrcs);
        Wroot_fine.alloc((size_t)ftotal);
        launch_pack_panels(nullptr, root_fine_items.p, fsrcs_dev.p, (int)ft.size(), Proot.p, Wroot_fine.p);
        HIP_CHECK(hipDeviceSynchronize());
        root_fine_rows = fine;
        // (live tiles of the coarse class below which the fine one is taken: the thresholds the class itself was chosen by)
        root_fine_below = rows == 64 ? env_int("DPGO_SPD_FINE_ROOT", 192) : env_int("DPGO_SPD_FINE_ROOT8", 64);
      }
      HIP_CHECK(hipDeviceSynchronize());
      if (!F.keep_numeric) Proot.release();   // (kept for repack() when the factor is re-done with new values)
    }
    if (getenv("DPGO_SPD_DUMP"))
      fprintf(stderr, "[spd] dof %d fused roots: %zu fronts, %zu tiles x %d rows, %.1f MB of panels\n", dof, roots.size(), tiles.size(), rows, total * 8e-6);
  }
  spd_release_device(F);
  // the panels are on the device now: the host copy of the factor (gigabytes at the headline size) can go
  std::vector<double>().swap(F.W);
  std::vector<double>().swap(F.WT);
  // both panel sets of a factor this small can live in the 256 MiB Infinity Cache from one solve to the next
  // (measured: G_tt with 177 MB of panels at two nodes per GPU still gains 2 % from staying; 288 MB does not)
  size_t keep = 200u << 20;
  if (const char *e = getenv("DPGO_SPD_KEEP_MB")) keep = (size_t)atol(e) << 20;
  stream_once = sizeof(double) * (W.n + WT.n + Wroot.n) > keep;
  dev.piv_idx = piv_idx.p; dev.upd_idx = upd_idx.p; dev.asm_ptr = asm_ptr.p; dev.ubuf_dst = ubuf_dst.p;
  dev.W = W.p; dev.WT = WT.p; dev.fwd_items = fwd_items.p; dev.bwd_items = bwd_items.p; dev.ubuf = ubuf.p;
  dev.root_items = root_items.p; dev.Wroot = Wroot.p;
}

static void spd_profile(int d, hipStream_t st, SpdSolverDev &S, double *vec);

// New values in the same factor (F.dev_W / F.dev_WT after spd_refactor_device): the panels of every tile are cut out
// again on the device, with the tile lists and sources the first upload() left there.  Enqueued on `st`.
int SpdSolverDev::repack(hipStream_t st) {
  if (!F.dev_W || !F.dev_WT || fwd_srcs.n != fwd_items.n || bwd_srcs.n != bwd_items.n) return -1;
  launch_pack_panels(st, fwd_items.p, fwd_srcs.p, (int)fwd_items.n, F.dev_WT, WT.p);
  launch_pack_panels(st, bwd_items.p, bwd_srcs.p, (int)bwd_items.n, F.dev_W, W.p);
  if (fused_root && root_items.n > 0) {
    const DevBuf<SpdItem> &cut = root_sym ? root_pack : root_items;   // (one triangle: a descriptor per block)
    if (root_srcs.n != cut.n || Proot.n == 0) return -1;
    launch_root_syrk(st, root_desc.p, (int)root_desc.n, root_max_w, F.dev_W, Proot.p);
    launch_pack_panels(st, cut.p, root_srcs.p, (int)cut.n, Proot.p, Wroot.p);
  }
  return 0;
}

// lambda_max of a symmetric matrix by Lanczos with full reorthogonalisation (stands in for the
// Spectra call of DPGOProblem.cpp:106-118, tolerance 1e-4).
static double lanczos_lambda_max(const CsrMatrix &A) {
  const int n = A.n, kmax = std::min(n, 60);
  if (n == 0) return 0.0;
  std::vector<std::vector<double>> Q;
  std::vector<double> alpha, beta, q(n), w(n);
  for (int i = 0; i < n; i++) q[i] = 1.0 + 0.37 * std::sin(1.7 * i + 0.3);
  double nrm = 0;
  for (double v : q) nrm += v * v;
  nrm = std::sqrt(nrm);
  for (double &v : q) v /= nrm;
  double lam_prev = 0, lam = 0;
  for (int k = 0; k < kmax; k++) {
    Q.push_back(q);
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int e = A.ptr[i]; e < A.ptr[i + 1]; e++) s += A.val[e] * q[A.col[e]];
      w[i] = s;
    }
    double a = 0;
    for (int i = 0; i < n; i++) a += w[i] * q[i];
    alpha.push_back(a);
    for (int pass = 0; pass < 2; pass++)
      for (const auto &qq : Q) {
        double c = 0;
        for (int i = 0; i < n; i++) c += w[i] * qq[i];
        for (int i = 0; i < n; i++) w[i] -= c * qq[i];
      }
    double b = 0;
    for (double v : w) b += v * v;
    b = std::sqrt(b);
    // largest eigenvalue of the tridiagonal (alpha, beta) by bisection on the Sturm sequence
    const int m = (int)alpha.size();
    double lo = -1e300, hi = -1e300;
    for (int i = 0; i < m; i++) {
      double r = (i > 0 ? std::fabs(beta[i - 1]) : 0) + (i < m - 1 ? std::fabs(beta[i]) : 0);
      hi = std::max(hi, alpha[i] + r);
      lo = std::max(lo, alpha[i] - r);
    }
    lo = std::min(lo, hi - 1.0);
    for (int it = 0; it < 200; it++) {
      const double mid = 0.5 * (lo + hi);
      int neg = 0;   // number of eigenvalues < mid
      double dd = 1.0;
      for (int i = 0; i < m; i++) {
        dd = alpha[i] - mid - (i > 0 ? beta[i - 1] * beta[i - 1] / dd : 0.0);
        if (dd == 0.0) dd = 1e-300;
        if (dd < 0) neg++;
      }
      if (neg == m) hi = mid; else lo = mid;
      if (hi - lo <= 1e-14 * std::fabs(hi)) break;
    }
    lam = 0.5 * (lo + hi);
    if (k > 3 && std::fabs(lam - lam_prev) <= 1e-6 * std::fabs(lam)) break;
    lam_prev = lam;
    if (b < 1e-12 * std::fabs(lam)) break;
    beta.push_back(b);
    for (int i = 0; i < n; i++) q[i] = w[i] / b;
  }
  return lam;
}

Group::Group(const Graph &g, const std::vector<int> &node_ids, const Options &opt, int device)
    : d_(g.d), device_(device), opt_(opt), nodes_(node_ids) {
  RS_ = (d_ + 1) * d_;
  B_ = d_ + 1;
  num_poses_global_ = g.num_poses;
  num_nodes_total_ = g.num_nodes;
  if (d_ != 2 && d_ != 3) {
    fprintf(stderr, "[dpgo_amd] ERROR: d must be 2 or 3.\n");
    return;
  }
  if (opt.preconditioner != 0 && opt.preconditioner != 1 && opt.preconditioner != 3) {
    fprintf(stderr, "[dpgo_amd] ERROR: preconditioner %d (IncompleteCholesky) is not implemented; use None (0), Jacobi (1) or "
                    "RegularizedCholesky (3).\n", opt.preconditioner);
    return;
  }
  if (opt.rescale != 0 && opt.rescale != 1) {
    fprintf(stderr, "[dpgo_amd] ERROR: rescale must be 0 (Static) or 1 (Dynamic).\n");
    return;
  }
  {
    std::set<int> uniq(node_ids.begin(), node_ids.end());
    if (uniq.size() != node_ids.size()) {
      fprintf(stderr, "[dpgo_amd] ERROR: a node appears twice in the group.\n");
      return;
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    fprintf(stderr, "[dpgo_amd] ERROR: no HIP device available; the DPGO hot path has no CPU fallback.\n");
    return;
  }
  HIP_CHECK(hipSetDevice(device));
  HIP_CHECK(hipStreamCreate(&st_));
  const int L = (int)nodes_.size();
  const bool trivial = (opt.loss == 0);
  info_.resize(L);
  ops_.resize(L);
  scale_.assign(L, {});
  rescale_count_.assign(L, 0);
  res_.assign(L, NodeResults());
  lambda_max_.assign(L, 0.0);
  g_index_.resize(L);
  own_off_.assign(L + 1, 0);
  nbr_off_.assign(L + 1, 0);
  for (int a = 0; a < L; a++) {
    local_of_node_[nodes_[a]] = a;
    if (generate_data_info(nodes_[a], d_, g.measurements[nodes_[a]], info_[a]) != 0) return;
    // Rescale::Dynamic (robust losses): one scale per inter-node edge, all ones at construction (DPGOProblem.cpp:34)
    if (!trivial && opt.rescale == 1) scale_[a].assign(info_[a].inter.size(), 1.0);
    g_index_[a] = g.g_index[nodes_[a]];
    own_off_[a + 1] = own_off_[a] + info_[a].n[0];
    nbr_off_[a + 1] = nbr_off_[a] + info_[a].n[1];
  }
  P0_ = own_off_[L];
  P1_ = nbr_off_[L];
  {   // the operators of the nodes, one host thread per node
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(L, host_threads()))) reduction(+ : bad)
    for (int a = 0; a < L; a++)
      bad += assemble_node(info_[a], opt.regularizer, trivial, ops_[a], scale_[a].empty() ? nullptr : scale_[a].data()) != 0;
    if (bad) return;
  }
  auto uni = [&](int a, int p) { return p < info_[a].n[0] ? own_off_[a] + p : P0_ + nbr_off_[a] + (p - info_[a].n[0]); };

  // ---- segments
  {
    std::vector<Seg> segs;
    std::vector<int> optr(L + 1, 0), nptr(L + 1, 0);
    for (int a = 0; a < L; a++) {
      for (int r = own_off_[a]; r < own_off_[a + 1]; r += SEG_ROWS)
        segs.push_back({r, std::min(r + SEG_ROWS, own_off_[a + 1]), a, 0});
      optr[a + 1] = (int)segs.size();
    }
    const int nown = (int)segs.size();
    for (int a = 0; a < L; a++) {
      nptr[a] = (int)segs.size();
      for (int r = nbr_off_[a]; r < nbr_off_[a + 1]; r += SEG_ROWS)
        segs.push_back({P0_ + r, P0_ + std::min(r + SEG_ROWS, nbr_off_[a + 1]), a, 0});
    }
    nptr[L] = (int)segs.size();
    // nbr_ptr is used as [nptr[a], nptr[a+1]) : make it monotone per node
    segs_.upload(segs);
    own_seg_ptr_.upload(optr);
    own_seg_ptr_host_ = optr;
    nbr_seg_ptr_.upload(nptr);
    T_.segs = segs_.p;
    T_.nseg_own = nown;
    T_.nseg_all = (int)segs.size();
    T_.rows_own = P0_;
    T_.rows_all = P0_ + P1_;
    T_.own_ptr = own_seg_ptr_.p;
    T_.nbr_ptr = nbr_seg_ptr_.p;
  }
  if (L > MAX_LOCAL_NODES) {
    fprintf(stderr, "[dpgo_amd] ERROR: a group hosts at most %d nodes (%d requested).\n", MAX_LOCAL_NODES, L);
    return;
  }
  cur_mask_ = ALL_NODES;
  // pinned: [scalars of k_reduce | a cache line | the flag's cache line | CG summaries | TNT summaries]
  const size_t nsc = (size_t)std::max(L, 1) * MAX_SLOTS;
  HIP_CHECK(hipHostMalloc((void **)&h_scal_, sizeof(double) * (nsc + 16 + (size_t)std::max(L, 1) * (CG_SUMMARY + TNT_SUMMARY + 1) + nsc + 8),
                          hipHostMallocMapped | hipHostMallocCoherent));
  h_flag_ = reinterpret_cast<unsigned long long *>(h_scal_ + nsc + 8);
  *h_flag_ = 0;
  h_cg_ = h_scal_ + nsc + 16;
  h_tnt_ = h_cg_ + (size_t)std::max(L, 1) * CG_SUMMARY;
  h_rs_ = h_tnt_ + (size_t)std::max(L, 1) * TNT_SUMMARY;
  h_upd_ = h_rs_ + std::max(L, 1);   // update()'s sums have a block of their own: the next refinement's sums may arrive before the host has read them
  h_gate_ = h_upd_ + nsc;            // the verdict of k_reduce_gate (group.h: SpecUpdate)
  h_gate_[0] = -1.0;
  for (int i = 0; i < std::max(L, 1) * (CG_SUMMARY + TNT_SUMMARY + 1); i++) h_cg_[i] = 0.0;
  reduce_arrived_.alloc(1);
  fused_ = env_int("DPGO_FUSED", 1) != 0;
  partials_.alloc((size_t)MAX_SLOTS * std::max(T_.nseg_all, 1));
  cg_.alloc(MAX_LOCAL_NODES);
  dmask_.alloc(4);
  dev_seq_.alloc(1);
  go_.alloc(1);
  dev_sums_.alloc((size_t)MAX_LOCAL_NODES * MAX_SLOTS);
  dev_tnt_.alloc((size_t)MAX_LOCAL_NODES * TNT_SUMMARY);
  spec_update_enabled_ = env_int("DPGO_SPEC_UPDATE", 1) != 0;
  coefs_dev_.alloc(MAX_LOCAL_NODES);

  upload_operators();
  // ---- inter-node edges (residual form) and their incidence lists
  {
    std::vector<int> tail, head;
    std::vector<double> R, t, kap, tau;
    std::vector<std::vector<int>> inc(P0_ + P1_);
    e_off_.assign(L + 1, 0);
    for (int a = 0; a < L; a++) e_off_[a + 1] = e_off_[a] + (int)info_[a].inter.size();
    for (int a = 0; a < L; a++)
      for (const auto &m : info_[a].inter) {
        const int e = (int)tail.size();
        const int p = uni(a, info_[a].tail(m)), q = uni(a, info_[a].head(m));
        tail.push_back(p); head.push_back(q);
        for (int k = 0; k < d_ * d_; k++) R.push_back(m.R[k]);
        for (int k = 0; k < d_; k++) t.push_back(m.t[k]);
        kap.push_back(m.kappa); tau.push_back(m.tau);
        inc[p].push_back(2 * e); inc[q].push_back(2 * e + 1);
      }
    std::vector<int> iptr(P0_ + P1_ + 1, 0), iv;
    for (int r = 0; r < P0_ + P1_; r++) {
      iptr[r + 1] = iptr[r] + (int)inc[r].size();
      iv.insert(iv.end(), inc[r].begin(), inc[r].end());
    }
    e_tail_.upload(tail); e_head_.upload(head); e_R_.upload(R); e_t_.upload(t); e_kappa_.upload(kap);
    e_tau_.upload(tau); e_inc_ptr_.upload(iptr); e_inc_.upload(iv);
    {
      // one 128-byte record per incidence, in the order of the incidence lists (kernels.h: InterInc)
      std::vector<InterInc> rec(std::max<size_t>(iv.size(), 1));
      std::memset(rec.data(), 0, sizeof(InterInc) * rec.size());
      for (size_t k = 0; k < iv.size(); k++) {
        const int e = iv[k] >> 1, role = iv[k] & 1;
        InterInc &r = rec[k];
        r.other = role ? tail[e] : head[e];
        r.osrc = -1;
        r.code = iv[k];
        r.tau = tau[e]; r.kappa = kap[e];
        for (int i = 0; i < d_; i++) r.t[i] = t[(size_t)e * d_ + i];
        for (int i = 0; i < d_ * d_; i++) r.R[i] = R[(size_t)e * d_ * d_ + i];
      }
      e_rec_.upload(rec);
      e_rec_host_ = rec;
      E_.rec = e_rec_.p;
    }
    E_.nrows_own = P0_; E_.nrows_all = P0_ + P1_;
    E_.m = (int)tail.size(); E_.tail = e_tail_.p; E_.head = e_head_.p; E_.R = e_R_.p; E_.t = e_t_.p;
    E_.kappa = e_kappa_.p; E_.tau = e_tau_.p; E_.inc_ptr = e_inc_ptr_.p; E_.inc = e_inc_.p;
    if (dynamic()) {
      e_w_.alloc(std::max<size_t>(tail.size(), 1));
      e_scale_.upload(std::vector<double>(std::max<size_t>(tail.size(), 1), 1.0));   // all ones at construction (DPGOProblem.cpp:34)
      e_off_dev_.upload(e_off_);
      rs_count_.alloc(std::max(L, 1));
      rs_flags_.alloc(std::max(L, 1));
    }
  }
  {
    std::vector<int> tail, head;
    std::vector<double> R, t, kap, tau;
    std::vector<std::vector<int>> inc(P0_);
    for (int a = 0; a < L; a++)
      for (const auto &m : info_[a].intra) {
        const int e = (int)tail.size();
        const int p = uni(a, info_[a].tail(m)), q = uni(a, info_[a].head(m));
        tail.push_back(p); head.push_back(q);
        for (int k = 0; k < d_ * d_; k++) R.push_back(m.R[k]);
        for (int k = 0; k < d_; k++) t.push_back(m.t[k]);
        kap.push_back(m.kappa); tau.push_back(m.tau);
        inc[p].push_back(2 * e);
      }
    std::vector<int> iptr(P0_ + 1, 0), iv;
    for (int r = 0; r < P0_; r++) {
      iptr[r + 1] = iptr[r] + (int)inc[r].size();
      iv.insert(iv.end(), inc[r].begin(), inc[r].end());
    }
    i_tail_.upload(tail); i_head_.upload(head); i_R_.upload(R); i_t_.upload(t); i_kappa_.upload(kap);
    i_tau_.upload(tau); i_inc_ptr_.upload(iptr); i_inc_.upload(iv);
    Ei_.nrows_own = P0_; Ei_.nrows_all = P0_;
    Ei_.m = (int)tail.size(); Ei_.tail = i_tail_.p; Ei_.head = i_head_.p; Ei_.R = i_R_.p; Ei_.t = i_t_.p;
    Ei_.kappa = i_kappa_.p; Ei_.tau = i_tau_.p; Ei_.inc_ptr = i_inc_ptr_.p; Ei_.inc = i_inc_.p;
  }
  // ---- SPD solvers: one block-diagonal system over all local nodes
  if (refactor_tt() != 0) return;
  if (dynamic() && Ltt_.F.numeric) setup_device_rescale();
  {
    SetupClock clk;
    CsrMatrix Arr;
    Arr.ptr.push_back(0);
    for (int a = 0; a < L; a++) {
      if (opt.preconditioner == 3 && opt.max_iterations > 0 && opt.max_iterations_accepted > 0) {
        const CsrMatrix &r = ops_[a].GRR;
        lambda_max_[a] = lanczos_lambda_max(r);
        const double shift = lambda_max_[a] / opt.reg_Cholesky_precon_max_condition_number;  // DPGOProblem.cpp:119-123
        for (int i = 0; i < r.n; i++) {
          for (int e = r.ptr[i]; e < r.ptr[i + 1]; e++) {
            Arr.col.push_back(own_off_[a] * d_ + r.col[e]);
            Arr.val.push_back(r.val[e] + (r.col[e] == i ? shift : 0.0));
          }
          Arr.ptr.push_back((int)Arr.col.size());
        }
      }
    }
    Arr.n = (int)Arr.ptr.size() - 1;
    clk.lap("G_RR: lambda_max (Lanczos) + shifted matrix");
    if (Arr.n > 0) {
      if (spd_factor(Arr, Lrr_.F, env_int("DPGO_SPD_LEAF_RR", 96), env_int("DPGO_SPD_COLLAPSE_RR", 0), env_int("DPGO_SPD_QUOTIENT", 1) ? d_ : 1,
                     env_int("DPGO_SPD_DEVICE_PANELS", 1) != 0) != 0) return;
      clk.lap("G_RR: ordering + symbolic + numeric factor");
      warn_conditioning("G_RR + lambda I", Lrr_.F);
      Lrr_.dof = d_;
      std::vector<int> node_of_row((size_t)P0_ * d_);
      for (int a = 0; a < L; a++)
        for (int p = 0; p < info_[a].n[0] * d_; p++) node_of_row[(size_t)own_off_[a] * d_ + p] = a;
      Lrr_.upload(d_, node_of_row);
      clk.lap("G_RR: panels (pack + upload)");
    }
  }
  if (opt.preconditioner == 1) {   // Preconditioner::Jacobi: diag(G_RR)^-1, fixed at construction (DPGOProblem.cpp:96-98)
    std::vector<double> dinv((size_t)P0_ * d_, 1.0);
    for (int a = 0; a < L; a++) {
      const CsrMatrix &r = ops_[a].GRR;
      for (int i = 0; i < r.n; i++)
        for (int e = r.ptr[i]; e < r.ptr[i + 1]; e++)
          if (r.col[e] == i) dinv[(size_t)own_off_[a] * d_ + i] = 1.0 / r.val[e];
    }
    jacobi_.upload(dinv);
  }
  // ---- halo lists
  {
    std::vector<int> gdst, gsrc;
    std::set<std::pair<int, int>> sent;
    for (int a = 0; a < L; a++) {
      for (int k = 0; k < info_[a].n[1]; k++) {
        const auto key = info_[a].nbr_key[k];
        auto it = local_of_node_.find(key.first);
        if (it != local_of_node_.end()) {
          gdst.push_back(P0_ + nbr_off_[a] + k);
          gsrc.push_back(own_off_[it->second] + info_[it->second].index.at(key));
        }
      }
      for (const auto &s : info_[a].sent)
        if (!local_of_node_.count(s.first))
          for (int row : s.second) sent.insert({nodes_[a], row});
    }
    gather_dst_.upload(gdst);
    gather_src_.upload(gsrc);
    for (const auto &s : sent) {
      const int a = local_of_node_.at(s.first);
      sent_rows_.push_back(own_off_[a] + s.second);
      sent_keys_.push_back({s.first, info_[a].own_pose[s.second]});
    }
    sent_rows_dev_.upload(sent_rows_);
  }
  // ---- vectors
  const size_t nall = (size_t)(P0_ + P1_) * RS_, nown = (size_t)P0_ * RS_;
  for (DevBuf<double> *b : {&Xk_, &Zc_, &Zp_, &Y_, &DfE_, &Tall_}) b->alloc(nall);
  for (DevBuf<double> *b : {&Xak_, &Xakh_, &gc_, &gp_, &Dfc_, &Dfp_, &gx_, &Dfx_, &T1_}) b->alloc(nown);
  if (keep_gx()) { GXc_.alloc(nown); GXp_.alloc(nown); }
  for (auto &b : tmp_) b.alloc(nown);
  if (getenv("DPGO_SPD_DUMP")) {
    spd_profile(d_, st_, Ltt_, T1_.p);
    if (Lrr_.F.n > 0) spd_profile(d_, st_, Lrr_, T1_.p);
    HIP_CHECK(hipMemsetAsync(T1_.p, 0, sizeof(double) * nown, st_));
  }
  HIP_CHECK(hipDeviceSynchronize());
  ok_ = true;
}

// The operators of every local node on the device (block-CSR G, S, P, P0, Q; the per-pose arrays D, T, N, V, robust Q).
// Called at construction and after a Dynamic rescale.
void Group::upload_operators() {
  graphs_invalidate();   // (captured launches carry the operators' addresses)
  const int L = num_local();
  const bool trivial = (opt_.loss == 0);
  auto uni = [&](int a, int p) { return p < info_[a].n[0] ? own_off_[a] + p : P0_ + nbr_off_[a] + (p - info_[a].n[0]); };
    std::vector<const BsrMatrix *> v(L);
    for (int a = 0; a < L; a++) v[a] = &ops_[a].G;
    upload_bsr(v, false, G_);
    if (trivial) {
      for (int a = 0; a < L; a++) v[a] = &ops_[a].S;
      upload_bsr(v, false, S_);
      for (int a = 0; a < L; a++) v[a] = &ops_[a].P;
      upload_bsr(v, true, P_);
      for (int a = 0; a < L; a++) v[a] = &ops_[a].P0;
      upload_bsr(v, true, P0m_);
      for (int a = 0; a < L; a++) v[a] = &ops_[a].Q;
      upload_bsr(v, true, Q_);
    }
    std::vector<double> Dd((size_t)P0_ * B_ * B_), Ti(P0_), Nn((size_t)P0_ * d_), Vv((size_t)P0_ * d_ * d_);
    std::vector<double> Qd((size_t)(P0_ + P1_) * B_ * B_, 0.0);
    for (int a = 0; a < L; a++) {
      const int n0 = info_[a].n[0];
      std::copy(ops_[a].D.begin(), ops_[a].D.end(), Dd.begin() + (size_t)own_off_[a] * B_ * B_);
      std::copy(ops_[a].Tinv.begin(), ops_[a].Tinv.end(), Ti.begin() + own_off_[a]);
      std::copy(ops_[a].N.begin(), ops_[a].N.end(), Nn.begin() + (size_t)own_off_[a] * d_);
      std::copy(ops_[a].V.begin(), ops_[a].V.end(), Vv.begin() + (size_t)own_off_[a] * d_ * d_);
      if (!trivial) {   // robust Q is block diagonal
        const BsrMatrix &Q = ops_[a].Q;
        for (int r = 0; r < Q.nrows; r++)
          for (int k = Q.ptr[r]; k < Q.ptr[r + 1]; k++)
            if (Q.col[k] == r)
              std::copy(&Q.val[(size_t)k * B_ * B_], &Q.val[(size_t)(k + 1) * B_ * B_],
                        Qd.begin() + (size_t)uni(a, r) * B_ * B_);
      }
      (void)n0;
    }
    Dd_.upload(Dd); Tinv_.upload(Ti); N_.upload(Nn); V_.upload(Vv); Qd_.upload(Qd);
}

// Factor G_tt of all local nodes (block diagonal): L_.compute / L_.factorize (DPGOProblem.cpp:93, 315, 479)
int Group::refactor_tt() {
  const int L = num_local();
  CsrMatrix Att;
  Att.ptr.push_back(0);
  for (int a = 0; a < L; a++) {
    const CsrMatrix &t = ops_[a].Gtt;
    for (int i = 0; i < t.n; i++) {
      for (int e = t.ptr[i]; e < t.ptr[i + 1]; e++) { Att.col.push_back(own_off_[a] + t.col[e]); Att.val.push_back(t.val[e]); }
      Att.ptr.push_back((int)Att.col.size());
    }
  }
  Att.n = (int)Att.ptr.size() - 1;
  SetupClock clk;
  // Rescale::Dynamic re-factors G_tt every few iterations: the numeric phase keeps its device state, and the values it
  // reads stay on the GPU where k_rescale_apply rewrites the diagonal (att_pos_: where each pose's diagonal entry is)
  const bool keep = dynamic() && env_int("DPGO_RESCALE_HOST", 0) == 0 && env_int("DPGO_SPD_DEVICE_PANELS", 1) != 0 &&
                    env_int("DPGO_SPD_HOST_FACTOR", 0) == 0;
  Ltt_.F.keep_numeric = keep;
  if (keep && att_pos_.n == 0) {
    std::vector<int> pos(std::max(Att.n, 1), 0);
    for (int i = 0; i < Att.n; i++)
      for (int e = Att.ptr[i]; e < Att.ptr[i + 1]; e++)
        if (Att.col[e] == i) pos[i] = e;
    att_pos_.upload(pos);
  }
  if (Ltt_.F.n == Att.n && Ltt_.F.nfronts > 0 && !Ltt_.F.children.empty()) {
    // same pattern, new values (a Dynamic rescale): numeric phase only, on the GPU
    if (spd_refactor(Att, Ltt_.F) != 0) return -1;
  } else {
    // A factor that is re-done in every iteration (Dynamic) is ordered for the refactorisation, not for the three solves
    // it serves: small leaves and NO merged levels keep the fronts at the top of the trees narrow, and the chain of
    // dependent block columns there -- 33 at the merged roots of the headline, two launches each -- is what a
    // refactorisation costs (5.29 -> 4.17 ms per iteration at the headline size; the cost model of
    // spd_factor knows solves only).
    const int leaf = env_int("DPGO_SPD_LEAF_TT", keep ? 64 : 128), collapse = env_int("DPGO_SPD_COLLAPSE_TT", keep ? 1 : 0);
    if (spd_factor(Att, Ltt_.F, leaf, collapse, 1, env_int("DPGO_SPD_DEVICE_PANELS", 1) != 0) != 0) return -1;
  }
  clk.lap("G_tt: ordering + symbolic + numeric factor");
  warn_conditioning("G_tt", Ltt_.F);
  Ltt_.dof = 1;
  std::vector<int> node_of_pose(P0_);
  for (int a = 0; a < L; a++)
    for (int p = 0; p < info_[a].n[0]; p++) node_of_pose[own_off_[a] + p] = a;
  graphs_invalidate();   // (... and the panels')
  Ltt_.upload(d_, node_of_pose);
  clk.lap("G_tt: panels (pack + upload)");
  return 0;
}

// Wait until the group's stream is idle, for at most `seconds`: true when it is.  (hipStreamSynchronize would wait for
// ever behind an exchange whose peer is gone.)
bool Group::drain(double seconds) const {
  if (!st_) return true;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st_);
    if (q != hipErrorNotReady) return q == hipSuccess;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

Group::~Group() {
  // A replay may still be running (its executable graph goes next), a kernel may still write the pinned block.  A group
  // whose stream never drains -- it waits for an exchange whose peer is gone, or a kernel faulted -- LEAKS what the
  // device might still touch instead of freeing it under the device's feet (or waiting for ever).
  if (host_timing_)
    fprintf(stderr, "[host] node group of %d: %ld replays %.3f s in hipGraphLaunch (%.1f us each), %ld eager segments %.3f s, %ld waits %.3f s (%.1f us each)\n",
            num_local(), seg_replays_, t_graph_launch_, seg_replays_ ? 1e6 * t_graph_launch_ / seg_replays_ : 0.0, seg_eager_, t_eager_seg_,
            n_wait_, t_wait_, n_wait_ ? 1e6 * t_wait_ / n_wait_ : 0.0);
  if (host_timing_)
    fprintf(stderr, "[host] updates enqueued ahead of the host's decision: %ld, of which the decision let stand: %ld\n", n_spec_enqueued_, n_spec_stood_);
  if (host_timing_)
    fprintf(stderr, "[host] waits of < 50 us / 200 us / 1 ms / 5 ms / 50 ms / longer: %ld %ld %ld %ld %ld %ld; segments replayed since the host was found to be the slower side: %s\n",
            wait_hist_[0], wait_hist_[1], wait_hist_[2], wait_hist_[3], wait_hist_[4], wait_hist_[5], host_bound_ ? "yes" : "no");
  const bool idle = drain(failed_ ? 2.0 : 60.0);
  if (!idle) dev_leak_buffers(true);   // (the members' buffers are destroyed after this body: hipFree would wait for the stuck stream)
  if (idle) {
    graphs_destroy();
  } else {
    fprintf(stderr, "[dpgo_amd] WARNING: the group's stream did not drain; its graphs, pinned block and stream are leaked.\n");
    seg_graphs_.clear();
  }
  chordal_release();
  if (h_scal_ && idle) (void)hipHostFree(h_scal_);
  if (st_ && idle) (void)hipStreamDestroy(st_);
}

void Group::upload_bsr(const std::vector<const BsrMatrix *> &per_node, bool rows_all, BsrBufs &out) {
  const int L = (int)per_node.size();
  const int nrows = rows_all ? P0_ + P1_ : P0_;
  auto uni = [&](int a, int p) { return p < info_[a].n[0] ? own_off_[a] + p : P0_ + nbr_off_[a] + (p - info_[a].n[0]); };
  std::vector<int> cnt(nrows, 0);
  for (int a = 0; a < L; a++)
    for (int r = 0; r < per_node[a]->nrows; r++) cnt[uni(a, r)] = per_node[a]->ptr[r + 1] - per_node[a]->ptr[r];
  std::vector<int> ptr(nrows + 1, 0);
  for (int r = 0; r < nrows; r++) ptr[r + 1] = ptr[r] + cnt[r];
  std::vector<int> col(ptr[nrows]);
  std::vector<double> val((size_t)ptr[nrows] * B_ * B_);
  for (int a = 0; a < L; a++) {
    const BsrMatrix &M = *per_node[a];
    for (int r = 0; r < M.nrows; r++) {
      int dst = ptr[uni(a, r)];
      for (int k = M.ptr[r]; k < M.ptr[r + 1]; k++, dst++) {
        col[dst] = uni(a, M.col[k]);
        std::copy(&M.val[(size_t)k * B_ * B_], &M.val[(size_t)(k + 1) * B_ * B_], &val[(size_t)dst * B_ * B_]);
      }
    }
  }
  out.ptr.upload(ptr);
  out.col.upload(col);
  if (&out == &G_) {   // compact copy of the translation column for launch_bsr_tcol
    std::vector<double> tc((size_t)ptr[nrows] * B_);
    for (size_t k = 0; k < (size_t)ptr[nrows]; k++)
      for (int r = 0; r < B_; r++) tc[k * B_ + r] = val[k * B_ * B_ + (size_t)r * B_];
    out.tcol.upload(tc);
  }
  {
    // k_bsr gives a block row to BSR_LPR lanes, lane j taking blocks j, j + BSR_LPR, ... of the row.  The values of the (up
    // to) BSR_LPR blocks one round reads are stored interleaved in 16-byte pieces (8-byte for the 3 x 3 blocks of SE(2)):
    // piece p of lane j at ((p * cnt + j) * PS), so that the quad's loads of one instruction are contiguous
    const int BB = B_ * B_, PS = BB % 2 == 0 ? 2 : 1;
    std::vector<double> il(val.size());
    for (int r = 0; r < nrows; r++)
      for (int k0 = ptr[r]; k0 < ptr[r + 1]; k0 += BSR_LPR) {
        const int cnt = std::min(BSR_LPR, ptr[r + 1] - k0);
        for (int j = 0; j < cnt; j++)
          for (int e = 0; e < BB; e++)
            il[(size_t)k0 * BB + (size_t)((e / PS) * cnt + j) * PS + e % PS] = val[(size_t)(k0 + j) * BB + e];
      }
    out.val.upload(il);
  }
  out.dev.nrows = nrows;
  out.dev.nnzb = ptr[nrows];
  out.dev.ptr = out.ptr.p;
  out.dev.col = out.col.p;
  out.dev.val = out.val.p;
}

void Group::sync() const {
  const_cast<Group *>(this)->flush_pending_recv();   // (whoever reads Xk next must find the neighbour rows in it)
  if (xchg_done_) HIP_CHECK(hipEventSynchronize(xchg_done_));   // an exchange on the communicator's stream
  HIP_CHECK(hipStreamSynchronize(st_));
  check_tt_verdict(false);
}

void Group::check_tt_verdict(bool wait) const {
  if (!tt_verdict_pending_) return;
  tt_verdict_pending_ = false;
  // (DPGO_DEBUG_FAIL_REFACTOR=1, a test hook: the verdict counts as "not positive definite")
  static const bool forced = env_int("DPGO_DEBUG_FAIL_REFACTOR", 0) != 0;
  if (spd_refactor_finish(const_cast<SpdFactor &>(Ltt_.F), wait) != 0 || forced) {
    failed_ = true;
    fprintf(stderr, "[dpgo_amd] ERROR: G_tt is not positive definite after a rescale; the group cannot go on (create a new one).\n");
    throw DeviceError("G_tt is not positive definite after a rescale");
  }
}

// The nodes the following launches work on: a bit mask passed to every kernel by value (no upload).
void Group::set_mask(const std::vector<int> &locals) {
  NodeBits m = 0;
  for (int a : locals) m |= 1ull << a;
  cur_mask_ = live_mask(m, nullptr);
}

NodeMask Group::live_mask(NodeBits bits, const NodeBits *p) const {
  NodeMask m{bits, p};
  const int L = num_local();
  int n = 0, idle = -1;
  if ((int)own_seg_ptr_host_.size() != L + 1) return m;
  for (int a = 0; a < L; a++) {
    if ((bits >> a) & 1ull) n++;
    // (the surplus workgroups of the launch land on the idle node's FIRST segment and leave at once: it must be a segment
    // of that node -- a node without own rows has none, its "first" one would be the next node's)
    else if (idle < 0 && own_seg_ptr_host_[a + 1] > own_seg_ptr_host_[a]) idle = a;
  }
  if (n == 0 || n > MAX_LIVE_SEGS || idle < 0) return m;   // (every node, too many, or nowhere to park: the whole grid)
  for (int a = 0; a < L; a++)
    if ((bits >> a) & 1ull) {
      m.seg0[m.nlive] = own_seg_ptr_host_[a];
      m.nseg[m.nlive] = own_seg_ptr_host_[a + 1] - own_seg_ptr_host_[a];
      m.nlive++;
    }
  m.idle_seg = own_seg_ptr_host_[idle];
  return m;
}

// group.h: SpecUpdate.  spec_update_possible: asked by run_tnt() before it enqueues the head of a refinement -- the trial
// point's reduction then waits for speculate_update(), which is called once update(k-1)'s scalars are taken (the gate gets
// them by value) and enqueues that reduction WITH the gate in one launch, then the continuation.
bool Group::spec_update_possible(const double *xprop) const {
  const int L = num_local();
  if (!spec_update_armed_ || !spec_update_enabled_ || !fused_ || !keep_gx() || star_ || capturing_ || iter_graph_wanted() ||
      xchg_done_ || pending_recv_ || !deferred_.empty() || pending_tail_.on || xprop != tmp_[7].p || L == 0)
    return false;
  for (int a = 0; a < L; a++)
    if (res_[a].iters < 1 || !res_[a].updated) return false;   // (every node: a later update, never a node's first)
  return true;
}

void Group::speculate_update(const double *xprop, int nslots_trial) {
  spec_upd_ = SpecUpdate();
  const int L = num_local();
  AmmGate G;
  G.nnodes = L; G.ds = 2 * MAX_DOTS; G.max_it = opt_.max_iterations; G.max_acc = opt_.max_iterations_accepted;
  G.max_hits0 = opt_.max_soft_restart_hits[0]; G.max_hits1 = opt_.max_soft_restart_hits[1];
  G.sqrt_eps = std::sqrt(std::numeric_limits<double>::epsilon()); G.eta1 = .05;   // TNT.h:83 (run_tnt's constants)
  G.rel_tol = opt_.rel_func_decrease_tol; G.step_tol = opt_.stepsize_tol; G.psi = opt_.psi; G.phi = opt_.phi;
  for (int a = 0; a < MAX_LOCAL_NODES; a++) {
    const bool in = a < L;
    G.f[a] = in ? res_[a].f : 0.0; G.Fk0[a] = in ? res_[a].Fk[0] : 0.0; G.Fk1[a] = in ? res_[a].Fk[1] : 0.0;
    G.fobj[a] = in ? res_[a].fobj : 0.0;
    G.hits0[a] = in ? res_[a].soft_restart_hits[0] : 0; G.hits1[a] = in ? res_[a].soft_restart_hits[1] : 0;
  }
  // the trial point's sums to the host (k_reduce's work, its flag) and the gate's verdict, one launch
  launch_reduce_gate(st_, T_, L, nslots_trial, partials_.p, h_scal_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p, dev_sums_.p, G,
                     dev_tnt_.p, cg_.p, go_.p, h_gate_);
  spec_upd_.seq_trial = fetch_seq_;
  // the common course from here: the accepted point is the trial buffer; iterate()'s tail and the local halo copy; the history
  // rotates (X[iter] <- what is X[iter-1] now, and so on); update()'s later-iteration sequence for the static robust surrogate
  const double *nxak = xprop, *zp = Zc_.p;
  double *zc = Zp_.p, *gc = gp_.p, *dfc = Dfp_.p, *gx = GXp_.p;
  const NodeMask m{all_bits(), go_.p};
  if (gather_dst_.n > 0) launch_copy_indexed(d_, st_, (int)gather_dst_.n, gather_dst_.p, gather_src_.p, nxak, Xk_.p, go_.p);
  double *const pupd = partials_.p + (size_t)UPD_SLOT0 * T_.nseg_all;   // (update()'s own slots: deferred_slots_ is 0 here, Dynamic is off)
  launch_bsr(d_, st_, T_, false, m, G_.dev, nxak, false, nullptr, gx, nxak, 0.5, nullptr, pupd, 5, Xk_.p, zc);
  InterFuse fz;
  fz.GX = gx; fz.X = nxak; fz.Df = dfc; fz.gn_slot = 4;
  launch_inter(d_, st_, T_, m, E_, opt_.loss, opt_.loss_reg, 0, true, zc, zp, Qd_.p, Dd_.p, DfE_.p, gc, pupd, nullptr, nullptr, nullptr,
               nullptr, nullptr, Xk_.p, nullptr, &fz);
  // (the closing reduction: left to the next refinement where update() itself would leave it, group.h: UpdLazy)
  spec_upd_.lazy = lazy_update_reduce();
  if (!spec_upd_.lazy) launch_reduce(st_, T_, L, true, 6, pupd, h_upd_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
  n_spec_enqueued_++;
  spec_upd_.on = true; spec_upd_.seq_last = fetch_seq_;
  spec_upd_.xak = nxak; spec_upd_.zc = zc; spec_upd_.gc = gc; spec_upd_.dfc = dfc; spec_upd_.gx = gx;
}

// The host has taken its decision: the enqueued continuation stands (the common course) or is forgotten (its launches fell
// through); either way the gate's verdict, which arrives behind the trial point's flag, is compared at the next wait that
// covers it.
void Group::check_gate(bool host_common) {
  if (!spec_upd_.on) return;
  spec_verdict_pending_ = true;
  spec_verdict_expected_ = host_common;
  spec_verdict_seq_ = spec_upd_.seq_trial;   // (the verdict is written in front of that flag)
  if (!host_common) spec_upd_ = SpecUpdate();
  else n_spec_stood_++;
}

// update()'s closing reduction left for the next refinement's k_cg_scal_begin (group.h: UpdLazy): eager launches of the fused
// sequence only (a replayed segment is a fixed list of launches)
bool Group::lazy_update_reduce() const {
  static const bool on = env_int("DPGO_LAZY_UPDATE_REDUCE", 1) != 0;   // (A/B hook)
  return on && fused_ && keep_gx() && !star_ && !capturing_ && !iter_graph_wanted();
}

void Group::flush_pending_tail() {
  if (!pending_tail_.on) return;
  const PendingTail p = pending_tail_;
  pending_tail_.on = false;
  launch_axpby(d_, st_, T_, false, p.m, 1.0, p.xak, 0.0, nullptr, p.xk, 0, p.z);
}

// ---------------------------------------------------------------------------
// Segments of an iteration as graph replays (group.h)
// ---------------------------------------------------------------------------
bool Group::iter_graph_wanted() const {
  static const int force = env_int("DPGO_ITER_GRAPH", -1);
  if (graphs_broken_ || force == 0 || prof_enabled()) return false;   // (never while launches are timed)
  if (force == 1) return true;
  // Where a segment streams gigabytes (the headline's eight nodes on one GPU) the host is never what bounds it, and its
  // launches shrink with the set of nodes that still iterate, which a replay's frozen grids cannot do.  Below that size:
  // replays once the host has been seen to be the slower side (host_bound_tick).
  return P0_ <= 40000 && host_bound_;
}

// Every 32 iterations: the share of its time inside iterate() / update() that this group's host thread spent waiting for
// read-backs.  Measured: 0.68 at one node per GPU of the headline on an idle host (eager launches are the faster way there:
// replays cost 4-8 %), between 0.4 and 0.55 for the same on a slower box, 0.06-0.15 for sphere2500, city10000, M3500.  Below
// 0.4 the host is what bounds the group.  (The CG steps of small multi-node groups are replayed whatever this says:
// cg_graph_wanted.)
void Group::host_bound_tick() {
  if (host_bound_ || ++win_iters_ < 32) return;
  static const double below = getenv("DPGO_HOST_BOUND_BELOW") ? atof(getenv("DPGO_HOST_BOUND_BELOW")) : 0.40;   // (test hook)
  // (round 6: ... and at least half of its waits found the flag already raised -- the GPU had been waiting for the HOST.  A host
  // that enqueues ahead of the GPU's decisions -- SpecUpdate -- spends less of its time waiting without being the slower side)
  const bool late = below >= 1.0 || 2 * win_nlate_ >= win_nwait_;
  if (win_lib_s_ > 0 && win_wait_s_ < below * win_lib_s_ && late) host_bound_ = true;
  win_iters_ = 0;
  win_wait_s_ = 0;
  win_lib_s_ = 0;
  win_nwait_ = win_nlate_ = 0;
}

void Group::graphs_destroy() {
  for (auto &g : seg_graphs_)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  seg_graphs_.clear();
}

// Whatever a captured launch carries by value has changed (operators re-uploaded, panels re-cut, options set): the
// graphs go.  A graph that may still be executing must not be destroyed, so the stream is drained first -- bounded;
// a stream that never drains keeps (leaks) its graphs.
void Group::graphs_invalidate() {
  graph_gen_++;
  seg_captures_live_ = 0;   // (the cap below is on captures of one generation of arguments, not of the group's life)
  if (seg_graphs_.empty()) return;
  if (drain(60.0)) graphs_destroy();
  else seg_graphs_.clear();
}

void Group::defer_or_launch(unsigned long long key, std::function<void()> fn) {
  if (!defer_armed_) { fn(); return; }
  deferred_.push_back(std::move(fn));
  deferred_key_ = deferred_key_ * 1000003ull + key;
}

void Group::flush_deferred() {
  if (deferred_.empty()) return;
  std::vector<std::function<void()>> d;
  d.swap(deferred_);
  deferred_key_ = 0;
  for (auto &f : d) f();
}

void Group::segment(int id, NodeBits bits, std::initializer_list<unsigned long long> extra, const std::function<void()> &body_in,
                    int wanted_in) {
  if (capturing_) { body_in(); return; }   // (a segment inside a segment is part of it)
  // launches that were waiting for a segment to carry them (step()) become its head
  std::vector<std::function<void()>> pro;
  pro.swap(deferred_);
  const unsigned long long pro_key = pro.empty() ? 0ull : deferred_key_;
  deferred_key_ = 0;
  const std::function<void()> with_pro = [&] {
    for (auto &f : pro) f();
    body_in();
  };
  const std::function<void()> &body = pro.empty() ? body_in : with_pro;
  const bool wanted = wanted_in < 0 ? iter_graph_wanted() : wanted_in != 0;
  if (!wanted || bits != all_bits()) {
    seg_eager_++;
    const auto t0 = std::chrono::steady_clock::now();
    body();
    if (host_timing_) t_eager_seg_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return;
  }
  std::vector<unsigned long long> key;
  key.reserve(16 + extra.size());
  key.push_back((unsigned long long)id);
  key.push_back(graph_gen_);
  // the buffers that rotate with the history or swap with an accepted step
  for (const DevBuf<double> *b : {&Zc_, &Zp_, &gc_, &gp_, &Dfc_, &Dfp_, &GXc_, &GXp_, &Xak_, &tmp_[7]})
    key.push_back((unsigned long long)(uintptr_t)b->p);
  key.insert(key.end(), extra.begin(), extra.end());
  key.push_back(pro_key);
  SegGraph *hit = nullptr;
  for (auto &g : seg_graphs_)
    if (g.key == key) { hit = &g; break; }
  if (!hit && seg_captures_live_ >= 256) {
    if (!capture_cap_warned_) {
      capture_cap_warned_ = true;
      fprintf(stderr, "[dpgo_amd] WARNING: more than 256 segment variants captured without the arguments changing; further new variants "
                      "run eagerly (the replayed ones stay).\n");
    }
    // (more variants than a steady state has: whatever keeps changing, capturing it again and again is not the cure)
    seg_eager_++;
    body();
    return;
  }
  if (!hit) {
    hipGraph_t graph = nullptr;
    capturing_ = true;
    captured_flags_ = 0;
    bool ok = hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      try {
        body();
      } catch (...) {
        ok = false;
      }
      if (hipStreamEndCapture(st_, &graph) != hipSuccess) ok = false;
    }
    capturing_ = false;
    hipGraphExec_t exec = nullptr;
    if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) ok = false;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
      // nothing of the body has run (a capture only records): run it eagerly, and stop trying
      (void)hipGetLastError();
      graphs_broken_ = true;
      fprintf(stderr, "[dpgo_amd] WARNING: a segment of the iteration could not be captured as a graph; eager launches from here on.\n");
      seg_eager_++;
      body();
      return;
    }
    if (seg_graphs_.size() >= 32) {   // (a handful of keys per segment is normal: the history rotates, the iterate swaps)
      size_t old = 0;
      for (size_t i = 1; i < seg_graphs_.size(); i++)
        if (seg_graphs_[i].used < seg_graphs_[old].used) old = i;
      // (never destroy a graph that may be executing: the least recently used one was replayed many read-backs ago -- the
      // flag says so -- and only if it does not is the stream waited for)
      if (__atomic_load_n(h_flag_, __ATOMIC_ACQUIRE) >= seg_graphs_[old].done_seq || drain(60.0)) (void)hipGraphExecDestroy(seg_graphs_[old].exec);
      seg_graphs_.erase(seg_graphs_.begin() + old);
    }
    seg_graphs_.push_back(SegGraph{key, exec, captured_flags_, 0, 0});
    hit = &seg_graphs_.back();
    seg_captures_++;
    seg_captures_live_++;
  }
  hit->used = ++seg_clock_;
  if (host_timing_) {
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipGraphLaunch(hit->exec, st_));
    t_graph_launch_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } else
  HIP_CHECK(hipGraphLaunch(hit->exec, st_));
  fetch_seq_ += hit->flags;   // the flag-raising kernels of the replay count on from the device's own word
  hit->done_seq = fetch_seq_ + 1;   // (a flag raised BEHIND the replay says it is over: its own flag need not be its last kernel)
  seg_replays_++;
}

unsigned long long Group::fetch_async(int nslots, bool all_rows) {
  finish_update();
  nslots = std::max(nslots, deferred_slots_);
  deferred_slots_ = 0;
  launch_reduce(st_, T_, num_local(), all_rows, nslots, partials_.p, h_scal_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
  return fetch_seq_;   // (under capture: the caller adds the replay's flags itself, segment())
}

// The deferred end of update(): wait for its reduction, then the scalar logic that needs the numbers.
void Group::finish_update() {
  if (!pending_update_) return;
  if (upd_lazy_.pending) {   // (nobody has taken update()'s reduction along: launch it now)
    upd_lazy_.pending = false;
    launch_reduce(st_, T_, num_local(), true, upd_lazy_.nslots, partials_.p + (size_t)UPD_SLOT0 * T_.nseg_all, h_upd_, reduce_arrived_.p, h_flag_,
                  next_seq(), dev_seq_.p);
    pending_seq_ = fetch_seq_;
  }
  std::function<void()> f;
  f.swap(pending_update_);
  wait_flag(pending_seq_);
  f();
}

void Group::fetch(int nslots, bool all_rows) {
  finish_update();   // (its scalars sit in the pinned slots the next reduction overwrites)
  nslots = std::max(nslots, deferred_slots_);
  deferred_slots_ = 0;
  launch_reduce(st_, T_, num_local(), all_rows, nslots, partials_.p, h_scal_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
  wait_flag(fetch_seq_);
}

// Wait until the kernel that raises the pinned flag to `seq` (or a later one of the in-order stream) has run: seeing
// the flag means everything enqueued before that kernel is done.
void Group::wait_flag(unsigned long long seq) {
  const auto t0 = std::chrono::steady_clock::now();
  struct Acc {   // (what the host-bound test and DPGO_HOST_TIMING need: the time spent in here)
    Group *g; std::chrono::steady_clock::time_point t;
    ~Acc() {
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
      g->win_wait_s_ += dt;
      if (!g->host_timing_) return;
      g->t_wait_ += dt; g->n_wait_++;
      g->wait_hist_[dt < 50e-6 ? 0 : dt < 200e-6 ? 1 : dt < 1e-3 ? 2 : dt < 5e-3 ? 3 : dt < 50e-3 ? 4 : 5]++;
    }
  } acc{this, t0};
  auto arrived = [&] { return __atomic_load_n(h_flag_, __ATOMIC_ACQUIRE) >= seq; };
  win_nwait_++;
  if (arrived()) win_nlate_++;
  auto last = t0;
  for (unsigned spins = 0; !arrived(); spins++) {
    __builtin_ia32_pause();
    if ((spins & 0xfffff) != 0xfffff) continue;
    const auto now = std::chrono::steady_clock::now();
    if (host_timing_ && now - last > std::chrono::milliseconds(1)) holes_total_++;   // (the thread was off its core: diagnostic)
    last = now;
    if (now - t0 > std::chrono::seconds(60)) {
      // surfaces a kernel fault, if that is why the flag never came -- without waiting for ever on a stream that is itself
      // waiting for an exchange whose peer is gone
      hipError_t q = hipStreamQuery(st_);
      for (int i = 0; i < 600 && q == hipErrorNotReady && !arrived(); i++) {
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        q = hipStreamQuery(st_);
      }
      if (q != hipErrorNotReady) HIP_CHECK(q);
      if (arrived()) break;
      fprintf(stderr, "[dpgo_amd] ERROR: read-back flag never arrived\n");
      if (stuck_fn_) stuck_fn_(stuck_user_);   // (a collective on this stream that never ends: its communicator aborts it now)
      throw DeviceError("read-back flag never arrived");
    }
  }
  // (debug hook: a host that comes late to every read-back -- the stream runs ahead of it by that much; the results must not
  // depend on it, tests/test_gpu_parity.py)
  static const int late_us = [] { const char *e = getenv("DPGO_DEBUG_LATE_HOST_US"); return e ? atoi(e) : 0; }();
  if (late_us > 0) {
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(late_us);
    while (std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
  }
  if (tt_verdict_pending_ && seq >= tt_verdict_seq_) check_tt_verdict(false);   // (the stream has passed the factorisation)
  if (spec_verdict_pending_ && seq >= spec_verdict_seq_) {   // (... and the gate of a speculative update)
    spec_verdict_pending_ = false;
    const double v = h_gate_[0];
    if ((v == 1.0) != spec_verdict_expected_) {
      failed_ = true;
      fprintf(stderr, "[dpgo_amd] ERROR: the device-side gate of the speculative update (%.0f) and the host (%d) disagree; the group cannot go on.\n",
              v, (int)spec_verdict_expected_);
      throw DeviceError("gate / host verdicts differ");
    }
  }
}

void Group::copy_rows(double *dst, const double *src, bool all_rows, int part) {
  launch_axpby(d_, st_, T_, all_rows, cur_mask_, 1.0, src, 0.0, nullptr, dst, part);
}

// out <- scale * A^-1 in (the unknowns' entries of the records; everything else in `out` is left alone).
// The forward sweep only reads `in`, the backward sweep only touches `out`: in == out solves in place.
bool SpdSolverDev::Level::map(NodeBits bits, SpdLevelMap &M, double *bytes) const {
  const int nw = spd_waves(rows);
  M.nlive = 0;
  M.pad = tile0;
  int maxw = 0, maxn = 0;
  double b = 0;
  for (int a = 0; a < (int)wcount.size() && a < MAX_LOCAL_NODES; a++) {
    if (!((bits >> a) & 1ull) || wcount[a] + ncount[a] == 0) continue;
    const int j = M.nlive++;
    M.node[j] = (unsigned char)a;
    M.wstart[j] = wstart[a]; M.wcount[j] = wcount[a];
    M.nstart[j] = nstart[a]; M.ncount[j] = ncount[a];
    maxw = std::max(maxw, wcount[a]);
    maxn = std::max(maxn, ncount[a]);
    b += node_bytes[a];
  }
  M.wide_wgs = M.nlive * maxw;
  M.narrow_wgs = M.nlive * ((maxn + nw - 1) / nw);
  if (bytes) *bytes = b;
  return M.nlive > 0;
}

// The tile class of the fused roots for a launch over the nodes `v`: the finer one where few roots are live.  The two classes
// split a row's sum differently (their results differ in the last bits), so the choice must be a function of what the
// ALGORITHM knows -- the nodes live after the last step the host has read (tnt.cpp: live_after) -- never of the launch's
// geometry: a captured CG step, whose launches cover every node, asks with the eager step's node set (class_of below).
bool SpdSolverDev::fine_root_for(NodeBits v) const {
  if (root_sym || !root_fine_rows) return false;
  SpdLevelMap rm;
  if (!root_level.map(v, rm)) return false;
  int live_tiles = 0;
  for (int j = 0; j < rm.nlive; j++) live_tiles += rm.wcount[j];
  return live_tiles * root_level.rows / 64 < root_fine_below;
}

void spd_run(int d, hipStream_t st, SpdSolverDev &S, NodeMask mask, double *in, double *out, double scale, const NodeBits *class_of) {
  // the launches of the nodes in mask.v (what the host knows); mask.p, if any, is the device's more recent word
  std::vector<SpdLevelMap> fm(S.fwd_levels.size()), bm(S.bwd_levels.size());
  std::vector<double> fby(fm.size(), 0.0), bby(bm.size(), 0.0);
  std::vector<char> fon(fm.size(), 0), bon(bm.size(), 0);
  double bf = 0, bb = 0;
  int nf = 0, nb = 0;
  for (size_t l = 0; l < fm.size(); l++)
    if ((fon[l] = S.fwd_levels[l].map(mask.v, fm[l], &fby[l]))) { bf += fby[l]; nf++; }
  for (size_t l = 0; l < bm.size(); l++)
    if ((bon[l] = S.bwd_levels[l].map(mask.v, bm[l], &bby[l]))) { bb += bby[l]; nb++; }
  SpdLevelMap rm, rrm;
  double rby = 0;
  bool ron = S.root_sym ? (S.root_sym_level.map(mask.v, rm, &rby) && S.root_rows_level.map(mask.v, rrm)) : S.root_level.map(mask.v, rm, &rby);
  // few live roots: the finer tile class (upload()), counted in 64-row tiles as the classes are chosen
  bool fine_root = false;
  if (ron && S.fine_root_for(class_of ? *class_of : mask.v)) {
    fine_root = S.root_fine_level.map(mask.v, rm, &rby);
    ron = fine_root;
  }
  if (ron && in == out) throw DeviceError("spd_run: the fused root step cannot solve in place");
  {
  ProfSweep sweep(true, st, bf + rby, nf + (ron ? (S.root_sym ? 2 : 1) : 0));
  for (size_t l = 0; l < fm.size(); l++)
    if (fon[l]) launch_spd_level(d, S.dof, st, S.dev, 0, fm[l], S.fwd_levels[l].rows, in, S.ytmp.p, scale, fby[l], S.stream_once, mask);
  // the roots: right-hand side from `in` (+ the children's updates), solution straight into `out`
  if (ron && S.root_sym) {
    launch_root_sym(d, S.dof, st, S.dev, rm, in, S.root_part.p, rby, S.stream_once, mask);
    launch_root_combine(d, S.dof, st, S.dev, rrm, S.root_rows.p, S.root_part.p, scale, out, mask);
  } else if (ron && fine_root) {
    SpdDev dv = S.dev;
    dv.root_items = S.root_fine_items.p;
    dv.Wroot = S.Wroot_fine.p;
    launch_spd_level(d, S.dof, st, dv, 2, rm, S.root_fine_rows, in, out, scale, rby, S.stream_once, mask);
  } else if (ron) launch_spd_level(d, S.dof, st, S.dev, 2, rm, S.root_level.rows, in, out, scale, rby, S.stream_once, mask);
  }
  ProfSweep sweep(false, st, bb, nb);
  for (size_t l = 0; l < bm.size(); l++)
    if (bon[l]) launch_spd_level(d, S.dof, st, S.dev, 1, bm[l], S.bwd_levels[l].rows, out, S.ytmp.p, scale, bby[l], S.stream_once, mask);
}

// DPGO_SPD_DUMP=1: time every launch of one solve on a zero vector (HIP events, best of 5) and print its
// algorithmic bytes and rate -- the per-level view behind bench.py's per-family roofline numbers.
static void spd_profile(int d, hipStream_t st, SpdSolverDev &S, double *vec) {
  const SpdFactor &F = S.F;
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  double tot_us = 0, tot_mb = 0;
  unsigned long long *trace = nullptr;
  if (getenv("DPGO_SPD_TRACE")) {
    size_t most = 1;
    for (const auto &v : S.fwd_levels) most = std::max(most, (size_t)(v.nwide + v.nnarrow));
    for (const auto &v : S.bwd_levels) most = std::max(most, (size_t)(v.nwide + v.nnarrow));
    most = std::max(most, (size_t)S.root_level.nwide);
    HIP_CHECK(hipMalloc(&trace, most * 6 * 8));
    HIP_CHECK(hipMemset(trace, 0, most * 6 * 8));
    spd_trace_set(trace);
  }
  auto is_root = [&](int f) { return S.fused_root && F.parent[f] < 0 && F.u[f] == 0 && F.w[f] > 0; };
  auto run = [&](int mode, size_t l, const SpdSolverDev::Level &v, const std::vector<int> &fronts_all) {
    if (v.nwide + v.nnarrow == 0) return;
    const bool fwd = mode != 1;
    int wmax = 0, mmax = 0;
    std::vector<int> fronts;
    for (int f : fronts_all)
      if (is_root(f) == (mode == 2)) fronts.push_back(f);
    for (int f : fronts) { wmax = std::max(wmax, F.w[f]); mmax = std::max(mmax, F.w[f] + F.u[f]); }
    double bytes = 0;
    for (double b : v.node_bytes) bytes += b;
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
      HIP_CHECK(hipEventRecord(e0, st));
      SpdLevelMap M;
      v.map(~0ull, M);
      // (mode 2 on a zero vector: in and out may be the same array here, nothing is compared)
      if (mode == 2 && S.root_sym) {
        SpdLevelMap R;
        S.root_rows_level.map(~0ull, R);
        launch_root_sym(d, S.dof, st, S.dev, M, vec, S.root_part.p, 0.0, S.stream_once, ALL_NODES);
        launch_root_combine(d, S.dof, st, S.dev, R, S.root_rows.p, S.root_part.p, 1.0, vec, ALL_NODES);
      } else
      launch_spd_level(d, S.dof, st, S.dev, mode, M, v.rows, vec, mode == 2 ? vec : S.ytmp.p, 1.0, 0.0, S.stream_once, ALL_NODES);
      HIP_CHECK(hipEventRecord(e1, st));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = std::min(best, ms);
    }
    tot_us += best * 1e3;
    tot_mb += bytes / 1e6;
    if (trace) {   // (-DSPD_TRACE builds) phase timestamps of the last repetition, 100 MHz ticks -> us
      const int nt = v.nwide + v.nnarrow;
      std::vector<unsigned long long> h((size_t)nt * 6);
      HIP_CHECK(hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long tmin = ~0ull;
      for (int t = 0; t < nt; t++) if (h[(size_t)t * 6]) tmin = std::min(tmin, h[(size_t)t * 6]);
      auto pct = [](std::vector<double> &x, double q) { if (x.empty()) return 0.0; std::sort(x.begin(), x.end()); return x[std::min(x.size() - 1, (size_t)(q * x.size()))]; };
      for (int cls = 0; cls < 2; cls++) {
        const int a = cls == 0 ? 0 : v.nwide, b = cls == 0 ? v.nwide : nt;
        if (a == b) continue;
        std::vector<double> ph[5], start, end;
        for (int t = a; t < b; t++) {
          const unsigned long long *q = &h[(size_t)t * 6];
          if (!q[0] || !q[5]) continue;
          for (int i = 0; i < 5; i++) ph[i].push_back((double)(q[i + 1] - q[i]) * 0.01);
          start.push_back((double)(q[0] - tmin) * 0.01);
          end.push_back((double)(q[5] - tmin) * 0.01);
        }
        fprintf(stderr, "[trace]   %s tiles %5zu: item %.2f/%.2f  gather %.2f/%.2f  stream %.2f/%.2f  reduce %.2f/%.2f  write %.2f/%.2f us (median/p90);"
                " start p10 %.1f p50 %.1f p90 %.1f max %.1f, end p10 %.1f p50 %.1f p90 %.1f max %.1f us\n", cls == 0 ? "wide  " : "narrow", start.size(),
                pct(ph[0], .5), pct(ph[0], .9), pct(ph[1], .5), pct(ph[1], .9), pct(ph[2], .5), pct(ph[2], .9), pct(ph[3], .5), pct(ph[3], .9),
                pct(ph[4], .5), pct(ph[4], .9), pct(start, .1), pct(start, .5), pct(start, .9), pct(start, 1.0), pct(end, .1), pct(end, .5),
                pct(end, .9), pct(end, 1.0));
      }
    }
    fprintf(stderr, "[spd] dof %d %s level %2zu fronts %5zu wide tiles %5d x %2d rows, narrow tiles %5d, max_w %4d max_m %4d  %7.2f MB %6.1f us %6.0f GB/s\n",
            S.dof, mode == 2 ? "root" : (fwd ? "fwd" : "bwd"), l, fronts.size(), v.nwide, v.rows, v.nnarrow, wmax, mmax, bytes / 1e6, best * 1e3, bytes / (best * 1e-3) / 1e9);
  };
  std::vector<int> every(F.nfronts);
  for (int f = 0; f < F.nfronts; f++) every[f] = f;
  for (size_t l = 0; l < S.fwd_levels.size(); l++) run(0, l, S.fwd_levels[l], F.by_height[l]);
  run(2, 0, S.root_level, every);
  for (size_t l = 0; l < S.bwd_levels.size(); l++) run(1, l, S.bwd_levels[l], F.by_depth[l]);
  fprintf(stderr, "[spd] dof %d total %.1f MB %.1f us %.0f GB/s (launches timed one by one)\n", S.dof, tot_mb, tot_us, tot_mb / tot_us * 1e3);
  if (trace) {
    spd_trace_set(nullptr);
    HIP_CHECK(hipFree(trace));
  }
  HIP_CHECK(hipEventDestroy(e0));
  HIP_CHECK(hipEventDestroy(e1));
}

// (fronts of nodes outside the current mask are skipped: their entries of `out` stay as they are)
void Group::solve_tt(double *in, double *out, double scale) { spd_run(d_, st_, Ltt_, cur_mask_, in, out, scale, class_tt_); }
void Group::solve_rr(double *in, double *out, double scale) { spd_run(d_, st_, Lrr_, cur_mask_, in, out, scale, class_rr_); }

// X.t = -G_tt^-1 (g_t + G_tR X.R)    (DPGOProblem.h:275-294)
// Leaves T1_ = G [0 ; X.R] + g on all rows (its translation rows are the right-hand side of the solve):
// with the new translations, G X + g = T1_ + G_{:,t} X.t, which apply_tcol() adds at a quarter of the
// cost of another G X.
void Group::recover_translations(double *X, const double *g) {
  launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, X, true, g, T1_.p, nullptr, 0.0, nullptr, nullptr, 0);
  solve_tt(T1_.p, X, -1.0);
}

// y = base + G_{:,t} xt.t, with the row-local epilogues of launch_bsr_tcol
void Group::apply_tcol(const double *xt, const double *base, double *y, int mode, const double *X, const double *nabla,
                       const double *Rdot, double *out2, const double *rres, double *partials, const double *dg,
                       const double *dga, const double *ds, const double *dgrad, const double *dhs) {
  launch_bsr_tcol(d_, st_, T_, cur_mask_, G_.dev, G_.tcol.p, xt, base, y, mode, X, nabla, Rdot, out2, rres, partials, dg, dga,
                  ds, dgrad, dhs);
}

// partial[slot] = tr(X^T (g + 1/2 G X))     (DPGOProblem.cpp:180-205; + f on the host)
void Group::eval_G(const double *X, const double *g, int slot) {
  launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, X, false, nullptr, nullptr, X, 0.5, g, partials_.p, slot);
}

// ---------------------------------------------------------------------------
// layout conversion (reference column-major <-> pose records)
// ---------------------------------------------------------------------------
static void to_records(int d, int n, const double *X, int ld, int row_t0, int row_r0, double *rec) {
  // X rows [row_t0, row_t0+n) translations, [row_r0 + d k, ...) rotation blocks
  const int RS = (d + 1) * d;
  for (int k = 0; k < n; k++)
    for (int c = 0; c < d; c++) {
      rec[(size_t)k * RS + c] = X[(size_t)c * ld + row_t0 + k];
      for (int r = 0; r < d; r++) rec[(size_t)k * RS + d + r * d + c] = X[(size_t)c * ld + row_r0 + k * d + r];
    }
}
static void from_records(int d, int n, const double *rec, double *X, int ld, int row_t0, int row_r0) {
  const int RS = (d + 1) * d;
  for (int k = 0; k < n; k++)
    for (int c = 0; c < d; c++) {
      X[(size_t)c * ld + row_t0 + k] = rec[(size_t)k * RS + c];
      for (int r = 0; r < d; r++) X[(size_t)c * ld + row_r0 + k * d + r] = rec[(size_t)k * RS + d + r * d + c];
    }
}

int Group::initialize(int a, const double *X, int ld) {
  finish_update();
  zc_ready_ = false;   // (whatever an earlier iterate() left in the history buffers is overwritten here)
  packed_ = false;     // (... and whatever it packed for an exchange)
  if (a < 0 || a >= num_local()) return -1;
  const int n0 = info_[a].n[0], n1 = info_[a].n[1];
  if (ld < (d_ + 1) * (n0 + n1)) {
    fprintf(stderr, "[dpgo_amd] ERROR: initialize: inconsistent size of X for node %d.\n", nodes_[a]);
    return -1;
  }
  sync();
  std::vector<double> own((size_t)n0 * RS_), nbr((size_t)std::max(n1, 1) * RS_);
  to_records(d_, n0, X, ld, 0, n0, own.data());
  to_records(d_, n1, X, ld, (d_ + 1) * n0, (d_ + 1) * n0 + n1, nbr.data());
  for (double *dst : {Xk_.p, Zc_.p, Zp_.p}) {
    HIP_CHECK(hipMemcpy(dst + (size_t)own_off_[a] * RS_, own.data(), sizeof(double) * n0 * RS_, hipMemcpyHostToDevice));
    if (n1) HIP_CHECK(hipMemcpy(dst + (size_t)(P0_ + nbr_off_[a]) * RS_, nbr.data(), sizeof(double) * n1 * RS_, hipMemcpyHostToDevice));
  }
  HIP_CHECK(hipMemcpy(Xak_.p + (size_t)own_off_[a] * RS_, own.data(), sizeof(double) * n0 * RS_, hipMemcpyHostToDevice));
  for (double *dst : {gc_.p, gp_.p, Dfc_.p, Dfp_.p})
    HIP_CHECK(hipMemset(dst + (size_t)own_off_[a] * RS_, 0, sizeof(double) * n0 * RS_));
  res_[a] = NodeResults();
  res_[a].updated = 0;
  spec_refined_ = false;   // (a fresh start: nothing to guess the next iteration's refinements from)
  rescale_count_[a] = 0;   // DPGOResult::clear (DPGO_types.h:301); the scales belong to the problem and stay
  if (device_rescale_) HIP_CHECK(hipMemset(rs_count_.p + a, 0, sizeof(int)));
  return 0;
}

// Rows of a global X ((d+1)N x d, reference layout) that node a works on, as a (d+1)(n0+n1) x d matrix Z in
// the node's own ordering (dist_pgo.cpp:435-446 + DPGO::communicate)
void Group::node_rows_of_global(int a, const double *X, int ld, std::vector<double> &Z) const {
  const int N = num_poses_global_;
  const int n0 = info_[a].n[0], n1 = info_[a].n[1], rows = (d_ + 1) * (n0 + n1);
  Z.assign((size_t)rows * d_, 0.0);
  auto put = [&](int trow, int rrow, int gid) {
    for (int c = 0; c < d_; c++) {
      Z[(size_t)c * rows + trow] = X[(size_t)c * ld + gid];
      for (int r = 0; r < d_; r++) Z[(size_t)c * rows + rrow + r] = X[(size_t)c * ld + N + gid * d_ + r];
    }
  };
  for (int k = 0; k < n0; k++) put(k, n0 + k * d_, g_index_[a].at(info_[a].own_pose[k]));
  // neighbours: global id from the partition rule (DPGO_utils.cpp:147-158)
  const int q = N / num_nodes_total_, inc_n = N - num_nodes_total_ * q;
  for (int k = 0; k < n1; k++) {
    const int node = info_[a].nbr_key[k].first, pose = info_[a].nbr_key[k].second;
    const int start = node < inc_n ? node * (q + 1) : inc_n * (q + 1) + (node - inc_n) * q;
    put((d_ + 1) * n0 + k, (d_ + 1) * n0 + n1 + k * d_, start + pose);
  }
}

int Group::initialize_global(const double *X, int ld) {
  finish_update();
  const int N = num_poses_global_;
  if (ld < (d_ + 1) * N) return -1;
  std::vector<double> Z;
  for (int a = 0; a < num_local(); a++) {
    node_rows_of_global(a, X, ld, Z);
    if (initialize(a, Z.data(), (d_ + 1) * (info_[a].n[0] + info_[a].n[1])) != 0) return -1;
  }
  return 0;
}

// DPGOStar::evaluate_f / evaluate_grad at an arbitrary global X (C++/DPGO/src/DPGOStar.cpp:713-829), without
// touching the optimizer state.  Every node evaluates its intra edges and its inter edges (objective: charged 1/2
// per node; gradient: the rows of its own poses, which are DfobjE_top + (G - D) X = g + G X);
// F and |grad F|^2 are the sums over the nodes of this group, and over all groups when collectives are set.
// grad (optional): global (d+1)N x d, only the rows of this group's own poses are written.
int Group::evaluate_global(const double *X, int ld, double *F, double *grad_sqnorm, double *grad, int ldg) {
  finish_update();
  const int N = num_poses_global_;
  if (ld < (d_ + 1) * N || (grad && ldg < (d_ + 1) * N)) {
    fprintf(stderr, "[dpgo_amd] ERROR: evaluate: inconsistent size of X.\n");
    return -1;
  }
  join_exchange();
  sync();
  {
    std::vector<double> rec((size_t)(P0_ + P1_) * RS_), Z;
    for (int a = 0; a < num_local(); a++) {
      node_rows_of_global(a, X, ld, Z);
      const int n0 = info_[a].n[0], n1 = info_[a].n[1], rows = (d_ + 1) * (n0 + n1);
      to_records(d_, n0, Z.data(), rows, 0, n0, rec.data() + (size_t)own_off_[a] * RS_);
      to_records(d_, n1, Z.data(), rows, (d_ + 1) * n0, (d_ + 1) * n0 + n1, rec.data() + (size_t)(P0_ + nbr_off_[a]) * RS_);
    }
    HIP_CHECK(hipMemcpy(Tall_.p, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice));
  }
  std::vector<int> all(num_local());
  for (int a = 0; a < num_local(); a++) all[a] = a;
  set_mask(all);
  double *g = tmp_[0].p, *Df = tmp_[1].p, *gr = tmp_[2].p;
  if (opt_.loss == 0)   // g = S Z
    launch_bsr(d_, st_, T_, false, cur_mask_, S_.dev, Tall_.p, false, nullptr, g, nullptr, 0, nullptr, nullptr, 0);
  else                  // g = (B1^T W B1 Z)_own - D X   (weights at X)
    launch_inter(d_, st_, T_, cur_mask_, E_, opt_.loss, opt_.loss_reg, 1, false, Tall_.p, nullptr, nullptr, Dd_.p, nullptr, g,
                 partials_.p);
  launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Tall_.p, false, g, Df, nullptr, 0, nullptr, nullptr, 0);
  launch_cost(d_, st_, T_, cur_mask_, Ei_, E_, opt_.loss == 0, opt_.loss, opt_.loss_reg, Tall_.p, partials_.p, 0);
  launch_tangent_full(d_, st_, T_, cur_mask_, Tall_.p, Df, gr, partials_.p, 2);
  fetch(3, true);
  double v[2] = {0, 0};
  for (int a = 0; a < num_local(); a++) {
    v[0] += 0.5 * scal(a, 0) + 0.25 * scal(a, 1);
    v[1] += scal(a, 2);
  }
  if (coll_allreduce_ && coll_allreduce_(coll_user_, v, 2) != 0) return -1;
  if (F) *F = v[0];
  if (grad_sqnorm) *grad_sqnorm = v[1];
  if (grad) {
    std::vector<double> own((size_t)P0_ * RS_);
    HIP_CHECK(hipMemcpy(own.data(), gr, sizeof(double) * own.size(), hipMemcpyDeviceToHost));
    for (int a = 0; a < num_local(); a++)
      for (int k = 0; k < info_[a].n[0]; k++) {
        const int gid = g_index_[a].at(info_[a].own_pose[k]);
        const double *rec = &own[(size_t)(own_off_[a] + k) * RS_];
        for (int c = 0; c < d_; c++) {
          grad[(size_t)c * ldg + gid] = rec[c];
          for (int r = 0; r < d_; r++) grad[(size_t)c * ldg + N + gid * d_ + r] = rec[d_ + r * d_ + c];
        }
      }
  }
  return 0;
}

// DPGOHash::set_options (DPGOHash.h:93-96).  The reference swaps the optimizer's options and keeps the problem it
// built at construction; here the fields baked into the problem (operators, factorizations) must not change.
int Group::set_options(const Options &o) {
  finish_update();
  if (o.loss != opt_.loss || o.loss_reg != opt_.loss_reg || o.regularizer != opt_.regularizer || o.rescale != opt_.rescale ||
      o.preconditioner != opt_.preconditioner ||
      o.reg_Cholesky_precon_max_condition_number != opt_.reg_Cholesky_precon_max_condition_number) {
    fprintf(stderr, "[dpgo_amd] ERROR: set_options: loss, loss_reg, regularizer, rescale and the preconditioner are part of "
                    "the problem built at construction; create a new group to change them.\n");
    return -1;
  }
  if (o.preconditioner == 3 && Lrr_.F.n == 0 && o.max_iterations > 0 && o.max_iterations_accepted > 0) {
    fprintf(stderr, "[dpgo_amd] ERROR: set_options: the group was created without refinement (max_iterations = 0), so "
                    "the preconditioner was never factorised.\n");
    return -1;
  }
  opt_ = o;
  graphs_invalidate();   // (captured launches carry tolerances and iteration limits by value)
  spec_refined_ = false;
  return 0;
}

int Group::get_Xk(int a, double *X, int ld) const {
  if (a < 0 || a >= num_local()) return -1;
  sync();
  const int n0 = info_[a].n[0], n1 = info_[a].n[1];
  std::vector<double> own((size_t)n0 * RS_), nbr((size_t)std::max(n1, 1) * RS_);
  HIP_CHECK(hipMemcpy(own.data(), Xk_.p + (size_t)own_off_[a] * RS_, sizeof(double) * n0 * RS_, hipMemcpyDeviceToHost));
  if (n1) HIP_CHECK(hipMemcpy(nbr.data(), Xk_.p + (size_t)(P0_ + nbr_off_[a]) * RS_, sizeof(double) * n1 * RS_, hipMemcpyDeviceToHost));
  from_records(d_, n0, own.data(), X, ld, 0, n0);
  from_records(d_, n1, nbr.data(), X, ld, (d_ + 1) * n0, (d_ + 1) * n0 + n1);
  return 0;
}

int Group::get_X_own(int a, double *X, int ld) const {
  if (a < 0 || a >= num_local()) return -1;
  sync();
  const int n0 = info_[a].n[0];
  std::vector<double> own((size_t)n0 * RS_);
  HIP_CHECK(hipMemcpy(own.data(), Xak_.p + (size_t)own_off_[a] * RS_, sizeof(double) * n0 * RS_, hipMemcpyDeviceToHost));
  from_records(d_, n0, own.data(), X, ld, 0, n0);
  return 0;
}

int Group::scatter_global(double *X, int ld) const {
  sync();
  const int N = num_poses_global_;
  std::vector<double> own((size_t)P0_ * RS_);
  HIP_CHECK(hipMemcpy(own.data(), Xk_.p, sizeof(double) * P0_ * RS_, hipMemcpyDeviceToHost));
  for (int a = 0; a < num_local(); a++) {
    for (int k = 0; k < info_[a].n[0]; k++) {
      const int gid = g_index_[a].at(info_[a].own_pose[k]);
      const double *rec = &own[(size_t)(own_off_[a] + k) * RS_];
      for (int c = 0; c < d_; c++) {
        X[(size_t)c * ld + gid] = rec[c];
        for (int r = 0; r < d_; r++) X[(size_t)c * ld + N + gid * d_ + r] = rec[d_ + r * d_ + c];
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// halo exchange
// ---------------------------------------------------------------------------
int Group::communicate_local() {
  // neighbour rows whose owner lives in this group: one indexed device copy (DPGOHash.h:64-82)
  if (gather_dst_.n == 0) return 0;
  if (spec_upd_.on) return 0;   // (enqueued ahead, under the gate: speculate_update)
  if (pending_tail_.on) {
    // Xk's own rows are still on their way (they ride on the next update()'s product with G): the neighbour rows come
    // from Xak, which holds the same records
    const double *src = pending_tail_.xak;
    defer_or_launch(0x6c6f63ull, [this, src] { launch_copy_indexed(d_, st_, (int)gather_dst_.n, gather_dst_.p, gather_src_.p, src, Xk_.p); });
    return 0;
  }
  defer_or_launch(0x6c6f63ull, [this] { launch_copy_indexed(d_, st_, (int)gather_dst_.n, gather_dst_.p, gather_src_.p, Xk_.p, Xk_.p); });
  return 0;
}

int Group::step(const std::vector<int> &locals, const std::function<int()> &exchange) {
  struct Disarm {   // (whatever happens in between -- an error return, an exception on its way to the C ABI -- nothing stays deferred)
    Group *g;
    ~Disarm() { g->defer_armed_ = false; g->tail_fusable_ = false; g->spec_update_armed_ = false; g->pending_tail_.on = false; g->deferred_.clear(); g->deferred_key_ = 0; }
  } disarm{this};
  defer_armed_ = !exchange && iter_graph_wanted();
  tail_fusable_ = !exchange;
  spec_update_armed_ = !exchange;
  int rc = iterate(locals);
  tail_fusable_ = false;
  if (rc == 0 && exchange) rc = exchange();
  if (rc == 0) rc = communicate_local();
  defer_armed_ = false;
  if (rc == 0) rc = update(locals);
  flush_pending_tail();   // (nothing, unless update() had nothing to do)
  flush_deferred();
  return rc;
}

int Group::num_recv(int a, int beta) const {
  if (a < 0 || a >= num_local()) return -1;
  auto it = info_[a].recv.find(beta);
  return it == info_[a].recv.end() ? 0 : (int)it->second.size();
}
int Group::num_send(int a, int beta) const {
  if (a < 0 || a >= num_local()) return -1;
  auto it = info_[a].sent.find(beta);
  return it == info_[a].sent.end() ? 0 : (int)it->second.size();
}

int Group::receive(int a, int beta, const double *msg, int ld) {
  finish_update();
  if (a < 0 || a >= num_local()) return -1;
  auto it = info_[a].recv.find(beta);
  if (it == info_[a].recv.end()) {
    fprintf(stderr, "[dpgo_amd] ERROR: Can not find information for node %d\n", beta);   // DPGOHash.cpp:77
    return -1;
  }
  const int np = (int)it->second.size();
  if (ld < (d_ + 1) * np) return -1;
  sync();
  // the poses of one neighbour occupy consecutive neighbour rows (ordering of generate_data_info)
  std::vector<double> rec((size_t)np * RS_);
  for (int k = 0; k < np; k++)
    for (int c = 0; c < d_; c++) {
      rec[(size_t)k * RS_ + c] = msg[(size_t)c * ld + k];
      for (int r = 0; r < d_; r++) rec[(size_t)k * RS_ + d_ + r * d_ + c] = msg[(size_t)c * ld + np + k * d_ + r];
    }
  const int first = it->second.front().second;
  HIP_CHECK(hipMemcpy(Xk_.p + (size_t)(P0_ + nbr_off_[a] + first) * RS_, rec.data(), sizeof(double) * rec.size(),
                      hipMemcpyHostToDevice));
  res_[a].updated = 0;
  return 0;
}

int Group::send(int a, int beta, double *msg, int ld) const {
  if (a < 0 || a >= num_local()) return -1;
  auto it = info_[a].sent.find(beta);
  if (it == info_[a].sent.end()) return -1;
  const int np = (int)it->second.size();
  if (ld < (d_ + 1) * np) return -1;
  sync();
  std::vector<double> own((size_t)info_[a].n[0] * RS_);
  HIP_CHECK(hipMemcpy(own.data(), Xk_.p + (size_t)own_off_[a] * RS_, sizeof(double) * own.size(), hipMemcpyDeviceToHost));
  for (int k = 0; k < np; k++) {
    const double *rec = &own[(size_t)it->second[k] * RS_];
    for (int c = 0; c < d_; c++) {
      msg[(size_t)c * ld + k] = rec[c];
      for (int r = 0; r < d_; r++) msg[(size_t)c * ld + np + k * d_ + r] = rec[d_ + r * d_ + c];
    }
  }
  return 0;
}

int Group::pack_sent(double *dev_buf, hipStream_t st) {
  launch_copy_indexed(d_, st ? st : st_, (int)sent_rows_.size(), nullptr, sent_rows_dev_.p, Xk_.p, dev_buf);
  return 0;
}

void Group::join_exchange() {
  if (!xchg_done_) return;
  HIP_CHECK(hipStreamWaitEvent(st_, xchg_done_, 0));
  xchg_done_ = nullptr;
}

int Group::set_recv_layout(int nranks, int stride, const int *counts, const int *nodes, const int *poses) {
  std::map<std::pair<int, int>, int> slot;
  int off = 0;
  for (int r = 0; r < nranks; r++) {
    for (int k = 0; k < counts[r]; k++) slot[{nodes[off + k], poses[off + k]}] = r * stride + k;
    off += counts[r];
  }
  std::vector<int> dst, src;
  for (int a = 0; a < num_local(); a++)
    for (int k = 0; k < info_[a].n[1]; k++) {
      const auto key = info_[a].nbr_key[k];
      if (local_of_node_.count(key.first)) continue;
      auto it = slot.find(key);
      if (it == slot.end()) {
        fprintf(stderr, "[dpgo_amd] ERROR: No information for pose [%d, %d].\n", key.first, key.second);
        return -1;
      }
      dst.push_back(P0_ + nbr_off_[a] + k);
      src.push_back(it->second);
    }
  recv_dst_.upload(dst);
  recv_src_.upload(src);
  return 0;
}

int Group::unpack_recv(const double *dev_gathered, hipStream_t st) {
  if ((!st || st == st_) && set_pending_recv(dev_gathered, (int)recv_dst_.n, recv_dst_.p, recv_src_.p) == 0) return 0;   // (lazily: below)
  launch_copy_indexed(d_, st ? st : st_, (int)recv_dst_.n, recv_dst_.p, recv_src_.p, dev_gathered, Xk_.p);
  return 0;
}

// A LAZY unpack: the neighbour rows an exchange on the group's own stream delivered stay in its receive buffer; the next
// update()'s inter-edge pass reads them from there and stores them into Xk on the way (kernels.h: InterEdgesDev::recv) -- no
// unpack kernel.  Whoever else looks at Xk's neighbour rows first (flush_pending_recv) gets the plain indexed copy.  The
// lists (dst: neighbour rows, src: slots of the buffer; device arrays that outlive the exchange) are digested once per list:
// a slot per neighbour row, and per incidence record the slot of its other pose.  Robust losses only (the trivial loss has
// no inter-edge pass); -1: not taken, the caller unpacks as before.
int Group::set_pending_recv(const double *buf, int count, const int *dst_dev, const int *src_dev) {
  flush_pending_recv();
  static const bool lazy = env_int("DPGO_LAZY_UNPACK", 1) != 0;   // (A/B hook)
  if (!lazy || !fused_ || opt_.loss == 0 || star_ || count <= 0 || e_rec_host_.empty()) return -1;
  if (recv_key_ != dst_dev || recv_count_ != count) {
    std::vector<int> dst(count), src(count), nsrc((size_t)std::max(P1_, 1), -1);
    HIP_CHECK(hipMemcpy(dst.data(), dst_dev, sizeof(int) * count, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(src.data(), src_dev, sizeof(int) * count, hipMemcpyDeviceToHost));
    for (int k = 0; k < count; k++) {
      if (dst[k] < P0_ || dst[k] >= P0_ + P1_) return -1;   // (not a neighbour row: not ours to interpret)
      nsrc[dst[k] - P0_] = src[k];
    }
    std::vector<InterInc> rec = e_rec_host_;
    for (auto &r : rec) r.osrc = r.other >= P0_ ? nsrc[r.other - P0_] : -1;
    graphs_invalidate();   // (captured launches carry the records' address -- unchanged -- but let nothing replay mid-upload)
    HIP_CHECK(hipMemcpyAsync(e_rec_.p, rec.data(), sizeof(InterInc) * rec.size(), hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    recv_nsrc_.upload(nsrc);
    recv_key_ = dst_dev; recv_count_ = count;
    recv_dst_dev_ = dst_dev; recv_src_dev_ = src_dev;
  }
  pending_recv_ = buf;
  return 0;
}

void Group::flush_pending_recv() {
  if (!pending_recv_) return;
  const double *buf = pending_recv_;
  pending_recv_ = nullptr;
  launch_copy_indexed(d_, st_, recv_count_, recv_dst_dev_, recv_src_dev_, buf, Xk_.p);
}

void Group::needed_keys(std::vector<std::pair<int, int>> &keys, std::vector<int> &rows) const {
  keys.clear();
  rows.clear();
  for (int a = 0; a < num_local(); a++)
    for (int k = 0; k < info_[a].n[1]; k++) {
      const auto key = info_[a].nbr_key[k];
      if (local_of_node_.count(key.first)) continue;
      keys.push_back(key);
      rows.push_back(P0_ + nbr_off_[a] + k);
    }
}

void Group::copy_records(hipStream_t st, int count, const int *didx, const int *sidx, const double *src, double *dst) const {
  launch_copy_indexed(d_, st, count, didx, sidx, src, dst);
}

int Group::set_collectives(double *send_dev, double *gathered_dev, AllGatherFn ag, AllReduceFn ar, void *user) {
  if ((ag && (!send_dev || !gathered_dev)) || (ag && !ar)) return -1;
  coll_send_ = send_dev;
  coll_gathered_ = gathered_dev;
  coll_allgather_ = ag;
  coll_allreduce_ = ar;
  coll_allreduce_dev_ = nullptr;   // (belongs to whoever lends the collectives: set again by set_device_allreduce)
  coll_user_ = user;
  return 0;
}

// ---------------------------------------------------------------------------
// DPGOHash::update  (DPGOHash.cpp:84-228)
// ---------------------------------------------------------------------------
// The part of the scalar logic that does not need the numbers update() reads back: the Nesterov sequence s[iter],
// s[iter+1] and gamma (DPGOHash.cpp:150-160, 206-216).  The next iterate() may start with it.
void Group::host_update_pre(int a) {
  NodeResults &r = res_[a];
  const int it = r.iters;
  r.pre_repeat = (r.hist_iter == it);
  r.pre_done = true;
  if (opt_.scheme == 1) {
    if (it == 0) r.s0 = 1.0;
    else if (!r.pre_repeat) r.s0 = r.s1;
    r.s1 = 0.5 + 0.5 * std::sqrt(4.0 * r.s0 * r.s0 + 1.0);
    r.gamma = (r.s0 - 1) / r.s1;
  } else {
    r.gamma = 0;
  }
}

void Group::host_update_logic(int a, double fobj, double f, double gradFnorm) {
  NodeResults &r = res_[a];
  const Options &o = opt_;
  const int it = r.iters;
  // update() may run again at the same iteration (update -> receive() -> update): X[iter-1], fobj[iter-1] and
  // s[iter] are those of the first call, everything else is re-done as the reference does (DPGOHash.cpp:99-225)
  const bool pre = r.pre_done;   // the Nesterov sequence was already advanced by host_update_pre
  const bool repeat = pre ? r.pre_repeat : (r.hist_iter == it);
  r.pre_done = false;
  r.hist_iter = it;
  if (!repeat) r.fobj_prev = r.fobj;
  r.fobj = fobj;
  r.f = f;
  r.gradFnorm = gradFnorm;
  if (star_) {   // update_n (DPGOStar.cpp:339-385): no restart counters, Gk = Fk = fobj every iteration
    r.Gk = fobj;
    if (o.scheme == 1) {
      if (!repeat) r.s0 = it == 0 ? 1.0 : r.s1;
      r.s1 = 0.5 + 0.5 * std::sqrt(4.0 * r.s0 * r.s0 + 1.0);
      r.gamma = (r.s0 - 1) / r.s1;
    }
    r.Fk[0] = r.Fk[1] = fobj;
    r.updated = 1;
    return;
  }
  if (it == 0) {
    r.Fk[0] = r.Fk[1] = fobj;
    r.Gk = fobj;
  }
  if (o.scheme == 1) {
    if (it == 0) {
      if (!pre) r.s0 = 1.0;
      if (!repeat) r.oscillations.assign(1, 1);
      else r.oscillations.push_back(1);   // the reference pushes again (DPGOHash.cpp:168-171)
    } else if (!repeat && !pre) {
      r.s0 = r.s1;
    }
    if (!pre) {
      r.s1 = 0.5 + 0.5 * std::sqrt(4.0 * r.s0 * r.s0 + 1.0);
      r.gamma = (r.s0 - 1) / r.s1;
    }
    if (fobj <= r.Fk[1]) r.soft_restart_hits[0] = r.soft_restart_hits[0] > 2 ? r.soft_restart_hits[0] - 2 : 0;
    else r.soft_restart_hits[0]++;
    if (it > 0) {
      if (fobj <= r.fobj_prev) { r.soft_restart_hits[1] = 0; r.oscillations.push_back(1); }
      else { r.soft_restart_hits[1]++; r.oscillations.push_back(0); }
      r.num_oscillations += (r.oscillations[it] != r.oscillations[it - 1]);
    }
    if (it > o.oscillation_cnt_period) {
      const int k = it - o.oscillation_cnt_period;
      r.num_oscillations -= (r.oscillations[k] != r.oscillations[k - 1]);
    }
    r.Fk[0] = r.Fk[0] * (1 - o.eta[0]) + fobj * o.eta[0];
    r.Fk[1] = std::max(fobj, r.Fk[1] * (1 - o.eta[1]) + fobj * o.eta[1]);
  } else {
    r.Fk[0] = r.Fk[1] = fobj;
    if (!pre) r.gamma = 0;
  }
  r.updated = 1;
}

// The rescale test of evaluate_g_and_f*_rescale (DPGOProblem.cpp:300-321, 464-485) for the nodes of `set`: a node
// is rescaled when its counter has reached max_rescale_count or some edge weight exceeds the edge's scale; its new
// scales are clamp(1.25 w, min_rescale_, max_rescale_) (DPGOProblem.h:17-18), update_quadratic_mat (:751-840) and
// L_.factorize follow.  The preconditioner keeps the factor of the constructor, as in the reference.
std::vector<int> Group::maybe_rescale(const std::vector<int> &set) {
  SetupClock clk;   // (DPGO_SETUP_TIMING=1)
  std::vector<int> changed;
  std::vector<double> w(std::max<size_t>(e_w_.n, 1));
  sync();
  if (E_.m > 0) HIP_CHECK(hipMemcpy(w.data(), e_w_.p, sizeof(double) * E_.m, hipMemcpyDeviceToHost));
  for (int a : set) {
    const int m1 = e_off_[a + 1] - e_off_[a];
    bool rescaled = rescale_count_[a] >= opt_.max_rescale_count;
    for (int e = 0; e < m1 && !rescaled; e++) rescaled = w[e_off_[a] + e] > scale_[a][e];
    if (!rescaled) {
      rescale_count_[a]++;
      continue;
    }
    for (int e = 0; e < m1; e++) scale_[a][e] = std::min(1.0, std::max(0.01, 1.25 * w[e_off_[a] + e]));
    rescale_count_[a] = 0;
    changed.push_back(a);
  }
  if (!changed.empty()) {
    int bad = 0;
    const int nc = (int)changed.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(nc, host_threads()))) reduction(+ : bad)
    for (int i = 0; i < nc; i++) {
      const int a = changed[i];
      bad += assemble_node(info_[a], opt_.regularizer, false, ops_[a], scale_[a].data()) != 0;
    }
    if (bad) throw DeviceError("assemble_node");
    clk.lap("rescale: weights + host assembly");
    upload_operators();
    clk.lap("rescale: operators to the device");
    if (refactor_tt() != 0) throw DeviceError("G_tt is not positive definite after a rescale");
  }
  return changed;
}

// What the device-side rescale needs beside the scales: the diagonal blocks of G and of the proximal majoriser H with
// every scale at zero (the intra-node part + the regulariser), and where the diagonal block of every own pose sits in
// the uploaded block values of G.
void Group::setup_device_rescale() {
  const int L = num_local(), BB = B_ * B_;
  std::vector<double> Gb((size_t)std::max(P0_, 1) * BB, 0.0), Hb((size_t)std::max(P0_, 1) * BB, 0.0);
  std::vector<int> gpos((size_t)std::max(P0_, 1) * 4, 0);
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(L, host_threads()))) reduction(+ : bad)
  for (int a = 0; a < L; a++) {
    NodeOperators base;
    const std::vector<double> zeros(std::max<size_t>(info_[a].inter.size(), 1), 0.0);
    if (assemble_node(info_[a], opt_.regularizer, false, base, zeros.data()) != 0) { bad++; continue; }
    for (int r = 0; r < info_[a].n[0]; r++) {
      for (int k = base.G.ptr[r]; k < base.G.ptr[r + 1]; k++)
        if (base.G.col[k] == r) std::copy(&base.G.val[(size_t)k * BB], &base.G.val[(size_t)(k + 1) * BB], &Gb[(size_t)(own_off_[a] + r) * BB]);
      std::copy(&base.Hd[(size_t)r * BB], &base.Hd[(size_t)(r + 1) * BB], &Hb[(size_t)(own_off_[a] + r) * BB]);
    }
  }
  if (bad) throw DeviceError("assemble_node");
  // the unified block-CSR of G (upload_bsr): rows of node a at own_off_[a], its blocks in the order of ops_[a].G
  int kuni = 0;
  for (int a = 0; a < L; a++) {
    const BsrMatrix &M = ops_[a].G;
    for (int r = 0; r < M.nrows; r++) {
      const int k0row = kuni, cntrow = M.ptr[r + 1] - M.ptr[r];
      for (int k = M.ptr[r]; k < M.ptr[r + 1]; k++)
        if (M.col[k] == r) {
          const int kk = k - M.ptr[r], round0 = k0row + (kk / BSR_LPR) * BSR_LPR;
          int *g = &gpos[(size_t)(own_off_[a] + r) * 4];
          g[0] = round0 * BB;
          g[1] = std::min(BSR_LPR, k0row + cntrow - round0);
          g[2] = kk % BSR_LPR;
          g[3] = k0row + kk;
        }
      kuni += cntrow;
    }
  }
  Gbase_.upload(Gb);
  Hbase_.upload(Hb);
  gpos_.upload(gpos);
  device_rescale_ = true;
}

// The decision of k_rescale_decide has arrived (h_rs_): rebuild the block-diagonal terms of the rescaled nodes on the
// device, re-factor G_tt from the values that are already there, cut the solve's panels again.
std::vector<int> Group::rescale_device(const std::vector<int> &set) {
  SetupClock clk;   // (DPGO_SETUP_TIMING=1)
  std::vector<int> changed;
  for (int a : set)
    if (h_rs_[a] != 0.0) changed.push_back(a);
  if (changed.empty()) return changed;
  RescaleArgs A;
  A.flags = rs_flags_.p; A.scale = e_scale_.p; A.Gbase = Gbase_.p; A.Hbase = Hbase_.p; A.gpos = gpos_.p; A.att_pos = att_pos_.p;
  A.Gval = G_.val.p; A.Gtcol = G_.tcol.p; A.Dd = Dd_.p; A.Qd = Qd_.p; A.Tinv = Tinv_.p; A.N = N_.p; A.V = V_.p;
  A.att_val = spd_numeric_values(Ltt_.F);
  A.xi = opt_.regularizer;
  launch_rescale_apply(d_, st_, T_, E_, A);
  // everything on the group's stream, enqueued in one go: the new values, the factorisation, the panels (three waits
  // before, with the GPU idle across each)
  if (spd_refactor_device(Ltt_.F, (void *)st_, true) != 0) throw DeviceError("G_tt: refactorisation could not be enqueued");
  if (Ltt_.repack(st_) != 0) throw DeviceError("repack");
  // the verdict is read at the next read-back that was enqueued behind the factorisation: nothing waits for it here
  tt_verdict_pending_ = true;
  tt_verdict_seq_ = fetch_seq_ + 1;
  if (clk.on) {
    sync();
    fprintf(stderr, "[setup] rescale: %zu of %zu nodes\n", changed.size(), set.size());
  }
  clk.lap("rescale: block-diagonal terms, numeric factorisation of G_tt, panels (device)");
  return changed;
}

int Group::update(const std::vector<int> &locals_in) {
  InLib in_lib(this);
  if (failed_) { flush_pending_tail(); flush_deferred(); pending_recv_ = nullptr; return -1; }
  finish_update();
  std::vector<int> locals;
  for (int a : locals_in)
    if (!res_[a].updated) locals.push_back(a);
  if (locals.empty()) {
    flush_pending_tail();
    flush_deferred();
    flush_pending_recv();
    join_exchange();   // a pending exchange must still be ordered before whatever the caller does next on this stream
    return 0;
  }
  const bool trivial = (opt_.loss == 0);
  // update()'s partial sums have slots of their own (UPD_SLOT0 ..): they may wait there for the next refinement's
  // k_cg_scal_begin to reduce them (`lazy` below) while that refinement's passes use the first slots.  Not with Dynamic
  // rescale (its fetch() in the middle reads the first slots) nor when parked sums ride along (they are in the first slots)
  double *const pupd = (dynamic() || deferred_slots_ != 0) ? partials_.p : partials_.p + (size_t)UPD_SLOT0 * T_.nseg_all;
  host_bound_tick();
  set_mask(locals);
  // history: X[iter-1] <- X[iter], X[iter] <- Xk ; same for g and Dfobj (masked nodes only).  A node whose
  // history already stands at this iteration (update() ran, then receive() cleared `updated`) only refreshes
  // X[iter]: the reference overwrites X[iter] / g[iter] in place and leaves X[iter-1] alone (DPGOHash.cpp:99-106).
  std::vector<int> adv;
  for (int a : locals)
    if (res_[a].hist_iter != res_[a].iters) adv.push_back(a);
  bool zc_done = false;
  NodeBits mask_locals_bits = 0;
  for (int a : locals) mask_locals_bits |= 1ull << a;
  // (launches that wait for this update()'s first segment -- step() -- go now if something eager comes before it)
  if ((int)adv.size() != num_local() || !zc_ready_ || xchg_done_ || dynamic() || star_) flush_deferred();
  const double *lazy_recv = nullptr;   // (robust losses: the receive buffer the inter-edge pass unpacks on the way)
  // the tail of iterate() rides on the product with G (group.h: PendingTail) where that product reads the very records the
  // tail copies: every node advances, the copy's second target is the buffer that becomes X[iter] below
  bool fuse_copy = false;
  if (pending_tail_.on) {
    fuse_copy = !trivial && (int)adv.size() == num_local() && zc_ready_ && !xchg_done_ && !star_ && pending_tail_.m.v == mask_locals_bits &&
                pending_tail_.z == Zp_.p && pending_tail_.xk == Xk_.p && pending_tail_.xak == Xak_.p;
    if (!fuse_copy) flush_pending_tail();
  }
  if ((int)adv.size() == num_local()) {
    // every node advances: rotate the buffers instead of copying them
    Zp_.swap(Zc_);
    gp_.swap(gc_);
    Dfp_.swap(Dfc_);
    if (keep_gx()) GXp_.swap(GXc_);
    zc_done = zc_ready_;   // iterate() already left Xk's own rows in what is X[iter] now
  } else if (!adv.empty()) {
    set_mask(adv);
    copy_rows(Zp_.p, Zc_.p, true);
    copy_rows(gp_.p, gc_.p, false);
    copy_rows(Dfp_.p, Dfc_.p, false);
    if (keep_gx()) copy_rows(GXp_.p, GXc_.p, false);
    set_mask(locals);
  }
  // The new linearisation point needs the neighbours' poses, which may still be on their way (an exchange on the
  // communicator's stream, comm.cpp).  What needs no neighbour row goes first -- X[iter] own rows and the product
  // with G, a third of the surrogate build -- then the stream waits for the exchange and takes the neighbour rows.
  std::vector<int> first, later;
  for (int a : locals) ((res_[a].iters == 0 || star_) ? first : later).push_back(a);
  // the closing read-back is deferred to the next reader (finish_update) where there is exactly one of them and nothing
  // depends on it at once: not for AMM-PGO* (the master decides on the sums right away) nor with Dynamic rescale
  static const bool defer_enabled = env_int("DPGO_DEFER_UPDATE", 1) != 0;
  const bool can_defer = defer_enabled && !star_ && !dynamic() && (first.empty() != later.empty());
  // `launches`: the rest of the surrogate build of the nodes in `set`, ending with the reduction of its sums -- a branch-free
  // sequence, replayed from a captured graph where the host's launch rate would bound it (segment()); it may be empty when
  // the caller has already enqueued everything but the reduction
  auto end_with = [&](int seg_id, unsigned long long variant, int nslots, const std::vector<int> &set,
                      const std::function<void()> &launches, std::function<void()> logic) {
    nslots = std::max(nslots, deferred_slots_);
    deferred_slots_ = 0;
    NodeBits bits = 0;
    for (int a : set) bits |= 1ull << a;
    // the closing reduction is left to the next refinement's k_cg_scal_begin (group.h: UpdLazy) where the read-back is
    // deferred anyway and the launches are eager: one launch less on the stream
    // (only where the next iterate() starts its refinement unasked -- every node was refined in this one: otherwise the host
    // wants these sums before it enqueues anything that could carry them)
    bool lazy = can_defer && lazy_update_reduce() && spec_refined_ && pupd != partials_.p && nslots <= 6 && bits == all_bits();
    // (launches that went out ahead decided for themselves: the policy may have changed since -- host_bound_tick above)
    if (spec_upd_.on) lazy = spec_upd_.lazy;
    if (spec_upd_.on) {
      // the launches of this sequence went out ahead of the host's decision (speculate_update) and the decision was the
      // common one: what they were given must be what this call would have given them
      const SpecUpdate sp = spec_upd_;
      spec_upd_ = SpecUpdate();
      const bool same = seg_id == 4 && fuse_copy && nslots == 6 && bits == all_bits() && !xchg_done_ && !lazy_recv && sp.seq_last == fetch_seq_ && can_defer && pupd != partials_.p &&
                        sp.xak == Xak_.p && sp.zc == Zc_.p && sp.gc == gc_.p && sp.dfc == Dfc_.p && sp.gx == GXc_.p && pending_tail_.on;
      if (!same) {
        failed_ = true;
        fprintf(stderr, "[dpgo_amd] ERROR: a speculative update was enqueued for another state than update() found (segment %d, copy %d, slots %d, "
                        "nodes %d, exchange %d, receive %d, flags %llu / %llu, deferred %d, own slots %d, tail %d, buffers %d %d %d %d %d)\n",
                seg_id, (int)fuse_copy, nslots, (int)(bits == all_bits()), (int)(xchg_done_ != nullptr), (int)(lazy_recv != nullptr), sp.seq_last,
                fetch_seq_, (int)can_defer, (int)(pupd != partials_.p), (int)pending_tail_.on, (int)(sp.xak == Xak_.p), (int)(sp.zc == Zc_.p),
                (int)(sp.gc == gc_.p), (int)(sp.dfc == Dfc_.p), (int)(sp.gx == GXc_.p));
        throw DeviceError("a speculative update was enqueued for another state than update() found");
      }
      pending_tail_.on = false;   // (it rode on the enqueued product with G)
    } else
    segment(seg_id, bits & mask_locals_bits, {bits, mask_locals_bits, variant, (unsigned long long)nslots, fuse_copy ? 1ull : 0ull, (unsigned long long)(uintptr_t)lazy_recv}, [&] {
      launches();
      if (!lazy) launch_reduce(st_, T_, num_local(), true, nslots, pupd, h_upd_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
    });
    if (lazy) { upd_lazy_.pending = true; upd_lazy_.nslots = nslots; }
    if (can_defer) {
      pending_seq_ = lazy ? 0ull : fetch_seq_;   // (lazy: whoever launches the reduction sets it -- run_tnt, or finish_update)
      for (int a : set) {
        host_update_pre(a);
        res_[a].updated = 1;
      }
      pending_update_ = std::move(logic);
    } else {
      wait_flag(fetch_seq_);
      logic();
    }
  };
  zc_ready_ = false;
  if (!zc_done) copy_rows(Zc_.p, Xk_.p, false);
  double *GX = (!trivial && keep_gx()) ? GXc_.p : T1_.p;
  // the product with G: the part of the build that needs no neighbour row, ahead of the exchange's arrival.  Without a
  // pending exchange it is simply the head of the segment below.
  const bool split = xchg_done_ != nullptr;
  auto product_with_G = [&] {
    if (trivial)   // T1 = G Xak and <Xak, 1/2 G Xak>   (half of evaluate_G, DPGOProblem.cpp:180-205)
      launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Xak_.p, false, nullptr, T1_.p, Xak_.p, 0.5, nullptr, pupd, 5);
    else if (fuse_copy) {   // ... on Xak's records (the same numbers), which go to Xk and X[iter] on the way
      const PendingTail pt = pending_tail_;
      pending_tail_.on = false;
      launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, pt.xak, false, nullptr, GX, pt.xak, 0.5, nullptr, pupd, 5, pt.xk, pt.z);
    } else         // T1 = G X and <X, 1/2 G X>  (kept as G X[k] where the next extrapolation reuses it)
      launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Zc_.p, false, nullptr, GX, Zc_.p, 0.5, nullptr, pupd, 5);
  };
  const NodeMask mask_locals = cur_mask_;
  if (split) {
    product_with_G();
    join_exchange();
  }
  auto head = [&] {   // (what a segment starts with when the product has not gone ahead)
    if (!split) { cur_mask_ = mask_locals; product_with_G(); }
  };
  if (trivial) {
    flush_pending_recv();
    // X[iter]'s neighbour rows <- Xk's: a launch of its own for the trivial loss; the robust losses' inter-edge pass does it
    // on the way (it reads the neighbour rows from Xk and stores them)
    // g = S Z  (evaluate_none_g_and_f0 / _f, DPGOProblem.cpp:269-287, 516-542), with <Xak, g> alongside
    auto common = [&] {
      head();
      cur_mask_ = mask_locals;
      launch_copy_nbr_rows(d_, st_, T_, cur_mask_, Xk_.p, Zc_.p);
      launch_bsr(d_, st_, T_, false, cur_mask_, S_.dev, Zc_.p, false, nullptr, gc_.p, Xak_.p, 1.0, nullptr, pupd, 1);
    };
    const bool both = !first.empty() && !later.empty();
    if (both) flush_deferred();
    if (both) common();   // (nodes at different iterations: two read-backs, nothing deferred, the shared part goes first)
    if (!first.empty()) {
      end_with(1, (split ? 1ull : 0ull) | (both ? 2ull : 0ull), 6, first, [&] {
        if (!both) common();
        set_mask(first);
        launch_bsr(d_, st_, T_, true, cur_mask_, P0m_.dev, Zc_.p, false, nullptr, nullptr, Zc_.p, 0.5, nullptr, pupd, 0);
        // fobj = G(Xak | g, f0) = f0 + <Xak, g> + <Xak, 1/2 G Xak>: slots 1 and 5; Dfobj = g + G Xak
        launch_tangent_full(d_, st_, T_, cur_mask_, Xak_.p, T1_.p, nullptr, pupd, 2, gc_.p, Dfc_.p);
      }, [this, first] {
        for (int a : first) {
          const double f0 = uscal(a, 0);
          host_update_logic(a, f0 + (uscal(a, 1) + uscal(a, 5)), f0, std::sqrt(uscal(a, 2)));
        }
      });
    }
    if (!later.empty()) {
      end_with(2, (split ? 1ull : 0ull) | (both ? 2ull : 0ull), 4, later, [&] {
        if (!both) common();
        set_mask(later);
        launch_axpby(d_, st_, T_, true, cur_mask_, 1.0, Zc_.p, -1.0, Zp_.p, Tall_.p, 0);
        launch_bsr(d_, st_, T_, true, cur_mask_, Q_.dev, Tall_.p, false, nullptr, nullptr, Tall_.p, 0.5, nullptr, pupd, 0);
        launch_bsr(d_, st_, T_, true, cur_mask_, P_.dev, Zc_.p, false, nullptr, nullptr, Zc_.p, 0.5, nullptr, pupd, 3);
        launch_tangent_full(d_, st_, T_, cur_mask_, Xak_.p, T1_.p, nullptr, pupd, 2, gc_.p, Dfc_.p);
      }, [this, later] {
        for (int a : later) {
          const double fobj = res_[a].Gk + uscal(a, 0);
          host_update_logic(a, fobj, fobj + uscal(a, 3), std::sqrt(uscal(a, 2)));
        }
      });
    }
  } else {
    // evaluate_g_and_f0 / evaluate_g_and_f (DPGOProblem.cpp:222-267, 360-424); _rescale variants (:289-358, :426-514)
    const bool both = !first.empty() && !later.empty();
    // a lazy unpack is taken by the one inter-edge pass that covers every node of the group (it delivers all neighbour rows
    // at once); anything else gets the plain copy first
    if (pending_recv_) {
      if (!both && !dynamic() && mask_locals_bits == all_bits()) { lazy_recv = pending_recv_; pending_recv_ = nullptr; }
      else flush_pending_recv();
    }
    if (both || dynamic()) {   // (the product covers every node of `locals`: it cannot sit inside one of two segments)
      flush_deferred();
      head();
    }
    const bool head_inside = !(both || dynamic());
    for (int pass = 0; pass < 2; pass++) {
      const std::vector<int> &set = pass == 0 ? first : later;
      if (set.empty()) continue;
      std::vector<double> rho(num_local(), 0.0), gap(num_local(), 0.0);
      // fz: what the pass does on the way (Dfobj and |grad F|^2)
      auto inter_pass = [&](const InterFuse *fz) {
        InterEdgesDev E = E_;
        if (lazy_recv) { E.recv = lazy_recv; E.nsrc = recv_nsrc_.p; }
        launch_inter(d_, st_, T_, cur_mask_, E, opt_.loss, opt_.loss_reg, 0, pass == 1, Zc_.p, Zp_.p, Qd_.p, Dd_.p, DfE_.p,
                     gc_.p, pupd, dynamic() ? e_w_.p : nullptr, nullptr, nullptr, nullptr, nullptr, Xk_.p, nullptr, fz);   // slots 0, 1 and 2 = <X, g>
      };
      if (dynamic()) {
        set_mask(set);
        inter_pass(nullptr);
        if (device_rescale_)   // the rescale test on the weights just computed; its verdict rides with the sums below
          launch_rescale_decide(st_, num_local(), cur_mask_.v, e_off_dev_.p, e_w_.p, e_scale_.p, rs_count_.p, opt_.max_rescale_count,
                                rs_flags_.p, h_rs_);
        // Rescale::Dynamic: the sum of rho and the majorisation gap (under the OLD Q) are final; whether the
        // surrogate is rescaled depends on the edge weights just computed (:300-321, :464-485).  Rescaled nodes get
        // their D, G, T, N, V, Q and the factor of G_tt rebuilt, and g, G X are taken again with the new operators.
        fetch(3, true);
        for (int a : set) { rho[a] = scal(a, 0); gap[a] = scal(a, 1); }
        const std::vector<int> changed = device_rescale_ ? rescale_device(set) : maybe_rescale(set);
        if (!changed.empty()) {
          set_mask(changed);
          launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Zc_.p, false, nullptr, GX, Zc_.p, 0.5, nullptr, pupd, 5);
          launch_inter(d_, st_, T_, cur_mask_, E_, opt_.loss, opt_.loss_reg, 1, false, Zc_.p, nullptr, nullptr, Dd_.p, nullptr,
                       gc_.p, pupd);   // g = DfobjE_own - D X with the new D (slot 2 = <X, g> again)
          set_mask(set);
        }
      }
      // (a node's first update: there is no X[k-1] yet -- gamma is 0 there, but the buffer must hold numbers)
      std::vector<int> fresh;
      if (keep_gx())
        for (int a : set)
          if (res_[a].iters == 0) fresh.push_back(a);
      const bool dyn = dynamic();
      NodeBits fresh_bits = 0;
      for (int a : fresh) fresh_bits |= 1ull << a;
      end_with(3 + pass, (split ? 1ull : 0ull) | (head_inside ? 2ull : 0ull) | (dyn ? 4ull : 0ull) | (fused_ ? 8ull : 0ull) | (lazy_recv ? 16ull : 0ull) | (fresh_bits << 5), 6, set, [&] {
        if (head_inside) head();
        set_mask(set);
        // Dfobj = G X + g, its tangent projection and norm: inside the inter-edge pass (kernels.h: InterFuse), or k_tangent_full
        const bool in_pass = !dyn && fused_;
        if (!dyn) {
          InterFuse fz;
          fz.GX = GX; fz.X = Xak_.p; fz.Df = Dfc_.p; fz.gn_slot = 4;
          inter_pass(in_pass ? &fz : nullptr);
        }
        if (pass == 0) launch_bdiag_dot(d_, st_, T_, cur_mask_, Dd_.p, Zc_.p, 0.5, DfE_.p, -1.0, pupd, 3);
        if (!in_pass) launch_tangent_full(d_, st_, T_, cur_mask_, Xak_.p, GX, nullptr, pupd, 4, gc_.p, Dfc_.p);   // Dfobj = G X + g
        if (!fresh.empty()) {
          set_mask(fresh);
          copy_rows(GXp_.p, GXc_.p, false);
          set_mask(set);
        }
      }, [this, set, pass, dyn, rho, gap] {
        for (int a : set) {
          NodeResults &r = res_[a];
          const double fobjE = 0.5 * (dyn ? rho[a] : uscal(a, 0));
          const double quad = uscal(a, 2) + uscal(a, 5);   // tr(X^T (g + 1/2 G X))
          double fobj, f;
          if (pass == 0) {
            f = 0.5 * fobjE + uscal(a, 3);
            fobj = f + quad;
          } else {
            fobj = r.Gk - 0.5 * r.fobjE - 0.5 * (dyn ? gap[a] : uscal(a, 1)) + 0.5 * fobjE;
            f = fobj - quad;
          }
          r.fobjE = fobjE;
          host_update_logic(a, fobj, f, std::sqrt(uscal(a, 4)));
        }
      });
    }
  }
  flush_pending_tail();   // (nothing, unless the product with G never came)
  return 0;
}

// ---------------------------------------------------------------------------
// DPGOHash::iterate  (DPGOHash.cpp:583-628)
// ---------------------------------------------------------------------------
int Group::iterate(const std::vector<int> &locals) {
  InLib in_lib(this);
  if (failed_) return -1;
  for (int a : locals)
    if (!res_[a].updated) {
      fprintf(stderr, "[dpgo_amd] ERROR: The optimizer has not been updated (node %d).\n", nodes_[a]);
      return -1;
    }
  if (locals.empty()) return 0;
  join_exchange();   // (iterate -> exchange -> iterate without an update(): the pack must not race with the new Xk)
  const int rc = opt_.scheme == 1 ? amm(locals) : mm(locals);
  if (rc != 0) return rc;
  set_mask(locals);
  // Xk.top = Xak (:614).  When every node of the group iterated, the same pass also writes the buffer that the next
  // update() turns into X[iter] (it rotates the history buffers when every node advances; X[iter-1], which that buffer
  // holds now, has had its last reader), so update() need not copy Xk's own rows again.
  zc_ready_ = (int)locals.size() == num_local() && !star_;
  {
    // (the pointers' values of NOW: the launch may run as the head of update()'s first segment, after the history rotated)
    const NodeMask m = cur_mask_;
    const double *xak = Xak_.p;
    double *xk = Xk_.p, *z = zc_ready_ ? Zp_.p : nullptr;
    flush_pending_tail();   // (an older one nobody took: iterate() twice without an update())
    if (tail_fusable_ && fused_ && zc_ready_ && opt_.loss != 0) {
      // the next update()'s product with G takes it along (group.h: PendingTail)
      pending_tail_.on = true; pending_tail_.m = m; pending_tail_.xak = xak; pending_tail_.xk = xk; pending_tail_.z = z;
    } else if (pack_dst_ && !defer_armed_) {
      // an exchange follows (step()): its pack rides on this launch (kernels.h: launch_tail_pack), Comm::exchange() finds it done
      launch_tail_pack(d_, st_, T_, m, xak, xk, z, pack_rows_, pack_n_, pack_dst_);
      packed_ = true;
    } else
    defer_or_launch(0x7461696cull ^ m.v, [this, m, xak, xk, z] { launch_axpby(d_, st_, T_, false, m, 1.0, xak, 0.0, nullptr, xk, 0, z); });
  }
  for (int a : locals) {
    res_[a].iters++;
    res_[a].updated = 0;
  }
  return 0;
}

// DPGOHash::mm_pgo  (DPGOHash.cpp:446-581)
int Group::mm(const std::vector<int> &locals) {
  const Options &o = opt_;
  set_mask(locals);
  segment(12, cur_mask_.v, {}, [&] {
    launch_proximal(d_, st_, T_, cur_mask_, Zc_.p, Dfc_.p, Tinv_.p, N_.p, V_.p, Xakh_.p, nullptr, nullptr, 0);
    recover_translations(Xakh_.p, gc_.p);
    copy_rows(Xak_.p, Xakh_.p, false);
  });
  finish_update();   // the scalars of the last update(): needed from here on (the GPU has the launches above to chew on)
  std::vector<int> plain;
  for (int a : locals) {
    NodeResults &r = res_[a];
    r.refined = ((r.gradFnorm * r.gradFnorm / r.fobj) > o.accepted_delta) && o.max_iterations > 0 &&
                o.max_iterations_accepted > 0;
    if (!r.refined) plain.push_back(a);
  }
  {
    std::vector<int> ref;
    for (int a : locals)
      if (res_[a].refined) ref.push_back(a);
    if (!ref.empty()) run_tnt(ref, Xak_.p, gc_.p, nullptr, true);   // sets Gk
  }
  if (!plain.empty()) {
    set_mask(plain);
    eval_G(Xak_.p, gc_.p, 0);
    fetch(1, false);
    for (int a : plain) res_[a].Gk = scal(a, 0) + res_[a].f;
  }
  return 0;
}

// Y = X[k] + gamma (X[k] - X[k-1]) and the surrogate gradient data at Y, for the masked nodes
// (gam_dev: the same gammas in device memory -- the launches may be replayed from a captured graph, whose by-value
// arguments are frozen: amm())
// prox_slot >= 0: the caller's next step is Xakh = proximal(Y, Df) with |Xakh - Xak|^2 into that partial slot and Xak's
// rotations <- Xakh's (amm()); returns true when the inter-edge pass took it along (kernels.h: InterFuse::Xout)
bool Group::prepare_extrapolated(const double *gam_dev, int prox_slot) {
  const Options &o = opt_;
  const bool trivial = (o.loss == 0);
  NodeCoefs gam;
  for (int a = 0; a < num_local(); a++) gam.a[a] = gam.b[a] = res_[a].gamma;
  // own AND neighbour rows are extrapolated with the local gamma (DPGOHash.cpp:255-256)
  if (!trivial && fused_) {
    // ... inside the inter-edge pass: it forms Y's records as it reads them (its own row, which it stores -- the proximal
    // step reads Y's own rows --, and the pose at the other end of every incidence): no pass of its own over X[k], X[k-1]
    InterFuse fz;
    fz.Zc = Zc_.p; fz.Zp = Zp_.p; fz.Yout = Y_.p;
    if (keep_gx()) {
      const bool prox = prox_slot >= 0;
      if (prox) { fz.Xout = Xakh_.p; fz.Xref = Xak_.p; fz.Tinv = Tinv_.p; fz.Nv = N_.p; fz.Vb = V_.p; fz.gn_slot = prox_slot; }
      launch_inter(d_, st_, T_, cur_mask_, E_, o.loss, o.loss_reg, 1, false, Y_.p, nullptr, nullptr, Dd_.p, nullptr, gx_.p,
                   partials_.p, nullptr, GXc_.p, GXp_.p, &gam, prox ? nullptr : Dfx_.p, nullptr, gam_dev, &fz);
      return prox;
    } else {
      launch_inter(d_, st_, T_, cur_mask_, E_, o.loss, o.loss_reg, 1, false, Y_.p, nullptr, nullptr, Dd_.p, nullptr, gx_.p,
                   partials_.p, nullptr, nullptr, nullptr, &gam, nullptr, nullptr, gam_dev, &fz);
      launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Y_.p, false, gx_.p, Dfx_.p, nullptr, 0, nullptr, nullptr, 0);
    }
    return false;
  }
  if (trivial && fused_) {   // Y, g and Dfobj at the extrapolated point in one launch (:255-262)
    launch_extrapolate3(d_, st_, T_, cur_mask_, gam, gam_dev, Zc_.p, Zp_.p, Y_.p, gc_.p, gp_.p, gx_.p, Dfc_.p, Dfp_.p, Dfx_.p);
    return false;
  }
  launch_extrapolate(d_, st_, T_, true, cur_mask_, gam, Zc_.p, Zp_.p, Y_.p, gam_dev);
  if (trivial) {
    launch_extrapolate(d_, st_, T_, false, cur_mask_, gam, gc_.p, gp_.p, gx_.p, gam_dev);      // :259-262
    launch_extrapolate(d_, st_, T_, false, cur_mask_, gam, Dfc_.p, Dfp_.p, Dfx_.p, gam_dev);
  } else {
    // evaluate_g_and_Df(Y) (:264 -> DPGOProblem.cpp:683-749)
    if (keep_gx()) {   // G Y = G X[k] + gamma (G X[k] - G X[k-1]): Df comes out of the inter-edge pass
      launch_inter(d_, st_, T_, cur_mask_, E_, o.loss, o.loss_reg, 1, false, Y_.p, nullptr, nullptr, Dd_.p, nullptr, gx_.p,
                   partials_.p, nullptr, GXc_.p, GXp_.p, &gam, Dfx_.p, nullptr, gam_dev);
    } else {
      launch_inter(d_, st_, T_, cur_mask_, E_, o.loss, o.loss_reg, 1, false, Y_.p, nullptr, nullptr, Dd_.p, nullptr, gx_.p,
                   partials_.p);
      launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Y_.p, false, gx_.p, Dfx_.p, nullptr, 0, nullptr, nullptr, 0);
    }
  }
  return false;
}

// DPGOHash::amm_pgo  (DPGOHash.cpp:230-444)
int Group::amm(const std::vector<int> &locals) {
  const Options &o = opt_;
  set_mask(locals);
  constexpr int DS = 2 * MAX_DOTS;
  // The head of the iteration -- extrapolation, proximal half step, translation solve -- is branch-free: one replay where
  // the host's launch rate would bound it.  The gammas then come from device memory, written by the eager launch in front.
  const double *gam_dev = nullptr;
  if (iter_graph_wanted()) {
    NodeCoefs gam;
    for (int a = 0; a < num_local(); a++) gam.a[a] = gam.b[a] = res_[a].gamma;
    launch_set_coefs(st_, gam, num_local(), coefs_dev_.p);
    gam_dev = coefs_dev_.p;
  }
  const NodeMask mask_locals = cur_mask_;
  auto head_of_iteration = [&, gam_dev, mask_locals] {
    cur_mask_ = mask_locals;
    // Xakh = proximal(Y, Df); Gkh = G(Xakh | g[k], f); |Xakh - Xak|^2    (:363-367)
    // (these three scalars sit in slots DS.. and are read back together with the first scalars of TNT)
    if (!prepare_extrapolated(gam_dev, DS))
      launch_proximal(d_, st_, T_, cur_mask_, Y_.p, Dfx_.p, Tinv_.p, N_.p, V_.p, Xakh_.p, Xak_.p, partials_.p, DS);
    // Gkh = G(Xakh | g[k]) needs G Xakh; the translations of Xak = [. ; Xakh.R] need G [0 ; Xakh.R] + g: one pass over
    // G gives both (T1_ = G [0 ; R] + gx, slot DS + 1 = <Xakh, 1/2 G Xakh + gc>), then the solve   (:363-372)
    launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Xakh_.p, 2, gx_.p, T1_.p, Xakh_.p, 0.5, gc_.p, partials_.p, DS + 1);
    solve_tt(T1_.p, Xak_.p, -1.0);
  };
  // (where the refinement starts unasked -- below -- and segments are replayed, the two sequences are ONE segment: the
  // head of the iteration rides at the front of the refinement's head, one graph launch and one start-up less)
  static const bool spec_on = env_int("DPGO_SPEC_REFINE", 1) != 0;
  const bool speculate = spec_on && spec_refined_ && (int)locals.size() == num_local() && pending_update_ && !star_ && !dynamic();
  // (the closure below refers to this frame: whatever happens -- an exception on its way to the C ABI before a segment has
  // taken it -- it does not outlive the frame)
  struct DropHead {
    Group *g; bool armed = false;
    ~DropHead() { if (armed && !g->deferred_.empty()) { g->deferred_.clear(); g->deferred_key_ = 0; } }
  } drop_head{this};
  if (speculate && iter_graph_wanted() && deferred_.empty()) {
    deferred_.push_back(head_of_iteration);
    deferred_key_ = 0x68656164ull ^ mask_locals.v;
    drop_head.armed = true;
  } else {
    segment(10, cur_mask_.v, {}, head_of_iteration);
  }
  // the scalars of the last update() are needed from here on: `refined` (:351-355)
  auto decide_refined = [&] {
    finish_update();
    bool all = true;
    for (int a : locals) {
      NodeResults &r = res_[a];
      r.refined = (((r.gradFnorm * r.gradFnorm / r.fobj) > o.accepted_delta) || (r.num_oscillations >= o.max_oscillations)) &&
                  o.max_iterations > 0 && o.max_iterations_accepted > 0;   // :351-355
      all = all && r.refined;
    }
    return all;
  };
  // Where every node of the group was refined in the last iteration, the refinement of this one starts UNASKED: its head --
  // model gradient, preconditioned gradient, first CG step, trial point -- goes to the GPU right behind the translation
  // solve, and only then does the host take update()'s read-back and decide whether the nodes are refined at all (they
  // are, for the whole early regime).  A wrong guess costs the GPU some work on scratch vectors; the stream never idles
  // waiting for the decision.  DPGO_SPEC_REFINE=0 switches it off (measurement hook).
  bool done_tnt = false, abandoned = false;
  if (speculate) {
    const std::function<bool()> confirm = decide_refined;
    deferred_slots_ = DS + 3;
    done_tnt = run_tnt(locals, Xak_.p, gx_.p, gc_.p, true, &confirm);
    abandoned = !done_tnt;
    if (done_tnt) {
      for (int a : locals) res_[a].Gk = res_[a].Gk_alt;
    } else {
      // A wrong guess.  What the abandoned head wrote is scratch -- except T1_, which the trial point's translation recovery
      // has overwritten and the refinement of the nodes that ARE refined starts from: the pass that made it runs again
      // (same operands, same bits; its sum lands in the same slot), so that a guess, right or wrong, never changes a bit
      // of the trajectory.
      deferred_slots_ = 0;
      cur_mask_ = mask_locals;
      launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Xakh_.p, 2, gx_.p, T1_.p, Xakh_.p, 0.5, gc_.p, partials_.p, DS + 1);
    }
  }
  std::vector<int> plain, ref;
  if (!done_tnt) {
    if (!abandoned) decide_refined();   // (an abandoned attempt has taken the read-back and the decision)
    for (int a : locals) (res_[a].refined ? ref : plain).push_back(a);
    // Gk for nodes that are not refined; refined nodes run TNT first (:374-383)
    if (ref.empty()) {   // (the regime once the gradient is small: the pass and its read-back as one segment)
      segment(11, cur_mask_.v, {}, [&] {
        eval_G(Xak_.p, gc_.p, DS + 2);
        launch_reduce(st_, T_, num_local(), false, DS + 3, partials_.p, h_scal_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
      });
      wait_flag(fetch_seq_);
    } else {
      if (!plain.empty()) eval_G(Xak_.p, gc_.p, DS + 2);
      deferred_slots_ = DS + 3;
      // TNT minimises G(. | g extrapolated); Gk is G(. | g[k]) at the refined point (:377-383)
      run_tnt(ref, Xak_.p, gx_.p, gc_.p, true);
      for (int a : ref) res_[a].Gk = res_[a].Gk_alt;
    }
  }
  spec_refined_ = true;
  for (int a : locals) spec_refined_ = spec_refined_ && res_[a].refined;
  spec_refined_ = spec_refined_ && (int)locals.size() == num_local();
  std::vector<double> Gkh(num_local(), 0.0), minG(num_local(), 0.0);
  for (int a : locals) {
    NodeResults &r = res_[a];
    Gkh[a] = scal(a, DS + 1) + r.f;
    minG[a] = r.Fk[0] - o.psi * scal(a, DS);
    if (!r.refined) r.Gk = scal(a, DS + 2) + r.f;
  }
  // adaptive restart of the half step (:386-389)
  std::vector<int> redo;
  for (int a : locals)
    if (Gkh[a] > minG[a]) redo.push_back(a);
  if (!redo.empty()) {
    set_mask(redo);
    launch_proximal(d_, st_, T_, cur_mask_, Zc_.p, Dfc_.p, Tinv_.p, N_.p, V_.p, Xakh_.p, nullptr, nullptr, 0);
    eval_G(Xakh_.p, gc_.p, 0);
    fetch(1, false);
    for (int a : redo) Gkh[a] = scal(a, 0) + res_[a].f;
  }
  // hard / soft restart (:391-432)
  std::vector<int> restart, use_half, use_prox;
  std::vector<char> hard(num_local(), 0);
  for (int a : locals) {
    NodeResults &r = res_[a];
    const bool hr = r.Gk > r.Fk[0];
    const bool sr = (r.Gk > r.Fk[1] && r.soft_restart_hits[0] >= o.max_soft_restart_hits[0]) ||
                    (r.Gk > r.fobj && r.soft_restart_hits[1] > o.max_soft_restart_hits[1]);
    if (hr || sr) {
      restart.push_back(a);
      hard[a] = hr;
      (Gkh[a] <= r.fobj ? use_half : use_prox).push_back(a);
    }
  }
  std::vector<char> g_is_current(num_local(), 0);
  if (!restart.empty()) {
    if (!use_half.empty()) {
      set_mask(use_half);
      copy_rows(Xak_.p, Xakh_.p, false);
    }
    if (!use_prox.empty()) {
      set_mask(use_prox);
      launch_proximal(d_, st_, T_, cur_mask_, Zc_.p, Dfc_.p, Tinv_.p, N_.p, V_.p, Xak_.p, nullptr, nullptr, 0);
    }
    set_mask(restart);
    recover_translations(Xak_.p, gc_.p);
    std::vector<int> plain_r;
    for (int a : restart) {
      g_is_current[a] = 1;
      res_[a].restarts++;
      if (!res_[a].refined) plain_r.push_back(a);
    }
    {
      std::vector<int> ref;
      for (int a : restart)
        if (res_[a].refined) ref.push_back(a);
      if (!ref.empty()) run_tnt(ref, Xak_.p, gc_.p, nullptr, true);   // Gk = Results.f (:420-421)
    }
    if (!plain_r.empty()) {
      set_mask(plain_r);
      eval_G(Xak_.p, gc_.p, 0);
      fetch(1, false);
      for (int a : plain_r) res_[a].Gk = scal(a, 0) + res_[a].f;
    }
    for (int a : restart) {
      NodeResults &r = res_[a];
      if (hard[a]) r.s1 = std::max(0.5 * r.s1, 1.0);
      r.soft_restart_hits[0] /= 3;
      r.soft_restart_hits[1] = 0;
    }
  }
  // fall back to the proximal rotations when the refined step gains too little (:434-441)
  std::vector<int> fb_x, fb_c;
  for (int a : locals) {
    NodeResults &r = res_[a];
    if ((r.Fk[0] - r.Gk) < o.phi * (r.Fk[0] - Gkh[a])) (g_is_current[a] ? fb_c : fb_x).push_back(a);
  }
  for (int pass = 0; pass < 2; pass++) {
    const std::vector<int> &set = pass == 0 ? fb_x : fb_c;
    if (set.empty()) continue;
    set_mask(set);
    copy_rows(Xak_.p, Xakh_.p, false, 2);
    recover_translations(Xak_.p, pass == 0 ? gx_.p : gc_.p);
    eval_G(Xak_.p, gc_.p, 0);
    fetch(1, false);
    for (int a : set) res_[a].Gk = scal(a, 0) + res_[a].f;
  }
  for (int a : locals) res_[a].Gkh = Gkh[a];
  // (a speculative update stands only if the iteration took the common course: group.h)
  check_gate(done_tnt && tnt_common_ && redo.empty() && restart.empty() && fb_x.empty() && fb_c.empty());
  return 0;
}

}  // namespace dpgo

namespace dpgo {

// Single operators on reference-layout inputs, for the parity tests.
int Group::debug_apply(int a, const char *op_c, const double *in, int ld_in, double *out, int ld_out) {
  finish_update();
  if (a < 0 || a >= num_local()) return -1;
  const std::string op(op_c);
  const int n0 = info_[a].n[0], n1 = info_[a].n[1];
  sync();
  set_mask({a});
  auto put_own = [&](double *dev, const double *X, int ld, int row_t0, int row_r0, bool has_t) {
    std::vector<double> rec((size_t)n0 * RS_, 0.0);
    for (int k = 0; k < n0; k++)
      for (int c = 0; c < d_; c++) {
        if (has_t) rec[(size_t)k * RS_ + c] = X[(size_t)c * ld + row_t0 + k];
        for (int r = 0; r < d_; r++) rec[(size_t)k * RS_ + d_ + r * d_ + c] = X[(size_t)c * ld + row_r0 + k * d_ + r];
      }
    (void)hipMemcpy(dev + (size_t)own_off_[a] * RS_, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice);
  };
  auto get_own = [&](const double *dev, double *X, int ld, int row_t0, int row_r0, bool has_t) {
    std::vector<double> rec((size_t)n0 * RS_);
    sync();
    (void)hipMemcpy(rec.data(), dev + (size_t)own_off_[a] * RS_, sizeof(double) * rec.size(), hipMemcpyDeviceToHost);
    for (int k = 0; k < n0; k++)
      for (int c = 0; c < d_; c++) {
        if (has_t) X[(size_t)c * ld + row_t0 + k] = rec[(size_t)k * RS_ + c];
        for (int r = 0; r < d_; r++) X[(size_t)c * ld + row_r0 + k * d_ + r] = rec[(size_t)k * RS_ + d_ + r * d_ + c];
      }
  };
  double *A = tmp_[0].p, *Bv = tmp_[1].p, *C = tmp_[2].p;
  if (op == "project") {
    (void)hipMemset(A + (size_t)own_off_[a] * RS_, 0, sizeof(double) * n0 * RS_);
    put_own(Bv, in, ld_in, 0, 0, false);
    launch_retract_rot(d_, st_, T_, cur_mask_, A, Bv, C);
    get_own(C, out, ld_out, 0, 0, false);
  } else if (op == "solve_tt" || op == "solve_rr") {
    put_own(A, in, ld_in, 0, n0, true);
    put_own(Bv, in, ld_in, 0, n0, true);   // (the solve writes the unknowns' entries only: the rest of the answer is the input's)
    if (op == "solve_tt") solve_tt(A, Bv, 1.0);
    else {
      if (Lrr_.F.n == 0) return -1;
      solve_rr(A, Bv, 1.0);
    }
    get_own(Bv, out, ld_out, 0, n0, true);
  } else if (op == "G") {
    put_own(A, in, ld_in, 0, n0, true);
    launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, A, false, nullptr, Bv, nullptr, 0, nullptr, nullptr, 0);
    get_own(Bv, out, ld_out, 0, n0, true);
  } else if (op == "proximal") {
    // in = [Z ((d+1)(n0+n1) rows) ; Df ((d+1) n0 rows)]
    const int zr = (d_ + 1) * (n0 + n1);
    put_own(A, in, ld_in, 0, n0, true);
    put_own(Bv, in, ld_in, zr, zr + n0, true);
    launch_proximal(d_, st_, T_, cur_mask_, A, Bv, Tinv_.p, N_.p, V_.p, C, nullptr, nullptr, 0);
    get_own(C, out, ld_out, 0, n0, true);
  } else {
    fprintf(stderr, "[dpgo_amd] ERROR: debug_apply: unknown operator %s\n", op_c);
    return -1;
  }
  return 0;
}

}  // namespace dpgo


namespace dpgo {

// ---------------------------------------------------------------------------
// AMM-PGO*  (DPGOStar, C++/DPGO/src/DPGOStar.cpp)
// ---------------------------------------------------------------------------
// Global objective F(X) (DPGOStar::evaluate_f, :713-761) at the point whose own rows are X_own:
// neighbour rows are gathered from the same trial vector, every node evaluates its intra edges and
// its inter edges (charged 1/2 per node), the host adds the per-node sums (the master's aggregate).
void Group::enqueue_objective(const double *X_own, int slot0) {
  copy_rows(Tall_.p, X_own, false, 0);
  launch_copy_indexed(d_, st_, (int)gather_dst_.n, gather_dst_.p, gather_src_.p, Tall_.p, Tall_.p);
  if (coll_allgather_) {   // boundary poses of the trial point hosted by other groups
    launch_copy_indexed(d_, st_, (int)sent_rows_.size(), nullptr, sent_rows_dev_.p, X_own, coll_send_);
    if (coll_allgather_(coll_user_) != 0) throw DeviceError("all-gather callback failed");
    launch_copy_indexed(d_, st_, (int)recv_dst_.n, recv_dst_.p, recv_src_.p, coll_gathered_, Tall_.p);
  }
  launch_cost(d_, st_, T_, cur_mask_, Ei_, E_, opt_.loss == 0, opt_.loss, opt_.loss_reg, Tall_.p, partials_.p, slot0);
}

// Everything the master's tests of one AMM-PGO* iteration need (DPGOStar.cpp:147-192), enqueued back to back and read
// ONCE: the passes leave their partial sums in slots 0..5, k_star_sums folds them over the nodes on the device, the sum
// over the groups -- if there are several -- is an all-reduce on the same stream (set_device_allreduce) or, for a host
// that lent only host collectives, one call after the read-back; k_publish raises the flag.  (Before: up to four
// read-backs and as many host-staged all-reduces per iteration.)
int Group::star_sums(const double *X1_own, const double *X2_own, const double *ref_own, double *F1, double *F2, double *d1, double *d2) {
  finish_update();
  std::vector<int> all(num_local());
  for (int a = 0; a < num_local(); a++) all[a] = a;
  set_mask(all);
  if (star_vals_.n == 0) star_vals_.alloc(8);
  // Slots: update() and evaluate_global() reduce slots 0..2 over ALL rows and count on the neighbour segments of slot 2
  // never having been written; k_cost writes every segment of its two slots.  So the first point's pair is 0, 1 (as it
  // always was) and the second point's pair the last two slots, which nobody else touches; the distances are own-row sums.
  static const int slots[6] = {0, 1, MAX_SLOTS - 2, MAX_SLOTS - 1, 4, 5};
  unsigned valid = 0;
  if (F1) { enqueue_objective(X1_own, slots[0]); valid |= 0x3; }
  if (X2_own && F2) { enqueue_objective(X2_own, slots[2]); valid |= 0xc; }
  if (ref_own) {
    if (d1) { launch_sqdist(d_, st_, T_, cur_mask_, X1_own, ref_own, partials_.p, slots[4]); valid |= 0x10; }
    if (X2_own && d2) { launch_sqdist(d_, st_, T_, cur_mask_, X2_own, ref_own, partials_.p, slots[5]); valid |= 0x20; }
  }
  launch_star_sums(st_, T_, num_local(), valid, slots, partials_.p, star_vals_.p);
  const bool dev_sum = coll_allreduce_dev_ != nullptr;
  if (dev_sum && coll_allreduce_dev_(coll_user_, star_vals_.p, 4) != 0) return -1;
  launch_publish(st_, star_vals_.p, 4, h_scal_, h_flag_, next_seq(), dev_seq_.p);
  wait_flag(fetch_seq_);
  double v[4] = {h_scal_[0], h_scal_[1], h_scal_[2], h_scal_[3]};
  if (!dev_sum && coll_allreduce_ && coll_allreduce_(coll_user_, v, 4) != 0) return -1;
  if (F1) *F1 = v[0];
  if (F2) *F2 = v[1];
  if (d1) *d1 = v[2];
  if (d2) *d2 = v[3];
  return 0;
}

double Group::global_objective(const double *X_own) {
  double F = NAN;
  if (star_sums(X_own, nullptr, nullptr, &F, nullptr, nullptr, nullptr) != 0) return NAN;
  return F;
}

int Group::star_initialize_global(const double *X, int ld) {
  finish_update();
  if (num_local() != num_nodes_total_ && !coll_allgather_) {
    fprintf(stderr, "[dpgo_amd] ERROR: AMM-PGO* needs every node of the graph in one group, or collectives "
                    "(dpgo_group_set_collectives) that connect the groups.\n");
    return -1;
  }
  if (initialize_global(X, ld) != 0) return -1;
  star_ = true;
  star_fobj_ = global_objective(Xk_.p);
  starF_ = star_fobj_;
  star_fobjh_ = star_fobj_;
  return 0;
}

int Group::star_update() {
  finish_update();
  if (!star_) return -1;
  std::vector<int> all(num_local());
  for (int a = 0; a < num_local(); a++) all[a] = a;
  return update(all);
}

int Group::star_iterate() {
  InLib in_lib(this);
  finish_update();
  if (!star_) return -1;
  const Options &o = opt_;
  const int L = num_local();
  std::vector<int> all(L);
  for (int a = 0; a < L; a++) all[a] = a;
  for (int a : all)
    if (!res_[a].updated) {
      fprintf(stderr, "[dpgo_amd] ERROR: The optimizer has not been updated (node %d).\n", nodes_[a]);
      return -1;
    }
  star_branches_ = 0;
  // ---- amm_pgo_n for every node (:392-550)
  set_mask(all);
  prepare_extrapolated();
  std::vector<int> ref;
  for (int a : all) {
    NodeResults &r = res_[a];
    r.refined = (r.gradFnorm * r.gradFnorm / r.fobj) > o.accepted_delta;   // :515-516
    if (r.refined) ref.push_back(a);
  }
  launch_proximal(d_, st_, T_, cur_mask_, Y_.p, Dfx_.p, Tinv_.p, N_.p, V_.p, Xakh_.p, nullptr, nullptr, 0);
  copy_rows(Xak_.p, Xakh_.p, false, 2);
  recover_translations(Xak_.p, gx_.p);
  if (!ref.empty()) run_tnt(ref, Xak_.p, gx_.p, nullptr, true);
  // ---- the master's tests (:147-192); Xk = own rows of X[k] (Zc_).  The two objectives and the two distances of the
  // common path come with one read-back (the first test's branch changes Xakh only, so F(Xak) may be taken before it)
  double fobjh = NAN, fobj = NAN, sqh = NAN, sq = NAN;
  if (star_sums(Xakh_.p, Xak_.p, Zc_.p, &fobjh, &fobj, &sqh, &sq) != 0) return -1;
  set_mask(all);
  if (fobjh > starF_ - o.psi * sqh) {
    star_branches_ |= 1;   // pm_pgo_n (:685-711)
    launch_proximal(d_, st_, T_, cur_mask_, Zc_.p, Dfc_.p, Tinv_.p, N_.p, V_.p, Xakh_.p, nullptr, nullptr, 0);
    fobjh = global_objective(Xakh_.p);
  }
  set_mask(all);
  if (fobj > starF_ - o.psi * sq) {
    star_branches_ |= 2;   // mm_pgo_n (:552-683) and halve s
    copy_rows(Xak_.p, Xakh_.p, false, 2);
    recover_translations(Xak_.p, gc_.p);
    std::vector<int> plain;
    for (int a : all)
      if (!res_[a].refined) plain.push_back(a);
    if (!ref.empty()) run_tnt(ref, Xak_.p, gc_.p, nullptr, true);   // sets Gk = Results.f
    if (!plain.empty()) {
      set_mask(plain);
      eval_G(Xak_.p, gc_.p, 0);
      fetch(1, false);
      for (int a : plain) res_[a].Gk = scal(a, 0) + res_[a].f;
    }
    for (int a : all) res_[a].s1 = std::max(0.5 * res_[a].s1, 1.0);
    fobj = global_objective(Xak_.p);
  }
  if (starF_ - fobj < o.phi * (starF_ - fobjh)) {
    star_branches_ |= 4;   // fall back to the proximal rotations (:171-192)
    set_mask(all);
    copy_rows(Xak_.p, Xakh_.p, false, 2);
    recover_translations(Xak_.p, gc_.p);
    fobj = global_objective(Xak_.p);
  }
  set_mask(all);
  copy_rows(Xk_.p, Xak_.p, false);
  for (int a : all) {
    res_[a].iters++;
    res_[a].updated = 0;
  }
  star_fobj_ = fobj;
  star_fobjh_ = fobjh;
  starF_ = starF_ * (1 - o.eta[0]) + fobj * o.eta[0];
  return 0;
}

}  // namespace dpgo
