// Hand-written gfx950 (CDNA4) kernels of the DPGO hot path.  fp64 throughout
// (the reference is `typedef double Scalar`, C++/DPGO/include/DPGO/DPGO_types.h:12).
//
// Everything here is HBM/L2-latency bound graph work on 96-byte pose records;
// the kernels are organised for coalesced 16-byte accesses, 64-wide wavefront
// shuffles for the reductions and LDS staging for the dense front mat-vecs.
// No atomics: every reduction is a fixed-order tree, so results are bitwise
// reproducible run to run.
#include "kernels.h"

#include <algorithm>

#include <cstdlib>
#include <vector>

namespace dpgo {
namespace {

template <int D>
struct Dim {
  static constexpr int B = D + 1;
  static constexpr int RS = (D + 1) * D;
};

// the nodes a launch works on: the by-value bits, and-ed with the device-resident mask when there is one
__device__ __forceinline__ NodeBits mask_bits(const NodeMask &m) { return m.p ? (m.v & *m.p) : m.v; }
__device__ __forceinline__ bool node_on(const NodeMask &m, int node) { return (mask_bits(m) >> node) & 1ull; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Sum NS per-thread values over a 256-thread workgroup and store to dst[s * stride].  Callers store only for segments of
// nodes inside the launch mask: a masked launch must leave the partials of the other nodes alone (a second, narrower
// pass over the same slots -- the nodes a Dynamic rescale changed -- would otherwise zero what the first pass left
// for the nodes it skips).
template <int NS, int NWAVES = SEG_ROWS / 64>
__device__ __forceinline__ void block_store(const double (&v)[NS], double *dst, int stride) {
  if constexpr (NWAVES == 1) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const double r = wave_sum(v[s]);
      if (threadIdx.x == 0) dst[(size_t)s * stride] = r;
    }
    return;
  }
  __shared__ double sm[NS][NWAVES];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < NS; s++) {
    double r = wave_sum(v[s]);
    if (lane == 0) sm[s][wid] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      double a = 0;
#pragma unroll
      for (int q = 0; q < NWAVES; q += 4) a += (sm[s][q] + sm[s][q + 1]) + (sm[s][q + 2] + sm[s][q + 3]);
      dst[(size_t)s * stride] = a;
    }
  }
}

template <int N>
__device__ __forceinline__ void load_vec(const double *p, double (&r)[N]) {
  if constexpr (N % 2 == 0) {   // records and SE(3) blocks: whole 16-byte loads
    const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
      double2 v = q[k];
      r[2 * k] = v.x;
      r[2 * k + 1] = v.y;
    }
  } else {                      // 3x3 blocks of SE(2): 72 bytes, only 8-byte aligned
#pragma unroll
    for (int k = 0; k < N; k++) r[k] = p[k];
  }
}
// One block of a round of k_bsr: the values of the round's cnt blocks are interleaved in pieces of 16 bytes (8 bytes
// when the block has an odd number of entries), piece p of lane j at (p * cnt + j) -- see Group::upload_bsr
template <int N>
__device__ __forceinline__ void load_block(const double *base, int cnt, int j, double (&r)[N]) {
  if constexpr (N % 2 == 0) {
    const double2 *q = reinterpret_cast<const double2 *>(base) + j;
#pragma unroll
    for (int p = 0; p < N / 2; p++) {
      double2 v = q[p * cnt];
      r[2 * p] = v.x;
      r[2 * p + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int p = 0; p < N; p++) r[p] = base[p * cnt + j];
  }
}
template <int N>
__device__ __forceinline__ void store_vec(double *p, const double (&r)[N]) {
  static_assert(N % 2 == 0, "records are multiples of 16 bytes");
  double2 *q = reinterpret_cast<double2 *>(p);
#pragma unroll
  for (int k = 0; k < N / 2; k++) q[k] = make_double2(r[2 * k], r[2 * k + 1]);
}

// out (B x D) += blk (B x B) * rec (B x D)
template <int D, bool SKIP_T>
__device__ __forceinline__ void blk_mul_acc(const double *blk, const double *rec, double *out) {
  constexpr int B = D + 1;
#pragma unroll
  for (int r = 0; r < B; r++)
#pragma unroll
    for (int j = SKIP_T ? 1 : 0; j < B; j++) {
      const double a = blk[r * B + j];
#pragma unroll
      for (int c = 0; c < D; c++) out[r * D + c] = fma(a, rec[j * D + c], out[r * D + c]);
    }
}

// ---------------------------------------------------------------------------
// SO(d) projection: nearest rotation in the Frobenius norm.
// Replaces project_to_SO3 / project_to_SO2 (C++/DPGO/src/internal/project_to_SOd.cpp:27-33,121-196):
// eigen-decompose M^T M with a fixed schedule of cyclic Jacobi rotations (no data-dependent
// loop), form B = M V = U Sigma, rebuild U by Gram-Schmidt with u3 = u1 x u2, return U V^T.
// u3 = u1 x u2 with det V = +1 is exactly the det(U V^T) fix of the reference's
// JacobiSVD path (C++/DPGO/include/DPGO/DPGO_utils.h:485-514).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void jacobi_pair(double &app, double &aqq, double &apq, double &arp, double &arq,
                                            double *V, int p, int q, bool live) {
  // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)) with theta = (aqq - app) / (2 apq), written without theta: one
  // square root and one division on the chain instead of two and two (the kernels that project are bound by this chain of
  // dependent fp64 operations, not by their bytes); S is scaled to trace ~ 1 by the caller, so a^2 + b^2 cannot overflow and
  // underflows only where the rotation would be the identity to 1e-150
  const double a = aqq - app, b = 2.0 * apq;
  const double h = sqrt(fma(a, a, b * b));
  const bool go = live && h > 1e-150 && apq != 0.0;
  double t = b / (a + (a < 0.0 ? -h : h));
  t = go ? t : 0.0;
  const double c = rsqrt(fma(t, t, 1.0)), s = t * c;
  app -= t * apq;
  aqq += t * apq;
  apq = 0.0;
  const double rp = c * arp - s * arq, rq = s * arp + c * arq;
  arp = rp;
  arq = rq;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double vp = V[k * 3 + p], vq = V[k * 3 + q];
    V[k * 3 + p] = c * vp - s * vq;
    V[k * 3 + q] = s * vp + c * vq;
  }
}

__device__ __forceinline__ void swap_cols_neg(double *A, int a, int b, bool doit) {
  // swap columns a and b, negating the one that lands in b: a proper rotation of the column space
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double x = A[k * 3 + a], y = A[k * 3 + b];
    A[k * 3 + a] = doit ? y : x;
    A[k * 3 + b] = doit ? -x : y;
  }
}

__device__ void project_so3(const double *M, double *R) {
  // S = M^T M
  double s00 = 0, s11 = 0, s22 = 0, s01 = 0, s02 = 0, s12 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    s00 = fma(M[k * 3 + 0], M[k * 3 + 0], s00);
    s11 = fma(M[k * 3 + 1], M[k * 3 + 1], s11);
    s22 = fma(M[k * 3 + 2], M[k * 3 + 2], s22);
    s01 = fma(M[k * 3 + 0], M[k * 3 + 1], s01);
    s02 = fma(M[k * 3 + 0], M[k * 3 + 2], s02);
    s12 = fma(M[k * 3 + 1], M[k * 3 + 2], s12);
  }
  double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  {
    // S / 2^e with 2^e <= trace < 2^(e+1): the eigenvectors are those of S, the pair's a^2 + b^2 stays in range
    const double tr = s00 + s11 + s22;
    const int e = (tr > 0.0 && tr < 1.7e308) ? -ilogb(tr) : 0;
    s00 = ldexp(s00, e); s11 = ldexp(s11, e); s22 = ldexp(s22, e);
    s01 = ldexp(s01, e); s02 = ldexp(s02, e); s12 = ldexp(s12, e);
  }
  // at most the reference's 8 x 3 conjugations; a lane stops turning once its off-diagonal part is below 1e-20 of the trace
  // (the sweeps converge quadratically: what is skipped would turn V by less than 1e-20 where the eigenvalues are apart, and
  // where they are not U V^T does not depend on the turn), the wave leaves when every lane has stopped -- a pose's result
  // depends on its own matrix only, not on the poses it shares a wave with
  bool live = true;
  for (int sweep = 0; sweep < 8; sweep++) {
    jacobi_pair(s00, s11, s01, s02, s12, V, 0, 1, live);
    jacobi_pair(s00, s22, s02, s01, s12, V, 0, 2, live);
    jacobi_pair(s11, s22, s12, s01, s02, V, 1, 2, live);
    live = live && fabs(s01) + fabs(s02) + fabs(s12) > 1e-20;
    if (!__any(live)) break;
  }
  double Bm[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++)
      Bm[r * 3 + c] = M[r * 3 + 0] * V[0 * 3 + c] + M[r * 3 + 1] * V[1 * 3 + c] + M[r * 3 + 2] * V[2 * 3 + c];
  double n0 = Bm[0] * Bm[0] + Bm[3] * Bm[3] + Bm[6] * Bm[6];
  double n1 = Bm[1] * Bm[1] + Bm[4] * Bm[4] + Bm[7] * Bm[7];
  double n2 = Bm[2] * Bm[2] + Bm[5] * Bm[5] + Bm[8] * Bm[8];
  // sort columns by decreasing norm (3-element network), keeping det V = +1
  bool sw = n0 < n1;
  swap_cols_neg(Bm, 0, 1, sw); swap_cols_neg(V, 0, 1, sw);
  { double a = sw ? n1 : n0, b = sw ? n0 : n1; n0 = a; n1 = b; }
  sw = n0 < n2;
  swap_cols_neg(Bm, 0, 2, sw); swap_cols_neg(V, 0, 2, sw);
  { double a = sw ? n2 : n0, b = sw ? n0 : n2; n0 = a; n2 = b; }
  sw = n1 < n2;
  swap_cols_neg(Bm, 1, 2, sw); swap_cols_neg(V, 1, 2, sw);
  { double a = sw ? n2 : n1, b = sw ? n1 : n2; n1 = a; n2 = b; }
  double u1[3], u2[3], u3[3];
  const bool ok1 = n0 > 1e-300;
  const double i1 = rsqrt(ok1 ? n0 : 1.0);
  u1[0] = ok1 ? Bm[0] * i1 : 1.0; u1[1] = ok1 ? Bm[3] * i1 : 0.0; u1[2] = ok1 ? Bm[6] * i1 : 0.0;
  const double pr = u1[0] * Bm[1] + u1[1] * Bm[4] + u1[2] * Bm[7];
  u2[0] = Bm[1] - pr * u1[0]; u2[1] = Bm[4] - pr * u1[1]; u2[2] = Bm[7] - pr * u1[2];
  double m2 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
  if (!(m2 > 1e-28 * n0) || !ok1) {
    // rank <= 1: any unit vector orthogonal to u1 completes a nearest rotation
    const double ax = fabs(u1[0]), ay = fabs(u1[1]), az = fabs(u1[2]);
    double e[3] = {0, 0, 0};
    if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
    u2[0] = u1[1] * e[2] - u1[2] * e[1]; u2[1] = u1[2] * e[0] - u1[0] * e[2]; u2[2] = u1[0] * e[1] - u1[1] * e[0];
    m2 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
  }
  const double i2 = rsqrt(m2);
  u2[0] *= i2; u2[1] *= i2; u2[2] *= i2;
  u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
  u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
  u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) R[r * 3 + c] = u1[r] * V[c * 3 + 0] + u2[r] * V[c * 3 + 1] + u3[r] * V[c * 3 + 2];
}

__device__ __forceinline__ void project_so2(const double *M, double *R) {
  // C++/DPGO/include/DPGO/internal/project_to_SO2.h:3-18, guard traits.cpp:10
  double c = M[0] + M[3], s = M[2] - M[1];
  double n2 = fma(s, s, c * c);
  const bool ok = n2 >= 1.0e-32;
  c = ok ? c : 1.0;
  s = ok ? s : 0.0;
  n2 = ok ? n2 : 1.0;
  const double inv = 1.0 / sqrt(n2);
  c *= inv;
  s *= inv;
  R[0] = c; R[1] = -s; R[2] = s; R[3] = c;
}

template <int D>
__device__ __forceinline__ void project_sod(const double *M, double *R) {
  if constexpr (D == 3) project_so3(M, R); else project_so2(M, R);
}

// out = F - sym(F Y^T) Y      (SOdProduct::Proj, C++/DPGO/include/DPGO/SOdProduct.h:96-103)
template <int D>
__device__ __forceinline__ void tangent_proj(const double *Y, const double *F, double *out) {
  double G[D * D], S[D * D];
#pragma unroll
  for (int r = 0; r < D; r++)
#pragma unroll
    for (int c = 0; c < D; c++) {
      double a = 0;
#pragma unroll
      for (int k = 0; k < D; k++) a = fma(F[r * D + k], Y[c * D + k], a);
      G[r * D + c] = a;
    }
#pragma unroll
  for (int r = 0; r < D; r++)
#pragma unroll
    for (int c = 0; c < D; c++) S[r * D + c] = 0.5 * (G[r * D + c] + G[c * D + r]);
#pragma unroll
  for (int r = 0; r < D; r++)
#pragma unroll
    for (int c = 0; c < D; c++) {
      double a = F[r * D + c];
#pragma unroll
      for (int k = 0; k < D; k++) a = fma(-S[r * D + k], Y[k * D + c], a);
      out[r * D + c] = a;
    }
}

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8).  Segment kernels therefore take segment
// xcd_seg(blockIdx) instead of segment blockIdx: XCD x works on one contiguous eighth of the rows, so the
// records its gathers touch (lattice neighbours, a few thousand rows away at most) stay in that XCD's own
// 4 MB L2 instead of being fetched by all eight.  A bijection on [0, n) for any n.
__device__ __forceinline__ int xcd_seg(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return x * q + min(x, r) + (b >> 3);
}
#define SEGB xcd_seg((int)blockIdx.x, (int)gridDim.x)
// ... or, for a launch that covers the own segments of a few nodes only (NodeMask::nlive), the node's r-th segment
__device__ __forceinline__ int seg_live(const NodeMask &m, int b, int n) {
  if (m.nlive == 0) return xcd_seg(b, n);
  const int j = b % m.nlive, r = b / m.nlive;
  return r < m.nseg[j] ? m.seg0[j] + r : m.idle_seg;
}
#define SEGM(mask) seg_live(mask, (int)blockIdx.x, (int)gridDim.x)

// ---------------------------------------------------------------------------
// Block-sparse operator apply over pose records.
// ---------------------------------------------------------------------------
// Four lanes share one block row: lane j takes the blocks k = j (mod 4) of the row and the partial
// (d+1) x d results are combined with two xor-shuffles.  A row has ~10 blocks, each needing two
// dependent loads (column index, then record); splitting them over lanes cuts the serial latency of a
// row by 4x, which is what bounds this kernel when a GPU holds a single node (12.5 k rows).
// MODE 0: y = A x.  MODE 1: the translation row of x counts as zero (y = A [0 ; x.R]).  MODE 2: both at once --
// y = A [0 ; x.R] (+ add), while the fused dot product sees the full A x (one pass over A instead of two).
#ifndef BSR_UNROLL
#define BSR_UNROLL 2   // blocks of a lane in flight together (k_bsr)
#endif
template <int D, int MODE>
__global__ __launch_bounds__(BSR_LPR * SEG_ROWS) void k_bsr(const Seg *segs, NodeMask mask, BsrDev A, const double *x,
                                              const double *addv, double *y, const double *dotv, double coef,
                                              const double *dotadd, double *partial, double *copy1, double *copy2) {
  constexpr int B = Dim<D>::B, RS = Dim<D>::RS;
  const int si = SEGM(mask);
  const Seg s = segs[si];
  const bool active = node_on(mask, s.node);
  double part[1] = {0.0};
  constexpr int LPR = BSR_LPR;
  const int row = s.begin + (int)threadIdx.x / LPR, j = threadIdx.x % LPR;
  if (active) {   // uniform per workgroup; rows past the segment end simply have no blocks
    double acc[RS], acct[MODE == 2 ? RS : 1];
#pragma unroll
    for (int k = 0; k < RS; k++) acc[k] = 0.0;
    if constexpr (MODE == 2) {
#pragma unroll
      for (int k = 0; k < RS; k++) acct[k] = 0.0;
    }
    const bool inrow = row < s.end;
    const int k1 = inrow ? A.ptr[row + 1] : 0;
    // A lane's blocks in groups of U: the group's records and block values are requested together, the column indices of
    // the NEXT group with them -- a group costs one round trip instead of two per block (index, then record); the products
    // run in the order of the row, the sums as before.
    constexpr int U = BSR_UNROLL;
    int kk = (inrow ? A.ptr[row] : 0) + j;
    int qn[U];
#pragma unroll
    for (int u = 0; u < U; u++) qn[u] = kk + u * LPR < k1 ? A.col[kk + u * LPR] : 0;
    for (; kk < k1; kk += U * LPR) {
      double xb[U][RS], blk[U][B * B];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = kk + u * LPR;
        if (k < k1) {
          load_vec<RS>(x + (size_t)qn[u] * RS, xb[u]);
          load_block<B * B>(A.val + (size_t)(k - j) * B * B, min(LPR, k1 - (k - j)), j, blk[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) qn[u] = kk + (U + u) * LPR < k1 ? A.col[kk + (U + u) * LPR] : 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (kk + u * LPR >= k1) break;
        blk_mul_acc<D, MODE != 0>(blk[u], xb[u], acc);
        if constexpr (MODE == 2) {   // what the translation row of x adds: first column of the block
#pragma unroll
          for (int r = 0; r < B; r++)
#pragma unroll
            for (int c = 0; c < D; c++) acct[r * D + c] = fma(blk[u][r * B], xb[u][c], acct[r * D + c]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < RS; k++)
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) acc[k] += __shfl_xor(acc[k], o, 64);
    if constexpr (MODE == 2) {
#pragma unroll
      for (int k = 0; k < RS; k++)
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) acct[k] += __shfl_xor(acct[k], o, 64);
    }
    if (inrow && j == 0) {
      if (copy1) {   // the row's own record goes to two more arrays on the way (the tail of iterate(): Xk <- Xak, X[iter] <- Xak)
        double xr[RS];
        load_vec<RS>(x + (size_t)row * RS, xr);
        store_vec<RS>(copy1 + (size_t)row * RS, xr);
        if (copy2) store_vec<RS>(copy2 + (size_t)row * RS, xr);
      }
      if (dotv) {
        double v[RS], da[RS];
        load_vec<RS>(dotv + (size_t)row * RS, v);
        if (dotadd) load_vec<RS>(dotadd + (size_t)row * RS, da);
        double p = 0;
#pragma unroll
        for (int k = 0; k < RS; k++) {
          const double full = MODE == 2 ? acc[k] + acct[k] : acc[k];
          p = fma(v[k], fma(coef, full, dotadd ? da[k] : 0.0), p);
        }
        part[0] = p;
      }
      if (y) {
        if (addv) {
          double av[RS];
          load_vec<RS>(addv + (size_t)row * RS, av);
#pragma unroll
          for (int k = 0; k < RS; k++) acc[k] += av[k];
        }
        store_vec<RS>(y + (size_t)row * RS, acc);
      }
    }
  }
  if (partial && active) block_store<1, LPR * SEG_ROWS / 64>(part, partial + si, 0);
}

// out (d x d) = Proj_R(E - sym(nabla R^T) Rdot): the rotation rows of the Riemannian Hessian-vector product
// (DPGOProblem.cpp:570-574; SymBlockDiagProduct(A = Rdot, B = R, C = nabla), SOdProduct.h:64-89)
template <int D>
__device__ __forceinline__ void hess_epilogue_rows(const double *R, const double *E, const double *nb, const double *rd,
                                                   double *out) {
  double G[D * D], F[D * D];
#pragma unroll
  for (int r = 0; r < D; r++)
#pragma unroll
    for (int c = 0; c < D; c++) {
      double a = 0;
#pragma unroll
      for (int k = 0; k < D; k++) a = fma(nb[r * D + k], R[c * D + k], a);
      G[r * D + c] = a;
    }
#pragma unroll
  for (int r = 0; r < D; r++)
#pragma unroll
    for (int c = 0; c < D; c++) {
      double a = E[r * D + c];
#pragma unroll
      for (int k = 0; k < D; k++) a = fma(-0.5 * (G[r * D + k] + G[k * D + r]), rd[k * D + c], a);
      F[r * D + c] = a;
    }
  tangent_proj<D>(R, F, out);
}

// more vectors for the dot-product epilogues of k_bsr_tcol
struct TcolDots {
  const double *s = nullptr, *grad = nullptr, *hs = nullptr, *g = nullptr, *ga = nullptr;
};
// y = base + A[:, translation column] t: what A x adds when only the translation rows of x change.  The
// first column of every (d+1) x (d+1) block is kept in a compact copy (tval, d+1 doubles per block), so
// this pass moves a quarter of the operator.  Used after a G_tt solve: G [t ; R] = G [0 ; R] + G_{:,t} t,
// with the first product already there as the right-hand side of that solve.
// Row-local epilogues (mode): 1 = also out2 = [0 ; Proj_X(y.R)] (the reduced Riemannian gradient when y is the
// model gradient); 2 = y is not stored, out2 = [0 ; Proj_X(y.R - sym(nabla.R X.R^T) Rdot.R)] (the Hessian-vector
// product when y = G [tdot ; Rdot]).
#ifndef TCOL_UNROLL
#define TCOL_UNROLL 4
#endif
template <int D>
__global__ __launch_bounds__(4 * SEG_ROWS) void k_bsr_tcol(const Seg *segs, NodeMask mask, BsrDev A, const double *tval,
                                                         const double *xt, const double *base, double *y, int mode,
                                                         const double *X, const double *nabla, const double *Rdot,
                                                         double *out2, const double *rres, double *partial,
                                                         int pstride, TcolDots E) {
  constexpr int B = Dim<D>::B, RS = Dim<D>::RS;
  const int si = SEGM(mask);
  const Seg s = segs[si];
  if (!node_on(mask, s.node)) return;   // (partials of a node outside the mask are never read)
  const int row = s.begin + (threadIdx.x >> 2), j = threadIdx.x & 3;
  double pr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double acc[RS];
#pragma unroll
  for (int k = 0; k < RS; k++) acc[k] = 0.0;
  const bool inrow = row < s.end;
  const int k1 = inrow ? A.ptr[row + 1] : 0;
  // (a lane's blocks in groups of U, the next group's column indices requested with this group's operands: k_bsr)
  constexpr int U = TCOL_UNROLL;
  int kk = (inrow ? A.ptr[row] : 0) + j;
  int qn[U];
#pragma unroll
  for (int u = 0; u < U; u++) qn[u] = kk + 4 * u < k1 ? A.col[kk + 4 * u] : 0;
  // (the row's own record of `base`: nothing in the loop depends on it)
  double bv[RS];
  if (inrow && j == 0) load_vec<RS>(base + (size_t)row * RS, bv);
  for (; kk < k1; kk += 4 * U) {
    double t[U][D], c0[U][B];
#pragma unroll
    for (int u = 0; u < U; u++)
      if (kk + 4 * u < k1) {
        load_vec<D>(xt + (size_t)qn[u] * RS, t[u]);
        load_vec<B>(tval + (size_t)(kk + 4 * u) * B, c0[u]);
      }
#pragma unroll
    for (int u = 0; u < U; u++) qn[u] = kk + 4 * (U + u) < k1 ? A.col[kk + 4 * (U + u)] : 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (kk + 4 * u >= k1) break;
#pragma unroll
      for (int r = 0; r < B; r++)
#pragma unroll
        for (int c = 0; c < D; c++) acc[r * D + c] = fma(c0[u][r], t[u][c], acc[r * D + c]);
    }
  }
#pragma unroll
  for (int k = 0; k < RS; k++) {
    acc[k] += __shfl_xor(acc[k], 1, 64);
    acc[k] += __shfl_xor(acc[k], 2, 64);
  }
  if (inrow && j == 0) {
#pragma unroll
    for (int k = 0; k < RS; k++) acc[k] += bv[k];
    if (mode != 2) store_vec<RS>(y + (size_t)row * RS, acc);
    if (mode == 0 && partial) {
      // the six sums of a trial point x+ (= xt), TNT.h:505-536: <s,s>, <grad,s>, <s,Hs> over the rotation rows,
      // <x+, g>, <x+, g_alt>, <x+, G x+ + g> over the whole record
      double xp[RS], sv[RS], gv[RS], hv[RS], g1[RS], g2[RS];
      load_vec<RS>(xt + (size_t)row * RS, xp);
      load_vec<RS>(E.s + (size_t)row * RS, sv);
      load_vec<RS>(E.grad + (size_t)row * RS, gv);
      load_vec<RS>(E.hs + (size_t)row * RS, hv);
      load_vec<RS>(E.g + (size_t)row * RS, g1);
      load_vec<RS>(E.ga + (size_t)row * RS, g2);
#pragma unroll
      for (int k = 0; k < RS; k++) {
        if (k >= D) {
          pr[0] = fma(sv[k], sv[k], pr[0]);
          pr[1] = fma(gv[k], sv[k], pr[1]);
          pr[2] = fma(sv[k], hv[k], pr[2]);
        }
        pr[3] = fma(xp[k], g1[k], pr[3]);
        pr[4] = fma(xp[k], g2[k], pr[4]);
        pr[5] = fma(xp[k], acc[k], pr[5]);
      }
    }
    if (mode != 0) {
      double x[RS], o[RS];
      load_vec<RS>(X + (size_t)row * RS, x);
#pragma unroll
      for (int k = 0; k < D; k++) o[k] = 0.0;
      if (mode == 1) {
        tangent_proj<D>(x + D, acc + D, o + D);
        if (partial) {   // |grad|^2 (rotation rows), <X, nabla>, <X, g>, <X, g_alt>: the start of a refinement (tnt.cpp)
          double g1[RS], g2[RS];
          load_vec<RS>(E.g + (size_t)row * RS, g1);
          load_vec<RS>(E.ga + (size_t)row * RS, g2);
#pragma unroll
          for (int k = 0; k < RS; k++) {
            if (k >= D) pr[0] = fma(o[k], o[k], pr[0]);
            pr[1] = fma(x[k], acc[k], pr[1]);
            pr[2] = fma(x[k], g1[k], pr[2]);
            pr[3] = fma(x[k], g2[k], pr[3]);
          }
        }
      } else {
        double nb[RS], rd[RS];
        load_vec<RS>(nabla + (size_t)row * RS, nb);
        load_vec<RS>(Rdot + (size_t)row * RS, rd);
        hess_epilogue_rows<D>(x + D, acc + D, nb + D, rd + D, o + D);
        if (partial) {   // <p, Hp>, <Hp, Hp>, <p, p>, <p, r> over the rotation rows
          double rr[RS];
          load_vec<RS>(rres + (size_t)row * RS, rr);
#pragma unroll
          for (int k = D; k < RS; k++) {
            pr[0] = fma(rd[k], o[k], pr[0]);
            pr[1] = fma(o[k], o[k], pr[1]);
            pr[2] = fma(rd[k], rd[k], pr[2]);
            pr[3] = fma(rd[k], rr[k], pr[3]);
          }
        }
      }
      store_vec<RS>(out2 + (size_t)row * RS, o);
    }
  }
  if (partial) block_store<6, 4 * SEG_ROWS / 64>(pr, partial + si, pstride);
}

// ---------------------------------------------------------------------------
// Robust inter-node edge pass (residual form).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void loss_weight(int loss, double dl, double s, double &w, double &rho) {
  if (loss == 1) {          // Huber   (DPGOProblem.cpp:651-658)
    const double rs = sqrt(fmax(s, dl)), sd = sqrt(dl);
    w = sd / rs;
    rho = fmin(2.0 * sd * rs - dl, s);
  } else if (loss == 2) {   // Geman-McClure (:659-664)
    const double q = s + dl;
    w = dl * dl / (q * q);
    rho = dl * (s / q);
  } else if (loss == 3) {   // Welsch (:665-670)
    w = exp(-s / dl);
    rho = dl - dl * w;
  } else {
    w = 1.0;
    rho = s;
  }
}

// The proximal step of one pose (DPGOProblem.cpp:600-632): Xout = proximal(z, df); with Xref, returns |Xout - Xref|^2, after
// which Xref takes over the new rotations (its translations follow from a solve: DPGOHash.cpp:369-372).  Shared by k_proximal
// and by the inter-edge pass that forms df itself (k_inter, InterFuse::Xout).
template <int D>
__device__ __forceinline__ double proximal_row(int row, const double *z, const double *df, const double *Tinv, const double *Nv,
                                               const double *Vb, double *Xout, double *Xref) {
  constexpr int RS = Dim<D>::RS;
  double N[D], V[D * D], M[D * D], R[D * D], out[RS];
#pragma unroll
  for (int k = 0; k < D; k++) N[k] = Nv[(size_t)row * D + k];
#pragma unroll
  for (int k = 0; k < D * D; k++) V[k] = Vb[(size_t)row * D * D + k];
  const double T = Tinv[row];
  // M = -Df_R + N^T Df_t + V R0      (DPGOProblem.cpp:618-620)
#pragma unroll
  for (int r = 0; r < D; r++)
#pragma unroll
    for (int c = 0; c < D; c++) {
      double a = fma(N[r], df[c], -df[D + r * D + c]);
#pragma unroll
      for (int k = 0; k < D; k++) a = fma(V[r * D + k], z[D + k * D + c], a);
      M[r * D + c] = a;
    }
  project_sod<D>(M, R);
  // t = t0 - N (R - R0) - T Df_t     (:627-629)
#pragma unroll
  for (int c = 0; c < D; c++) {
    double a = fma(-T, df[c], z[c]);
#pragma unroll
    for (int k = 0; k < D; k++) a = fma(-N[k], R[k * D + c] - z[D + k * D + c], a);
    out[c] = a;
  }
#pragma unroll
  for (int k = 0; k < D * D; k++) out[D + k] = R[k];
  store_vec<RS>(Xout + (size_t)row * RS, out);
  double p = 0;
  if (Xref) {
    double ref[RS];
    load_vec<RS>(Xref + (size_t)row * RS, ref);
#pragma unroll
    for (int k = 0; k < RS; k++) { const double dd = out[k] - ref[k]; p = fma(dd, dd, p); }
    // the reference point takes over the new rotations
#pragma unroll
    for (int k = D; k < RS; k++) ref[k] = out[k];
    store_vec<RS>(Xref + (size_t)row * RS, ref);
  }
  return p;
}

struct InterLin {   // (k_inter mode 1) Df at the extrapolated point from the kept products G X[k], G X[k-1]
  const double *GXc = nullptr, *GXp = nullptr;
  double *out = nullptr;
  const double *gamma_dev = nullptr;   // per-node gamma in device memory (a launch that may be replayed), else `gamma`
  NodeCoefs gamma;
};
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_inter(const Seg *segs, NodeMask mask, InterEdgesDev E, int loss,
                                               double dl, int mode, int quad, int nseg_own, const double *Z,
                                               const double *Zprev, const double *Qd, const double *Dd,
                                               double *DfE, double *g, double *partial, int pstride, double *wout,
                                               InterLin lin, const double *Znbr, InterFuse F) {
  constexpr int B = Dim<D>::B, RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  const bool active = node_on(mask, s.node);
  const bool own = SEGB < nseg_own;
  double part[3] = {0.0, 0.0, 0.0};   // sum rho ; the quadratic term ; <z, g> over own rows
  double gn[1] = {0.0};               // (F.Df) |grad F|^2 of the row
  const int row = s.begin + threadIdx.x;
  if (active && row < s.end) {
    double zp[RS], acc[RS];
    const bool xfuse = F.Zc != nullptr;
    const double gm = (xfuse || lin.GXc) ? (lin.gamma_dev ? lin.gamma_dev[s.node] : lin.gamma.a[s.node]) : 0.0;
    // Znbr (mode 0): the neighbour rows are taken from there (the iterate the exchange just delivered) and copied into Z
    // on the way -- update()'s halo copy without a launch of its own.  Nobody reads Z's neighbour rows in this pass.
    const bool from_nbr = Znbr && row >= E.nrows_own;
    if (xfuse) {
      double zq[RS];
      load_vec<RS>(F.Zc + (size_t)row * RS, zp);
      load_vec<RS>(F.Zp + (size_t)row * RS, zq);
#pragma unroll
      for (int k = 0; k < RS; k++) zp[k] = fma(gm, zp[k] - zq[k], zp[k]);
      store_vec<RS>(F.Yout + (size_t)row * RS, zp);
    } else if (from_nbr) {
      // (a lazy unpack: the row may still be in the exchange's receive buffer; it then also goes where an unpack would have put it)
      const int src = E.recv ? E.nsrc[row - E.nrows_own] : -1;
      load_vec<RS>(src >= 0 ? E.recv + (size_t)src * RS : Znbr + (size_t)row * RS, zp);
      store_vec<RS>(const_cast<double *>(Z) + (size_t)row * RS, zp);
      if (src >= 0) store_vec<RS>(const_cast<double *>(Znbr) + (size_t)row * RS, zp);
    } else {
      load_vec<RS>(Z + (size_t)row * RS, zp);
    }
#pragma unroll
    for (int k = 0; k < RS; k++) acc[k] = 0.0;
    const int k0 = E.inc_ptr[row], k1 = E.inc_ptr[row + 1];
    // the incidence's record: eight 16-byte loads of one line, then the other pose.  A pose's incidences are a chain of
    // dependent round trips (record -> other pose -> arithmetic), and the wave waits for its pose with the most
    // incidences: the NEXT record is requested together with this one's other pose, so an incidence costs one round trip,
    // not two
    union Rec { double2 q[8]; InterInc r; };
    Rec nxt;
    if (k0 < k1) {
      const double2 *rq = reinterpret_cast<const double2 *>(E.rec + k0);
#pragma unroll
      for (int i = 0; i < 8; i++) nxt.q[i] = rq[i];
    }
    for (int k = k0; k < k1; k++) {
      Rec u8;
#pragma unroll
      for (int i = 0; i < 8; i++) u8.q[i] = nxt.q[i];
      const int code = u8.r.code, e = code >> 1, role = code & 1;
      const int other = u8.r.other;
      double zo[RS], Re[D * D], te[D];
      if (xfuse) {
        double zq[RS];
        load_vec<RS>(F.Zc + (size_t)other * RS, zo);
        load_vec<RS>(F.Zp + (size_t)other * RS, zq);
#pragma unroll
        for (int i = 0; i < RS; i++) zo[i] = fma(gm, zo[i] - zq[i], zo[i]);
      } else {
        const int osrc = u8.r.osrc;
        const double *po = (Znbr && other >= E.nrows_own) ? ((E.recv && osrc >= 0) ? E.recv + (size_t)osrc * RS : Znbr + (size_t)other * RS)
                                                          : Z + (size_t)other * RS;
        load_vec<RS>(po, zo);
      }
      if (k + 1 < k1) {
        const double2 *rq = reinterpret_cast<const double2 *>(E.rec + k + 1);
#pragma unroll
        for (int i = 0; i < 8; i++) nxt.q[i] = rq[i];
      }
#pragma unroll
      for (int i = 0; i < D * D; i++) Re[i] = u8.r.R[i];
#pragma unroll
      for (int i = 0; i < D; i++) te[i] = u8.r.t[i];
      const double tau = u8.r.tau, kap = u8.r.kappa;
      const double *zi = role ? zo : zp;   // tail record
      const double *zj = role ? zp : zo;   // head record
      double u[D], W[D * D];
      double sn = 0;
#pragma unroll
      for (int c = 0; c < D; c++) {
        double a = zi[c] - zj[c];
#pragma unroll
        for (int q = 0; q < D; q++) a = fma(te[q], zi[D + q * D + c], a);
        u[c] = a;
        sn = fma(tau * a, a, sn);
      }
#pragma unroll
      for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) {
          double a = -zj[D + r * D + c];
#pragma unroll
          for (int q = 0; q < D; q++) a = fma(Re[q * D + r], zi[D + q * D + c], a);
          W[r * D + c] = a;
          sn = fma(kap * a, a, sn);
        }
      double w, rho;
      loss_weight(loss, dl, sn, w, rho);
      if (role == 0) {
        part[0] += rho;
        if (wout) wout[e] = w;   // the weight of every edge, once (Rescale::Dynamic reads them, DPGOProblem.cpp:300-306)
#pragma unroll
        for (int c = 0; c < D; c++) acc[c] = fma(w * tau, u[c], acc[c]);
#pragma unroll
        for (int q = 0; q < D; q++)
#pragma unroll
          for (int c = 0; c < D; c++) {
            double a = tau * te[q] * u[c];
#pragma unroll
            for (int r = 0; r < D; r++) a = fma(kap * Re[q * D + r], W[r * D + c], a);
            acc[D + q * D + c] = fma(w, a, acc[D + q * D + c]);
          }
      } else {
#pragma unroll
        for (int c = 0; c < D; c++) acc[c] = fma(-w * tau, u[c], acc[c]);
#pragma unroll
        for (int i = 0; i < D * D; i++) acc[D + i] = fma(-w * kap, W[i], acc[D + i]);
      }
    }
    if (mode == 0) {
      if (quad) {
        double zq[RS], old[RS], dz[RS], qz[RS];
        load_vec<RS>(Zprev + (size_t)row * RS, zq);
        load_vec<RS>(DfE + (size_t)row * RS, old);
#pragma unroll
        for (int k = 0; k < RS; k++) { dz[k] = zp[k] - zq[k]; qz[k] = 0.0; }
        blk_mul_acc<D, false>(Qd + (size_t)row * B * B, dz, qz);
        double p = 0;
#pragma unroll
        for (int k = 0; k < RS; k++) p = fma(dz[k], fma(0.5, qz[k], old[k]), p);
        part[1] = p;
      }
      store_vec<RS>(DfE + (size_t)row * RS, acc);
    }
    if (own) {
      double dz[RS];
#pragma unroll
      for (int k = 0; k < RS; k++) dz[k] = 0.0;
      blk_mul_acc<D, false>(Dd + (size_t)row * B * B, zp, dz);
      double zg = 0;
#pragma unroll
      for (int k = 0; k < RS; k++) {
        acc[k] -= dz[k];
        zg = fma(zp[k], acc[k], zg);
      }
      part[2] = zg;
      store_vec<RS>(g + (size_t)row * RS, acc);
      if (F.Df) {
        // Dfobj = G X + g, gradF = [Dfobj.x ; Proj_R(Dfobj.Y)], |gradF|^2 (k_tangent_full: the same operations in the same order)
        double v[RS], x[RS], o[RS];
        load_vec<RS>(F.GX + (size_t)row * RS, v);
        load_vec<RS>(F.X + (size_t)row * RS, x);
#pragma unroll
        for (int k = 0; k < RS; k++) v[k] += acc[k];
        store_vec<RS>(F.Df + (size_t)row * RS, v);
#pragma unroll
        for (int k = 0; k < D; k++) o[k] = v[k];
        tangent_proj<D>(x + D, v + D, o + D);
        double p = 0;
#pragma unroll
        for (int k = 0; k < RS; k++) p = fma(o[k], o[k], p);
        gn[0] = p;
      }
      if (lin.GXc) {
        // Df = g + G Y at the extrapolated point Y = X[k] + gamma (X[k] - X[k-1]), without another pass over G:
        // G Y = G X[k] + gamma (G X[k] - G X[k-1]), both products kept from the last two update()s
        double a[RS], b[RS];
        load_vec<RS>(lin.GXc + (size_t)row * RS, a);
        load_vec<RS>(lin.GXp + (size_t)row * RS, b);
#pragma unroll
        for (int k = 0; k < RS; k++) acc[k] += fma(gm, a[k] - b[k], a[k]);
        if (lin.out) store_vec<RS>(lin.out + (size_t)row * RS, acc);
        if (F.Xout) {
          // the proximal half step at this row -- k_proximal's operations in its order, on the Df just formed
          gn[0] = proximal_row<D>(row, zp, acc, F.Tinv, F.Nv, F.Vb, F.Xout, F.Xref);
        }
      }
    }
  }
  if (active) {
    block_store<3>(part, partial + SEGB, pstride);
    if ((F.Df || F.Xout) && own) block_store<1>(gn, partial + (size_t)F.gn_slot * pstride + SEGB, 0);
  }
}

// ---------------------------------------------------------------------------
// Rescale::Dynamic on the device (DPGOProblem.cpp:289-358, 426-514: the test; :751-840: update_quadratic_mat).
// k_rescale_decide: one workgroup per node.  A node is rescaled when its counter has reached max_count or the loss
// weight of one of its inter-node edges exceeds the edge's scale (:300-321); its new scales are clamp(1.25 w, 0.01, 1)
// (DPGOProblem.h:17-18).  flags[a] (device) and host_flags[a] (pinned) = 1 for a rescaled node.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rescale_decide(NodeBits nodes, const int *e_off, const double *w, double *scale,
                                                        int *count, int max_count, int *flags, double *host_flags) {
  const int a = blockIdx.x;
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  const bool mine = (nodes >> a) & 1ull;
  const int e0 = e_off[a], e1 = e_off[a + 1];
  if (mine) {
    int hit = 0;
    for (int e = e0 + threadIdx.x; e < e1; e += 256) hit |= w[e] > scale[e];
    if (hit) atomicOr(&any, 1);
  }
  __syncthreads();
  const bool rescaled = mine && (count[a] >= max_count || any);
  __syncthreads();   // (count[a] is read by every thread above, written by one below)
  if (rescaled)
    for (int e = e0 + threadIdx.x; e < e1; e += 256) scale[e] = fmin(1.0, fmax(0.01, 1.25 * w[e]));
  if (threadIdx.x == 0) {
    if (mine) count[a] = rescaled ? 0 : count[a] + 1;
    flags[a] = rescaled ? 1 : 0;
    __hip_atomic_store(host_flags + a, rescaled ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// k_rescale_apply: the rescaled nodes' surrogate with the new scales -- only block-diagonal terms change.  Every inter-node
// edge e adds 2 s_e E_end(e) to the diagonal block of its own endpoint in G, D and the proximal majoriser H, and to Q on
// both endpoints (assemble.cpp, the `scale` branch); here one thread per pose sums its incidences (records of k_inter) and
// writes: the diagonal block of G (interleaved block values + the translation column copy), D, Q, T / N / V from H, and the
// diagonal entry of G_tt among the values the numeric factorisation reads.  Gbase / Hbase: the blocks with all scales zero.
struct RescaleDev {
  const int *flags;            // per node
  const double *scale;         // per inter edge
  const double *Gbase, *Hbase; // per own pose, B x B
  const int4 *gpos;            // per own pose: offset of the round's values, blocks in the round, lane, CSR index of the block
  const int *att_pos;          // per own pose: index of its diagonal entry in att_val
  double *Gval, *Gtcol, *Dd, *Qd, *Tinv, *N, *V, *att_val;
  double xi;
};
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_rescale_apply(const Seg *segs, InterEdgesDev E, RescaleDev R, int nseg_own) {
  constexpr int B = Dim<D>::B, BB = B * B, PS = BB % 2 == 0 ? 2 : 1;
  const Seg s = segs[SEGB];
  if (!R.flags[s.node]) return;
  const bool own = SEGB < nseg_own;
  const int row = s.begin + threadIdx.x;
  if (row >= s.end) return;
  double acc[BB];
#pragma unroll
  for (int k = 0; k < BB; k++) acc[k] = 0.0;
  const int k1 = E.inc_ptr[row + 1];
  for (int k = E.inc_ptr[row]; k < k1; k++) {
    union { double2 q[8]; InterInc r; } u8;
    const double2 *rq = reinterpret_cast<const double2 *>(E.rec + k);
#pragma unroll
    for (int i = 0; i < 8; i++) u8.q[i] = rq[i];
    const int e = u8.r.code >> 1, role = u8.r.code & 1;
    const double w2 = 2.0 * R.scale[e], tau = u8.r.tau, kap = u8.r.kappa;
    // E_end: AA for the tail [tau, tau t^T; tau t, kappa I + tau t t^T], II for the head [tau, 0; 0, kappa I]
    acc[0] = fma(w2, tau, acc[0]);
#pragma unroll
    for (int r = 0; r < D; r++) {
      acc[(1 + r) * B + 1 + r] = fma(w2, kap, acc[(1 + r) * B + 1 + r]);
      if (role == 0) {
        const double tt = tau * u8.r.t[r];
        acc[1 + r] = fma(w2, tt, acc[1 + r]);
        acc[(1 + r) * B] = fma(w2, tt, acc[(1 + r) * B]);
#pragma unroll
        for (int c = 0; c < D; c++) acc[(1 + r) * B + 1 + c] = fma(w2, tt * u8.r.t[c], acc[(1 + r) * B + 1 + c]);
      }
    }
  }
  double q[BB];
#pragma unroll
  for (int k = 0; k < BB; k++) q[k] = acc[k];
  if (own) {
#pragma unroll
    for (int k = 0; k < B; k++) q[k * B + k] += 2.0 * R.xi;
  }
#pragma unroll
  for (int k = 0; k < BB; k++) R.Qd[(size_t)row * BB + k] = q[k];
  if (!own) return;
  double g[BB], h[BB], dd[BB];
#pragma unroll
  for (int k = 0; k < BB; k++) {
    g[k] = R.Gbase[(size_t)row * BB + k] + acc[k];
    h[k] = R.Hbase[(size_t)row * BB + k] + acc[k];
    dd[k] = acc[k];
  }
#pragma unroll
  for (int k = 0; k < B; k++) dd[k * B + k] += R.xi;
#pragma unroll
  for (int k = 0; k < BB; k++) R.Dd[(size_t)row * BB + k] = dd[k];
  const int4 gp = R.gpos[row];
#pragma unroll
  for (int e = 0; e < BB; e++) R.Gval[(size_t)gp.x + (size_t)((e / PS) * gp.y + gp.z) * PS + e % PS] = g[e];
#pragma unroll
  for (int r = 0; r < B; r++) R.Gtcol[(size_t)gp.w * B + r] = g[r * B];
  R.att_val[R.att_pos[row]] = g[0];
  // T = 1 / H_tt, N = T H_tR, V = H_RR - H_Rt T H_tR   (DPGO_utils.cpp:2958-2964)
  const double T = 1.0 / h[0];
  R.Tinv[row] = T;
#pragma unroll
  for (int k = 0; k < D; k++) R.N[(size_t)row * D + k] = T * h[1 + k];
#pragma unroll
  for (int r = 0; r < D; r++)
#pragma unroll
    for (int c = 0; c < D; c++) R.V[((size_t)row * D + r) * D + c] = h[(1 + r) * B + 1 + c] - h[(1 + r) * B] * (T * h[1 + c]);
}

// ---------------------------------------------------------------------------
// Objective of a node at an arbitrary point Z (DPGOStar::evaluate_f, C++/DPGO/src/DPGOStar.cpp:713-761).
// Every edge is charged to its tail pose.  slot 0: sum over intra edges of the quadratic cost,
// slot 1: sum over inter edges of rho(|r_e|^2).  eform = 1 uses the quadratic form of the data
// matrix M (kappa |Y_i|^2 instead of kappa |R^T Y_i|^2: trivial loss, :722), eform = 0 the residual
// form of B0 / B1 (robust losses, :729-736).
// ---------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ double edge_cost(const InterEdgesDev &E, int e, const double *zi, const double *zj,
                                            int eform) {
  double Re[D * D], te[D];
#pragma unroll
  for (int i = 0; i < D * D; i++) Re[i] = E.R[(size_t)e * D * D + i];
#pragma unroll
  for (int i = 0; i < D; i++) te[i] = E.t[(size_t)e * D + i];
  const double tau = E.tau[e], kap = E.kappa[e];
  double sn = 0;
#pragma unroll
  for (int c = 0; c < D; c++) {
    double a = zi[c] - zj[c];
#pragma unroll
    for (int q = 0; q < D; q++) a = fma(te[q], zi[D + q * D + c], a);
    sn = fma(tau * a, a, sn);
  }
  if (eform) {
    // kappa (|Y_i|^2 + |Y_j|^2 - 2 <Y_i, R Y_j>)
    double s = 0;
#pragma unroll
    for (int q = 0; q < D; q++)
#pragma unroll
      for (int c = 0; c < D; c++) {
        double ry = 0;
#pragma unroll
        for (int r = 0; r < D; r++) ry = fma(Re[q * D + r], zj[D + r * D + c], ry);
        const double yi = zi[D + q * D + c], yj = zj[D + q * D + c];
        s += yi * yi + yj * yj - 2.0 * yi * ry;
      }
    sn = fma(kap, s, sn);
  } else {
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
      for (int c = 0; c < D; c++) {
        double a = -zj[D + r * D + c];
#pragma unroll
        for (int q = 0; q < D; q++) a = fma(Re[q * D + r], zi[D + q * D + c], a);
        sn = fma(kap * a, a, sn);
      }
  }
  return sn;
}

template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_cost(const Seg *segs, NodeMask mask, InterEdgesDev Ei, InterEdgesDev Ee,
                                              int eform, int loss, double dl, const double *Z, double *partial,
                                              int pstride) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  const bool active = node_on(mask, s.node);
  double part[2] = {0.0, 0.0};
  const int row = s.begin + threadIdx.x;
  if (active && row < s.end) {
    double zp[RS], zo[RS];
    load_vec<RS>(Z + (size_t)row * RS, zp);
    if (row < Ei.nrows_own)
      for (int k = Ei.inc_ptr[row]; k < Ei.inc_ptr[row + 1]; k++) {
        const int e = Ei.inc[k] >> 1;
        load_vec<RS>(Z + (size_t)Ei.head[e] * RS, zo);
        part[0] += edge_cost<D>(Ei, e, zp, zo, eform);
      }
    for (int k = Ee.inc_ptr[row]; k < Ee.inc_ptr[row + 1]; k++) {
      const int code = Ee.inc[k];
      if (code & 1) continue;   // charged to the tail only
      const int e = code >> 1;
      load_vec<RS>(Z + (size_t)Ee.head[e] * RS, zo);
      double w, rho;
      loss_weight(loss, dl, edge_cost<D>(Ee, e, zp, zo, eform), w, rho);
      part[1] += rho;
    }
  }
  if (active) block_store<2>(part, partial + SEGB, pstride);
}

// partial = sum |a_p - b_p|^2 over own rows
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_sqdist(const Seg *segs, NodeMask mask, const double *a, const double *b,
                                                double *partial) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  const bool active = node_on(mask, s.node);
  double pr[1] = {0.0};
  const int row = s.begin + threadIdx.x;
  if (active && row < s.end) {
    double va[RS], vb[RS];
    load_vec<RS>(a + (size_t)row * RS, va);
    load_vec<RS>(b + (size_t)row * RS, vb);
    double p = 0;
#pragma unroll
    for (int k = 0; k < RS; k++) { const double dd = va[k] - vb[k]; p = fma(dd, dd, p); }
    pr[0] = p;
  }
  if (active) block_store<1>(pr, partial + SEGB, 0);
}

// ---------------------------------------------------------------------------
// Per-pose kernels.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_proximal(const Seg *segs, NodeMask mask, const double *Z,
                                                  const double *Df, const double *Tinv, const double *Nv,
                                                  const double *Vb, double *Xout, double *Xref,
                                                  double *partial) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  const bool active = node_on(mask, s.node);
  double part[1] = {0.0};
  const int row = s.begin + threadIdx.x;
  if (active && row < s.end) {
    double z[RS], df[RS];
    load_vec<RS>(Z + (size_t)row * RS, z);
    load_vec<RS>(Df + (size_t)row * RS, df);
    part[0] = proximal_row<D>(row, z, df, Tinv, Nv, Vb, Xout, Xref);
  }
  if (partial && active) block_store<1>(part, partial + SEGB, 0);
}

template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_extrapolate(const Seg *segs, NodeMask mask, NodeCoefs gamma, const double *gamma_dev,
                                                     const double *a, const double *b, double *out) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  if (!node_on(mask, s.node)) return;
  const int row = s.begin + threadIdx.x;
  if (row >= s.end) return;
  const double gm = gamma_dev ? gamma_dev[s.node] : gamma.a[s.node];   // (gamma_dev: a launch that may be replayed, k_set_coefs)
  double va[RS], vb[RS];
  load_vec<RS>(a + (size_t)row * RS, va);
  load_vec<RS>(b + (size_t)row * RS, vb);
#pragma unroll
  for (int k = 0; k < RS; k++) va[k] = fma(gm, va[k] - vb[k], va[k]);
  store_vec<RS>(out + (size_t)row * RS, va);
}

// the trivial loss's extrapolated point in ONE launch (DPGOHash.cpp:255-262): Y = X[k] + gamma (X[k] - X[k-1]) over all rows,
// and over the own rows also g and Dfobj (three launches of k_extrapolate before round 6: the same operations)
struct Extrap3 {
  const double *a[3], *b[3];
  double *out[3];
};
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_extrapolate3(const Seg *segs, NodeMask mask, NodeCoefs gamma, const double *gamma_dev,
                                                      int nseg_own, Extrap3 E) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  if (!node_on(mask, s.node)) return;
  const int row = s.begin + threadIdx.x;
  if (row >= s.end) return;
  const double gm = gamma_dev ? gamma_dev[s.node] : gamma.a[s.node];
  const int n = SEGB < nseg_own ? 3 : 1;
#pragma unroll
  for (int q = 0; q < 3; q++)
    if (q < n) {
      double va[RS], vb[RS];
      load_vec<RS>(E.a[q] + (size_t)row * RS, va);
      load_vec<RS>(E.b[q] + (size_t)row * RS, vb);
#pragma unroll
      for (int k = 0; k < RS; k++) va[k] = fma(gm, va[k] - vb[k], va[k]);
      store_vec<RS>(E.out[q] + (size_t)row * RS, va);
    }
}

// out = alpha a + beta b on the whole record (PART 0), the translation row (1) or the rotation rows (2);
// whole 16-byte loads and stores, the untouched part of `out` is carried through registers
template <int D, int PART>
__global__ __launch_bounds__(SEG_ROWS) void k_axpby(const Seg *segs, NodeMask mask, double alpha, const double *a,
                                                    double beta, const double *b, double *out, double *out2) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  if (!node_on(mask, s.node)) return;
  const int row = s.begin + threadIdx.x;
  if (row >= s.end) return;
  double va[RS], vb[RS], vo[RS];
  load_vec<RS>(a + (size_t)row * RS, va);
  if (b) load_vec<RS>(b + (size_t)row * RS, vb);
  if (PART != 0) load_vec<RS>(out + (size_t)row * RS, vo);
#pragma unroll
  for (int k = 0; k < RS; k++) {
    const bool in = PART == 0 || (PART == 1 ? k < D : k >= D);
    const double v = b ? fma(beta, vb[k], alpha * va[k]) : alpha * va[k];
    va[k] = in ? v : vo[k];
  }
  store_vec<RS>(out + (size_t)row * RS, va);
  if (PART == 0 && out2) store_vec<RS>(out2 + (size_t)row * RS, va);   // a second copy of the result
}

// the tail of iterate() and the exchange's pack in one launch (kernels.h: launch_tail_pack): workgroups [0, nseg_own) copy
// the masked nodes' own records xak -> xk (and z), the rest copy the exported records xak[pack_rows[k]] -> pack[k]
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_tail_pack(const Seg *segs, NodeMask mask, int nseg_own, const double *xak, double *xk,
                                                        double *z, const int *pack_rows, int npack, double *pack) {
  constexpr int RS = Dim<D>::RS;
  double v[RS];
  if ((int)blockIdx.x >= nseg_own) {
    const int k = ((int)blockIdx.x - nseg_own) * SEG_ROWS + (int)threadIdx.x;
    if (k >= npack) return;
    load_vec<RS>(xak + (size_t)pack_rows[k] * RS, v);
    store_vec<RS>(pack + (size_t)k * RS, v);
    return;
  }
  const Seg s = segs[xcd_seg((int)blockIdx.x, nseg_own)];
  if (!node_on(mask, s.node)) return;
  const int row = s.begin + threadIdx.x;
  if (row >= s.end) return;
  load_vec<RS>(xak + (size_t)row * RS, v);
  store_vec<RS>(xk + (size_t)row * RS, v);
  if (z) store_vec<RS>(z + (size_t)row * RS, v);
}

// out = alpha[node] * a + beta[node] * b  (per-node coefficients: batched CG updates)
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_axpby_node(const Seg *segs, NodeMask mask, NodeCoefs C,
                                                    const double *a, const double *b, double *out) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  if (!node_on(mask, s.node)) return;
  const int row = s.begin + threadIdx.x;
  if (row >= s.end) return;
  const double al = C.a[s.node], be = C.b[s.node];
  double va[RS], vb[RS];
  load_vec<RS>(a + (size_t)row * RS, va);
  load_vec<RS>(b + (size_t)row * RS, vb);
#pragma unroll
  for (int k = 0; k < RS; k++) va[k] = fma(be, vb[k], al * va[k]);
  store_vec<RS>(out + (size_t)row * RS, va);
}

// out.Y row (pose, r) = dinv[pose * D + r] * in.Y row: the Jacobi preconditioner diag(G_RR)^-1 (DPGOProblem.cpp:96-98, 583-585)
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_rot_rowscale(const Seg *segs, NodeMask mask, const double *dinv, const double *in,
                                                      double *out) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  if (!node_on(mask, s.node)) return;
  const int row = s.begin + threadIdx.x;
  if (row >= s.end) return;
#pragma unroll
  for (int r = 0; r < D; r++) {
    const double a = dinv[(size_t)row * D + r];
#pragma unroll
    for (int c = 0; c < D; c++) out[(size_t)row * RS + D + r * D + c] = a * in[(size_t)row * RS + D + r * D + c];
  }
}

// one CG step of every node in the mask (IterativeSolvers.h:340-390): s += c p, H s += c H p and, where the node
// goes on (cr != 0), r += cr H p
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_cg_step(const Seg *segs, NodeMask mask, NodeCoefs C, const double *p,
                                                      const double *Hp, double *s, double *hs, double *r,
                                                      const CgNode *cg, const double *r0, const double *X, double *xprop,
                                                      const NodeBits *rmask) {
  constexpr int RS = Dim<D>::RS;
  const Seg sg = segs[SEGM(mask)];
  if (!node_on(mask, sg.node)) return;
  const int row = sg.begin + threadIdx.x;
  if (row >= sg.end) return;
  // xprop: the nodes of *rmask (their CG ended with this step) go straight on to their trial point's rotations,
  // xprop.Y = proj_SO(d)(X.Y + s.Y) -- k_rot_op mode 2 without a launch of its own
  const bool retract = xprop && ((*rmask >> sg.node) & 1ull);
  double xr[RS];
  if (retract) load_vec<RS>(X + (size_t)row * RS, xr);
  const double cc = cg ? cg[sg.node].c1 : C.a[sg.node], cc_r = cg ? cg[sg.node].cr : C.b[sg.node];
  double vp[RS], vh[RS], v[RS];
  load_vec<RS>(p + (size_t)row * RS, vp);
  load_vec<RS>(Hp + (size_t)row * RS, vh);
  // r0 != nullptr: the first step of a CG run -- s and H s start from zero (not read), r from r0 (the gradient)
  if (r0) {
#pragma unroll
    for (int k = 0; k < RS; k++) v[k] = 0.0;
  } else {
    load_vec<RS>(s + (size_t)row * RS, v);
  }
#pragma unroll
  for (int k = 0; k < RS; k++) v[k] = fma(cc, vp[k], 1.0 * v[k]);
  store_vec<RS>(s + (size_t)row * RS, v);
  if (retract) {
    double M[D * D], o[RS];
#pragma unroll
    for (int k = 0; k < D; k++) o[k] = 0.0;
#pragma unroll
    for (int k = 0; k < D * D; k++) M[k] = xr[D + k] + v[D + k];
    project_sod<D>(M, o + D);
    store_vec<RS>(xprop + (size_t)row * RS, o);
  }
  if (r0) {
#pragma unroll
    for (int k = 0; k < RS; k++) v[k] = 0.0;
  } else {
    load_vec<RS>(hs + (size_t)row * RS, v);
  }
#pragma unroll
  for (int k = 0; k < RS; k++) v[k] = fma(cc, vh[k], 1.0 * v[k]);
  store_vec<RS>(hs + (size_t)row * RS, v);
  if (cc_r != 0.0) {
    load_vec<RS>((r0 ? r0 : r) + (size_t)row * RS, v);
#pragma unroll
    for (int k = 0; k < RS; k++) v[k] = fma(cc_r, vh[k], 1.0 * v[k]);
    store_vec<RS>(r + (size_t)row * RS, v);
  }
}

// p = -v + beta[node] p with the device-resident beta of k_cg_scal (IterativeSolvers.h:386-388)
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_cg_dir(const Seg *segs, NodeMask mask, const CgNode *cg, const double *v,
                                                     double *p) {
  constexpr int RS = Dim<D>::RS;
  const Seg sg = segs[SEGM(mask)];
  if (!node_on(mask, sg.node)) return;
  const int row = sg.begin + threadIdx.x;
  if (row >= sg.end) return;
  const double be = cg[sg.node].be;
  double vv[RS], vp[RS];
  load_vec<RS>(v + (size_t)row * RS, vv);
  load_vec<RS>(p + (size_t)row * RS, vp);
#pragma unroll
  for (int k = 0; k < RS; k++) vp[k] = fma(be, vp[k], -1.0 * vv[k]);
  store_vec<RS>(p + (size_t)row * RS, vp);
}

template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_cg_init(const Seg *segs, NodeMask mask, const double *grad,
                                                      const double *pgrad, double *s, double *hs, double *r, double *v,
                                                      double *p) {
  constexpr int RS = Dim<D>::RS;
  const Seg sg = segs[SEGB];
  if (!node_on(mask, sg.node)) return;
  const int row = sg.begin + threadIdx.x;
  if (row >= sg.end) return;
  double g[RS], pg[RS], z[RS];
  load_vec<RS>(pgrad + (size_t)row * RS, pg);
  if (s) {   // (s == nullptr: only p = -P grad; the first k_cg_step then takes s = H s = 0 and r = grad as given)
    load_vec<RS>(grad + (size_t)row * RS, g);
#pragma unroll
    for (int k = 0; k < RS; k++) z[k] = 0.0;
    store_vec<RS>(s + (size_t)row * RS, z);
    store_vec<RS>(hs + (size_t)row * RS, z);
    store_vec<RS>(r + (size_t)row * RS, g);
    store_vec<RS>(v + (size_t)row * RS, pg);
  }
#pragma unroll
  for (int k = 0; k < RS; k++) pg[k] = -1.0 * pg[k];
  store_vec<RS>(p + (size_t)row * RS, pg);
}

// up to MAX_DOTS dot products in one pass (TNT / CG scalars); part[q]: 0 whole record, 1 translation row,
// 2 rotation rows.  Always stores MAX_DOTS slots (zeros behind the n-th).
struct DotPairs {
  const double *a[MAX_DOTS];
  const double *b[MAX_DOTS];
  int part[MAX_DOTS];
  int n;
};
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_dots(const Seg *segs, NodeMask mask, DotPairs P, double *partial,
                                                   int pstride) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  const bool active = node_on(mask, s.node);
  double pr[MAX_DOTS];
#pragma unroll
  for (int q = 0; q < MAX_DOTS; q++) pr[q] = 0.0;
  const int row = s.begin + threadIdx.x;
  if (active && row < s.end) {
#pragma unroll
    for (int q = 0; q < MAX_DOTS; q++)
      if (q < P.n) {
        const int k0 = P.part[q] == 2 ? D : 0, k1 = P.part[q] == 1 ? D : RS;
        double va[RS], vb[RS];
        load_vec<RS>(P.a[q] + (size_t)row * RS, va);
        load_vec<RS>(P.b[q] + (size_t)row * RS, vb);
        double p = 0;
#pragma unroll
        for (int k = 0; k < RS; k++) p = (k >= k0 && k < k1) ? fma(va[k], vb[k], p) : p;
        pr[q] = p;
      }
  }
  if (active) block_store<MAX_DOTS>(pr, partial + SEGB, pstride);
}

template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_tangent_full(const Seg *segs, NodeMask mask, const double *X,
                                                      const double *V, const double *add, double *sum_out,
                                                      double *out, double *partial) {
  constexpr int RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  const bool active = node_on(mask, s.node);
  double pr[1] = {0.0};
  const int row = s.begin + threadIdx.x;
  if (active && row < s.end) {
    double x[RS], v[RS], o[RS];
    load_vec<RS>(X + (size_t)row * RS, x);
    load_vec<RS>(V + (size_t)row * RS, v);
    if (add) {   // the vector is V + add (and is wanted: Dfobj = G X + g)
      double w[RS];
      load_vec<RS>(add + (size_t)row * RS, w);
#pragma unroll
      for (int k = 0; k < RS; k++) v[k] += w[k];
      if (sum_out) store_vec<RS>(sum_out + (size_t)row * RS, v);
    }
#pragma unroll
    for (int k = 0; k < D; k++) o[k] = v[k];
    tangent_proj<D>(x + D, v + D, o + D);
    double p = 0;
#pragma unroll
    for (int k = 0; k < RS; k++) p = fma(o[k], o[k], p);
    pr[0] = p;
    if (out) store_vec<RS>(out + (size_t)row * RS, o);
  }
  if (partial && active) block_store<1>(pr, partial + SEGB, 0);
}

// mode 0: out.Y = Proj_R(in.Y)                         (reduced_tangent_space_projection)
// mode 2: out.Y = proj_SO(d)(R + in.Y)                  (SOdProduct::retract)
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_rot_op(const Seg *segs, NodeMask mask, int mode, const double *X,
                                                const double *in, const double *dotv, double *partial,
                                                double *out, int pstride, int two, double *neg) {
  constexpr int RS = Dim<D>::RS;
  const int si = SEGM(mask);
  const Seg s = segs[si];
  if (!node_on(mask, s.node)) return;
  const int row = s.begin + threadIdx.x;
  double pr[2] = {0.0, 0.0};
  if (row < s.end) {
    double x[RS], v[RS], o[RS];
    load_vec<RS>(X + (size_t)row * RS, x);
    load_vec<RS>(in + (size_t)row * RS, v);
#pragma unroll
    for (int k = 0; k < D; k++) o[k] = 0.0;
    const double *R = x + D;
    if (mode == 0) {
      tangent_proj<D>(R, v + D, o + D);
    } else {
      double M[D * D];
#pragma unroll
      for (int k = 0; k < D * D; k++) M[k] = R[k] + v[D + k];
      project_sod<D>(M, o + D);
    }
    store_vec<RS>(out + (size_t)row * RS, o);
    if (partial) {   // <dotv, out> over the rotation rows; two: <out, out> first, then <dotv, out>
      double dv[RS];
      load_vec<RS>(dotv + (size_t)row * RS, dv);
      double p = 0, q = 0;
#pragma unroll
      for (int k = D; k < RS; k++) {
        p = fma(dv[k], o[k], p);
        q = fma(o[k], o[k], q);
      }
      pr[0] = two ? q : p;
      pr[1] = p;
    }
    if (neg) {   // the first CG direction p_0 = -P grad (IterativeSolvers.h:258)
#pragma unroll
      for (int k = 0; k < RS; k++) o[k] = -1.0 * o[k];
      store_vec<RS>(neg + (size_t)row * RS, o);
    }
  }
  if (partial) {
    if (two) block_store<2>(pr, partial + si, pstride);
    else block_store<1>(reinterpret_cast<const double(&)[1]>(pr), partial + si, 0);
  }
}

// dst[didx[k]] = src[sidx[k]]  (didx == nullptr: dst[k]); halo copies, pack and unpack
template <int D>
__global__ __launch_bounds__(256) void k_copy_indexed(int count, const int *didx, const int *sidx,
                                                      const double *src, double *dst, const NodeBits *gate) {
  constexpr int RS = Dim<D>::RS;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= count || (gate && *gate == 0ull)) return;   // gate: a device word that switches the whole launch off (k_reduce_gate)
  double r[RS];
  load_vec<RS>(src + (size_t)sidx[k] * RS, r);
  store_vec<RS>(dst + (size_t)(didx ? didx[k] : k) * RS, r);
}

// partial = sum_p < x_p , coef * (D_p x_p) + addcoef * add_p >   (own rows, D block diagonal)
template <int D>
__global__ __launch_bounds__(SEG_ROWS) void k_bdiag_dot(const Seg *segs, NodeMask mask, const double *Dd,
                                                   const double *x, double coef, const double *add,
                                                   double addcoef, double *partial) {
  constexpr int B = Dim<D>::B, RS = Dim<D>::RS;
  const Seg s = segs[SEGB];
  const bool active = node_on(mask, s.node);
  double pr[1] = {0.0};
  const int row = s.begin + threadIdx.x;
  if (active && row < s.end) {
    double xv[RS], dx[RS], av[RS];
    load_vec<RS>(x + (size_t)row * RS, xv);
    load_vec<RS>(add + (size_t)row * RS, av);
#pragma unroll
    for (int k = 0; k < RS; k++) dx[k] = 0.0;
    blk_mul_acc<D, false>(Dd + (size_t)row * B * B, xv, dx);
    double p = 0;
#pragma unroll
    for (int k = 0; k < RS; k++) p = fma(xv[k], fma(coef, dx[k], addcoef * av[k]), p);
    pr[0] = p;
  }
  if (active) block_store<1>(pr, partial + SEGB, 0);
}

// One wave per (node, slot): sums the node's per-segment partials in a fixed order and writes the scalar
// straight into pinned host memory; the last wave to finish raises the host-visible flag to `seq`, so the
// host gets the numbers by polling a cache line instead of a copy + stream synchronisation.
__global__ __launch_bounds__(64) void k_reduce(SegTable T, int all_rows, int nslots, const double *partials,
                                               double *host_scalars, unsigned *arrived, unsigned long long *host_flag,
                                               unsigned long long seq, unsigned long long *dev_seq) {
  const int a = blockIdx.x / nslots, s = blockIdx.x % nslots, lane = threadIdx.x;
  const double *p = partials + (size_t)s * T.nseg_all;
  double v = 0;
  for (int k = T.own_ptr[a] + lane; k < T.own_ptr[a + 1]; k += 64) v += p[k];
  if (all_rows)
    for (int k = T.nbr_ptr[a] + lane; k < T.nbr_ptr[a + 1]; k += 64) v += p[k];
  v = wave_sum(v);
  if (lane == 0) {
    __hip_atomic_store(host_scalars + a * MAX_SLOTS + s, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope: the scalar is on its way before the count
    const unsigned done = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (seq == 0: a launch replayed from a captured graph takes the next value from the device's own count, see k_cg_scal)
      if (seq == 0) seq = *dev_seq + 1;
      *dev_seq = seq;
      __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The per-iteration coefficients of a node (the Nesterov gamma of the extrapolation) as a DEVICE array: launches replayed from
// a captured graph cannot carry fresh by-value arguments, so the one eager launch of an iteration that knows them writes
// them here and the replayed kernels read them (k_extrapolate, k_inter).
__global__ __launch_bounds__(64) void k_set_coefs(NodeCoefs C, int n, double *dev) {
  if ((int)threadIdx.x < n) dev[threadIdx.x] = C.a[threadIdx.x];
}

// ---------------------------------------------------------------------------
// Device-side control of the Steihaug-Toint CG (IterativeSolvers.h:207-390): the scalar recurrences of every
// node live in CgNode records, the set of still-iterating nodes in two device masks (dmask[0]: the nodes of the
// step under way -- Hessian product and s / H s update; dmask[1]: the nodes that go on to the preconditioner and
// to the next step).  The host enqueues
// whole CG steps and only polls the summary these kernels write to pinned memory.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cg_begin(int nnodes, NodeBits bits, CgStart S, int max_it, CgNode *cg,
                                                 NodeBits *dmask) {
  const int a = threadIdx.x;
  if (a < nnodes && ((bits >> a) & 1ull)) {
    CgNode c;
    c.sk_M_pk = 0.0; c.sk_M_2 = 0.0; c.pk_M_2 = S.rv[a]; c.rv = S.rv[a];
    c.Delta = S.Delta[a]; c.Delta_2 = S.Delta[a] * S.Delta[a]; c.target = S.target[a]; c.h_M_norm = 0.0;
    c.c1 = 0.0; c.cr = 0.0; c.al = 0.0; c.kap = 0.0; c.be = 0.0;
    c.cg_it = 0; c.max_it = max_it; c.stop_ord = 0;
    // the stopping test of the first step (:285-291)
    c.live = !(c.cg_it >= max_it || sqrt(c.rv) <= c.target);
    if (!c.live) c.h_M_norm = sqrt(c.sk_M_2);
    cg[a] = c;
  }
  const bool live = a < nnodes && ((bits >> a) & 1ull) && cg[a].live;
  const NodeBits m = __ballot(live);
  if (a == 0) { dmask[0] = m; dmask[1] = m; dmask[2] = bits & ~m; }
}

// The start of a trust-region iteration without a host round trip (TNT.h:446-484, IterativeSolvers.h:230-291): per node the
// sums |grad|^2, <X, nabla>, <X, g>, <X, g_alt> (partial slots 0..3) and |P grad|^2, <grad, P grad> (slots MAX_DOTS,
// MAX_DOTS + 1) are reduced in the order of k_reduce and written to pinned host memory (the host reads them whenever it
// next waits, and forms the same norms from the same bits); the gradient-norm tests decide which nodes iterate at all;
// those get the start values of their CG (k_cg_begin's job, with <r_0, v_0> taken from the sums).
// dmask[0] = dmask[1] = the nodes whose CG runs, dmask[2] = active nodes whose CG is over.
struct TntBegin {
  NodeBits bits;          // the candidates (iteration limits are the host's business)
  int use_precon, max_it;
  double grad_tol, pgrad_tol, kappa, theta;
  double Delta[MAX_LOCAL_NODES];
};
// the per-node part: v = the six sums (|grad|^2, <X, nabla>, <X, g>, <X, g_alt>, |P grad|^2, <grad, P grad>); one thread
__device__ __forceinline__ void tnt_begin_node(int a, bool mine, const double (&v)[6], int use_precon, int max_it, double grad_tol,
                                               double pgrad_tol, double kappa, double theta, double Delta, CgNode *cg,
                                               NodeBits *dmask, double *host_tnt, double *dev_tnt = nullptr) {
  bool active = false, live = false;
  if (mine) {
    const double gnorm = sqrt(v[0]), pgnorm = use_precon ? sqrt(v[4]) : gnorm, rv0 = use_precon ? v[5] : v[0];
    active = !(gnorm < grad_tol) && !(pgnorm < pgrad_tol);
    CgNode c;
    c.sk_M_pk = 0.0; c.sk_M_2 = 0.0; c.pk_M_2 = rv0; c.rv = rv0;
    c.Delta = Delta; c.Delta_2 = Delta * Delta;
    const double r0 = sqrt(rv0);
    c.target = r0 * fmin(kappa, pow(r0, theta));
    c.h_M_norm = 0.0;
    c.c1 = 0.0; c.cr = 0.0; c.al = 0.0; c.kap = 0.0; c.be = 0.0;
    c.cg_it = 0; c.max_it = max_it; c.stop_ord = 0;
    c.live = active && !(c.cg_it >= max_it || sqrt(c.rv) <= c.target);
    if (!c.live) c.h_M_norm = sqrt(c.sk_M_2);
    cg[a] = c;
    live = c.live;
#pragma unroll
    for (int q = 0; q < 6; q++) __hip_atomic_store(host_tnt + a * TNT_SUMMARY + q, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_tnt + a * TNT_SUMMARY + 6, active ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (dev_tnt) {   // (k_reduce_gate)
#pragma unroll
      for (int q = 0; q < 6; q++) dev_tnt[a * TNT_SUMMARY + q] = v[q];
      dev_tnt[a * TNT_SUMMARY + 6] = active ? 1.0 : 0.0;
    }
  }
  // every node owns its bit of the three masks (a node outside `bits` clears it): no word is written as a whole, so the
  // workgroups need not meet
  const NodeBits bit = 1ull << a;
  if (live) { atomicOr(dmask + 0, bit); atomicOr(dmask + 1, bit); atomicAnd(dmask + 2, ~bit); }
  else {
    atomicAnd(dmask + 0, ~bit); atomicAnd(dmask + 1, ~bit);
    if (active) atomicOr(dmask + 2, bit);
    else atomicAnd(dmask + 2, ~bit);
  }
}
__global__ __launch_bounds__(384) void k_tnt_begin(SegTable T, int nnodes, TntBegin B, const double *partials, CgNode *cg,
                                                   NodeBits *dmask, double *host_tnt) {
  // one workgroup per node, one wave per sum (six independent reductions side by side; each in the order of k_reduce)
  __shared__ double sums[6];
  const int a = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool mine = (B.bits >> a) & 1ull;
  if (mine && (wv < 4 || B.use_precon)) {
    const int slot = wv < 4 ? wv : MAX_DOTS + (wv - 4);
    const double *p = partials + (size_t)slot * T.nseg_all;
    double t = 0;
    for (int k = T.own_ptr[a] + lane; k < T.own_ptr[a + 1]; k += 64) t += p[k];
    t = wave_sum(t);
    if (lane == 0) sums[wv] = t;
  } else if (lane == 0) {
    sums[wv] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double v[6];
#pragma unroll
  for (int q = 0; q < 6; q++) v[q] = sums[q];
  tnt_begin_node(a, mine, v, B.use_precon, B.max_it, B.grad_tol, B.pgrad_tol, B.kappa, B.theta, B.Delta[a], cg, dmask, host_tnt);
}

// The scalar step of STPCG for one node from the sums of a phase (v[0..3]: phase 0 <p, H p>, <H p, H p>, <p, p>, <p, r>;
// phase 1: v[0] = <r, v>), IterativeSolvers.h:296-390 -- shared by k_cg_scal and by the vector kernels that take the
// step (kept apart from the launch plumbing).
__device__ __forceinline__ void cg_scal_logic(int phase, const double (&v)[4], CgNode &c) {
  if (phase == 0) {
    const double kappa_k = v[0];
    bool stop = false;
    if (sqrt(v[1]) / sqrt(v[2]) < 1e-8) {   // p is (numerically) in the kernel of H (:305-338)
      double sgn = 1.0;
      if (v[3] < 0) { sgn = -1.0; c.sk_M_pk = -c.sk_M_pk; }
      const double sigma = (-c.sk_M_pk + sqrt(c.sk_M_pk * c.sk_M_pk + c.pk_M_2 * (c.Delta_2 - c.sk_M_2))) / c.pk_M_2;
      c.c1 = sgn * sigma;
      stop = true;
    } else {
      const double alpha = c.rv / kappa_k;
      const double skp1 = c.sk_M_2 + 2 * alpha * c.sk_M_pk + alpha * alpha * c.pk_M_2;
      if (kappa_k <= 0 || skp1 > c.Delta_2) {   // negative curvature or the step leaves the region (:347-362)
        c.c1 = (-c.sk_M_pk + sqrt(c.sk_M_pk * c.sk_M_pk + c.pk_M_2 * (c.Delta_2 - c.sk_M_2))) / c.pk_M_2;
        stop = true;
      } else {
        c.c1 = alpha; c.cr = alpha; c.al = alpha; c.kap = kappa_k; c.sk_M_2 = skp1;
      }
    }
    if (stop) { c.cr = 0.0; c.h_M_norm = c.Delta; c.live = 0; c.stop_ord = 2 * c.cg_it + 1; }
  } else {
    const double rk_vk = v[0];
    const double be = rk_vk / (c.al * c.kap);   // (:364-390)
    c.sk_M_pk = be * (c.sk_M_pk + c.al * c.pk_M_2);
    c.pk_M_2 = rk_vk + be * be * c.pk_M_2;
    c.rv = rk_vk;
    c.be = be;
    c.cg_it++;
    if (c.cg_it >= c.max_it || sqrt(c.rv) <= c.target) {   // the stopping test of the next step (:285-291)
      c.h_M_norm = sqrt(c.sk_M_2);
      c.live = 0;
      c.stop_ord = 2 * c.cg_it;
    }
  }
}

// the per-node part of a scalar step: one thread.  `mine`: the node is part of this phase (its bit of dmask[phase])
__device__ __forceinline__ void cg_scal_node(int a, int phase, bool mine, const double (&v)[4], CgNode *cg, NodeBits *dmask,
                                             double *host_scalars) {
  CgNode c = cg[a];
  if (mine) {
    cg_scal_logic(phase, v, c);
    cg[a] = c;
    // a node that stops leaves dmask[1] now; dmask[0] (the nodes of the step under way, which still take the
    // s / H s update of this step) follows at the end of phase 1
    if (!c.live) {
      atomicAnd(dmask + 1, ~(1ull << a));
      atomicOr(dmask + 2, 1ull << a);    // ... and joins the nodes whose trial point can be taken
    }
  }
  // (the summary has its own pinned area, CG_SUMMARY doubles per node: it must survive the read-back of a trial point
  // that was enqueued behind this step)
  __hip_atomic_store(host_scalars + a * CG_SUMMARY + 0, c.live ? CG_LIVE_ORD : (double)c.stop_ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_scalars + a * CG_SUMMARY + 1, c.h_M_norm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_scalars + a * CG_SUMMARY + 2, (double)c.cg_it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the last of `expected` arrivals raises the host's flag (shared by k_reduce's successors; one thread)
// seq: the value the host's flag is raised to.  A launch replayed from a captured graph cannot carry a fresh value in its
// arguments: with seq == 0 the value is the device word *dev_seq + 1; either way *dev_seq ends up holding the value used,
// so that eager launches and replays can follow each other (the host counts along: Group::fetch_seq_).
__device__ __forceinline__ void flag_arrive(unsigned *arrived, unsigned expected, int phase, NodeBits *dmask,
                                            unsigned long long *host_flag, unsigned long long seq, unsigned long long *dev_seq) {
  __atomic_thread_fence(__ATOMIC_RELEASE);
  const unsigned done = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
  if (done == expected - 1) {
    if (phase == 1) dmask[0] = __hip_atomic_load(dmask + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next step's nodes
    __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seq == 0) seq = *dev_seq + 1;
    *dev_seq = seq;
    __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// one wave per node; the partial sums are combined in the order of k_reduce
__global__ __launch_bounds__(256) void k_cg_scal(SegTable T, int phase, const double *partials, CgNode *cg,
                                                 NodeBits *dmask, double *host_scalars, unsigned *arrived,
                                                 unsigned long long *host_flag, unsigned long long seq,
                                                 unsigned long long *dev_seq) {
  // one workgroup per node; the (up to) four sums of a phase are reduced side by side, one wave each
  __shared__ double sums[4];
  const int a = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const NodeBits on = dmask[phase];      // (read before anyone clears a bit of it: bits are only ever cleared)
  const bool mine = (on >> a) & 1ull;
  const int ns = phase == 0 ? 4 : 1;
  if (mine && wv < ns) {
    const double *p = partials + (size_t)wv * T.nseg_all;
    double t = 0;
    for (int k = T.own_ptr[a] + lane; k < T.own_ptr[a + 1]; k += 64) t += p[k];
    t = wave_sum(t);
    if (lane == 0) sums[wv] = t;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (mine)
    for (int q = 0; q < ns; q++) v[q] = sums[q];
  cg_scal_node(a, phase, mine, v, cg, dmask, host_scalars);
  flag_arrive(arrived, gridDim.x, phase, dmask, host_flag, seq, dev_seq);
}

// k_tnt_begin and the first step's k_cg_scal in one launch: the start of a refinement (norms, gradient tests, CG start
// values) is only needed from the step-length logic of the first CG step on -- the first Hessian product runs for every
// candidate (one that fails the gradient tests has done it for nothing; it happens at the very end of a run) -- so both sit
// behind that product: ten sums per node side by side (the refinement's six from slots 0..3 and MAX_DOTS.., the step's four
// from slot cg_slot0 on: the product must not overwrite the refinement's), then tnt_begin_node and, for a node whose CG runs,
// the phase-0 logic.  One workgroup per node, a wave per sum.
constexpr int CG_FIRST_SLOT = 16;   // (six consecutive slots nobody else uses: the product's epilogue always stores six)
__global__ __launch_bounds__(1024) void k_cg_scal_begin(SegTable T, TntBegin B, const double *partials, CgNode *cg, NodeBits *dmask,
                                                        double *host_tnt, double *host_scalars, unsigned *arrived,
                                                        unsigned long long *host_flag, unsigned long long seq,
                                                        unsigned long long *dev_seq, double *dev_tnt, int upd_nslots, double *upd_host) {
  // waves 0..9: the refinement's six sums and the step's four; waves 10..: the sums of the last update() (own AND neighbour
  // segments, k_reduce's order), for every node -- update() covers the whole group whatever the refinement's candidates are
  __shared__ double sums[16];
  const int a = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool cand = (B.bits >> a) & 1ull;
  if (wv >= 10) {
    const int q = wv - 10;
    if (q < upd_nslots) {
      const double *p = partials + (size_t)(UPD_SLOT0 + q) * T.nseg_all;
      double t = 0;
      for (int k = T.own_ptr[a] + lane; k < T.own_ptr[a + 1]; k += 64) t += p[k];
      for (int k = T.nbr_ptr[a] + lane; k < T.nbr_ptr[a + 1]; k += 64) t += p[k];
      t = wave_sum(t);
      if (lane == 0) sums[wv] = t;
    }
  } else if (cand && (wv < 4 || wv >= 6 || B.use_precon)) {
    const int slot = wv < 4 ? wv : (wv < 6 ? MAX_DOTS + (wv - 4) : CG_FIRST_SLOT + (wv - 6));
    const double *p = partials + (size_t)slot * T.nseg_all;
    double t = 0;
    for (int k = T.own_ptr[a] + lane; k < T.own_ptr[a + 1]; k += 64) t += p[k];
    t = wave_sum(t);
    if (lane == 0) sums[wv] = t;
  } else if (lane == 0) {
    sums[wv] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int q = 0; q < upd_nslots; q++) __hip_atomic_store(upd_host + a * MAX_SLOTS + q, sums[10 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  double v6[6];
#pragma unroll
  for (int q = 0; q < 6; q++) v6[q] = sums[q];
  tnt_begin_node(a, cand, v6, B.use_precon, B.max_it, B.grad_tol, B.pgrad_tol, B.kappa, B.theta, B.Delta[a], cg, dmask, host_tnt, dev_tnt);
  // (the node's own bit of dmask[0], as tnt_begin_node has just left it: nobody else writes it)
  const bool mine = cand && cg[a].live;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (mine)
    for (int q = 0; q < 4; q++) v[q] = sums[6 + q];
  cg_scal_node(a, 0, mine, v, cg, dmask, host_scalars);
  flag_arrive(arrived, gridDim.x, 0, dmask, host_flag, seq, dev_seq);
}

#ifdef SPD_TRACE   /* measurement build only: per-tile phase timestamps (100 MHz wall clock), see spd_profile() */
__device__ unsigned long long *g_spd_trace = nullptr;
#define SPD_T(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr[i] = wall_clock64(); }
#define SPD_TRACE_DECL unsigned long long tr[6] = {trace_t0, 0, 0, 0, 0, 0};
#define SPD_TRACE_FLUSH(cond) if (g_spd_trace && (cond)) { _Pragma("unroll") for (int q = 0; q < 6; q++) g_spd_trace[(size_t)trace_slot * 6 + q] = tr[q]; }
#else
#define SPD_T(i)
#define SPD_TRACE_DECL
#define SPD_TRACE_FLUSH(cond)
#endif
// ---------------------------------------------------------------------------
// Multifrontal SPD solve.  A tile = ROWS rows (forward: rows of W_s^T's transpose, i.e. outputs
// [y_s ; dupd]; backward: pivots x_s) of one front; NW waves share a tile and split the reduction
// length between them chunk by chunk.  The front's input vector is staged through LDS in chunks;
// W / WT are read coalesced (lane = row / column).
//
// What bounds a small level is not bytes but the chain of dependent loads inside a tile, so:
//   * a tile's whole description (front sizes, offsets) travels in its 64-byte SpdItem: one load;
//   * the first batch of matrix loads of every chunk is issued BEFORE the chunk's input vector is
//     gathered (the two are independent) where a level has few tiles;
//   * the update buffer is stored in pull order (see pull_updates): a parent reads its children's
//     contributions as a contiguous run, added in list order (the order of the host solve: results do
//     not depend on the schedule).
// ---------------------------------------------------------------------------
// NT: the panels are read once per solve; non-temporal loads keep a factor that cannot stay in the Infinity
// Cache anyway from displacing what the kernels between two solves re-read (operators, vectors, and a
// smaller factor that does fit)
#define LDW(p) ((double)(NT ? __builtin_nontemporal_load(p) : *(p)))
// waves per workgroup of the 16-row class (16 waves per tile measured 5 % slower than 8)
#ifndef SPD_NW16
#define SPD_NW16 8
#endif
#ifndef SPD_NW64
#define SPD_NW64 8   // (4 waves per 64-row tile, six workgroups per CU: +2 % at the headline and at one node per GPU, round 6)
#endif
#define SPD_NW(ROWS) ((ROWS) == 16 ? SPD_NW16 : SPD_NW64)
// loads per half-batch of the streaming loop (a lane has HB..2 HB loads in flight)
#ifndef SPD_HB
#define SPD_HB 8
#endif
#ifndef SPD_PULL16
#define SPD_PULL16 4   // measured best of 2, 4, 8 (one node per GPU: -2 %)
#endif
#ifndef SPD_PULL64
#define SPD_PULL64 2
#endif
#ifndef SPD_PRE64
#define SPD_PRE64 0
#endif
#ifndef SPD_WPE
#define SPD_WPE 6    // waves per SIMD the 64-row solve kernels are compiled for: 85 VGPRs, 3 workgroups per CU (measured best of 5, 6, 8)
#endif
template <int D, int DOF>
__device__ __forceinline__ size_t vaddr(int i) {
  constexpr int RS = Dim<D>::RS;
  if constexpr (DOF == 1) return (size_t)i * RS;
  else return (size_t)(i / DOF) * RS + D + (size_t)(i % DOF) * D;
}

__device__ __forceinline__ SpdItem load_item(const SpdItem *p) {
  const int4 *q = reinterpret_cast<const int4 *>(p);
  union { int4 v[4]; SpdItem it; } u;
  u.v[0] = q[0]; u.v[1] = q[1]; u.v[2] = q[2]; u.v[3] = q[3];
  return u.it;
}

// v += the children's contributions to front position `pos`.  The update buffer is laid out in PULL order: the
// rows a position receives are consecutive (rows asm_ptr[pos] .. asm_ptr[pos + 1]), in the fixed order of the
// host's assembly lists; the children scatter their update rows there (ubuf_dst), which is off the critical path.
template <int D, int PB>
__device__ __forceinline__ void pull_updates(const SpdDev &S, int pos, double (&v)[D]) {
  const int a0 = S.asm_ptr[pos], a1 = S.asm_ptr[pos + 1];
  for (int a = a0; a < a1; a += PB) {
    double t[PB][D];
#pragma unroll
    for (int q = 0; q < PB; q++)
      if (a + q < a1) {
#pragma unroll
        for (int c = 0; c < D; c++) t[q][c] = S.ubuf[(size_t)(a + q) * D + c];
      }
#pragma unroll
    for (int q = 0; q < PB; q++)
      if (a + q < a1) {
#pragma unroll
        for (int c = 0; c < D; c++) v[c] += t[q][c];
      }
  }
}

// The streaming dot products of one chunk: acc[c] += sum_k W[k][lane] f[k][c], k = kq, kq + KQ, ... < kn, the
// panel rows ld doubles apart.  Two half-batches of HB loads alternate: the loads of the next half are issued
// before the products of the current one, so a lane always has HB..2 HB loads in flight (a single batch that is
// waited for as a whole drains to zero between batches).  `pre`: the first half-batch, issued by stream_first()
// before the chunk's input vector was gathered.
template <int D, int KQ, bool NT, int HB, typename PT>
__device__ __forceinline__ bool stream_first(const PT *wp, int ld, int kq, int kn, double (&a)[HB]) {
  const bool full = kq + (HB - 1) * KQ < kn;
  if (full) {
#pragma unroll
    for (int q = 0; q < HB; q++) a[q] = LDW(wp + (size_t)(kq + q * KQ) * ld);
  }
  return full;
}
template <int D, int KQ, bool NT, int HB, typename PT>
__device__ __forceinline__ void stream_rest(const PT *wp, int ld, int kq, int kn, const double *fw, bool have,
                                            double (&a)[HB], double (&acc)[D]) {
  double b[HB];
  int kk = kq;
#ifdef SPD_PROBE_NOFMA   /* measurement only (wrong results): what the loop costs without LDS reads and FMAs */
#define SPD_FMA(buf, k0) _Pragma("unroll") for (int q = 0; q < HB; q++) acc[0] += buf[q];
#else
#define SPD_FMA(buf, k0)                                                                     \
  _Pragma("unroll") for (int q = 0; q < HB; q++)                                            \
      _Pragma("unroll") for (int c = 0; c < D; c++) acc[c] = fma(buf[q], fw[((k0) + q * KQ) * D + c], acc[c]);
#endif
#define SPD_LOAD(buf, k0) \
  _Pragma("unroll") for (int q = 0; q < HB; q++) buf[q] = LDW(wp + (size_t)((k0) + q * KQ) * ld);
  while (have) {
    int k2 = kk + HB * KQ;
    bool more = k2 + (HB - 1) * KQ < kn;
    if (more) { SPD_LOAD(b, k2) }
    SPD_FMA(a, kk)
    kk = k2;
    have = more;
    if (!have) break;
    k2 = kk + HB * KQ;
    more = k2 + (HB - 1) * KQ < kn;
    if (more) { SPD_LOAD(a, k2) }
    SPD_FMA(b, kk)
    kk = k2;
    have = more;
  }
#undef SPD_FMA
#undef SPD_LOAD
  if (kk < kn) {   // the rest: one predicated half-batch (independent loads, never one at a time)
#pragma unroll
    for (int q = 0; q < HB; q++) b[q] = kk + q * KQ < kn ? LDW(wp + (size_t)(kk + q * KQ) * ld) : 0.0;
#pragma unroll
    for (int q = 0; q < HB; q++)
      if (kk + q * KQ < kn) {
#pragma unroll
        for (int c = 0; c < D; c++) acc[c] = fma(b[q], fw[(kk + q * KQ) * D + c], acc[c]);
      }
  }
}

// chunk length for a reduction of `len` entries dealt to NW waves: len / NW rounded up to 8, within [8, CH]
template <int NW, int CH>
__device__ __forceinline__ int spd_chunk(int len) {
  if constexpr (NW == 1) return CH;
  const int c = ((len + NW - 1) / NW + 7) & ~7;
  return min(max(c, 8), CH);
}

// fw: this wave's staging area (SPD_CH * D doubles); red: NW x (ROWS * D) doubles shared by the tile's waves.
// ROOT: the tile belongs to the root of a node's elimination tree.  A root has no update rows and nobody above it, so
// its forward step y = L11^-1 f is followed at once by its backward step x = L11^-T y: the tile streams rows of the
// explicit product L11^-T L11^-1 (= the inverse of the root's Schur complement, S.Wroot) and writes scale * x straight
// into `out` -- one launch instead of two, the same bytes (w^2 entries against two triangles).
template <int D, int DOF, int NW, int SPD_CH, int ROWS, bool NT, bool ROOT = false, typename PT = double>
__device__ __forceinline__ void spd_fwd_tile(const SpdDev &S, const SpdItem &it, const double *vec, double *ytmp,
                                             double *fw, double *red, const int wv, const int lane,
                                             int trace_slot = 0, unsigned long long trace_t0 = 0, double *out = nullptr,
                                             double scale = 1.0) {
  SPD_TRACE_DECL
  SPD_T(1)
  constexpr int KQ = 64 / ROWS, HB = SPD_HB;
  constexpr bool PRE = ROWS < 64 || SPD_PRE64;   // levels with few tiles: latency matters, registers do not
  constexpr int PULLB = ROWS < 64 ? SPD_PULL16 : SPD_PULL64;   // children's contributions fetched per round (top fronts have many)
  const int r = lane % ROWS, kq = lane / ROWS;
  const int p = it.first + r;
  const bool valid = r < it.count;
  const int w = it.w;
  const PT *WT = reinterpret_cast<const PT *>(ROOT ? S.Wroot : S.WT) + it.mat_off + r;   // the tile's panel: [k][r], rows ldm apart
  const int ldm = it.ld;
  const int *piv = S.piv_idx + it.piv_ptr;
  const int pos0 = it.pos_off;
  double acc[D];
#pragma unroll
  for (int c = 0; c < D; c++) acc[c] = 0.0;
  // the lane that will write this row: where its update row goes is looked up now, not at the end
  const bool writer = valid && kq == 0 && wv == 0;
  const int udst = (writer && p >= w) ? S.ubuf_dst[it.ubuf_off + p - w] : 0;
  // the pivot block of W_s is L11^-1, lower triangular: rows of a tile that lies inside it only need
  // the columns up to the tile's last row
  const int kend = (!ROOT && it.first + ROWS <= w) ? it.first + ROWS : w;   // (the root's product matrix is full)
  // the reduction is dealt to the tile's NW waves in equal chunks (at most SPD_CH long): a front with few
  // pivots still keeps every wave busy
  const int cl = spd_chunk<NW, SPD_CH>(kend);
  for (int k0 = wv * cl; k0 < kend; k0 += NW * cl) {
    const int kn = min(cl, kend - k0);
    // first batch of this chunk's matrix entries: in flight while the input vector is gathered
    const PT *wp = WT + (size_t)k0 * ldm;
    double w0[HB];
    bool have0 = false;
    if (PRE && valid) have0 = stream_first<D, KQ, NT, HB, PT>(wp, ldm, kq, kn, w0);
    for (int kk = lane; kk < kn; kk += 64) {
      const int k = k0 + kk;
      double v[D];
#ifdef SPD_PROBE_DENSE   /* measurement only (wrong results): the front's input already in front order */
      const double *src = S.ubuf + (size_t)(it.piv_ptr + k) * D;
#else
      const double *src = vec + vaddr<D, DOF>(piv[k]);
#pragma unroll
      for (int c = 0; c < D; c++) v[c] = *(src + c);
      pull_updates<D, PULLB>(S, pos0 + k, v);
#pragma unroll
      for (int c = 0; c < D; c++) fw[kk * D + c] = v[c];
    }
#endif
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef SPD_TRACE
    if (tr[2] == 0) SPD_T(2)
#endif
    if (valid) {
      if (!PRE) have0 = stream_first<D, KQ, NT, HB, PT>(wp, ldm, kq, kn, w0);
      stream_rest<D, KQ, NT, HB, PT>(wp, ldm, kq, kn, fw, have0, w0, acc);
    }
    __builtin_amdgcn_wave_barrier();
  }
  SPD_T(3)
  if constexpr (KQ > 1) {
#pragma unroll
    for (int c = 0; c < D; c++)
#pragma unroll
      for (int o = ROWS; o < 64; o <<= 1) acc[c] += __shfl_xor(acc[c], o, 64);
  }
  // update rows also receive the children's contributions for that row: fetched while the other
  // waves finish
  double extra[D];
#pragma unroll
  for (int c = 0; c < D; c++) extra[c] = 0.0;
  if (writer && p >= w) pull_updates<D, PULLB>(S, pos0 + p, extra);
  if constexpr (NW > 1) {
    if (kq == 0) {
#pragma unroll
      for (int c = 0; c < D; c++) red[wv * ROWS * D + r * D + c] = acc[c];
    }
    __syncthreads();
    if (wv != 0) return;
    if (kq == 0) {
#pragma unroll
      for (int c = 0; c < D; c++) {
        double a = red[r * D + c];
        for (int q = 1; q < NW; q++) a += red[q * ROWS * D + r * D + c];
        acc[c] = a;
      }
    }
  }
  SPD_T(4)
  if (ROOT) {
    if (writer) {
      double *dst = out + vaddr<D, DOF>(piv[p]);
#pragma unroll
      for (int c = 0; c < D; c++) dst[c] = scale * acc[c];
    }
  } else if (writer) {
    if (p < w) {
      double *dst = ytmp + (size_t)(it.piv_ptr + p) * D;   // y in front order: the backward sweep reads it back contiguously
#pragma unroll
      for (int c = 0; c < D; c++) dst[c] = acc[c];
    } else {
      double *dst = S.ubuf + (size_t)udst * D;
#pragma unroll
      for (int c = 0; c < D; c++) dst[c] = acc[c] + extra[c];
    }
  }
  SPD_T(5)
  SPD_TRACE_FLUSH(wv == 0 && lane == 0)
}

template <int D, int DOF, int NW, int SPD_CH, int ROWS, bool NT, typename PT = double>
__device__ __forceinline__ void spd_bwd_tile(const SpdDev &S, const SpdItem &it, double scale, const double *ytmp,
                                             double *vec, double *fw, double *red, const int wv, const int lane,
                                             int trace_slot = 0, unsigned long long trace_t0 = 0) {
  SPD_TRACE_DECL
  SPD_T(1)
  constexpr int KQ = 64 / ROWS, HB = SPD_HB;
  constexpr bool PRE = ROWS < 64 || SPD_PRE64;   // levels with few tiles: latency matters, registers do not
  const int r = lane % ROWS, kq = lane / ROWS;
  const int k = it.first + r;
  const bool valid = r < it.count;
  const int w = it.w, m = w + it.u;
  const PT *W = reinterpret_cast<const PT *>(S.W) + it.mat_off + r;     // the tile's panel: [p - first][r], rows ldw apart
  const int ldw = it.ld;
  const int *piv = S.piv_idx + it.piv_ptr;
  const int *upd = S.upd_idx + it.upd_ptr;
  double acc[D];
#pragma unroll
  for (int c = 0; c < D; c++) acc[c] = 0.0;
  // columns of a tile starting at c0 are zero in the rows above c0 (L11^-1 is lower triangular)
  const int cl = spd_chunk<NW, SPD_CH>(m - it.first);
  for (int p0 = it.first + wv * cl; p0 < m; p0 += NW * cl) {
    const int pn = min(cl, m - p0);
    const PT *wp = W + (size_t)(p0 - it.first) * ldw;
    double w0[HB];
    bool have0 = false;
    if (PRE && valid) have0 = stream_first<D, KQ, NT, HB, PT>(wp, ldw, kq, pn, w0);
    for (int pp = lane; pp < pn; pp += 64) {
      const int p = p0 + pp;
      // ancestors' entries were scaled when they were written: undo by linearity (scale is +-1)
      const double sc = p < w ? 1.0 : scale;
#ifdef SPD_PROBE_DENSE   /* measurement only (wrong results) */
      const double *src = p < w ? ytmp + (size_t)(it.piv_ptr + p) * D : S.ubuf + (size_t)(it.ubuf_off + p - w) * D;
#else
      const double *src = p < w ? ytmp + (size_t)(it.piv_ptr + p) * D : vec + vaddr<D, DOF>(upd[p - w]);
#endif
#pragma unroll
      for (int c = 0; c < D; c++) fw[pp * D + c] = sc * src[c];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef SPD_TRACE
    if (tr[2] == 0) SPD_T(2)
#endif
    if (valid) {
      if (!PRE) have0 = stream_first<D, KQ, NT, HB, PT>(wp, ldw, kq, pn, w0);
      stream_rest<D, KQ, NT, HB, PT>(wp, ldw, kq, pn, fw, have0, w0, acc);
    }
    __builtin_amdgcn_wave_barrier();
  }
  SPD_T(3)
  if constexpr (KQ > 1) {
#pragma unroll
    for (int c = 0; c < D; c++)
#pragma unroll
      for (int o = ROWS; o < 64; o <<= 1) acc[c] += __shfl_xor(acc[c], o, 64);
  }
  if constexpr (NW > 1) {
    if (kq == 0) {
#pragma unroll
      for (int c = 0; c < D; c++) red[wv * ROWS * D + r * D + c] = acc[c];
    }
    __syncthreads();
    if (wv != 0) return;
    if (kq == 0) {
#pragma unroll
      for (int c = 0; c < D; c++) {
        double a = red[r * D + c];
        for (int q = 1; q < NW; q++) a += red[q * ROWS * D + r * D + c];
        acc[c] = a;
      }
    }
  }
  SPD_T(4)
  if (valid && kq == 0) {
    double *dst = vec + vaddr<D, DOF>(piv[k]);
#pragma unroll
    for (int c = 0; c < D; c++) dst[c] = scale * acc[c];
  }
  SPD_T(5)
  SPD_TRACE_FLUSH(wv == 0 && lane == 0)
}

// One level of a sweep in one launch.  A workgroup (8 waves) takes a PACK: either one tile of a wide front,
// shared by the 8 waves (ROWS-row tiles), or up to 8 tiles of narrow fronts (reduction length <= 96), one
// per wave.  Wide packs come first, longest first; the narrow ones fill the tail of the launch.
template <int D, int DOF, int ROWS, int MODE, bool NT, typename PT = double>   // MODE 0: forward level, 1: backward level, 2: the roots (fused); PT: panel storage
__global__ __launch_bounds__(64 * SPD_NW(ROWS), ROWS == 64 ? SPD_WPE : 4) void k_spd_level(SpdDev S, NodeMask mask, SpdLevelMap M,
                                                                                      double scale, double *vec, double *ytmp) {
  constexpr int CH = 128, NW = SPD_NW(ROWS);
  __shared__ double f[NW][CH * D];
  __shared__ double red[NW * ROWS * D];
  // workgroups [0, wide_wgs) take one wide tile each, the rest NW narrow tiles each; slot = b % nlive picks the node,
  // b / nlive the tile within the node (no work list to read: a tile's index follows from blockIdx and the arguments)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const SpdItem *items = MODE == 0 ? S.fwd_items : (MODE == 1 ? S.bwd_items : S.root_items);
  int b = blockIdx.x;
#ifdef SPD_TRACE
  const unsigned long long tt0 = wall_clock64();
#define SPD_TRACE_ARGS(slot) , (slot) - M.pad, tt0
#else
#define SPD_TRACE_ARGS(slot) , 0, 0ull
#endif
  const bool is_wide = b < M.wide_wgs;
  if (!is_wide) b -= M.wide_wgs;
  const int j = b % M.nlive, r = b / M.nlive;
  if (is_wide ? r >= M.wcount[j] : r * NW + wv >= M.ncount[j]) return;
  // a node that left the device-side mask after the host sized this launch: nothing of it is read
  if (mask.p && !((*mask.p >> M.node[j]) & 1ull)) return;
  // (MODE 2: `vec` is the right-hand side, `ytmp` the record array that receives the solution)
  if (!is_wide) {
    const int t = M.nstart[j] + r * NW + wv;
    const SpdItem it = load_item(items + t);
    if constexpr (MODE == 0) spd_fwd_tile<D, DOF, 1, CH, 64, NT, false, PT>(S, it, vec, ytmp, f[wv], red, 0, lane SPD_TRACE_ARGS(t));
    else if constexpr (MODE == 1) spd_bwd_tile<D, DOF, 1, CH, 64, NT, PT>(S, it, scale, ytmp, vec, f[wv], red, 0, lane SPD_TRACE_ARGS(t));
    else spd_fwd_tile<D, DOF, 1, CH, 64, NT, true, PT>(S, it, vec, nullptr, f[wv], red, 0, lane SPD_TRACE_ARGS(t), ytmp, scale);
  } else {
    const int t = M.wstart[j] + r;
    const SpdItem it = load_item(items + t);
    if constexpr (MODE == 0) spd_fwd_tile<D, DOF, NW, CH, ROWS, NT, false, PT>(S, it, vec, ytmp, f[wv], red, wv, lane SPD_TRACE_ARGS(t));
    else if constexpr (MODE == 1) spd_bwd_tile<D, DOF, NW, CH, ROWS, NT, PT>(S, it, scale, ytmp, vec, f[wv], red, wv, lane SPD_TRACE_ARGS(t));
    else spd_fwd_tile<D, DOF, NW, CH, ROWS, NT, true, PT>(S, it, vec, nullptr, f[wv], red, wv, lane SPD_TRACE_ARGS(t), ytmp, scale);
  }
#undef SPD_TRACE_ARGS
}

// ---- the roots as one triangle (kernels.h: RootRow) ---------------------------------------------------------------
// One workgroup of four waves per item: blocks (I, J0 .. J0 + nJ - 1) of the lower triangle of P = L11^-T L11^-1, 32 KB
// each.  A block is eight 8-column strips; wave v takes strips v and v + 4 (16 loads per lane, all in flight at once: a
// block is ONE memory round trip for the workgroup).  Lane = row r of block row I:
//   direct      acc[c]  += P[I r][J k] f_J[k][c]                       (f_J broadcast from LDS; the four waves' sums meet
//                                                                       through LDS at the end, in a fixed order)
//   transposed  t_k[c]  += P[I r][J k] f_I[r][c]  summed over r: the strip goes through LDS, lane (k, q) takes rows
//               8q .. 8q + 7 of column k; the two strips of a wave meet across lane bit 5 (each half keeps one strip),
//               two more exchanges complete the sums over q: the wave ends with the 16 column sums of its strips.
// The input entries f = rhs + the children's contributions are gathered once per item, one 64-entry chunk per wave (chunk
// 0: block row I, chunks 1 .. nJ: the block columns): index loads, then vector loads, then the pulls round by round.
template <int D>
__device__ __forceinline__ void root_xchg(double (&a)[D], const double (&b)[D], int lane, int bit) {
  // lanes with `bit` clear keep a (+ the partner's a), the others keep b (+ the partner's b)
  const bool hi = (lane & bit) != 0;
#pragma unroll
  for (int c = 0; c < D; c++) {
    const double keep = hi ? b[c] : a[c], send = hi ? a[c] : b[c];
    a[c] = keep + __shfl_xor(send, bit, 64);
  }
}

template <int D, int DOF, bool NT>
__global__ __launch_bounds__(256, 3) void k_root_sym(SpdDev S, NodeMask mask, SpdLevelMap M, const double *vec, double *part) {
  constexpr int NWV = 4, TS = 66, MAXJ = ROOT_SYM_MAXJ, NCH = (MAXJ + 1 + NWV - 1) / NWV;   // NCH: gather chunks per wave
  typedef double PT;
  __shared__ double f_s[MAXJ + 1][64 * D];                            // chunk 0: f_I, chunks 1..: f_J
  // a wave's two strips on their way to lane = column; at the end the same memory carries the waves' direct sums
  __shared__ __attribute__((aligned(16))) double T_s[NWV][2][8 * TS];
  static_assert(sizeof(double) * NWV * 2 * 8 * TS >= sizeof(double) * NWV * 64 * D, "the direct sums fit the strip buffers");
  double (*red)[64 * D] = reinterpret_cast<double (*)[64 * D]>(&T_s[0][0][0]);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = blockIdx.x, j = b % M.nlive, rr = b / M.nlive;
  if (rr >= M.ncount[j]) return;
  if (mask.p && !((*mask.p >> M.node[j]) & 1ull)) return;
  SpdItem it = load_item(S.root_items + M.nstart[j] + rr);
  // (the descriptor is the same in every lane: in scalar registers its tests become scalar branches)
  it.first = __builtin_amdgcn_readfirstlane(it.first); it.w = __builtin_amdgcn_readfirstlane(it.w);
  it.u = __builtin_amdgcn_readfirstlane(it.u); it.ld = __builtin_amdgcn_readfirstlane(it.ld);
  it.piv_ptr = __builtin_amdgcn_readfirstlane(it.piv_ptr); it.pos_off = __builtin_amdgcn_readfirstlane(it.pos_off);
  it.upd_ptr = __builtin_amdgcn_readfirstlane(it.upd_ptr); it.ubuf_off = __builtin_amdgcn_readfirstlane(it.ubuf_off);
  {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(it.mat_off & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((unsigned)(it.mat_off >> 32));
    it.mat_off = (int64_t)(((unsigned long long)hi << 32) | lo);
  }
  const int w = it.w, I = it.first >> 6, J0 = it.u, nJ = it.ld;
  // this wave's strips of block jb: rows (wv * 8 + i) and ((wv + 4) * 8 + i) of the k-major block
  const PT *blk = reinterpret_cast<const PT *>(S.Wroot) + it.mat_off + (size_t)wv * 512 + lane;
  // the first block's loads go out before anything else: they depend on nothing but the descriptor
  double ma[8], mb[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { ma[i] = LDW(blk + i * 64); mb[i] = LDW(blk + 2048 + i * 64); }
  // ---- gather: wave v takes chunks v, v + 4, ...; every stage of the chain for all of them at once
  {
    int idx[NCH], a0[NCH], a1[NCH];
    double f[NCH][D];
#pragma unroll
    for (int q = 0; q < NCH; q++) {
      const int ch = wv + NWV * q;
      const int k = (ch == 0 ? I : J0 + ch - 1) * 64 + lane;
      const bool v = ch <= nJ && k < w;
      idx[q] = v ? S.piv_idx[it.piv_ptr + k] : -1;
      a0[q] = v ? S.asm_ptr[it.pos_off + k] : 0;
      a1[q] = v ? S.asm_ptr[it.pos_off + k + 1] : 0;
    }
#pragma unroll
    for (int q = 0; q < NCH; q++)
#pragma unroll
      for (int c = 0; c < D; c++) f[q][c] = idx[q] >= 0 ? vec[vaddr<D, DOF>(idx[q]) + c] : 0.0;
    // the children's contributions, in list order (the order of the host solve), two list positions per round
    for (int r = 0;; r += 2) {
      bool any = false;
      double u[NCH][2][D];
#pragma unroll
      for (int q = 0; q < NCH; q++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const bool on = a0[q] + r + e < a1[q];
          any |= on;
#pragma unroll
          for (int c = 0; c < D; c++) u[q][e][c] = on ? S.ubuf[(size_t)(a0[q] + r + e) * D + c] : 0.0;
        }
      if (!__any(any)) break;
#pragma unroll
      for (int q = 0; q < NCH; q++)
#pragma unroll
        for (int e = 0; e < 2; e++)
          if (a0[q] + r + e < a1[q]) {
#pragma unroll
            for (int c = 0; c < D; c++) f[q][c] += u[q][e][c];
          }
    }
#pragma unroll
    for (int q = 0; q < NCH; q++)
      if (wv + NWV * q <= nJ) {
#pragma unroll
        for (int c = 0; c < D; c++) f_s[wv + NWV * q][lane * D + c] = f[q][c];
      }
  }
  __syncthreads();
  const int k8 = lane & 7, q8 = lane >> 3;
  double fIt[8][D];   // rows 8 q8 .. 8 q8 + 7 of f_I: what this lane multiplies its strip column with
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int c = 0; c < D; c++) fIt[i][c] = f_s[0][(8 * q8 + i) * D + c];
  double acc[D];
#pragma unroll
  for (int c = 0; c < D; c++) acc[c] = 0.0;
  // the diagonal block, if the item has it, is its last one; the blocks before it take both products
  const bool has_diag = J0 + nJ - 1 == I;
  for (int jb = 0; jb < nJ; jb++) {
    const double *fj = f_s[jb + 1];
    const bool offdiag = !(has_diag && jb == nJ - 1);
    double na[8], nb2[8];
    if (jb + 1 < nJ) {   // the next block's strips
#pragma unroll
      for (int i = 0; i < 8; i++) { na[i] = LDW(blk + (size_t)(jb + 1) * 4096 + i * 64); nb2[i] = LDW(blk + (size_t)(jb + 1) * 4096 + 2048 + i * 64); }
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int c = 0; c < D; c++) {
        acc[c] = fma(ma[i], fj[(wv * 8 + i) * D + c], acc[c]);
        acc[c] = fma(mb[i], fj[((wv + 4) * 8 + i) * D + c], acc[c]);
      }
    // (pins the products here: left alone, the compiler defers them behind the transposed part and spills their LDS operands)
#pragma unroll
    for (int c = 0; c < D; c++) asm volatile("" : "+v"(acc[c]) :: "memory");
    if (offdiag) {
#pragma unroll
      for (int i = 0; i < 8; i++) { T_s[wv][0][i * TS + lane] = ma[i]; T_s[wv][1][i * TS + lane] = mb[i]; }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double ta[D], tb[D];
#pragma unroll
      for (int c = 0; c < D; c++) ta[c] = tb[c] = 0.0;
      const double2 *pa = reinterpret_cast<const double2 *>(&T_s[wv][0][k8 * TS + 8 * q8]);
      const double2 *pb = reinterpret_cast<const double2 *>(&T_s[wv][1][k8 * TS + 8 * q8]);
#pragma unroll
      for (int h = 0; h < 4; h++) {
        const double2 va = pa[h], vb = pb[h];
#pragma unroll
        for (int c = 0; c < D; c++) {
          ta[c] = fma(va.x, fIt[2 * h][c], ta[c]);
          ta[c] = fma(va.y, fIt[2 * h + 1][c], ta[c]);
          tb[c] = fma(vb.x, fIt[2 * h][c], tb[c]);
          tb[c] = fma(vb.y, fIt[2 * h + 1][c], tb[c]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // lanes 0..31 keep strip wv, lanes 32..63 strip wv + 4; then the sums over the remaining two bits of q
      root_xchg<D>(ta, tb, lane, 32);
#pragma unroll
      for (int c = 0; c < D; c++) ta[c] += __shfl_xor(ta[c], 16, 64);
#pragma unroll
      for (int c = 0; c < D; c++) ta[c] += __shfl_xor(ta[c], 8, 64);
      if ((q8 & 3) == 0) {
        const int col = (wv + (lane >= 32 ? 4 : 0)) * 8 + k8;
        double *dst = part + ((size_t)(it.ubuf_off + jb) * 64 + col) * D;
#pragma unroll
        for (int c = 0; c < D; c++) dst[c] = ta[c];
      }
    }
    if (jb + 1 < nJ) {
#pragma unroll
      for (int i = 0; i < 8; i++) { ma[i] = na[i]; mb[i] = nb2[i]; }
    }
  }
  __syncthreads();   // (every wave is done with its strip buffers)
#pragma unroll
  for (int c = 0; c < D; c++) red[wv][lane * D + c] = acc[c];
  __syncthreads();
  if (wv != 0) return;
  double *dst = part + ((size_t)it.upd_ptr * 64 + lane) * D;
#pragma unroll
  for (int c = 0; c < D; c++) dst[c] = ((red[0][lane * D + c] + red[1][lane * D + c]) + red[2][lane * D + c]) + red[3][lane * D + c];
}

// x on block row R of a root = scale * (its direct slots, then the transposed slots of blocks (R + 1 .. nb - 1, R)): four
// waves take every fourth slot each and meet through LDS, always in the same order.
template <int D, int DOF>
__global__ __launch_bounds__(256) void k_root_combine(SpdDev S, NodeMask mask, SpdLevelMap M, const RootRow *rows, const double *part,
                                                      double scale, double *out) {
  __shared__ double red[4][64 * D];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x, j = b % M.nlive, rr = b / M.nlive;
  if (rr >= M.wcount[j]) return;
  if (mask.p && !((*mask.p >> M.node[j]) & 1ull)) return;
  union { int4 v[3]; RootRow r; } u;
  const int4 *rp = reinterpret_cast<const int4 *>(rows + M.wstart[j] + rr);
  u.v[0] = rp[0]; u.v[1] = rp[1]; u.v[2] = rp[2];
  const RootRow row = u.r;
  const int nslots = row.ndslots + row.nb - 1 - row.R;
  // (the pivot's record is looked up while the slots arrive)
  const int pidx = (wv == 0 && lane < row.count) ? S.piv_idx[row.piv_ptr + row.first + lane] : -1;
  double acc[D];
#pragma unroll
  for (int c = 0; c < D; c++) acc[c] = 0.0;
  auto slot_of = [&](int s) {
    if (s < row.ndslots) return (long long)(row.dslot + s);
    const long long I = row.R + 1 + (s - row.ndslots);
    return (long long)row.tbase + I * (I - 1) / 2 + row.R;
  };
  for (int s0 = wv; s0 < nslots; s0 += 4 * 4) {
    double t[4][D];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int s = s0 + 4 * q;
#pragma unroll
      for (int c = 0; c < D; c++) t[q][c] = s < nslots ? part[((size_t)slot_of(s) * 64 + lane) * D + c] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int c = 0; c < D; c++) acc[c] += t[q][c];
  }
#pragma unroll
  for (int c = 0; c < D; c++) red[wv][lane * D + c] = acc[c];
  __syncthreads();
  if (wv != 0 || pidx < 0) return;
  double *dst = out + vaddr<D, DOF>(pidx);
#pragma unroll
  for (int c = 0; c < D; c++) dst[c] = scale * (((red[0][lane * D + c] + red[1][lane * D + c]) + red[2][lane * D + c]) + red[3][lane * D + c]);
}

}  // namespace


// ---------------------------------------------------------------------------
// Optional per-launch timing with HIP events on the launch stream (bench.py's
// roofline pass).  Disabled by default: no events are recorded in timed runs.
// ---------------------------------------------------------------------------
namespace {
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;      // pairs
  std::vector<int> kind;
  size_t used = 0;
  double ms[PK_COUNT] = {0};
  double bytes[PK_COUNT] = {0};
  double obytes[PK_COUNT] = {0};   // operand-by-operand bytes of the fused passes
  long count[PK_COUNT] = {0};
  int nested = 0;
  void flush() {
    if (!used) return;
    (void)hipEventSynchronize(ev[2 * used - 1]);
    for (size_t i = 0; i < used; i++) {
      float t = 0;
      (void)hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]);
      ms[kind[i]] += t;
    }
    used = 0;
  }
};
Prof g_prof;
struct ProfScope {
  hipStream_t st;
  bool on;
  // n launches share the scope (a sweep of the solve: one event pair around its back-to-back launches, so that the
  // events do not sit between the kernels they time); scopes opened inside such a scope do nothing
  ProfScope(int kind, hipStream_t s, double bytes, int n = 1, double operand_bytes = 0.0) : st(s), on(g_prof.on && g_prof.nested == 0) {
    if (g_prof.on && n > 1) { g_prof.nested++; outer = true; }
    if (!on) return;
    if (g_prof.used == g_prof.kind.size()) {
      if (g_prof.kind.size() >= 32768) g_prof.flush();
      else {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        g_prof.ev.push_back(a);
        g_prof.ev.push_back(b);
        g_prof.kind.push_back(0);
      }
    }
    g_prof.kind[g_prof.used] = kind;
    g_prof.count[kind] += n;
    g_prof.bytes[kind] += bytes;
    g_prof.obytes[kind] += operand_bytes;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used], st);
  }
  bool outer = false;
  ~ProfScope() {
    if (outer) g_prof.nested--;
    if (!on) return;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], st);
    g_prof.used++;
  }
};
}  // namespace

struct ProfSweep::Impl { ProfScope ps; Impl(int k, hipStream_t st, double b, int n) : ps(k, st, b, n) {} };
ProfSweep::ProfSweep(bool forward, hipStream_t st, double bytes, int launches)
    : p(g_prof.on && launches > 1 ? new Impl(forward ? PK_SPD_FWD : PK_SPD_BWD, st, bytes, launches) : nullptr) {}
ProfSweep::~ProfSweep() { delete p; }

void prof_enable(bool on) {
  g_prof.flush();
  g_prof.on = on;
}
bool prof_enabled() { return g_prof.on; }
void prof_reset() {
  g_prof.flush();
  for (int k = 0; k < PK_COUNT; k++) { g_prof.ms[k] = 0; g_prof.bytes[k] = 0; g_prof.count[k] = 0; }
}
void prof_collect(double *ms, double *bytes, long *count) {
  g_prof.flush();
  for (int k = 0; k < PK_COUNT; k++) { ms[k] = g_prof.ms[k]; bytes[k] = g_prof.bytes[k]; count[k] = g_prof.count[k]; }
}
void prof_collect_operands(double *operand_bytes) {
  for (int k = 0; k < PK_COUNT; k++) operand_bytes[k] = g_prof.obytes[k];
}

// ---------------------------------------------------------------------------
// Launch wrappers
// ---------------------------------------------------------------------------
#define DPGO_DISPATCH_D(d, ...)            \
  do {                                     \
    if ((d) == 3) { constexpr int D = 3; __VA_ARGS__; } \
    else { constexpr int D = 2; __VA_ARGS__; }          \
  } while (0)

static inline int nseg(const SegTable &T, bool all_rows) { return all_rows ? T.nseg_all : T.nseg_own; }
// the grid of a launch over own segments: all of them, or -- NodeMask::nlive -- the live nodes' only
static inline int own_grid(const SegTable &T, const NodeMask &m) {
  if (m.nlive == 0) return T.nseg_own;
  int mx = 0;
  for (int j = 0; j < m.nlive; j++) mx = std::max(mx, m.nseg[j]);
  return m.nlive * mx;
}
static inline NodeMask whole_grid(NodeMask m) { m.nlive = 0; return m; }   // (launches that also cover the neighbour segments)

void launch_bsr(int d, hipStream_t st, const SegTable &T, bool all_rows, NodeMask mask, const BsrDev &A,
                const double *x, int mode, const double *addv, double *y, const double *dotv,
                double coef, const double *dotadd, double *partials, int slot, double *copy1, double *copy2) {
  if (all_rows) mask = whole_grid(mask);
  const int nb = all_rows ? T.nseg_all : own_grid(T, mask);
  if (nb == 0) return;
  double *part = (dotv && partials) ? partials + (size_t)slot * T.nseg_all : nullptr;
  ProfScope ps(PK_BSR, st, (double)A.nnzb * (8.0 * (d + 1) * (d + 1) + 4) + 2.0 * A.nrows * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, {
    if (mode == 1)
      hipLaunchKernelGGL((k_bsr<D, 1>), dim3(nb), dim3(BSR_LPR * SEG_ROWS), 0, st, T.segs, mask, A, x, addv, y, dotv, coef, dotadd,
                         part, copy1, copy2);
    else if (mode == 2)
      hipLaunchKernelGGL((k_bsr<D, 2>), dim3(nb), dim3(BSR_LPR * SEG_ROWS), 0, st, T.segs, mask, A, x, addv, y, dotv, coef, dotadd,
                         part, copy1, copy2);
    else
      hipLaunchKernelGGL((k_bsr<D, 0>), dim3(nb), dim3(BSR_LPR * SEG_ROWS), 0, st, T.segs, mask, A, x, addv, y, dotv, coef, dotadd,
                         part, copy1, copy2);
  });
}

void launch_bsr_tcol(int d, hipStream_t st, const SegTable &T, NodeMask mask, const BsrDev &A, const double *tval,
                     const double *xt, const double *base, double *y, int mode, const double *X, const double *nabla,
                     const double *Rdot, double *out2, const double *rres, double *partials, const double *dg,
                     const double *dga, const double *ds, const double *dgrad, const double *dhs) {
  if (T.nseg_own == 0) return;
  // which epilogue sums are produced: mode 2 with rres (a CG step's four), mode 1 with dg (a refinement's start), mode 0 with ds (a trial point's six)
  const bool sums = partials && ((mode == 2 && rres) || (mode == 1 && dg) || (mode == 0 && ds));
  // SURVEY 8(d)'s formula prices the bare pass (the blocks' first columns, the gathered translations, two vectors).  What
  // the fused pass moves, operand by operand: per block its first column and index; per row the translation its
  // neighbours gather (once: the rest are cache hits), `base`, y where it is stored, and the epilogue's vectors -- mode 1
  // (a refinement's start): X, the tangent gradient written, g and g_alt for the sums; mode 2 (a Hessian product): X, the
  // model gradient, the direction's and the residual's rotation rows read, the product written; mode 0 with sums (a trial
  // point): the step's, the gradient's and H s's rotation rows, g, g_alt and the point's own record.
  const double P = 8.0 * (d + 1) * d, Pr = 8.0 * d * d, Pt = 8.0 * d;
  double per_row = Pt + P + (mode != 2 ? P : 0.0);
  if (mode == 1) per_row += 2.0 * P + (sums ? (dg ? P : 0.0) + ((dga && dga != dg) ? P : 0.0) : 0.0);
  if (mode == 2) per_row += 3.0 * P + Pr + (rres ? Pr : 0.0);
  if (mode == 0 && sums) per_row += 3.0 * Pr + P + (dg ? P : 0.0) + ((dga && dga != dg) ? P : 0.0);
  ProfScope ps(PK_BSR_TCOL, st, (double)A.nnzb * (8.0 * (d + 1) + 4 + 8.0 * d) + 2.0 * A.nrows * 8.0 * (d + 1) * d, 1,
               (double)A.nnzb * (8.0 * (d + 1) + 4) + (double)A.nrows * per_row);
  TcolDots E;
  E.g = dg; E.ga = dga; E.s = ds; E.grad = dgrad; E.hs = dhs;
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_bsr_tcol<D>), dim3(own_grid(T, mask)), dim3(4 * SEG_ROWS), 0, st, T.segs, mask, A, tval,
                                        xt, base, y, mode, X, nabla, Rdot, out2, rres, sums ? partials : nullptr, T.nseg_all, E));
}

void launch_inter(int d, hipStream_t st, const SegTable &T, NodeMask mask, const InterEdgesDev &E, int loss,
                  double loss_reg, int mode, bool quad, const double *Z, const double *Zprev,
                  const double *Qdiag, const double *Ddiag, double *DfE, double *g, double *partials, double *wout,
                  const double *GXc, const double *GXp, const NodeCoefs *gamma, double *Df_out, const double *Znbr,
                  const double *gamma_dev, const InterFuse *fuse) {
  const int nb = mode == 0 ? T.nseg_all : T.nseg_own;
  if (nb == 0) return;
  InterFuse F;
  if (fuse) {
    if (mode == 0) { F.GX = fuse->GX; F.X = fuse->X; F.Df = fuse->Df; F.gn_slot = fuse->gn_slot; }
    else {
      F.Zc = fuse->Zc; F.Zp = fuse->Zp; F.Yout = fuse->Yout;
      if (fuse->Xout && Df_out == nullptr && GXc && GXp && gamma) {   // (the proximal step needs the Df this pass forms)
        F.Xout = fuse->Xout; F.Xref = fuse->Xref; F.Tinv = fuse->Tinv; F.Nv = fuse->Nv; F.Vb = fuse->Vb; F.gn_slot = fuse->gn_slot;
      }
    }
  }
  // operand by operand: per incidence its 128-byte record and the other endpoint's pose; per own pose its record, the previous
  // iterate (majorisation gap), the previous DfobjE read and the new one written, the Q and D blocks, g written, the
  // incidence pointer, and (iterate()) the two kept products G X read and Df written; per neighbour row the same without
  // D and g but with the halo copy it performs on the way
  const double P = 8.0 * (d + 1) * d, B = 8.0 * (d + 1) * (d + 1);
  const double own_b = P + (Zprev ? P : 0) + (DfE ? 2 * P : 0) + (Qdiag ? B : 0) + (Ddiag ? B : 0) + (g ? P : 0) + 8 +
                       ((mode == 1 && Df_out && GXc && GXp) ? 3 * P : 0) +
                       ((fuse && mode == 0 && fuse->Df) ? 3 * P : 0) +     // the product G X and the own record read, Dfobj written
                       ((fuse && mode == 1 && fuse->Zc) ? 2 * P : 0);      // X[k-1] read, the extrapolated record written
  const double prox_b = (fuse && mode == 1 && fuse->Xout) ? (2 * P + 8.0 * (1 + d + d * d) + 8.0 * d * d - P) : 0.0;   // Xakh written, Xak read and its rotations written, T / N / V; Df no longer written
  const double nbr_b = P + (Zprev ? P : 0) + (DfE ? 2 * P : 0) + (Qdiag ? B : 0) + 8 + (Znbr ? 2 * P : 0);
  const double operands = (double)E.m * (mode == 0 ? 2 : 1) * (128.0 + ((fuse && mode == 1 && fuse->Zc) ? 2 * P : P)) + (double)E.nrows_own * (own_b + prox_b) +
                          (mode == 0 ? (double)(E.nrows_all - E.nrows_own) * nbr_b : 0.0);
  ProfScope ps(PK_INTER, st, (double)E.m * (8.0 * (d * d + d + 2) + 8) + 2.0 * (mode == 0 ? E.nrows_all : E.nrows_own) * 8.0 * (d + 1) * d, 1, operands);
  InterLin lin;
  if (mode == 1 && (Df_out || F.Xout) && GXc && GXp && gamma) { lin.GXc = GXc; lin.GXp = GXp; lin.out = Df_out; lin.gamma = *gamma; lin.gamma_dev = gamma_dev; }
  else if (mode == 1 && F.Zc && gamma) { lin.gamma = *gamma; lin.gamma_dev = gamma_dev; }   // (the extrapolation's gamma alone)
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_inter<D>), dim3(nb), dim3(SEG_ROWS), 0, st, T.segs, mask, E, loss, loss_reg,
                                        mode, quad ? 1 : 0, T.nseg_own, Z, Zprev, Qdiag, Ddiag, DfE, g, partials,
                                        T.nseg_all, mode == 0 ? wout : nullptr, lin, mode == 0 ? Znbr : nullptr, F));
}

void launch_rescale_decide(hipStream_t st, int nnodes, NodeBits nodes, const int *e_off, const double *w, double *scale,
                           int *count, int max_count, int *flags, double *host_flags) {
  if (nnodes > 0) hipLaunchKernelGGL(k_rescale_decide, dim3(nnodes), dim3(256), 0, st, nodes, e_off, w, scale, count, max_count, flags, host_flags);
}
void launch_rescale_apply(int d, hipStream_t st, const SegTable &T, const InterEdgesDev &E, const RescaleArgs &A) {
  if (T.nseg_all == 0) return;
  RescaleDev R;
  R.flags = A.flags; R.scale = A.scale; R.Gbase = A.Gbase; R.Hbase = A.Hbase; R.gpos = reinterpret_cast<const int4 *>(A.gpos);
  R.att_pos = A.att_pos; R.Gval = A.Gval; R.Gtcol = A.Gtcol; R.Dd = A.Dd; R.Qd = A.Qd; R.Tinv = A.Tinv; R.N = A.N; R.V = A.V;
  R.att_val = A.att_val; R.xi = A.xi;
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_rescale_apply<D>), dim3(T.nseg_all), dim3(SEG_ROWS), 0, st, T.segs, E, R, T.nseg_own));
}

void launch_cost(int d, hipStream_t st, const SegTable &T, NodeMask mask, const InterEdgesDev &Ei,
                 const InterEdgesDev &Ee, bool eform, int loss, double loss_reg, const double *Z, double *partials,
                 int slot0) {
  if (T.nseg_all == 0) return;
  ProfScope ps(PK_INTER, st, (double)(Ei.m + Ee.m) * (8.0 * (d * d + d + 2) + 8) + 1.0 * T.rows_all * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_cost<D>), dim3(T.nseg_all), dim3(SEG_ROWS), 0, st, T.segs, mask, Ei, Ee,
                                        eform ? 1 : 0, loss, loss_reg, Z, partials + (size_t)slot0 * T.nseg_all,
                                        T.nseg_all));
}

void launch_sqdist(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *a, const double *b,
                   double *partials, int slot) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_DOT, st, 2.0 * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_sqdist<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, a, b,
                                        partials + (size_t)slot * T.nseg_all));
}

void launch_proximal(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *Z, const double *Df,
                     const double *Tinv, const double *N, const double *V, double *Xout, double *Xref,
                     double *partials, int slot) {
  if (T.nseg_own == 0) return;
  double *part = (Xref && partials) ? partials + (size_t)slot * T.nseg_all : nullptr;
  // operand by operand: Z, Df and Xout records, the coefficients T (1), N (d), V (d x d) and, with Xref, its record read and
  // its rotation rows written
  ProfScope ps(PK_PROX, st, 3.0 * T.rows_own * 8.0 * (d + 1) * d, 1,
               (double)T.rows_own * (3.0 * 8.0 * (d + 1) * d + 8.0 * (1 + d + d * d) + (part ? 8.0 * (d + 1) * d + 8.0 * d * d : 0.0)));
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_proximal<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, Z, Df,
                                        Tinv, N, V, Xout, part ? Xref : nullptr, part));
}

void launch_extrapolate(int d, hipStream_t st, const SegTable &T, bool all_rows, NodeMask mask,
                        const NodeCoefs &gamma, const double *a, const double *b, double *out, const double *gamma_dev) {
  const int nb = nseg(T, all_rows);
  if (nb == 0) return;
  ProfScope ps(PK_AXPBY, st, 3.0 * (all_rows ? T.rows_all : T.rows_own) * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_extrapolate<D>), dim3(nb), dim3(SEG_ROWS), 0, st, T.segs, mask, gamma, gamma_dev,
                                        a, b, out));
}

void launch_extrapolate3(int d, hipStream_t st, const SegTable &T, NodeMask mask, const NodeCoefs &gamma, const double *gamma_dev,
                         const double *za, const double *zb, double *zout, const double *ga, const double *gb, double *gout,
                         const double *da, const double *db, double *dout) {
  if (T.nseg_all == 0) return;
  Extrap3 E;
  E.a[0] = za; E.b[0] = zb; E.out[0] = zout;
  E.a[1] = ga; E.b[1] = gb; E.out[1] = gout;
  E.a[2] = da; E.b[2] = db; E.out[2] = dout;
  ProfScope ps(PK_AXPBY, st, 3.0 * (T.rows_all + 2.0 * T.rows_own) * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_extrapolate3<D>), dim3(T.nseg_all), dim3(SEG_ROWS), 0, st, T.segs, whole_grid(mask), gamma,
                                        gamma_dev, T.nseg_own, E));
}

void launch_axpby(int d, hipStream_t st, const SegTable &T, bool all_rows, NodeMask mask, double alpha,
                  const double *a, double beta, const double *b, double *out, int part, double *out2) {
  const int nb = nseg(T, all_rows);
  if (nb == 0) return;
  ProfScope ps(PK_AXPBY, st, (b ? 3.0 : 2.0) * (all_rows ? T.rows_all : T.rows_own) * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, {
    if (part == 0) hipLaunchKernelGGL((k_axpby<D, 0>), dim3(nb), dim3(SEG_ROWS), 0, st, T.segs, mask, alpha, a, beta, b, out, out2);
    else if (part == 1) hipLaunchKernelGGL((k_axpby<D, 1>), dim3(nb), dim3(SEG_ROWS), 0, st, T.segs, mask, alpha, a, beta, b, out, nullptr);
    else hipLaunchKernelGGL((k_axpby<D, 2>), dim3(nb), dim3(SEG_ROWS), 0, st, T.segs, mask, alpha, a, beta, b, out, nullptr);
  });
}

void launch_tail_pack(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *xak, double *xk, double *z,
                      const int *pack_rows, int npack, double *pack) {
  const int nb = T.nseg_own + (npack + SEG_ROWS - 1) / SEG_ROWS;
  if (nb == 0) return;
  ProfScope ps(PK_AXPBY, st, (2.0 * T.rows_own + 2.0 * npack) * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_tail_pack<D>), dim3(nb), dim3(SEG_ROWS), 0, st, T.segs, whole_grid(mask), T.nseg_own, xak, xk, z,
                                        pack_rows, npack, pack));
}

void launch_axpby_node(int d, hipStream_t st, const SegTable &T, NodeMask mask, const NodeCoefs &C, const double *a,
                       const double *b, double *out) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_AXPBY, st, 3.0 * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_axpby_node<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, C, a, b,
                                        out));
}

void launch_rot_rowscale(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *dinv, const double *in, double *out) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_AXPBY, st, 2.0 * T.rows_own * 8.0 * d * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_rot_rowscale<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, dinv, in, out));
}

void launch_cg_step(int d, hipStream_t st, const SegTable &T, NodeMask mask, const NodeCoefs &C, const double *p,
                    const double *Hp, double *s, double *hs, double *r, const CgNode *cg, const double *r0, const double *X,
                    double *xprop, const NodeBits *rmask) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_AXPBY, st, ((r0 ? 6.0 : 8.0) + (xprop ? 2.0 : 0.0)) * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_cg_step<D>), dim3(own_grid(T, mask)), dim3(SEG_ROWS), 0, st, T.segs, mask, C, p, Hp, s, hs,
                                        r, cg, r0, X, xprop, rmask));
}

void launch_cg_dir(int d, hipStream_t st, const SegTable &T, NodeMask mask, const CgNode *cg, const double *v, double *p) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_AXPBY, st, 3.0 * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_cg_dir<D>), dim3(own_grid(T, mask)), dim3(SEG_ROWS), 0, st, T.segs, mask, cg, v, p));
}

void launch_cg_begin(hipStream_t st, int nnodes, NodeBits bits, const CgStart &S, int max_it, CgNode *cg, NodeBits *dmask) {
  hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(64), 0, st, nnodes, bits, S, max_it, cg, dmask);
}

void launch_tnt_begin(hipStream_t st, const SegTable &T, int nnodes, NodeBits bits, bool use_precon, int max_it, double grad_tol,
                      double pgrad_tol, double kappa, double theta, const double *Delta, const double *partials, CgNode *cg,
                      NodeBits *dmask, double *host_tnt) {
  TntBegin B;
  B.bits = bits; B.use_precon = use_precon; B.max_it = max_it;
  B.grad_tol = grad_tol; B.pgrad_tol = pgrad_tol; B.kappa = kappa; B.theta = theta;
  for (int a = 0; a < MAX_LOCAL_NODES; a++) B.Delta[a] = a < nnodes ? Delta[a] : 0.0;
  ProfScope ps(PK_REDUCE, st, 8.0 * 6 * T.nseg_own);
  hipLaunchKernelGGL(k_tnt_begin, dim3(nnodes), dim3(384), 0, st, T, nnodes, B, partials, cg, dmask, host_tnt);
}

void launch_cg_scal_begin(hipStream_t st, const SegTable &T, int nnodes, NodeBits bits, bool use_precon, int max_it, double grad_tol,
                           double pgrad_tol, double kappa, double theta, const double *Delta, const double *partials, CgNode *cg,
                           NodeBits *dmask, double *host_tnt, double *host_scalars, unsigned *arrived, unsigned long long *host_flag,
                           unsigned long long seq, unsigned long long *dev_seq, double *dev_tnt, int upd_nslots, double *upd_host) {
  TntBegin B;
  B.bits = bits; B.use_precon = use_precon; B.max_it = max_it;
  B.grad_tol = grad_tol; B.pgrad_tol = pgrad_tol; B.kappa = kappa; B.theta = theta;
  for (int a = 0; a < MAX_LOCAL_NODES; a++) B.Delta[a] = a < nnodes ? Delta[a] : 0.0;
  if (upd_nslots > 6) { fprintf(stderr, "[dpgo_amd] ERROR: k_cg_scal_begin carries at most six sums of an update.\n"); return; }
  ProfScope ps(PK_REDUCE, st, 8.0 * (10 * T.nseg_own + upd_nslots * T.nseg_all));
  hipLaunchKernelGGL(k_cg_scal_begin, dim3(nnodes), dim3(upd_nslots > 0 ? 1024 : 640), 0, st, T, B, partials, cg, dmask, host_tnt, host_scalars,
                     arrived, host_flag, seq, dev_seq, dev_tnt, upd_nslots, upd_host);
}
int cg_first_slot() { return CG_FIRST_SLOT; }

void launch_cg_scal(hipStream_t st, const SegTable &T, int nnodes, int phase, const double *partials, CgNode *cg,
                    NodeBits *dmask, double *host_scalars, unsigned *arrived, unsigned long long *host_flag,
                    unsigned long long seq, unsigned long long *dev_seq) {
  ProfScope ps(PK_REDUCE, st, 8.0 * (phase == 0 ? 4 : 1) * T.nseg_own);
  hipLaunchKernelGGL(k_cg_scal, dim3(nnodes), dim3(256), 0, st, T, phase, partials, cg, dmask, host_scalars, arrived,
                     host_flag, seq, dev_seq);
}

void launch_dots(int d, hipStream_t st, const SegTable &T, NodeMask mask, int n, const double *const *a,
                 const double *const *b, const int *parts, double *partials, int slot0) {
  if (T.nseg_own == 0 || n <= 0) return;
  DotPairs P;
  P.n = n;
  for (int q = 0; q < MAX_DOTS; q++) {
    P.a[q] = a[q < n ? q : 0];
    P.b[q] = b[q < n ? q : 0];
    P.part[q] = parts[q < n ? q : 0];
  }
  ProfScope ps(PK_DOT, st, 2.0 * n * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_dots<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, P,
                                        partials + (size_t)slot0 * T.nseg_all, T.nseg_all));
}

void launch_cg_init(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *grad, const double *pgrad,
                    double *s, double *hs, double *r, double *v, double *p) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_AXPBY, st, (s ? 7.0 : 2.0) * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_cg_init<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, grad,
                                        pgrad, s, hs, r, v, p));
}

void launch_tangent_full(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *X,
                         const double *V, double *out, double *partials, int slot, const double *add, double *sum_out) {
  if (T.nseg_own == 0) return;
  double *part = partials ? partials + (size_t)slot * T.nseg_all : nullptr;
  ProfScope ps(PK_ROTOP, st, (add ? 4.0 : 2.0) * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_tangent_full<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, X,
                                        V, add, sum_out, out, part));
}

void launch_copy_nbr_rows(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *src, double *dst) {
  const int nb = T.nseg_all - T.nseg_own;
  if (nb <= 0) return;
  ProfScope ps(PK_AXPBY, st, 2.0 * (T.rows_all - T.rows_own) * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_axpby<D, 0>), dim3(nb), dim3(SEG_ROWS), 0, st, T.segs + T.nseg_own, mask, 1.0, src,
                                        0.0, nullptr, dst, nullptr));
}

void launch_tangent_rot(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *X,
                        const double *in, double *out, const double *dotv, double *partials, int slot, bool two, double *neg) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_ROTOP, st, (dotv ? 4.0 : 3.0) * T.rows_own * 8.0 * (d + 1) * d);
  double *part = (dotv && partials) ? partials + (size_t)slot * T.nseg_all : nullptr;
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_rot_op<D>), dim3(own_grid(T, mask)), dim3(SEG_ROWS), 0, st, T.segs, mask, 0, X, in,
                                        dotv, part, out, T.nseg_all, two ? 1 : 0, neg));
}

void launch_retract_rot(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *X,
                        const double *V, double *out) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_ROTOP, st, 3.0 * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_rot_op<D>), dim3(own_grid(T, mask)), dim3(SEG_ROWS), 0, st, T.segs, mask, 2, X, V,
                                        nullptr, nullptr, out, 0, 0, nullptr));
}

void launch_copy_indexed(int d, hipStream_t st, int count, const int *didx, const int *sidx, const double *src,
                         double *dst, const NodeBits *gate) {
  if (count == 0) return;
  ProfScope ps(PK_COPYIDX, st, 2.0 * count * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_copy_indexed<D>), dim3((count + 255) / 256), dim3(256), 0, st, count,
                                        didx, sidx, src, dst, gate));
}

void launch_bdiag_dot(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *Dd, const double *x,
                      double coef, const double *add, double addcoef, double *partials, int slot) {
  if (T.nseg_own == 0) return;
  ProfScope ps(PK_BDIAG, st, 3.0 * T.rows_own * 8.0 * (d + 1) * d);
  DPGO_DISPATCH_D(d, hipLaunchKernelGGL((k_bdiag_dot<D>), dim3(T.nseg_own), dim3(SEG_ROWS), 0, st, T.segs, mask, Dd, x,
                                        coef, add, addcoef, partials + (size_t)slot * T.nseg_all));
}

// ---- AMM-PGO*: the master's sums on the device (DPGOStar.cpp:147-192, 713-761) ---------------------------------------
// The global objective of up to two trial points and up to two squared distances, from the partial sums their passes left
// (six partial-sum slots: k_cost's intra and inter sums of point 1 / 2, over all rows; k_sqdist's of pair 1 / 2, own rows):
//   out[0] = sum_a 1/2 s0 + 1/4 s1,  out[1] = sum_a 1/2 s2 + 1/4 s3,  out[2] = sum_a s4,  out[3] = sum_a s5
// One wave per slot sums a node's partials in k_reduce's order, one thread adds the nodes in index order: the same
// numbers whatever the schedule.  The four values stay in device memory -- an all-reduce over the groups can follow on
// the same stream -- and k_publish hands them to the host with the usual flag.
struct StarSlots { int s[6]; };   // where the six partial sums are (the second point's pair lives in slots nobody else reduces over all rows)
__global__ __launch_bounds__(384) void k_star_sums(SegTable T, int nnodes, unsigned valid, StarSlots sl, const double *partials, double *out) {
  __shared__ double ns[6][MAX_LOCAL_NODES];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double *p = partials + (size_t)sl.s[wv] * T.nseg_all;
  const bool on = (valid >> wv) & 1u;   // (a slot no pass has filled counts as zero)
  for (int a = 0; a < nnodes; a++) {
    double v = 0;
    if (on) {
      for (int k = T.own_ptr[a] + lane; k < T.own_ptr[a + 1]; k += 64) v += p[k];
      if (wv < 4)
        for (int k = T.nbr_ptr[a] + lane; k < T.nbr_ptr[a + 1]; k += 64) v += p[k];
    }
    v = wave_sum(v);
    if (lane == 0) ns[wv][a] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double f1 = 0, f2 = 0, d1 = 0, d2 = 0;
    for (int a = 0; a < nnodes; a++) {
      f1 += 0.5 * ns[0][a] + 0.25 * ns[1][a];
      f2 += 0.5 * ns[2][a] + 0.25 * ns[3][a];
      d1 += ns[4][a];
      d2 += ns[5][a];
    }
    out[0] = f1; out[1] = f2; out[2] = d1; out[3] = d2;
  }
}
__global__ void k_publish(const double *vals, int n, double *host, unsigned long long *host_flag, unsigned long long seq,
                          unsigned long long *dev_seq) {
  if (threadIdx.x < n) __hip_atomic_store(host + threadIdx.x, vals[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);
    if (seq == 0) seq = *dev_seq + 1;   // (replayed from a graph: see k_reduce)
    *dev_seq = seq;
    __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void launch_star_sums(hipStream_t st, const SegTable &T, int nnodes, unsigned valid_slots, const int *slots6, const double *partials, double *out) {
  ProfScope ps(PK_REDUCE, st, 8.0 * 6 * T.nseg_all);
  StarSlots sl;
  for (int q = 0; q < 6; q++) sl.s[q] = slots6[q];
  hipLaunchKernelGGL(k_star_sums, dim3(1), dim3(384), 0, st, T, nnodes, valid_slots, sl, partials, out);
}
void launch_publish(hipStream_t st, const double *vals, int n, double *host, unsigned long long *host_flag, unsigned long long seq,
                    unsigned long long *dev_seq) {
  ProfScope ps(PK_REDUCE, st, 8.0 * n);
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, vals, n, host, host_flag, seq, dev_seq);
}

// ---- the gate of a speculative update (Group::speculate_update): one lane per node evaluates, from the trial point's sums, the refinement's start (k_cg_scal_begin) and the CG's state, whether the iteration
// takes its COMMON course -- the node was refined, its CG ended with the first step, the step is accepted and ends the
// refinement (TNT.h:537-607), the half step needs no redo, there is no restart and no fallback (DPGOHash.cpp:386-441) -- with
// the host's expressions in the host's order (amm(), run_tnt()'s judge): the two must reach the same verdict from the same
// bits, and the host checks that they did.  *go = all ones if every node does, else 0: the launches of the next update(),
// enqueued behind this kernel under that word, run or fall through.
__device__ __forceinline__ bool amm_gate_node(const AmmGate &G, int a, const double *sums, const double *tnt, const CgNode *cg) {
#pragma clang fp contract(off)   // (the host rounds every product before it adds: so must this)
  double t[16], b[TNT_SUMMARY];   // (the trial point's six sums and the half step's parked ones, slots 2 * MAX_DOTS ..)
  static_assert(2 * MAX_DOTS + 2 <= 16, "the gate's slots");
  // (written by other workgroups of this launch: read past this XCD's L2)
#pragma unroll
  for (int q = 0; q < 16; q++) t[q] = __hip_atomic_load(sums + a * MAX_SLOTS + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int q = 0; q < TNT_SUMMARY; q++) b[q] = tnt[a * TNT_SUMMARY + q];
  const double f = G.f[a], Fk0 = G.Fk0[a], Fk1 = G.Fk1[a], fobj = G.fobj[a];
  const bool active = b[6] != 0.0, cg_over = cg[a].live == 0;
  const double fx = 0.5 * (b[1] + b[2]) + f;                 // norms_take
  const double fx_prop = 0.5 * (t[5] + t[3]) + f;            // judge
  const double h_norm = sqrt(t[0]);
  const double dm = -t[1] - 0.5 * t[2];
  const double df = fx - fx_prop;
  const double rel_dec = df / (G.sqrt_eps + fabs(fx));
  const double rho = df / dm;
  const bool ok = (!isnan(rho)) && rho > G.eta1;
  const bool stop = rel_dec < G.rel_tol || h_norm < G.step_tol;
  const bool ends = stop || !(1 < G.max_it && 1 < G.max_acc);   // no further trust-region iteration behind the accepted one
  const double Gk = fx_prop - t[3] + t[4];                    // f(X | g[k]): only the linear term depends on g
  const double Gkh = t[G.ds + 1] + f;
  const double minG = Fk0 - G.psi * t[G.ds];
  const bool redo = Gkh > minG;
  const bool hr = Gk > Fk0;
  const bool sr = (Gk > Fk1 && G.hits0[a] >= G.max_hits0) || (Gk > fobj && G.hits1[a] > G.max_hits1);
  const bool fb = (Fk0 - Gk) < G.phi * (Fk0 - Gkh);
  return active && cg_over && ok && ends && !redo && !(hr || sr) && !fb;
}
// The trial point's k_reduce and the gate in ONE launch: every (node, slot) workgroup sums its slot exactly as k_reduce does
// (host scalars, and the copy in device memory); the last one to arrive -- the sums of every node are there -- takes the
// verdict, a lane per node, sets *go and the host's copy of it, and raises the flag.
__global__ __launch_bounds__(64) void k_reduce_gate(SegTable T, int nslots, const double *partials, double *host_scalars,
                                                    unsigned *arrived, unsigned long long *host_flag, unsigned long long seq,
                                                    unsigned long long *dev_seq, double *dev_scalars, AmmGate G, const double *tnt,
                                                    const CgNode *cg, NodeBits *go, double *host_out) {
  const int a = blockIdx.x / nslots, s = blockIdx.x % nslots, lane = threadIdx.x;
  const double *p = partials + (size_t)s * T.nseg_all;
  double v = 0;
  for (int k = T.own_ptr[a] + lane; k < T.own_ptr[a + 1]; k += 64) v += p[k];
  v = wave_sum(v);
  int last = 0;
  if (lane == 0) {
    __hip_atomic_store(host_scalars + a * MAX_SLOTS + s, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dev_scalars + a * MAX_SLOTS + s, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    last = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM) == gridDim.x - 1;
  }
  last = __shfl(last, 0, 64);
  if (!last) return;
  const bool common = lane < G.nnodes ? amm_gate_node(G, lane, dev_scalars, tnt, cg) : true;
  const bool all = __all(common);
  if (lane == 0) {
    *go = all ? ~0ull : 0ull;
    __hip_atomic_store(host_out, all ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seq == 0) seq = *dev_seq + 1;
    *dev_seq = seq;
    __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void launch_reduce_gate(hipStream_t st, const SegTable &T, int nnodes, int nslots, const double *partials, double *host_scalars,
                        unsigned *arrived, unsigned long long *host_flag, unsigned long long seq, unsigned long long *dev_seq,
                        double *dev_scalars, const AmmGate &G, const double *tnt, const CgNode *cg, NodeBits *go, double *host_out) {
  ProfScope ps(PK_REDUCE, st, 8.0 * nslots * T.nseg_own);
  hipLaunchKernelGGL(k_reduce_gate, dim3(nnodes * nslots), dim3(64), 0, st, T, nslots, partials, host_scalars, arrived, host_flag, seq,
                     dev_seq, dev_scalars, G, tnt, cg, go, host_out);
}
void launch_reduce(hipStream_t st, const SegTable &T, int nnodes, bool all_rows, int nslots, const double *partials,
                   double *host_scalars, unsigned *arrived, unsigned long long *host_flag, unsigned long long seq,
                   unsigned long long *dev_seq) {
  ProfScope ps(PK_REDUCE, st, 8.0 * nslots * T.nseg_all);
  hipLaunchKernelGGL(k_reduce, dim3(nnodes * nslots), dim3(64), 0, st, T, all_rows ? 1 : 0, nslots, partials, host_scalars,
                     arrived, host_flag, seq, dev_seq);
}

void launch_set_coefs(hipStream_t st, const NodeCoefs &C, int n, double *dev) {
  hipLaunchKernelGGL(k_set_coefs, dim3(1), dim3(64), 0, st, C, n, dev);
}

__global__ __launch_bounds__(256) void k_pack_panels(const SpdItem *items, const PanelSrc *srcs, const double *src, double *panels) {
  const SpdItem it = load_item(items + blockIdx.x);
  const PanelSrc ps = srcs[blockIdx.x];
  const double *in = src + ps.src_off;
  double *out = panels + it.mat_off;
  const int cnt = it.count, total = ps.len * cnt;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int k = i / cnt, r = i - k * cnt;
    out[(size_t)k * it.ld + r] = in[(size_t)k * ps.src_ld + r];
  }
}
// The dense product behind the fused root tiles: P = L11^-T L11^-1, P[k][c] = sum_{j >= max(k, c)} Linv[j][k] Linv[j][c],
// for every root front (Linv = L11^-1 = the front's W_s, w x ld row-major, zeros above the diagonal), written as a dense
// w x w matrix; launch_pack_panels then cuts the tiles' panels out of it.  64 x 64 output tiles, 4 x 4 per thread, the
// sum taken in increasing j (a fixed order).  Set-up work (and a Dynamic rescale's): about a millisecond.
__global__ __launch_bounds__(256) void k_root_syrk(const RootDesc *rd, const double *src, double *dst) {
  const RootDesc d = rd[blockIdx.z];
  const int w = d.w, k0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (k0 >= w || c0 >= w) return;
  __shared__ double As[16][65], Bs[16][65];
  const double *L = src + d.src_off;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
  for (int j0 = (max(k0, c0) / 16) * 16; j0 < w; j0 += 16) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int idx = threadIdx.x + q * 256, jj = idx / 64, kk = idx % 64, j = j0 + jj;
      As[jj][kk] = (j < w && k0 + kk < w) ? L[(size_t)j * d.ld + k0 + kk] : 0.0;
      Bs[jj][kk] = (j < w && c0 + kk < w) ? L[(size_t)j * d.ld + c0 + kk] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 16; jj++) {
      double a[4], bq[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { a[i] = As[jj][ty * 4 + i]; bq[i] = Bs[jj][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fma(a[i], bq[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = k0 + ty * 4 + i, c = c0 + tx * 4 + j;
      if (k < w && c < w) dst[d.dst_off + (size_t)k * w + c] = acc[i][j];
    }
}
void launch_root_syrk(hipStream_t st, const RootDesc *rd, int nroots, int max_w, const double *src, double *dst) {
  const int nt = (max_w + 63) / 64;
  if (nroots > 0 && nt > 0) hipLaunchKernelGGL(k_root_syrk, dim3(nt, nt, nroots), dim3(256), 0, st, rd, src, dst);
}

void launch_pack_panels(hipStream_t st, const SpdItem *items, const PanelSrc *srcs, int ntiles, const double *src, double *panels) {
  if (ntiles > 0) hipLaunchKernelGGL(k_pack_panels, dim3(ntiles), dim3(256), 0, st, items, srcs, src, panels);
}

#ifdef SPD_TRACE
void spd_trace_set(unsigned long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spd_trace), &p, sizeof(p)); }
#else
void spd_trace_set(unsigned long long *) {}
#endif
int spd_waves(int rows) { return SPD_NW(rows); }
void launch_root_sym(int d, int dof, hipStream_t st, const SpdDev &S, const SpdLevelMap &M, const double *vec, double *part,
                     double bytes, bool stream_once, NodeMask mask) {
  int maxn = 0;
  for (int j = 0; j < M.nlive; j++) maxn = std::max(maxn, M.ncount[j]);
  const int grid = M.nlive * maxn;
  if (grid == 0) return;
  ProfScope ps(PK_SPD_FWD, st, bytes);
#define ROOT_SYM(DOFV)                                                                                                       \
  do {                                                                                                                       \
    if (stream_once) hipLaunchKernelGGL((k_root_sym<D, DOFV, true>), dim3(grid), dim3(256), 0, st, S, mask, M, vec, part);    \
    else hipLaunchKernelGGL((k_root_sym<D, DOFV, false>), dim3(grid), dim3(256), 0, st, S, mask, M, vec, part);               \
  } while (0)
  DPGO_DISPATCH_D(d, {
    if (dof == 1) ROOT_SYM(1);
    else ROOT_SYM(D);
  });
#undef ROOT_SYM
}
void launch_root_combine(int d, int dof, hipStream_t st, const SpdDev &S, const SpdLevelMap &M, const RootRow *rows,
                         const double *part, double scale, double *out, NodeMask mask) {
  if (M.wide_wgs == 0 || M.nlive == 0) return;
  ProfScope ps(PK_SPD_FWD, st, 0.0);
  DPGO_DISPATCH_D(d, {
    if (dof == 1) hipLaunchKernelGGL((k_root_combine<D, 1>), dim3(M.wide_wgs), dim3(256), 0, st, S, mask, M, rows, part, scale, out);
    else hipLaunchKernelGGL((k_root_combine<D, D>), dim3(M.wide_wgs), dim3(256), 0, st, S, mask, M, rows, part, scale, out);
  });
}
void launch_spd_level(int d, int dof, hipStream_t st, const SpdDev &S, int mode, const SpdLevelMap &M, int rows,
                      double *vec, double *ytmp, double scale, double level_bytes, bool stream_once, NodeMask mask) {
  const int npacks = M.wide_wgs + M.narrow_wgs;
  if (npacks == 0 || M.nlive == 0) return;
  ProfScope ps(mode == 1 ? PK_SPD_BWD : PK_SPD_FWD, st, level_bytes);
#define SPD_LAUNCH3(DOFV, ROWSV, NTV, PTV)                                                                       \
  do {                                                                                                         \
    if (mode == 0)                                                                                             \
      hipLaunchKernelGGL((k_spd_level<D, DOFV, ROWSV, 0, NTV, PTV>), dim3(npacks), dim3(64 * SPD_NW(ROWSV)), 0, st, S, mask, M, scale, vec, ytmp);  \
    else if (mode == 1)                                                                                        \
      hipLaunchKernelGGL((k_spd_level<D, DOFV, ROWSV, 1, NTV, PTV>), dim3(npacks), dim3(64 * SPD_NW(ROWSV)), 0, st, S, mask, M, scale, vec, ytmp); \
    else                                                                                                       \
      hipLaunchKernelGGL((k_spd_level<D, DOFV, ROWSV, 2, NTV, PTV>), dim3(npacks), dim3(64 * SPD_NW(ROWSV)), 0, st, S, mask, M, scale, vec, ytmp); \
  } while (0)
#define SPD_LAUNCH2(DOFV, ROWSV, NTV) SPD_LAUNCH3(DOFV, ROWSV, NTV, double)
#define SPD_LAUNCH(DOFV, ROWSV)                \
  do {                                         \
    if (stream_once) SPD_LAUNCH2(DOFV, ROWSV, true); \
    else SPD_LAUNCH2(DOFV, ROWSV, false);      \
  } while (0)
/* 8-row tiles exist for the fused roots only (a single root front per GPU: twice the workgroups on it), fp64 panels */
#define SPD_ROOT8(DOFV, NTV) \
  hipLaunchKernelGGL((k_spd_level<D, DOFV, 8, 2, NTV, double>), dim3(npacks), dim3(64 * SPD_NW(8)), 0, st, S, mask, M, scale, vec, ytmp)
#define SPD_PICK(DOFV)                  \
  do {                                  \
    if (rows == 8) {                    \
      if (mode != 2) { fprintf(stderr, "[dpgo_amd] ERROR: 8-row solve tiles are a root-level class.\n"); return; } \
      if (stream_once) SPD_ROOT8(DOFV, true); \
      else SPD_ROOT8(DOFV, false);      \
    } else if (rows == 16) SPD_LAUNCH(DOFV, 16); \
    else SPD_LAUNCH(DOFV, 64);          \
  } while (0)
  DPGO_DISPATCH_D(d, {
    if (dof == 1) SPD_PICK(1);
    else SPD_PICK(D);
  });
#undef SPD_PICK
#undef SPD_ROOT8
#undef SPD_LAUNCH
#undef SPD_LAUNCH2
#undef SPD_LAUNCH3
}

}  // namespace dpgo
