// Numeric phase of the multifrontal Cholesky on the GPU (gfx950, fp64, MFMA).
//
// Replaces the host loop of spd.cpp for the factorisations the reference does with CHOLMOD
//   L_.compute / L_.factorize(G_tt)          C++/DPGO/src/DPGOProblem.cpp:93, 315, 479
//   reg_Chol_precon_.compute(G_RR + lambda I) C++/DPGO/src/DPGOProblem.cpp:119-123
// The symbolic analysis (nested dissection, fronts, assembly lists) stays on the host (spd.cpp); this file turns
// matrix values into the front matrices W_s = [L11^-1 ; -L21 L11^-1] the device solve streams.
//
// Level by level (fronts of one tree height are independent, one launch serves them all):
//   1. assemble   F_s <- entries of A (k_fa_scatter) + the children's Schur complements (k_fa_extend; one launch per
//                 child slot, so two children never add to the same entry at once: deterministic, no atomics)
//   2. eliminate  in block columns of 32.  Levels with thousands of fronts (the leaves), right-looking: k_fa_potrf_reg factors
//                 and inverts the 32 x 32 diagonal block (one wave per front), k_fa_potrf_panel scales the rows below
//                 (through LDS, on the matrix cores), k_fa_abt subtracts the rank-32 update P_I P_J^T from the rest of the
//                 128-wide super-block with v_mfma_f64_16x16x4_f64, and once per super-block from everything right of it
//                 (K = 128).  Levels with few fronts (the top of a tree: a launch is what costs), left-looking inside the
//                 super-block: k_fa_panel_ll applies the earlier block columns' updates to the strip itself, factors,
//                 inverts and scales in ONE launch per block column; k_fa_abt only runs the wide pass.
//      Each front carries w extra rows holding the identity: after the elimination they hold L11^-T (the same
//      row operations that turn F21 into L21 = F21 L11^-T), so the triangular inverse costs no kernel of its own.
//   3. finish     W_bottom = -L21 (L11^-T)^T is the same "A B^T" tile product (k_fa_abt, K = w) written straight into
//                 W / W^T; the top of W is the transpose of the identity rows (k_fa_wtop).
// The MFMA work is the batched dense block products of the factor set-up: 2/3 w^3 + 2 u w^2 + u^2 w flops per front.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "spd.h"

namespace dpgo {

namespace {
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NB = 32;    // block column
constexpr int SB = 128;   // super-block: the update of its block columns reaches the rest of the front in one pass
constexpr int TS = 64;    // MFMA tile: 64 x 64 per workgroup, 4 waves of 32 x 32 (2 x 2 instructions of 16 x 16 x 4)
constexpr int LDT = NB + 1;
constexpr int PIV_SLOTS = 64;   // the pivot range is collected in this many (min, max) pairs: 600 fronts on ONE address is 10 us of atomics

struct FrontDesc {
  int w, u, m, slot;          // slot: index inside its level (scratch of the diagonal block inverse)
  long long fm_off;           // front matrix: (m + w) x m row-major (rows m.. hold the identity)
  long long w_off, wt_off;    // outputs, host lay-out of SpdFactor::W / WT
  int ldw, ldm;
  int ent_ptr, ent_end;       // entries of A that land in this front
};

#define FA_OK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      fprintf(stderr, "[dpgo_amd] ERROR: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                                   \
    }                                                                                              \
  } while (0)

// F_s <- A entries (src >= 0) and the ones of the identity rows (src < 0)
__global__ __launch_bounds__(256) void k_fa_scatter(const FrontDesc *fd, const int *lvl, const long long *dst,
                                                    const int *src, const double *aval, double *Fm) {
  const FrontDesc f = fd[lvl[blockIdx.y]];
  const int e = f.ent_ptr + blockIdx.x * 256 + threadIdx.x;
  if (e >= f.ent_end) return;
  const int s = src[e];
  Fm[f.fm_off + dst[e]] = s >= 0 ? aval[s] : 1.0;
}

// parent += child's Schur complement: pair q = (parent, child); child row a -> parent position cmap[a]
struct ExtendPair {
  int parent, child, cmap_off, pad;
};
__global__ __launch_bounds__(256) void k_fa_extend(const FrontDesc *fd, const ExtendPair *pairs, const int *cmap, double *Fm) {
  const ExtendPair pr = pairs[blockIdx.y];
  const FrontDesc c = fd[pr.child], p = fd[pr.parent];
  const int a = blockIdx.x;
  if (a >= c.u) return;
  const int la = cmap[pr.cmap_off + a];
  const double *urow = Fm + c.fm_off + (long long)(c.w + a) * c.m + c.w;
  double *P = Fm + p.fm_off;
  for (int b = threadIdx.x; b <= a; b += 256) {
    const int lb = cmap[pr.cmap_off + b];
    const int r = max(la, lb), cc = min(la, lb);
    P[(long long)r * p.m + cc] += urow[b];
  }
}

// The same sums gathered by the PARENT's rows, every child of a level in ONE launch: a wave per parent row r walks the
// parent's children in slot order and, from each child that has a row a with cmap[a] == r (inv: the inverse map, -1
// where there is none), adds that row to its own.  Needs cmap to ascend with a (then row a's entries b <= a land in the
// columns cmap[b] <= r of row r and nowhere else: one writer per entry, the children in the order of the per-slot
// launches, so the bits are theirs); a tree where it does not keeps the per-slot launches.
struct ExtendRow {
  long long ubase;            // the child's Schur complement: Fm + ubase + a * cm
  int cm, cmap_off, inv_off, pad;
};
__global__ __launch_bounds__(256) void k_fa_extend_rows(const FrontDesc *fd, const int *lvl, const int *row_ptr, const ExtendRow *rows,
                                                        const int *cmap, const int *inv, double *Fm) {
  const int fi = lvl[blockIdx.y];
  const FrontDesc p = fd[fi];
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= p.m) return;
  double *Prow = Fm + p.fm_off + (long long)r * p.m;
  const int q0 = row_ptr[fi], q1 = row_ptr[fi + 1];
  for (int qb = q0; qb < q1; qb += 4) {
    // (the records and the inverse maps of up to four children are requested together: one memory latency, not four)
    ExtendRow er[4];
    int a[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      er[j] = rows[min(qb + j, q1 - 1)];
      a[j] = qb + j < q1 ? inv[er[j].inv_off + r] : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (a[j] < 0) continue;
      const double *urow = Fm + er[j].ubase + (long long)a[j] * er[j].cm;
      // (two children may add into the same entry of the parent's row from different lanes: the additions stay in child order
      // because a wave's loads and stores of one address are performed in the order the wave issues them, and the compiler
      // keeps may-alias accesses in program order; tests/test_gpu_parity.py holds this launch to the bits of the per-slot
      // launches, DPGO_SPD_EXTEND_SLOTS=1, where every child is a kernel of its own)
      for (int b = lane; b <= a[j]; b += 64) Prow[cmap[er[j].cmap_off + b]] += urow[b];
    }
  }
}

// a double from lane `src` (uniform) to every lane
__device__ __forceinline__ double bcast(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// The factor and the inverse of the diagonal block [kb, kb + nb) of front f by ONE wave, in registers and LDS.
// Lane = (row r = lane & 31, half h = lane >> 5): half h holds the columns j of its row with j % 2 == h, so an elimination
// step issues half the multiply-adds of a row-per-lane scheme (one wave is bound by the number of fp64 instructions it
// issues, ~9 cycles each, not by their latency: tools/probes/potrf_probe.hip); what a lane needs from another -- the scaled
// column k, row i of L -- goes through LDS (`Ls`, NB x (NB + 2) doubles of the wave's own), published once and read with
// broadcast loads off the dependent chain; 1 / l_kk comes from v_rsq_f64 and two Newton steps (error of X against a
// long-double inverse: 2.6e-16 of its largest entry, 1.8e-16 with sqrt and a division) and is kept for the inversion,
// which splits its sums over the halves the same way.  On return lane (c, h) holds Xh[q] = X[2 q + h][c], X = L_kk^-1.
// 11.5 us per launch against 30 us for the row-per-lane version of round 2 (48 us before its v_readlane broadcasts went).
// Returns false on a non-positive pivot.
// A, lda: the block (global memory, or the copy a left-looking panel kernel has just updated in LDS).
__device__ __forceinline__ bool potrf_inv_wave(const double *A, long long lda, int nb, int lane, double (&Xh)[NB / 2],
                                               double &dmin, double &dmax, double (*Ls)[NB + 2]) {
  const int r = lane & 31, h = lane >> 5;
  // row r of the diagonal block (rows and columns >= nb: the identity, which factors to itself)
  double L[NB / 2];
#pragma unroll
  for (int q = 0; q < NB / 2; q++) {
    const int j = 2 * q + h;
    L[q] = (r < nb && j <= r) ? A[(long long)r * lda + j] : ((r == j) ? 1.0 : 0.0);
  }
  dmin = 1e300;
  dmax = 0.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < NB; k++) {
    // column k lives in half k & 1, register k >> 1; its diagonal entry in lane k + 32 (k & 1)
    const double dkk = bcast(L[k >> 1], k + 32 * (k & 1));
    if (k < nb) {
      ok = ok && (dkk > 0.0);
      dmin = fmin(dmin, dkk);
      dmax = fmax(dmax, dkk);
    }
    double inv = __builtin_amdgcn_rsq(dkk);
    inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
    inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
    const double lkk = dkk * inv;
    if (h == (k & 1)) {
      const double v = r == k ? lkk : L[k >> 1] * inv;   // column k: the pivot, and the rows below it scaled
      L[k >> 1] = v;
      Ls[r][k] = v;                                      // L[r][k], final
      if (r == 0) Ls[k][NB] = inv;                       // 1 / L[k][k] for the inversion
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double lrk = Ls[r][k];   // this row's entry of column k (the other half holds it for every other k)
#pragma unroll
    for (int q = (k + 1) >> 1; q < NB / 2; q++) {
      const int j = 2 * q + h;
      // (no select for the entries above the diagonal: zeros that stay zeros, and nobody reads them)
      if (j > k) L[q] = fma(-lrk, Ls[j][k], L[q]);
    }
  }
  // X = L^-1, lane (c = r, h) its column c: X[i][c] = (delta_ic - sum_{k<i} L[i][k] X[k][c]) / L[i][i]; half h keeps X[k][c]
  // for k % 2 == h and sums over those k, the halves meet in a swap (zeros above the diagonal come out by themselves)
#pragma unroll
  for (int q = 0; q < NB / 2; q++) Xh[q] = 0.0;
#pragma unroll
  for (int i = 0; i < NB; i++) {
    double sum = (r == i && h == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < (i + 1) / 2; q++) {
      const int k = 2 * q + h;
      const double t = fma(-Ls[i][k], Xh[q], sum);
      sum = k < i ? t : sum;
    }
    sum += __shfl_xor(sum, 32, 64);
    const double x = sum * Ls[i][NB];
    if ((i & 1) == h) Xh[i >> 1] = x;
  }
  return ok;
}

// One block column of the elimination: the Cholesky of the diagonal block [kb, kb + nb) and its inverse, then the rows below
// times the transposed inverse,  F[i, kb:ke] <- F[i, kb:ke] L_kk^-T.
// FUSED (levels with few fronts: a launch is what costs): every workgroup of a front (256 rows each) first factors the
// 32 x 32 diagonal block itself -- wave 0, in registers (potrf_inv_wave), a few microseconds against a launch of its own and
// a trip through memory for the inverse -- and hands X = L_kk^-1 to the other waves through LDS.  Not FUSED (levels with
// thousands of fronts: throughput is what costs, and a workgroup per 256 rows that idles three of its four waves during
// the factorisation wastes it): k_fa_potrf_reg has done the diagonal blocks, one wave per front, and left X in `dinv`.
// (The factored diagonal block itself is never read again -- later steps touch the rows below and the identity rows
// only -- so nobody writes it back.)
// fail[0]: 1 + front of a non-positive pivot; pivr[0] / pivr[1]: smallest / largest pivot d_kk seen (bit patterns of
// positive doubles order like integers); both reported once per front.
__global__ __launch_bounds__(64) void k_fa_potrf_reg(const FrontDesc *fd, const int *lvl, int kb, const double *Fm, double *dinv,
                                                     int *fail, unsigned long long *pivr) {
  const FrontDesc f = fd[lvl[blockIdx.x]];
  if (f.w <= kb) return;
  const int nb = min(NB, f.w - kb), lane = threadIdx.x;
  __shared__ double Ls[NB][NB + 2];
  double Xh[NB / 2], dmin, dmax;
  const bool ok = potrf_inv_wave(Fm + f.fm_off + (long long)kb * f.m + kb, f.m, nb, lane, Xh, dmin, dmax, Ls);
  double *D = dinv + (long long)f.slot * NB * NB;
#pragma unroll
  for (int q = 0; q < NB / 2; q++) D[(2 * q + (lane >> 5)) * NB + (lane & 31)] = ok ? Xh[q] : 0.0;
  if (lane == 0) {
    if (!ok) atomicExch(fail, 1 + lvl[blockIdx.x]);
    else {
      unsigned long long *pr = pivr + 2 * (blockIdx.x % PIV_SLOTS);
      atomicMin(pr, (unsigned long long)__double_as_longlong(dmin));
      atomicMax(pr + 1, (unsigned long long)__double_as_longlong(dmax));
    }
  }
}

template <bool FUSED>
__global__ __launch_bounds__(FUSED ? 320 : 256) void k_fa_potrf_panel(const FrontDesc *fd, const int *lvl, int kb, double *Fm, const double *dinv,
                                                                      int *fail, unsigned long long *pivr) {
  const FrontDesc f = fd[lvl[blockIdx.y]];
  if (f.w <= kb) return;
  const int nb = min(NB, f.w - kb), ke = kb + nb;
  // regular rows [ke, m) and the identity rows whose one has come into play: [m, m + ke)
  const int nrows = (f.m - ke) + ke;
  if ((int)blockIdx.x * 256 >= nrows) return;
  __shared__ double D[NB][LDT];
  __shared__ double T[4][NB][NB + 1];   // a wave's 32 rows of the strip
  __shared__ double Ls[FUSED ? NB : 1][NB + 2];
  __shared__ int bad;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  // The rows times the transposed inverse: X <- X D^T, a wave taking 64 rows in two passes of 32.  A row of the strip is
  // 256 contiguous bytes and the rows are a front's width apart, so the strip moves through LDS: loads and stores with
  // half a wave per row (two whole rows per instruction, against 64 rows x 8 bytes for a row per lane), the product on the
  // matrix cores -- 32 v_mfma_f64_16x16x4 per pass, A = the rows, B = D (lower triangular, its zeros included).
  // FUSED: a fifth wave factors and inverts the diagonal block meanwhile (every workgroup of the front its own copy: the
  // 11 us run beside the first pass's loads instead of in a launch of their own).
  // (which of the five waves factors: not the same one in every workgroup -- a wave's number picks its SIMD, and three
  // workgroups of a CU would queue their 11 us on one SIMD)
  const int pw = FUSED ? 1 + (int)((blockIdx.x + blockIdx.y) % 4) : -1;
  const int rw = (FUSED && wv > pw) ? wv - 1 : wv;   // the wave's number among the four that take rows
  double (*Tw)[NB + 1] = T[rw & 3];
  double *Fb = Fm + f.fm_off + kb;
  const int rr0 = lane >> 5, c = lane & 31;
  auto load_pass = [&](int q0) {
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const int rr = 2 * p + rr0, q = q0 + rr;
      Tw[rr][c] = (q < nrows && c < nb) ? Fb[(long long)(ke + q) * f.m + c] : 0.0;
    }
  };
  const int qw = blockIdx.x * 256 + (rw & 3) * 64;   // this wave's first row (q); its passes: qw, qw + 32
  if constexpr (FUSED) {
    if (wv == pw) {
      double Xh[NB / 2], dmin, dmax;
      const bool ok = potrf_inv_wave(Fm + f.fm_off + (long long)kb * f.m + kb, f.m, nb, lane, Xh, dmin, dmax, Ls);
#pragma unroll
      for (int q = 0; q < NB / 2; q++) D[2 * q + (lane >> 5)][lane & 31] = Xh[q];
      if (lane == 0) {
        bad = !ok;
        if (blockIdx.x == 0) {
          if (!ok) atomicExch(fail, 1 + lvl[blockIdx.y]);
          else {
            unsigned long long *pr = pivr + 2 * (blockIdx.y % PIV_SLOTS);
            atomicMin(pr, (unsigned long long)__double_as_longlong(dmin));
            atomicMax(pr + 1, (unsigned long long)__double_as_longlong(dmax));
          }
        }
      }
    } else if (qw < nrows) {
      load_pass(qw);
    }
  } else {
    const double *Dg = dinv + (long long)f.slot * NB * NB;
    for (int idx = t; idx < NB * NB; idx += 256) D[idx / NB][idx % NB] = Dg[idx];
    if (t == 0) bad = 0;   // (a failed front holds zeros: its rows are zeroed, the failure is already reported)
  }
  __syncthreads();
  if (bad || wv == pw) return;
#define WAVE_SYNC()                                              \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)
  for (int half = 0; half < 2; half++) {
    const int q0 = qw + half * 32;   // rows ke + q0 .. (regular rows, then the identity rows in play)
    if (q0 >= nrows) break;
    if (!(FUSED && half == 0)) load_pass(q0);
    WAVE_SYNC();
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
      const int kq = kk + (lane >> 4), r16 = lane & 15;
      double av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; a++) av[a] = Tw[a * 16 + r16][kq];
#pragma unroll
      for (int b = 0; b < 2; b++) bv[b] = D[b * 16 + r16][kq];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    WAVE_SYNC();
    // C/D lay-out of v_mfma_f64_16x16x4_f64: register r of lane l = C[(l >> 4) + 4 r][l & 15]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) Tw[a * 16 + (lane >> 4) + 4 * r][b * 16 + (lane & 15)] = acc[a][b][r];
    WAVE_SYNC();
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const int rr = 2 * p + rr0, q = q0 + rr;
      if (q < nrows && c < nb) Fb[(long long)(ke + q) * f.m + c] = Tw[rr][c];
    }
    WAVE_SYNC();
  }
#undef WAVE_SYNC
}

// One block column, LEFT-LOOKING inside its super-block (levels with few fronts, where a launch is what costs): the
// updates the block columns [sb, kb) of the super-block owe the strip [kb, ke) are applied here, to the rows this
// workgroup holds and -- by every workgroup of the front again -- to the diagonal block, which is then factored and
// inverted (potrf_inv_wave) and multiplied into the rows.  The right-looking scheme above needs a launch of k_fa_abt per
// block column for the same updates (K = 32 each, 17-30 us at the top of a tree); here they ride in the launch that was
// there anyway, and the rest of the front gets the super-block's update in one wide pass (K = 128) as before.
//   rows of a workgroup: 128 (four waves x 32), of [ke, m + ke): the regular rows below the block and the identity rows.
//   Bs: rows kb.. of the panels [sb, kb) (32 x K, K <= 96); Dg: the diagonal block, then its inverse; T / U: a wave's
//   32 x 32 tile of the strip / of a panel.
// LDS: 76 KB, and 51 KB for the first block column of a super-block (FIRST: no earlier panels, no Bs) -- two and three
// workgroups per CU; a level of 600 - 1 200 workgroups ran in 2.4 - 4.9 rounds of one workgroup per CU with the 110 KB of a
// strip tile and a panel tile of their own per wave (the panel chunks now pass through the strip's tile, which waits in
// registers until the last of them is done).
template <bool FIRST>
__global__ __launch_bounds__(320) void k_fa_panel_ll(const FrontDesc *fd, const int *lvl, int sb, int kb, double *Fm, int *fail,
                                                     unsigned long long *pivr) {
  // five waves: four take 32 rows each, the fifth the diagonal block -- its 11 us run beside the rows' loads and updates,
  // not in front of them
  const FrontDesc f = fd[lvl[blockIdx.y]];
  if (f.w <= kb) return;
  const int nb = min(NB, f.w - kb), ke = kb + nb, K = kb - sb;
  const int nrows = f.m;   // rows [ke, m + ke)
  if ((int)blockIdx.x * 128 >= nrows) return;
  __shared__ double Bs[FIRST ? 1 : NB][FIRST ? 1 : SB - NB + 1];
  __shared__ double Dg[NB][LDT];
  __shared__ double Ls[NB][NB + 2];
  __shared__ double T[4][NB][NB + 1];
  __shared__ int bad;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  double *F = Fm + f.fm_off;
#define WAVE_SYNC()                                              \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)
  const int kq4 = lane >> 4, r16 = lane & 15;
  const int q0 = blockIdx.x * 128 + wv * 32;
  const bool rows_on = wv < 4 && q0 < nrows;
  const int rr0 = lane >> 5, c = lane & 31;
  // every load of the launch is requested before the first is waited for (a loop with a run-time trip count takes a
  // memory latency per trip): the 32 rows kb.. of the panels [sb, kb) and the diagonal block, eight threads per row; the
  // wave's tile of the strip and its rows' own entries of the panels, two whole rows per load
  double bq[(SB - NB) / 8], dq[NB / 8], tq[16], ua[(SB - NB) / NB][16];
  if (wv < 4) {
    const int r = t >> 3, l8 = t & 7;
    const double *src = F + (long long)(kb + r) * f.m + sb;
#pragma unroll
    for (int j = 0; j < (SB - NB) / 8; j++) bq[j] = (r < nb && j * 8 + l8 < K) ? src[j * 8 + l8] : 0.0;
#pragma unroll
    for (int j = 0; j < NB / 8; j++) {
      const int cc = j * 8 + l8;
      dq[j] = (r < nb && cc < nb) ? src[K + cc] : (r == cc ? 1.0 : 0.0);
    }
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const int q = q0 + 2 * p + rr0;
      tq[p] = (rows_on && q < nrows && c < nb) ? F[(long long)(ke + q) * f.m + kb + c] : 0.0;
    }
#pragma unroll
    for (int ch = 0; ch < (SB - NB) / NB; ch++)
#pragma unroll
      for (int p = 0; p < 16; p++) {
        const int q = q0 + 2 * p + rr0;
        ua[ch][p] = (rows_on && ch * NB < K && q < nrows) ? F[(long long)(ke + q) * f.m + sb + ch * NB + c] : 0.0;
      }
    if (!FIRST) {
#pragma unroll
      for (int j = 0; j < (SB - NB) / 8; j++) Bs[r][j * 8 + l8] = bq[j];
    }
#pragma unroll
    for (int j = 0; j < NB / 8; j++) Dg[r][j * 8 + l8] = dq[j];
  }
  __syncthreads();
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  if (wv == 4) {
    if (!FIRST && K > 0) {   // the diagonal block's share of the updates: Dg -= Bs Bs^T
      for (int kk = 0; kk < K; kk += 4) {
        double av[2];
#pragma unroll
        for (int a = 0; a < 2; a++) av[a] = Bs[a * 16 + r16][kk + kq4];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], av[b], acc[a][b], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) Dg[a * 16 + kq4 + 4 * r][b * 16 + r16] -= acc[a][b][r];
      WAVE_SYNC();
    }
    double Xh[NB / 2], dmin, dmax;
    const bool ok = potrf_inv_wave(&Dg[0][0], LDT, nb, lane, Xh, dmin, dmax, Ls);
    WAVE_SYNC();   // (every lane has read its row of Dg: the inverse may take its place)
#pragma unroll
    for (int q = 0; q < NB / 2; q++) Dg[2 * q + (lane >> 5)][lane & 31] = Xh[q];
    if (lane == 0) {
      bad = !ok;
      if (blockIdx.x == 0) {
        if (!ok) atomicExch(fail, 1 + lvl[blockIdx.y]);
        else {
          unsigned long long *pr = pivr + 2 * (blockIdx.y % PIV_SLOTS);
          atomicMin(pr, (unsigned long long)__double_as_longlong(dmin));
          atomicMax(pr + 1, (unsigned long long)__double_as_longlong(dmax));
        }
      }
    }
  } else if (rows_on) {
    double (*Tw)[NB + 1] = T[wv];
    double (*Uw)[NB + 1] = T[wv];   // (the panel chunks go through the same tile: the strip's own waits in tq)
#pragma unroll
    for (int ch = 0; ch < (FIRST ? 0 : (SB - NB) / NB); ch++) {
      const int kc = ch * NB;
      if (kc >= K) break;
#pragma unroll
      for (int p = 0; p < 16; p++) Uw[2 * p + rr0][c] = ua[ch][p];
      WAVE_SYNC();
#pragma unroll
      for (int kk = 0; kk < NB; kk += 4) {
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; a++) av[a] = Uw[a * 16 + r16][kk + kq4];
#pragma unroll
        for (int b = 0; b < 2; b++) bv[b] = Bs[b * 16 + r16][kc + kk + kq4];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
      WAVE_SYNC();
    }
#pragma unroll
    for (int p = 0; p < 16; p++) Tw[2 * p + rr0][c] = tq[p];
    WAVE_SYNC();
    if (!FIRST && K > 0) {
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) Tw[a * 16 + kq4 + 4 * r][b * 16 + r16] -= acc[a][b][r];
    }
  }
  __syncthreads();   // the inverse of the diagonal block is in Dg; the strips carry their updates
  if (bad || !rows_on) return;
  double (*Tw)[NB + 1] = T[wv];
  // times the transposed inverse
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    double av[2], bv[2];
#pragma unroll
    for (int a = 0; a < 2; a++) av[a] = Tw[a * 16 + r16][kk + kq4];
#pragma unroll
    for (int b = 0; b < 2; b++) bv[b] = Dg[b * 16 + r16][kk + kq4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
  }
  WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) Tw[a * 16 + kq4 + 4 * r][b * 16 + r16] = acc[a][b][r];
  WAVE_SYNC();
#pragma unroll
  for (int p = 0; p < 16; p++) {
    const int rr = 2 * p + rr0, q = q0 + rr;
    if (q < nrows && c < nb) F[(long long)(ke + q) * f.m + kb + c] = Tw[rr][c];
  }
#undef WAVE_SYNC
}

// C[I, J] -= A[I, K] B[J, K]^T on 64 x 64 tiles with v_mfma_f64_16x16x4_f64.
//  MODE 0 (trailing update of block column kb): A = B = the panel F[:, kb:ke]; rows I over [ke, m + ke), columns J
//         over [ke, m); an element (i, j) is touched iff (i < m and j <= i) or (i >= m and j < w).
//  MODE 1 (W_bottom): out[a, j] = -sum_k F[w + a, k] F[m + j, k], k over [0, w); a over [0, u), j over [0, w);
//         written to W[(w + a) ldw + j] and WT[j ldm + w + a].
//  MODE 0, narrow pass (wide = 0): K = the block column [k_lo, k_hi) (clipped to the front's w); columns from its end to
//  the end of the 128-wide super-block sb_end -- or to the end of the front when the super-block holds the front's
//  last pivots.  Wide pass (wide = 1): K = the whole super-block [k_lo, k_hi = sb_end), columns right of it; only
//  for fronts with pivots beyond the super-block.  The wide pass is what keeps the read-modify-write of the
//  trailing matrix from bounding the kernel (K = 128 instead of 32).
//  MODE 1 runs once, behind the last level, for every front at once (nobody reads W on the way up the tree: the parents
//  take the Schur complements out of the fronts themselves): lvl = (front, tile) pairs, one per workgroup.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_fa_abt(const FrontDesc *fd, const int *lvl, int k_lo, int k_hi, int sb_end, int wide,
                                                double *Fm, double *Wout, double *WTout) {
  const FrontDesc f = fd[MODE == 1 ? lvl[2 * blockIdx.x] : lvl[blockIdx.y]];
  int r0, c0, nrt, nct, kbeg, kend, cend = 0;
  if (MODE == 0) {
    if (f.w <= k_lo) return;
    if (wide == 1 && f.w <= sb_end) return;   // (wide == 2, behind left-looking block columns: every front with pivots here)
    const int ke = min(k_hi, f.w);
    c0 = ke;
    cend = (wide || sb_end >= f.w) ? f.m : sb_end;
    if (c0 >= cend) return;
    r0 = c0;                                       // lower triangle: rows from the first touched column on
    nrt = (f.m + ke - r0 + TS - 1) / TS;           // rows [r0, m + ke): regular rows and the identity rows < ke
    nct = (cend - c0 + TS - 1) / TS;
    kbeg = k_lo; kend = ke;
  } else {
    if (f.u == 0) return;
    r0 = f.w; c0 = f.m;                           // A rows: L21 (w .. m); B rows: the identity rows (m .. m + w)
    nrt = (f.u + TS - 1) / TS;
    nct = (f.w + TS - 1) / TS;
    kbeg = 0; kend = f.w;
  }
  const int tile = MODE == 1 ? lvl[2 * blockIdx.x + 1] : (int)blockIdx.x;
  if (tile >= nrt * nct) return;
  const int ti = tile / nct, tj = tile % nct;
  const int i0 = r0 + ti * TS, j0 = c0 + tj * TS;          // first row of the A tile / of the B tile (rows of F)
  int ilim, jlim;
  if (MODE == 0) {
    ilim = f.m + kend;                                     // rows < m + ke
    jlim = cend;
    // tiles strictly above the diagonal carry nothing (for the identity rows every column < w counts)
    if (i0 < f.m && j0 > min(i0 + TS - 1, f.m - 1)) return;
    if (i0 >= f.m && j0 >= f.w) return;
  } else {
    ilim = f.m;
    jlim = f.m + f.w;
  }
  __shared__ double As[TS][LDT], Bs[TS][LDT];
  const double *F = Fm + f.fm_off;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;       // this wave's 32 x 32 quadrant
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  // MODE 0: the entries this lane subtracts from are requested first of all (their latency passes behind the products
  // instead of in front of the stores)
  double cpre[2][2][4];
  if (MODE == 0) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int ii = i0 + wr + a * 16 + (lane >> 4) + 4 * r, jj = j0 + wc + b * 16 + (lane & 15);
          const bool on = (ii < f.m) ? (jj <= ii) : (ii < ilim && jj < f.w);
          cpre[a][b][r] = (on && jj < cend) ? F[(long long)ii * f.m + jj] : 0.0;
        }
  }
  // 64 rows x 32 columns of each operand per step, coalesced along the row; the next step's are in flight (registers)
  // while this step's products run
  double pa[TS * NB / 256], pb[TS * NB / 256];
  auto fetch = [&](int k0) {
    const int kn = min(NB, kend - k0);
#pragma unroll
    for (int j = 0; j < TS * NB / 256; j++) {
      const int idx = t + 256 * j, r = idx / NB, k = idx % NB;
      const int ia = i0 + r, ib = j0 + r;
      pa[j] = (ia < ilim && k < kn) ? F[(long long)ia * f.m + k0 + k] : 0.0;
      pb[j] = (ib < jlim && k < kn) ? F[(long long)ib * f.m + k0 + k] : 0.0;
    }
  };
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += NB) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TS * NB / 256; j++) {
      const int idx = t + 256 * j;
      As[idx / NB][idx % NB] = pa[j];
      Bs[idx / NB][idx % NB] = pb[j];
    }
    __syncthreads();
    if (k0 + NB < kend) fetch(k0 + NB);
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
      const int kq = kk + (lane >> 4), rr = lane & 15;
      double av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; a++) av[a] = As[wr + a * 16 + rr][kq];
#pragma unroll
      for (int b = 0; b < 2; b++) bv[b] = Bs[wc + b * 16 + rr][kq];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  // C/D lay-out of v_mfma_f64_16x16x4_f64: register r of lane l = C[(l >> 4) + 4 r][l & 15]
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ii = i0 + wr + a * 16 + (lane >> 4) + 4 * r, jj = j0 + wc + b * 16 + (lane & 15);
        const double v = acc[a][b][r];
        if (MODE == 0) {
          const bool on = (ii < f.m) ? (jj <= ii) : (ii < ilim && jj < f.w);
          if (on && jj < cend) Fm[f.fm_off + (long long)ii * f.m + jj] = cpre[a][b][r] - v;
        } else {
          const int arow = ii - f.w, jcol = jj - f.m;
          if (ii < f.m && jcol < f.w) {
            Wout[f.w_off + (long long)(f.w + arow) * f.ldw + jcol] = -v;
            WTout[f.wt_off + (long long)jcol * f.ldm + f.w + arow] = -v;
          }
        }
      }
}

// top of W: L11^-1 = (identity rows)^T; and its transpose into WT.  Once for every front, behind the last level (as
// MODE 1 of k_fa_abt): items = (front, first row) pairs, WTOP_ROWS identity rows per workgroup, a wave per row.
constexpr int WTOP_ROWS = 8;
__global__ __launch_bounds__(256) void k_fa_wtop(const FrontDesc *fd, const int *items, const double *Fm, double *Wout, double *WTout) {
  const FrontDesc f = fd[items[2 * blockIdx.x]];
  const int k0 = items[2 * blockIdx.x + 1], wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kend = min(k0 + WTOP_ROWS, f.w);
  for (int k = k0 + wv; k < kend; k += 4) {   // identity row k holds L11^-T[k, :] = column k of L11^-1
    const double *row = Fm + f.fm_off + (long long)(f.m + k) * f.m;
    for (int p = lane; p < f.w; p += 64) {
      const double v = p >= k ? row[p] : 0.0;
      WTout[f.wt_off + (long long)k * f.ldm + p] = v;             // WT[k][p] = W[p][k]
      if (p >= k) Wout[f.w_off + (long long)p * f.ldw + k] = v;   // W[p][k], lower triangle
    }
  }
}
}  // namespace

// The state of a numeric factorisation on the device: the maps from the entries of A into the fronts, the front
// storage, the outputs W / WT.  One-shot callers build it, run it and drop it; a factor that is re-done with new values
// every few iterations (Rescale::Dynamic re-factors G_tt) keeps it (SpdFactor::keep_numeric): a refactorisation is
// then nothing but the kernels -- no host pass over A, no allocation, no upload -- and the values of A themselves
// live on the device (spd_numeric_values), where the kernel that rescales the surrogate writes the new diagonal.
struct SpdNumericCtx {
  int nt = 0, maxh = 0;
  long long fm_total = 0;
  size_t max_lvl = 1, n_aval = 0;
  int64_t w_total = 0, wt_total = 0;
  std::vector<FrontDesc> fd;
  std::vector<std::vector<int>> lvl;
  std::vector<int> lvl_ptr;
  std::vector<std::vector<std::pair<int, int>>> pair_rng;   // per level: (first pair, count) per child slot
  std::vector<std::vector<int>> pair_maxu;
  FrontDesc *d_fd = nullptr;
  long long *d_dst = nullptr;
  int *d_src = nullptr, *d_cmap = nullptr, *d_lvl = nullptr, *d_fail = nullptr;
  double *d_aval = nullptr, *d_Fm = nullptr, *d_dinv = nullptr, *d_W = nullptr, *d_WT = nullptr;
  ExtendPair *d_pairs = nullptr;
  int *d_wtop_items = nullptr, *d_wbot_items = nullptr;   // (front, first row) / (front, tile): the outputs, all fronts at once
  ExtendRow *d_xrows = nullptr;                           // k_fa_extend_rows: a parent's children in slot order, ...
  int *d_xrow_ptr = nullptr, *d_inv = nullptr;            // ... where they start per front, the inverse maps
  bool extend_by_rows = false;
  int n_wtop_items = 0, n_wbot_items = 0, max_ent = 0;
  hipStream_t st = nullptr;
  bool outputs_zeroed = false;
  ~SpdNumericCtx() {
    for (void *q : {(void *)d_fd, (void *)d_dst, (void *)d_src, (void *)d_cmap, (void *)d_lvl, (void *)d_fail, (void *)d_aval,
                    (void *)d_Fm, (void *)d_dinv, (void *)d_W, (void *)d_WT, (void *)d_pairs, (void *)d_wtop_items, (void *)d_wbot_items, (void *)d_xrows, (void *)d_xrow_ptr, (void *)d_inv})
      if (q) (void)hipFree(q);
    if (st) (void)hipStreamDestroy(st);
    if (h_io) (void)hipHostFree(h_io);
  }
  int build(const CsrMatrix &A, const SpdFactor &F, const std::vector<std::vector<int>> &children);
  // on: the stream to run on (nullptr: the context's own); defer: return once everything is enqueued -- finish() waits
  // and reads the verdict (a caller that has more to enqueue behind the factorisation, Rescale::Dynamic)
  int factor(SpdFactor &F, const double *aval_host, double *flops_out, double *mfma_ms_out, hipStream_t on = nullptr, bool defer = false);
  int finish(SpdFactor &F, bool wait = true);   // wait = false: the caller knows the stream has passed the read-back
  unsigned long long *h_io = nullptr;   // pinned: the start values of d_fail, and its read-back
  hipStream_t pending_on = nullptr;
  bool pending = false;
};

int SpdNumericCtx::build(const CsrMatrix &A, const SpdFactor &F, const std::vector<std::vector<int>> &children) {
  nt = F.nfronts;
  const int n = A.n;
  fd.assign(nt, FrontDesc());
  std::vector<int> height(nt, 0);
  for (int f = 0; f < nt; f++)
    if (F.parent[f] >= 0) height[F.parent[f]] = std::max(height[F.parent[f]], height[f] + 1);
  maxh = 0;
  for (int f = 0; f < nt; f++) maxh = std::max(maxh, height[f]);
  lvl.assign(maxh + 1, {});
  fm_total = 0;
  for (int f = 0; f < nt; f++) {
    FrontDesc &d = fd[f];
    d.w = F.w[f]; d.u = F.u[f]; d.m = d.w + d.u;
    d.slot = (int)lvl[height[f]].size();
    if (d.w == 0 && d.u > 0) {
      fprintf(stderr, "[dpgo_amd] ERROR: spd_factor: a front without pivots.\n");
      return -1;
    }
    if (d.m > 0) lvl[height[f]].push_back(f);
    d.fm_off = fm_total;
    fm_total += (long long)(d.m + d.w) * d.m;
    d.w_off = F.w_off[f]; d.wt_off = F.wt_off[f]; d.ldw = F.ldw[f]; d.ldm = F.ldm[f];
  }
  w_total = F.w_off[nt];
  wt_total = F.wt_off[nt];
  // entries of A per front (and the ones of the identity rows), child -> parent position maps
  std::vector<long long> dst;
  std::vector<int> src, cmap(std::max(F.total_upd, 1), 0), loc(n, -1);
  std::vector<int> cmap_off(nt, 0);
  for (int f = 0; f < nt; f++) cmap_off[f] = F.ubuf_off[f];
  for (int f = 0; f < nt; f++) {
    FrontDesc &d = fd[f];
    const int *piv = &F.piv_idx[F.piv_ptr[f]];
    const int *up = d.u ? &F.upd_idx[F.upd_ptr[f]] : nullptr;
    for (int k = 0; k < d.w; k++) loc[piv[k]] = k;
    for (int k = 0; k < d.u; k++) loc[up[k]] = d.w + k;
    d.ent_ptr = (int)dst.size();
    for (int k = 0; k < d.w; k++) {
      const int v = piv[k];
      for (int e = A.ptr[v]; e < A.ptr[v + 1]; e++) {
        const int l = loc[A.col[e]];
        if (l >= k) { dst.push_back((long long)l * d.m + k); src.push_back(e); }
      }
      dst.push_back((long long)(d.m + k) * d.m + k);
      src.push_back(-1);
    }
    d.ent_end = (int)dst.size();
    for (int c : children[f]) {
      const int uc = F.u[c];
      const int *upc = uc ? &F.upd_idx[F.upd_ptr[c]] : nullptr;
      for (int a = 0; a < uc; a++) cmap[cmap_off[c] + a] = loc[upc[a]];
    }
    for (int k = 0; k < d.w; k++) loc[piv[k]] = -1;
    for (int k = 0; k < d.u; k++) loc[up[k]] = -1;
  }
  max_lvl = 1;
  for (const auto &l : lvl) max_lvl = std::max(max_lvl, l.size());
  if (max_lvl > 65535) {
    // the per-level launches index the fronts of a level with the grid's y extent (only the scatter over ALL fronts is cut
    // into slices): a tree with more fronts in one level -- about 8 M unknowns at the default leaf size -- is refused here,
    // with a message, rather than at the first launch, as a HIP error
    fprintf(stderr, "[dpgo_amd] ERROR: the elimination tree has %zu fronts in one level; the device factorisation handles at most 65535 "
                    "(larger leaves -- DPGO_SPD_LEAF_TT / _RR -- give fewer fronts).\n", max_lvl);
    return -1;
  }
  std::vector<int> lvl_flat;
  lvl_ptr.assign(1, 0);
  for (const auto &l : lvl) { lvl_flat.insert(lvl_flat.end(), l.begin(), l.end()); lvl_ptr.push_back((int)lvl_flat.size()); }
  // extend pairs per (level, child slot)
  std::vector<ExtendPair> pairs;
  pair_rng.assign(maxh + 1, {});
  pair_maxu.assign(maxh + 1, {});
  for (int h = 0; h <= maxh; h++) {
    size_t maxc = 0;
    for (int f : lvl[h]) maxc = std::max(maxc, children[f].size());
    for (size_t sl = 0; sl < maxc; sl++) {
      const int first = (int)pairs.size();
      int mu = 0;
      for (int f : lvl[h])
        if (children[f].size() > sl && F.u[children[f][sl]] > 0) {
          const int c = children[f][sl];
          pairs.push_back({f, c, cmap_off[c], 0});
          mu = std::max(mu, F.u[c]);
        }
      pair_rng[h].push_back({first, (int)pairs.size() - first});
      pair_maxu[h].push_back(mu);
    }
  }
  // the children of a front in slot order with the inverse of their maps (k_fa_extend_rows), if every map ascends
  std::vector<ExtendRow> xrows;
  std::vector<int> xrow_ptr(nt + 1, 0), inv;
  extend_by_rows = getenv("DPGO_SPD_EXTEND_SLOTS") == nullptr;
  for (int f = 0; f < nt && extend_by_rows; f++) {
    xrow_ptr[f] = (int)xrows.size();
    for (int c : children[f]) {
      if (F.u[c] == 0) continue;
      if (inv.size() + (size_t)fd[f].m > (size_t)0x7fffffff) { extend_by_rows = false; break; }
      const int off = (int)inv.size();
      inv.resize(inv.size() + fd[f].m, -1);
      for (int a = 0; a < F.u[c]; a++) {
        const int la = cmap[cmap_off[c] + a];
        if (la < 0 || la >= fd[f].m || (a > 0 && la <= cmap[cmap_off[c] + a - 1])) { extend_by_rows = false; break; }
        inv[off + la] = a;
      }
      xrows.push_back({fd[c].fm_off + (long long)fd[c].w * fd[c].m + fd[c].w, fd[c].m, cmap_off[c], off, 0});
    }
  }
  xrow_ptr[nt] = (int)xrows.size();
  // the outputs are written once, behind the last level: a row list for the tops of W, a tile list for the bottoms
  std::vector<int> wtop_items, wbot_items;
  max_ent = 0;
  for (int f : lvl_flat) {
    const FrontDesc &d = fd[f];
    max_ent = std::max(max_ent, d.ent_end - d.ent_ptr);
    for (int k0 = 0; k0 < d.w; k0 += WTOP_ROWS) { wtop_items.push_back(f); wtop_items.push_back(k0); }
  }
  {
    // (the tiles with the longest sums first: a merged root's K is its 1 000 - 2 000 pivots, a leaf's 64 - 128)
    std::vector<int> by_w(lvl_flat);
    std::stable_sort(by_w.begin(), by_w.end(), [&](int a, int b) { return fd[a].w > fd[b].w; });
    for (int f : by_w) {
      const FrontDesc &d = fd[f];
      if (d.u == 0) continue;
      const int tiles = ((d.u + TS - 1) / TS) * ((d.w + TS - 1) / TS);
      for (int t = 0; t < tiles; t++) { wbot_items.push_back(f); wbot_items.push_back(t); }
    }
  }
  n_wtop_items = (int)wtop_items.size() / 2;
  n_wbot_items = (int)wbot_items.size() / 2;
  n_aval = A.val.size();
  if (extend_by_rows) {
    FA_OK(hipMalloc((void **)&d_xrows, sizeof(ExtendRow) * std::max<size_t>(xrows.size(), 1)));
    FA_OK(hipMalloc((void **)&d_xrow_ptr, sizeof(int) * xrow_ptr.size()));
    FA_OK(hipMalloc((void **)&d_inv, sizeof(int) * std::max<size_t>(inv.size(), 1)));
  }
  FA_OK(hipMalloc((void **)&d_wtop_items, sizeof(int) * std::max<size_t>(wtop_items.size(), 2)));
  FA_OK(hipMalloc((void **)&d_wbot_items, sizeof(int) * std::max<size_t>(wbot_items.size(), 2)));
  FA_OK(hipMalloc((void **)&d_fd, sizeof(FrontDesc) * std::max(nt, 1)));
  FA_OK(hipMalloc((void **)&d_dst, sizeof(long long) * std::max<size_t>(dst.size(), 1)));
  FA_OK(hipMalloc((void **)&d_src, sizeof(int) * std::max<size_t>(src.size(), 1)));
  FA_OK(hipMalloc((void **)&d_cmap, sizeof(int) * cmap.size()));
  FA_OK(hipMalloc((void **)&d_lvl, sizeof(int) * std::max<size_t>(lvl_flat.size(), 1)));
  FA_OK(hipMalloc((void **)&d_fail, 8 + 16 * PIV_SLOTS));   // int fail; then PIV_SLOTS pairs of 64-bit words: smallest / largest pivot
  FA_OK(hipMalloc((void **)&d_aval, sizeof(double) * std::max<size_t>(n_aval, 1)));
  FA_OK(hipMalloc((void **)&d_Fm, sizeof(double) * std::max<long long>(fm_total, 1)));
  FA_OK(hipMalloc((void **)&d_dinv, sizeof(double) * max_lvl * NB * NB));
  FA_OK(hipMalloc((void **)&d_W, sizeof(double) * std::max<int64_t>(w_total, 1)));
  FA_OK(hipMalloc((void **)&d_WT, sizeof(double) * std::max<int64_t>(wt_total, 1)));
  FA_OK(hipMalloc((void **)&d_pairs, sizeof(ExtendPair) * std::max<size_t>(pairs.size(), 1)));
  FA_OK(hipStreamCreate(&st));
  FA_OK(hipMemcpyAsync(d_fd, fd.data(), sizeof(FrontDesc) * nt, hipMemcpyHostToDevice, st));
  FA_OK(hipMemcpyAsync(d_dst, dst.data(), sizeof(long long) * dst.size(), hipMemcpyHostToDevice, st));
  FA_OK(hipMemcpyAsync(d_src, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice, st));
  FA_OK(hipMemcpyAsync(d_cmap, cmap.data(), sizeof(int) * cmap.size(), hipMemcpyHostToDevice, st));
  FA_OK(hipMemcpyAsync(d_lvl, lvl_flat.data(), sizeof(int) * lvl_flat.size(), hipMemcpyHostToDevice, st));
  if (!pairs.empty()) FA_OK(hipMemcpyAsync(d_pairs, pairs.data(), sizeof(ExtendPair) * pairs.size(), hipMemcpyHostToDevice, st));
  if (extend_by_rows) {
    if (!xrows.empty()) FA_OK(hipMemcpyAsync(d_xrows, xrows.data(), sizeof(ExtendRow) * xrows.size(), hipMemcpyHostToDevice, st));
    FA_OK(hipMemcpyAsync(d_xrow_ptr, xrow_ptr.data(), sizeof(int) * xrow_ptr.size(), hipMemcpyHostToDevice, st));
    if (!inv.empty()) FA_OK(hipMemcpyAsync(d_inv, inv.data(), sizeof(int) * inv.size(), hipMemcpyHostToDevice, st));
  }
  if (!wtop_items.empty()) FA_OK(hipMemcpyAsync(d_wtop_items, wtop_items.data(), sizeof(int) * wtop_items.size(), hipMemcpyHostToDevice, st));
  if (!wbot_items.empty()) FA_OK(hipMemcpyAsync(d_wbot_items, wbot_items.data(), sizeof(int) * wbot_items.size(), hipMemcpyHostToDevice, st));
  FA_OK(hipStreamSynchronize(st));   // (the host vectors go out of scope)
  return 0;
}

// aval_host: the values of A in CSR order (nullptr: they are already in d_aval).  Leaves W / WT in d_W / d_WT.
int SpdNumericCtx::factor(SpdFactor &F, const double *aval_host, double *flops_out, double *mfma_ms_out, hipStream_t on, bool defer) {
  if (pending && finish(F) != 0) return -1;
  // (everything below is written against `st`: for the length of this call it names the stream to run on)
  struct Swap {
    hipStream_t &ref, keep;
    Swap(hipStream_t &r, hipStream_t v) : ref(r), keep(r) { if (v) ref = v; }
    ~Swap() { ref = keep; }
  } swap(st, on);
  constexpr int NIO = 1 + 2 * PIV_SLOTS;
  if (!h_io) {
    FA_OK(hipHostMalloc((void **)&h_io, sizeof(unsigned long long) * 2 * NIO, hipHostMallocDefault));
    h_io[0] = 0ull;
    for (int q = 0; q < PIV_SLOTS; q++) { h_io[1 + 2 * q] = 0x7ff0000000000000ull; h_io[2 + 2 * q] = 0ull; }
  }
  if (aval_host && n_aval) FA_OK(hipMemcpyAsync(d_aval, aval_host, sizeof(double) * n_aval, hipMemcpyHostToDevice, st));
  FA_OK(hipMemcpyAsync(d_fail, h_io, sizeof(unsigned long long) * NIO, hipMemcpyHostToDevice, st));   // (pinned, kept: no wait)
  FA_OK(hipMemsetAsync(d_Fm, 0, sizeof(double) * std::max<long long>(fm_total, 1), st));
  if (!outputs_zeroed) {
    // (what the kernels do not write -- the padding of a row, a zero triangle -- they never write: once is enough for a
    // context that is used again, Rescale::Dynamic)
    FA_OK(hipMemsetAsync(d_W, 0, sizeof(double) * std::max<int64_t>(w_total, 1), st));
    FA_OK(hipMemsetAsync(d_WT, 0, sizeof(double) * std::max<int64_t>(wt_total, 1), st));
    outputs_zeroed = true;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (mfma_ms_out) { FA_OK(hipEventCreate(&e0)); FA_OK(hipEventCreate(&e1)); }
  double flops = 0, mfma_ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  {
  // the entries of A and the ones of the identity rows, every front of every level at once (the children's Schur
  // complements are ADDED later, level by level)
  for (int f0 = 0; f0 < lvl_ptr.back() && max_ent > 0; f0 += 65535)   // (a grid's y extent ends at 65 535)
    hipLaunchKernelGGL(k_fa_scatter, dim3((max_ent + 255) / 256, std::min(65535, lvl_ptr.back() - f0)), dim3(256), 0, st, d_fd,
                       d_lvl + f0, d_dst, d_src, d_aval, d_Fm);
  for (int h = 0; h <= maxh; h++) {
    const int nf = (int)lvl[h].size();
    if (nf == 0) continue;
    const int *L = d_lvl + lvl_ptr[h];
    int max_w = 0, max_m = 0;
    for (int f : lvl[h]) {
      max_w = std::max(max_w, fd[f].w);
      max_m = std::max(max_m, fd[f].m);
    }
    if (extend_by_rows) {
      if (h > 0) hipLaunchKernelGGL(k_fa_extend_rows, dim3((max_m + 3) / 4, nf), dim3(256), 0, st, d_fd, L, d_xrow_ptr, d_xrows, d_cmap, d_inv, d_Fm);
    } else {
      for (size_t s = 0; s < pair_rng[h].size(); s++)
        if (pair_rng[h][s].second > 0)
          hipLaunchKernelGGL(k_fa_extend, dim3(pair_maxu[h][s], pair_rng[h][s].second), dim3(256), 0, st, d_fd,
                             d_pairs + pair_rng[h][s].first, d_cmap, d_Fm);
    }
    auto abt0 = [&](int k_lo, int k_hi, int sb_end, int wide) -> int {
      // upper bounds over the fronts of the level (a front's block column may end before k_hi: columns from k_lo + 1 on)
      const int nrt = (max_m + TS - 1) / TS, nct = (max_m - k_lo + TS - 1) / TS;
      if (nct <= 0) return 0;
      hipEvent_t a = nullptr, b = nullptr;
      if (mfma_ms_out) {
        FA_OK(hipEventCreate(&a)); FA_OK(hipEventCreate(&b));
        FA_OK(hipEventRecord(a, st));
      }
      hipLaunchKernelGGL((k_fa_abt<0>), dim3(nrt * nct, nf), dim3(256), 0, st, d_fd, L, k_lo, k_hi, sb_end, wide, d_Fm, d_W, d_WT);
      if (mfma_ms_out) {
        FA_OK(hipEventRecord(b, st));
        evs.push_back({a, b});
      }
      return 0;
    };
    static const bool ll_enabled = getenv("DPGO_SPD_LEFT_LOOKING") ? atoi(getenv("DPGO_SPD_LEFT_LOOKING")) != 0 : true;
    // right-looking levels: up to this many workgroups factor the diagonal block themselves, beside their rows' loads (one
    // launch per block column instead of two); beyond, one wave per front does it first (measured on the leaf level of the
    // headline's G_tt, 1 256 workgroups: 63-110 us fused against 22 + 39 us in two launches -- three workgroups per CU
    // each bring a factoring wave)
    static const long long fuse_limit = getenv("DPGO_SPD_FUSE_POTRF_WGS") ? atoll(getenv("DPGO_SPD_FUSE_POTRF_WGS")) : 768;
    // (measured in round 5 with the limit raised so that the leaf level -- 1 256 workgroups at the headline -- runs
    // left-looking too: a Dynamic iteration goes from 5.15 to 5.47 ms)
    const bool left_looking = ll_enabled && (long long)((max_m + 255) / 256) * nf <= 768;
    for (int sb = 0; sb < max_w; sb += SB) {
      const int se = sb + SB;
      for (int kb = sb; kb < std::min(se, max_w); kb += NB) {
        // few workgroups in the launch: every one factors the diagonal block itself (one launch); many: one wave per
        // front does it first (the workgroups of the panel would idle three waves of four meanwhile)
        const int bx = (max_m + 255) / 256;
        if (left_looking) {
          // few workgroups in the level: the block column takes the super-block's pending updates itself (no k_fa_abt
          // per block column); the rest of the front gets them in the wide pass below
          if (kb == sb)
            hipLaunchKernelGGL(k_fa_panel_ll<true>, dim3((max_m + 127) / 128, nf), dim3(320), 0, st, d_fd, L, sb, kb, d_Fm, d_fail,
                               reinterpret_cast<unsigned long long *>(d_fail) + 1);
          else
            hipLaunchKernelGGL(k_fa_panel_ll<false>, dim3((max_m + 127) / 128, nf), dim3(320), 0, st, d_fd, L, sb, kb, d_Fm, d_fail,
                               reinterpret_cast<unsigned long long *>(d_fail) + 1);
          continue;
        }
        if ((long long)bx * nf <= fuse_limit) {
          hipLaunchKernelGGL((k_fa_potrf_panel<true>), dim3(bx, nf), dim3(320), 0, st, d_fd, L, kb, d_Fm, d_dinv, d_fail,
                             reinterpret_cast<unsigned long long *>(d_fail) + 1);
        } else {
          hipLaunchKernelGGL(k_fa_potrf_reg, dim3(nf), dim3(64), 0, st, d_fd, L, kb, d_Fm, d_dinv, d_fail,
                             reinterpret_cast<unsigned long long *>(d_fail) + 1);
          hipLaunchKernelGGL((k_fa_potrf_panel<false>), dim3(bx, nf), dim3(256), 0, st, d_fd, L, kb, d_Fm, d_dinv, d_fail,
                             reinterpret_cast<unsigned long long *>(d_fail) + 1);
        }
        if (abt0(kb, kb + NB, se, 0) != 0) return -1;      // the rest of the super-block (of the front, if it ends here), K = 32
      }
      if (left_looking) { if (abt0(sb, se, se, 2) != 0) return -1; }   // ... of every front with pivots in the super-block
      else if (se < max_w && abt0(sb, se, se, 1) != 0) return -1;   // everything right of the super-block, K = 128
    }
    // the products the MFMA kernel carries for this level (useful flops: lower triangle of the trailing update)
    for (int f : lvl[h]) {
      const double w = fd[f].w, u = fd[f].u;
      flops += w * w * w / 3.0 + u * w * w + u * u * w      // trailing updates of regular rows (2 flops per multiply-add, half by symmetry)
               + w * w * w / 3.0                              // identity rows
               + 2.0 * u * w * w / 2.0;                       // W_bottom (triangular operand)
      if (left_looking) {
        // ... less what the left-looking block columns carry themselves (k_fa_panel_ll: the strip's updates from the
        // block columns of its super-block; not in the time of the tile kernel either)
        for (int kb = 0; kb < fd[f].w; kb += NB) {
          const double nbk = std::min(NB, fd[f].w - kb), K = kb % SB;
          flops -= 2.0 * K * nbk * fd[f].m;   // rows [kb, m) and the identity rows [m, m + ke): ~m of them, K x nb products each
        }
      }
    }
  }
  // the outputs: W = [L11^-1 ; -L21 L11^-1] and its transpose, of every front at once
  if (n_wtop_items > 0) hipLaunchKernelGGL(k_fa_wtop, dim3(n_wtop_items), dim3(256), 0, st, d_fd, d_wtop_items, d_Fm, d_W, d_WT);
  if (n_wbot_items > 0) {
    hipEvent_t a = nullptr, b = nullptr;
    if (mfma_ms_out) {
      FA_OK(hipEventCreate(&a)); FA_OK(hipEventCreate(&b));
      FA_OK(hipEventRecord(a, st));
    }
    hipLaunchKernelGGL((k_fa_abt<1>), dim3(n_wbot_items), dim3(256), 0, st, d_fd, d_wbot_items, 0, 0, 0, 0, d_Fm, d_W, d_WT);
    if (mfma_ms_out) {
      FA_OK(hipEventRecord(b, st));
      evs.push_back({a, b});
    }
  }
  }
  FA_OK(hipMemcpyAsync(h_io + NIO, d_fail, sizeof(unsigned long long) * NIO, hipMemcpyDeviceToHost, st));
  pending = true;
  pending_on = st;
  if (defer && !mfma_ms_out) {
    if (flops_out) *flops_out = flops;
    return 0;
  }
  if (finish(F) != 0) return -1;
  for (auto &ev : evs) {
    float t = 0;
    (void)hipEventElapsedTime(&t, ev.first, ev.second);
    mfma_ms += t;
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
  if (flops_out) *flops_out = flops;
  if (mfma_ms_out) *mfma_ms_out = mfma_ms;
  return 0;
}

// waits for the factorisation enqueued by factor(..., defer) and reads its verdict: 0, or -1 on a non-positive pivot
int SpdNumericCtx::finish(SpdFactor &F, bool wait) {
  if (!pending) return 0;
  pending = false;
  constexpr int NIO = 1 + 2 * PIV_SLOTS;
  if (wait) FA_OK(hipStreamSynchronize(pending_on));
  const unsigned long long *back = h_io + NIO;
  const int fail = (int)(back[0] & 0xffffffffull);
  {
    double lo = 1e300, hi = 0.0;
    for (int q = 0; q < PIV_SLOTS; q++) {
      double a, b;
      memcpy(&a, &back[1 + 2 * q], 8);
      memcpy(&b, &back[2 + 2 * q], 8);
      lo = std::min(lo, a);   // (an unused slot still holds +inf / 0)
      hi = std::max(hi, b);
    }
    F.pivot_min = lo;
    F.pivot_max = hi;
  }
  FA_OK(hipGetLastError());
  if (fail) {
    fprintf(stderr, "[dpgo_amd] ERROR: spd_factor (device): non-positive pivot in front %d\n", fail - 1);
    return -1;
  }
  return 0;
}

// children[f]: the fronts whose update rows are assembled into f.  Fills F.W and F.WT (host vectors), or -- keep_device --
// leaves them on the GPU as dev_W / dev_WT.  keep_numeric: the context stays with the factor (F.numeric) and dev_W / dev_WT
// are BORROWED from it (dev_borrowed; spd_release_device then only forgets them).
// flops (optional): floating-point operations of the MFMA kernel; mfma_ms: time spent in it.
int spd_factor_numeric_device(const CsrMatrix &A, SpdFactor &F, const std::vector<std::vector<int>> &children,
                              double *flops_out, double *mfma_ms_out) {
  spd_release_device(F);
  SpdNumericCtx *ctx = F.numeric;
  if (ctx && (ctx->nt != F.nfronts || ctx->n_aval != A.val.size())) { spd_release_numeric(F); ctx = nullptr; }
  if (!ctx) {
    ctx = new SpdNumericCtx();
    if (ctx->build(A, F, children) != 0) { delete ctx; return -1; }
  }
  const bool kept = F.keep_numeric && F.keep_device;
  if (kept) F.numeric = ctx;
  int rc = ctx->factor(F, A.val.data(), flops_out, mfma_ms_out);
  if (rc == 0) {
    const int nt = F.nfronts;
    if (kept) {
      F.dev_W = ctx->d_W;
      F.dev_WT = ctx->d_WT;
      F.dev_borrowed = true;
      std::vector<double>().swap(F.W);
      std::vector<double>().swap(F.WT);
    } else if (F.keep_device) {   // the caller packs its solve panels from the device copies (Group: SpdSolverDev::upload)
      F.dev_W = ctx->d_W;
      F.dev_WT = ctx->d_WT;
      F.dev_borrowed = false;
      ctx->d_W = ctx->d_WT = nullptr;
      std::vector<double>().swap(F.W);
      std::vector<double>().swap(F.WT);
    } else {
      F.W.assign(F.w_off[nt], 0.0);
      F.WT.assign(F.wt_off[nt], 0.0);
      if (F.w_off[nt] > 0 && hipMemcpy(F.W.data(), ctx->d_W, sizeof(double) * F.w_off[nt], hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
      if (F.wt_off[nt] > 0 && hipMemcpy(F.WT.data(), ctx->d_WT, sizeof(double) * F.wt_off[nt], hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    }
  }
  if (!kept) {
    if (F.numeric == ctx) F.numeric = nullptr;
    delete ctx;
  }
  return rc;
}

double *spd_numeric_values(SpdFactor &F) { return F.numeric ? F.numeric->d_aval : nullptr; }

// the numeric phase again, from the values in spd_numeric_values(F); dev_W / dev_WT (borrowed) hold the new factor
int spd_refactor_device(SpdFactor &F, void *stream, bool defer) {
  if (!F.numeric) return -1;
  if (F.numeric->factor(F, nullptr, nullptr, nullptr, (hipStream_t)stream, defer) != 0) return -1;
  F.dev_W = F.numeric->d_W;
  F.dev_WT = F.numeric->d_WT;
  F.dev_borrowed = true;
  return 0;
}
int spd_refactor_finish(SpdFactor &F, bool wait) { return F.numeric ? F.numeric->finish(F, wait) : -1; }

void spd_release_numeric(SpdFactor &F) {
  if (F.dev_borrowed) { F.dev_W = F.dev_WT = nullptr; F.dev_borrowed = false; }
  delete F.numeric;
  F.numeric = nullptr;
}

void spd_release_device(SpdFactor &F) {
  if (!F.dev_borrowed) {
    if (F.dev_W) (void)hipFree(F.dev_W);
    if (F.dev_WT) (void)hipFree(F.dev_WT);
  }
  F.dev_W = F.dev_WT = nullptr;
  F.dev_borrowed = false;
}

}  // namespace dpgo
