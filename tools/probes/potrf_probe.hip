// Where the time of the 32 x 32 diagonal-block kernel goes (k_fa_potrf_reg / potrf_inv_wave of spd_dev.hip): the same code
// in variants -- PH1 only, PH2 only, fast reciprocal square root, no loads -- timed with wall_clock64 inside one wave and
// with HIP events over back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/potrf_probe.hip -o /tmp/potrf_probe && /tmp/potrf_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
constexpr int NB = 32;
__device__ __forceinline__ double bcast(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// the value the lane with the same (lane & 31) in half `owner_hi` holds, in every lane: two v_permlane32_swap (gfx950)
// instead of a trip through the LDS crossbar (__shfl_xor(x, 32))
__device__ __forceinline__ double from_half(double x, int owner_hi) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)(b & 0xffffffffll), hi = (unsigned)(b >> 32);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const unsigned l = owner_hi ? r0[1] : r0[0], h = owner_hi ? r1[1] : r1[0];
  return __longlong_as_double(((long long)h << 32) | l);
}
template <int VAR>
__global__ __launch_bounds__(64) void k_potrf(const double *A, int m, double *out, unsigned long long *stamps) {
  __shared__ double Ls[NB][NB + 1];
  const int lane = threadIdx.x;
  const double *Ab = A + (size_t)blockIdx.x * m * m;
  unsigned long long t0 = wall_clock64();
  double L[NB], X[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) L[j] = (lane < NB && j <= lane) ? Ab[(long long)lane * m + j] : ((lane == j) ? 1.0 : 0.0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = wall_clock64();
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const double dkk = bcast(L[k], k);
    double lkk, inv;
    if (VAR == 1) { inv = __builtin_amdgcn_rsq(dkk); inv = inv * fma(-0.5 * dkk * inv, inv, 1.5); inv = inv * fma(-0.5 * dkk * inv, inv, 1.5); lkk = dkk * inv; }
    else { lkk = sqrt(dkk); inv = 1.0 / lkk; }
    L[k] = lane == k ? lkk : L[k] * inv;
    if (lane < NB) Ls[lane][k] = L[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = k + 1; j < NB; j++) L[j] = fma(-L[k], Ls[j][k], L[j]);
  }
  unsigned long long t2 = wall_clock64();
#pragma unroll
  for (int i = 0; i < NB; i++) {
    double sum = lane == i ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; k++) sum = fma(-Ls[i][k], X[k], sum);
    if (VAR == 2) X[i] = sum * Ls[i][i]; else X[i] = sum / Ls[i][i];
  }
  unsigned long long t3 = wall_clock64();
  if (lane < NB) {
#pragma unroll
    for (int i = 0; i < NB; i++) out[(size_t)blockIdx.x * NB * NB + i * NB + lane] = X[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t4 = wall_clock64();
  if (lane == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = t4 - t3; }
}

// VAR 3: both halves of the wave work -- lane = (row r = lane & 31, half h = lane >> 5); half h holds the columns j of its
// row with j % 2 == h (16 registers), so every elimination step issues half the multiply-adds; 1 / l_kk from v_rsq_f64 + two
// Newton steps; the inversion splits its sums over the halves the same way and multiplies by the kept reciprocals.
__global__ __launch_bounds__(64) void k_potrf3(const double *A, int m, double *out, unsigned long long *stamps) {
  __shared__ double Ls[NB][NB + 2];   // [.][NB]: 1 / L[k][k]
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const double *Ab = A + (size_t)blockIdx.x * m * m;
  unsigned long long t0 = wall_clock64();
  double L[NB / 2];
#pragma unroll
  for (int q = 0; q < NB / 2; q++) { const int j = 2 * q + h; L[q] = j <= r ? Ab[(long long)r * m + j] : 0.0; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = wall_clock64();
#pragma unroll
  for (int k = 0; k < NB; k++) {
    // column k lives in half k & 1, register k >> 1; its diagonal entry in lane k + 32 (k & 1)
    const double dkk = bcast(L[k >> 1], k + 32 * (k & 1));
    double inv = __builtin_amdgcn_rsq(dkk);
    inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
    inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
    const double lkk = dkk * inv;
    if (h == (k & 1)) {
      const double v = r == k ? lkk : L[k >> 1] * inv;
      L[k >> 1] = v;
      Ls[r][k] = v;
      if (r == 0) Ls[k][NB] = inv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double lrk = Ls[r][k];   // this row's entry of column k (held by the other half for every other k)
#pragma unroll
    for (int q = (k + 1) >> 1; q < NB / 2; q++) {
      const int j = 2 * q + h;
      if (j > k) L[q] = fma(-lrk, Ls[j][k], L[q]);
    }
  }
  unsigned long long t2 = wall_clock64();
  // X = L^-1, lane (c = r, h): column c; half h keeps X[k][c] for k % 2 == h and sums over those k, the halves meet in a swap
  double Xh[NB / 2];
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
  unsigned long long t3 = wall_clock64();
#pragma unroll
  for (int q = 0; q < NB / 2; q++) out[(size_t)blockIdx.x * NB * NB + (2 * q + h) * NB + r] = Xh[q];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t4 = wall_clock64();
  if (lane == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = t4 - t3; }
}


// VAR 4: TWO waves.  Wave 0 factors as VAR 3; wave 1 carries the identity rows through the same column operations one
// step behind it ([A; I] L^-T = [L; L^-T]: lane (i, h) of wave 1 ends with Y[i][j] = X[j][i], j % 2 == h -- the lay-out
// VAR 3's inversion returns), reading the scaled column and 1 / l_kk that wave 0 publishes in LDS and a step counter
// behind them: the inversion's 32 dependent steps run beside the factorisation instead of after it.
__global__ __launch_bounds__(128) void k_potrf4(const double *A, int m, double *out, unsigned long long *stamps) {
  __shared__ double Ls[NB][NB + 2];   // [.][NB]: 1 / L[k][k]
  __shared__ volatile int step;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
  const double *Ab = A + (size_t)blockIdx.x * m * m;
  if (t == 0) step = 0;
  __syncthreads();
  unsigned long long t0 = wall_clock64(), t1 = t0, t2 = t0, t3 = t0;
  if (wv == 0) {
    double L[NB / 2];
#pragma unroll
    for (int q = 0; q < NB / 2; q++) { const int j = 2 * q + h; L[q] = j <= r ? Ab[(long long)r * m + j] : 0.0; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = wall_clock64();
#pragma unroll
    for (int k = 0; k < NB; k++) {
      const double dkk = bcast(L[k >> 1], k + 32 * (k & 1));
      double inv = __builtin_amdgcn_rsq(dkk);
      inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
      inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
      const double lkk = dkk * inv;
      if (h == (k & 1)) {
        const double v = r == k ? lkk : L[k >> 1] * inv;
        L[k >> 1] = v;
        Ls[r][k] = v;
        if (r == 0) Ls[k][NB] = inv;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) step = k + 1;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double lrk = Ls[r][k];
#pragma unroll
      for (int q = (k + 1) >> 1; q < NB / 2; q++) {
        const int j = 2 * q + h;
        if (j > k) L[q] = fma(-lrk, Ls[j][k], L[q]);
      }
    }
    t2 = wall_clock64();
  } else {
    double Y[NB / 2];
#pragma unroll
    for (int q = 0; q < NB / 2; q++) Y[q] = (2 * q + h == r) ? 1.0 : 0.0;
    t1 = wall_clock64();
#pragma unroll
    for (int k = 0; k < NB; k++) {
      while (step <= k) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double inv = Ls[k][NB];
      // Y[i][k] lives in half k & 1; both halves need the scaled value
      double yk = Y[k >> 1] * inv;
      const double other = __shfl_xor(yk, 32, 64);
      if (h == (k & 1)) Y[k >> 1] = yk; else yk = other;
#pragma unroll
      for (int q = (k + 1) >> 1; q < NB / 2; q++) {
        const int j = 2 * q + h;
        if (j > k) Y[q] = fma(-yk, Ls[j][k], Y[q]);
      }
    }
    t2 = wall_clock64();
#pragma unroll
    for (int q = 0; q < NB / 2; q++) out[(size_t)blockIdx.x * NB * NB + (2 * q + h) * NB + r] = Y[q];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t3 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = 0; stamps[3] = t3 - t2; }
  }
}

// VAR 5: VAR 4 with v_permlane32_swap for the exchange between the halves
__global__ __launch_bounds__(128) void k_potrf5(const double *A, int m, double *out, unsigned long long *stamps) {
  __shared__ double Ls[NB][NB + 2];   // [.][NB]: 1 / L[k][k]
  __shared__ volatile int step;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
  const double *Ab = A + (size_t)blockIdx.x * m * m;
  if (t == 0) step = 0;
  __syncthreads();
  unsigned long long t0 = wall_clock64(), t1 = t0, t2 = t0, t3 = t0;
  if (wv == 0) {
    double L[NB / 2];
#pragma unroll
    for (int q = 0; q < NB / 2; q++) { const int j = 2 * q + h; L[q] = j <= r ? Ab[(long long)r * m + j] : 0.0; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = wall_clock64();
#pragma unroll
    for (int k = 0; k < NB; k++) {
      const double dkk = bcast(L[k >> 1], k + 32 * (k & 1));
      double inv = __builtin_amdgcn_rsq(dkk);
      inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
      inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
      const double lkk = dkk * inv;
      if (h == (k & 1)) {
        const double v = r == k ? lkk : L[k >> 1] * inv;
        L[k >> 1] = v;
        Ls[r][k] = v;
        if (r == 0) Ls[k][NB] = inv;
      }
      // (the counter goes out right behind the column, in the same in-order LDS queue: no wait of its own on wave 0's chain)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (lane == 0) step = k + 1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const double lrk = Ls[r][k];
#pragma unroll
      for (int q = (k + 1) >> 1; q < NB / 2; q++) {
        const int j = 2 * q + h;
        if (j > k) L[q] = fma(-lrk, Ls[j][k], L[q]);
      }
    }
    t2 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) { stamps[4] = t1 - t0; stamps[5] = t2 - t1; }
  } else {
    double Y[NB / 2];
#pragma unroll
    for (int q = 0; q < NB / 2; q++) Y[q] = (2 * q + h == r) ? 1.0 : 0.0;
    int seen = 0;
    t1 = wall_clock64();
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (seen <= k) {   // (behind wave 0: what it published up to `seen` is visible since the fence that followed that read)
        do { seen = step; } while (seen <= k);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      const double inv = Ls[k][NB];
      // Y[i][k] lives in half k & 1; both halves need the scaled value
      const double mine = Y[k >> 1] * inv;
      if (h == (k & 1)) Y[k >> 1] = mine;
      const double yk = from_half(mine, k & 1);
#pragma unroll
      for (int q = (k + 1) >> 1; q < NB / 2; q++) {
        const int j = 2 * q + h;
        if (j > k) Y[q] = fma(-yk, Ls[j][k], Y[q]);
      }
    }
    t2 = wall_clock64();
#pragma unroll
    for (int q = 0; q < NB / 2; q++) out[(size_t)blockIdx.x * NB * NB + (2 * q + h) * NB + r] = Y[q];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t3 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = 0; stamps[3] = t3 - t2; }
  }
}

// VAR 6: VAR 3 with v_permlane32_swap where the halves' sums meet
__global__ __launch_bounds__(64) void k_potrf6(const double *A, int m, double *out, unsigned long long *stamps) {
  __shared__ double Ls[NB][NB + 2];   // [.][NB]: 1 / L[k][k]
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const double *Ab = A + (size_t)blockIdx.x * m * m;
  unsigned long long t0 = wall_clock64();
  double L[NB / 2];
#pragma unroll
  for (int q = 0; q < NB / 2; q++) { const int j = 2 * q + h; L[q] = j <= r ? Ab[(long long)r * m + j] : 0.0; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = wall_clock64();
#pragma unroll
  for (int k = 0; k < NB; k++) {
    // column k lives in half k & 1, register k >> 1; its diagonal entry in lane k + 32 (k & 1)
    const double dkk = bcast(L[k >> 1], k + 32 * (k & 1));
    double inv = __builtin_amdgcn_rsq(dkk);
    inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
    inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
    const double lkk = dkk * inv;
    if (h == (k & 1)) {
      const double v = r == k ? lkk : L[k >> 1] * inv;
      L[k >> 1] = v;
      Ls[r][k] = v;
      if (r == 0) Ls[k][NB] = inv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double lrk = Ls[r][k];   // this row's entry of column k (held by the other half for every other k)
#pragma unroll
    for (int q = (k + 1) >> 1; q < NB / 2; q++) {
      const int j = 2 * q + h;
      if (j > k) L[q] = fma(-lrk, Ls[j][k], L[q]);
    }
  }
  unsigned long long t2 = wall_clock64();
  // X = L^-1, lane (c = r, h): column c; half h keeps X[k][c] for k % 2 == h and sums over those k, the halves meet in a swap
  double Xh[NB / 2];
#pragma unroll
  for (int i = 0; i < NB; i++) {
    double sum = (r == i && h == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < (i + 1) / 2; q++) {
      const int k = 2 * q + h;
      const double t = fma(-Ls[i][k], Xh[q], sum);
      sum = k < i ? t : sum;
    }
    sum = from_half(sum, 0) + from_half(sum, 1);
    const double x = sum * Ls[i][NB];
    if ((i & 1) == h) Xh[i >> 1] = x;
  }
  unsigned long long t3 = wall_clock64();
#pragma unroll
  for (int q = 0; q < NB / 2; q++) out[(size_t)blockIdx.x * NB * NB + (2 * q + h) * NB + r] = Xh[q];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t4 = wall_clock64();
  if (lane == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = t4 - t3; }
}

template <int VAR>
void run(const char *name, const double *dA, int m, int nf, double *dout, unsigned long long *dst) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(a);
    for (int i = 0; i < 200; i++) {
      if (VAR == 5) hipLaunchKernelGGL(k_potrf5, dim3(nf), dim3(128), 0, 0, dA, m, dout, dst);
      else if (VAR == 6) hipLaunchKernelGGL(k_potrf6, dim3(nf), dim3(64), 0, 0, dA, m, dout, dst);
      else if (VAR == 4) hipLaunchKernelGGL(k_potrf4, dim3(nf), dim3(128), 0, 0, dA, m, dout, dst);
      else if (VAR == 3) hipLaunchKernelGGL(k_potrf3, dim3(nf), dim3(64), 0, 0, dA, m, dout, dst);
      else hipLaunchKernelGGL(k_potrf<VAR>, dim3(nf), dim3(64), 0, 0, dA, m, dout, dst);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  unsigned long long st[6]; hipMemcpy(st, dst, 48, hipMemcpyDeviceToHost);
  if (VAR == 5) printf("  [wave 0: load %.2f cholesky %.2f]", st[4] / 100.0, st[5] / 100.0);
  {   // against the host's inverse of the Cholesky factor of front 0
    std::vector<double> X(NB * NB); hipMemcpy(X.data(), dout, NB * NB * 8, hipMemcpyDeviceToHost);
    extern std::vector<double> g_ref; double err = 0, big = 0;
    for (int i = 0; i < NB * NB; i++) { err = std::max(err, std::fabs(X[i] - g_ref[i])); big = std::max(big, std::fabs(g_ref[i])); }
    printf("  max |X - X_host| / max |X_host| = %.2e  ", err / big);
  }
  printf("%-28s fronts %4d: %6.2f us per launch; inside one wave (100 MHz ticks -> us): load %.2f  cholesky %.2f  inverse %.2f  store %.2f\n",
         name, nf, best * 1e3 / 200, st[0] / 100.0, st[1] / 100.0, st[2] / 100.0, st[3] / 100.0);
}
std::vector<double> g_ref;
int main() {
  const int m = 400, nf = 628;
  std::vector<double> A((size_t)nf * m * m, 0.0);
  for (int f = 0; f < nf; f++)
    for (int i = 0; i < NB; i++)
      for (int j = 0; j <= i; j++) A[(size_t)f * m * m + (size_t)i * m + j] = (i == j ? 40.0 + i : 1.0 / (1 + i + j));
  {   // host reference in long double: L, then X = L^-1 stored as X[i][c] at [i * NB + c]
    std::vector<long double> Lh(NB * NB, 0.0L), Xh(NB * NB, 0.0L);
    for (int i = 0; i < NB; i++) for (int j = 0; j <= i; j++) Lh[i * NB + j] = A[(size_t)i * m + j];
    for (int k = 0; k < NB; k++) {
      Lh[k * NB + k] = sqrtl(Lh[k * NB + k]);
      for (int i = k + 1; i < NB; i++) Lh[i * NB + k] /= Lh[k * NB + k];
      for (int j = k + 1; j < NB; j++) for (int i = j; i < NB; i++) Lh[i * NB + j] -= Lh[i * NB + k] * Lh[j * NB + k];
    }
    for (int c = 0; c < NB; c++) for (int i = 0; i < NB; i++) {
      long double sacc = (i == c) ? 1.0L : 0.0L;
      for (int k = 0; k < i; k++) sacc -= Lh[i * NB + k] * Xh[k * NB + c];
      Xh[i * NB + c] = sacc / Lh[i * NB + i];
    }
    g_ref.resize(NB * NB);
    for (int i = 0; i < NB * NB; i++) g_ref[i] = (double)Xh[i];
  }
  double *dA, *dout; unsigned long long *dst;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dout, (size_t)nf * NB * NB * 8); hipMalloc(&dst, 64);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  for (int n : {1, 8, 64, 628}) {
    run<0>("as shipped", dA, m, n, dout, dst);
    run<1>("rsq + 2 Newton steps", dA, m, n, dout, dst);
    run<2>("inverse without divisions", dA, m, n, dout, dst);
    run<3>("both half-waves, rsq", dA, m, n, dout, dst);
    run<4>("two waves, inverse beside", dA, m, n, dout, dst);
    run<5>("two waves, permlane32", dA, m, n, dout, dst);
    run<6>("one wave, permlane32", dA, m, n, dout, dst);
  }
  {   // the same two kernels alternating: does a long straight-line kernel pay for a cold instruction cache?
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int n : {8, 628}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        for (int i = 0; i < 100; i++) {
          hipLaunchKernelGGL(k_potrf3, dim3(n), dim3(64), 0, 0, dA, m, dout, dst);
          hipLaunchKernelGGL(k_potrf<0>, dim3(n), dim3(64), 0, 0, dA, m, dout, dst);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
      }
      printf("alternating half-wave / as-shipped, fronts %4d: %6.2f us per pair\n", n, best * 1e3 / 100);
    }
  }
  return 0;
}
