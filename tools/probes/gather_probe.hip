// Probe: what a wave pays for gathering 64 records of 96 bytes (12 doubles, a pose record) by per-lane index,
//  (a) lane-major: every lane loads the six 16-byte pieces of ITS record (six wave-instructions, each touching 64 records);
//  (b) record-major: instruction t loads pieces (t * 64 + lane) -- ten records' contiguous bytes per instruction -- into LDS,
//      and every lane then reads its record back from LDS.
// The table is L2 / Infinity-Cache resident (lattice-like locality: index = row + small offsets), as x is for k_bsr.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe tools/probes/gather_probe.hip && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const double *x, const int *idx, int per_row, int nrows, double *out) {
  __shared__ double2 stage[4][64 * 6 + 8];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 256 + threadIdx.x;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  for (int it = 0; it < per_row; it++) {
    const int q = row < nrows ? idx[(size_t)it * nrows + row] : 0;
    double2 r[6];
    if (MODE == 0) {
      const double2 *p = reinterpret_cast<const double2 *>(x + (size_t)q * 12);
#pragma unroll
      for (int c = 0; c < 6; c++) r[c] = p[c];
    } else {
#pragma unroll
      for (int t = 0; t < 6; t++) {
        const int piece = t * 64 + lane, rr = piece / 6, c = piece - rr * 6;
        const int qq = __shfl(q, rr, 64);
        stage[wv][piece] = reinterpret_cast<const double2 *>(x + (size_t)qq * 12)[c];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 6; c++) r[c] = stage[wv][lane * 6 + c];
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < 6; c++) { acc[2 * c] += r[c].x; acc[2 * c + 1] += r[c].y; }
  }
  if (row < nrows) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) s += acc[k];
    out[row] = s;
  }
}

int main() {
  const int nrows = 100000, per_row = 12;
  std::vector<double> hx((size_t)nrows * 12);
  for (size_t i = 0; i < hx.size(); i++) hx[i] = (double)(i % 97);
  std::vector<int> hidx((size_t)per_row * nrows);
  srand(1);
  const int offs[12] = {0, 1, -1, 50, -50, 2500, -2500, 51, -49, 2, 100, -2};
  for (int it = 0; it < per_row; it++)
    for (int r = 0; r < nrows; r++) { int q = r + offs[it]; if (q < 0) q += nrows; if (q >= nrows) q -= nrows; hidx[(size_t)it * nrows + r] = q; }
  double *x, *out; int *idx;
  CK(hipMalloc(&x, hx.size() * 8)); CK(hipMalloc(&out, nrows * 8)); CK(hipMalloc(&idx, hidx.size() * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> ref(nrows), got(nrows);
  for (int mode = 0; mode < 2; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; rep++) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3((nrows + 255) / 256), dim3(256), 0, 0, x, idx, per_row, nrows, out);
      else hipLaunchKernelGGL(k_gather<1>, dim3((nrows + 255) / 256), dim3(256), 0, 0, x, idx, per_row, nrows, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 2 && ms < best) best = ms;
    }
    CK(hipMemcpy((mode == 0 ? ref : got).data(), out, nrows * 8, hipMemcpyDeviceToHost));
    printf("mode %d (%s): %.2f us for %d x %d records of 96 B = %.1f MB gathered (%.2f TB/s of gathered bytes)\n", mode,
           mode == 0 ? "lane-major" : "record-major through LDS", best * 1e3, nrows, per_row, nrows * per_row * 96e-6,
           nrows * per_row * 96.0 / (best * 1e-3) / 1e12);
  }
  int bad = 0;
  for (int r = 0; r < nrows; r++) bad += ref[r] != got[r];
  printf("results %s\n", bad ? "DIFFER" : "agree");
  return 0;
}
