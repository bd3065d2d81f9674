// HBM read ceiling for the access shapes of the solve kernels: every workgroup (512 threads) sums one contiguous
// slab of doubles; 8 or 16 bytes per lane per load, plain or non-temporal, 16 loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VEC, bool NT>
__global__ __launch_bounds__(512) void k_sum(const double *x, size_t slab, double *out) {
  const double *p = x + (size_t)blockIdx.x * slab;
  double acc = 0;
  if constexpr (VEC == 1) {
    for (size_t i = threadIdx.x; i + 15 * 512 < slab; i += 16 * 512) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = NT ? __builtin_nontemporal_load(p + i + q * 512) : p[i + q * 512];
#pragma unroll
      for (int q = 0; q < 16; q++) acc += v[q];
    }
  } else {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d *p2 = reinterpret_cast<const v2d *>(p);
    const size_t n2 = slab / 2;
    for (size_t i = threadIdx.x; i + 7 * 512 < n2; i += 8 * 512) {
      v2d v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = NT ? __builtin_nontemporal_load(p2 + i + q * 512) : p2[i + q * 512];
#pragma unroll
      for (int q = 0; q < 8; q++) acc += v[q].x + v[q].y;
    }
  }
  if (acc == 12345.678) out[blockIdx.x] = acc;   // keep the loads
}

template <int VEC, bool NT>
void run(const char *name, const double *x, size_t n, int blocks, double *out) {
  const size_t slab = n / blocks / 8192 * 8192;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_sum<VEC, NT>), dim3(blocks), dim3(512), 0, 0, x, slab, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep) best = ms < best ? ms : best;
  }
  printf("%-28s blocks %5d slab %7.1f KB: %7.1f us  %6.0f GB/s\n", name, blocks, slab * 8 / 1024.0, best * 1e3,
         (double)slab * blocks * 8 / (best * 1e-3) / 1e9);
}

int main() {
  const size_t n = (size_t)256 << 20;   // 2 GiB of doubles
  double *x, *out;
  hipMalloc(&x, n * 8);
  hipMalloc(&out, 1 << 20);
  hipMemset(x, 0, n * 8);
  for (int blocks : {512, 768, 1024, 2048, 8192}) {
    run<1, false>("8 B/lane plain", x, n, blocks, out);
    run<1, true>("8 B/lane non-temporal", x, n, blocks, out);
    run<2, false>("16 B/lane plain", x, n, blocks, out);
    run<2, true>("16 B/lane non-temporal", x, n, blocks, out);
  }
  return 0;
}
