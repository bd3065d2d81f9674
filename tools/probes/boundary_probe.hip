// The DEPENDENT KERNEL BOUNDARY of one stream, measured on the GPU's own clock (the 2.9-3.1 us of
// barrier_probe.hip are back-to-back submissions of an EMPTY kernel, i.e. the host's launch rate).  Here a chain of N
// launches of a streaming kernel (G workgroups x 512 threads, each summing `per_wg` bytes: 5-20 us per launch) runs in one
// stream; every workgroup stamps the 100 MHz wall clock when it starts and when its last store is acknowledged, and a
// per-launch min(start) / max(end) gives
//     gap_i = first wave of launch i+1  -  last wave of launch i        (the boundary proper)
//     span_i = last end - first start of launch i                      (what the kernel itself takes)
// next to the host's view: (event time of the chain) / N.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/boundary_probe.hip -o /tmp/boundary_probe && /tmp/boundary_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int MAXG = 3072;
__global__ __launch_bounds__(512) void k_stream(const double *src, size_t per_wg_doubles, double *sink, unsigned long long *stamps,
                                                int launch) {
  const unsigned long long t0 = wall_clock64();
  const double *p = src + (size_t)blockIdx.x * per_wg_doubles;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < per_wg_doubles; i += 512 * 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = i + (size_t)q * 512 < per_wg_doubles ? __builtin_nontemporal_load(p + i + (size_t)q * 512) : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) acc += v[q];
  }
  sink[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  // (one slot per workgroup: same-address atomics from every workgroup would themselves take microseconds)
  if (threadIdx.x == 0) {
    stamps[((size_t)launch * MAXG + blockIdx.x) * 2] = t0;
    stamps[((size_t)launch * MAXG + blockIdx.x) * 2 + 1] = t1;
  }
}
__global__ void k_tiny(double *sink, unsigned long long *stamps, int launch) {
  const unsigned long long t0 = wall_clock64();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = 1.0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    stamps[((size_t)launch * MAXG + blockIdx.x) * 2] = t0;
    stamps[((size_t)launch * MAXG + blockIdx.x) * 2 + 1] = t1;
  }
}

int main() {
  const int N = 200;
  const size_t pool = (size_t)2 << 30;
  double *src, *sink;
  unsigned long long *stamps;
  hipMalloc(&src, pool);
  hipMemset(src, 0, pool);
  hipMalloc(&sink, 8192 * 512 * 8);
  hipMalloc(&stamps, (size_t)N * MAXG * 16);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  std::vector<unsigned long long> h(2 * N), raw((size_t)N * MAXG * 2);
  int curG = 0;
  auto report = [&](const char *what, float ms) {
    hipMemcpy(raw.data(), stamps, raw.size() * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; i++) {
      unsigned long long lo = ~0ull, hi = 0;
      for (int g = 0; g < curG; g++) {
        lo = std::min(lo, raw[((size_t)i * MAXG + g) * 2]);
        hi = std::max(hi, raw[((size_t)i * MAXG + g) * 2 + 1]);
      }
      h[2 * i] = lo; h[2 * i + 1] = hi;
    }
    std::vector<double> gap, span;
    for (int i = 0; i < N; i++) span.push_back((double)(h[2 * i + 1] - h[2 * i]) * 0.01);
    for (int i = 0; i + 1 < N; i++) gap.push_back(((double)h[2 * i + 2] - (double)h[2 * i + 1]) * 0.01);
    std::sort(gap.begin(), gap.end());
    std::sort(span.begin(), span.end());
    printf("%-46s host %6.2f us/launch | kernel span median %6.2f | boundary gap p10 %5.2f median %5.2f p90 %5.2f us\n", what,
           ms * 1e3 / N, span[N / 2], gap[gap.size() / 10], gap[gap.size() / 2], gap[gap.size() * 9 / 10]);
  };
  auto reset = [&] {};
  for (int G : {256, 768, 3072})
    for (double mb : {8.0, 32.0, 128.0}) {
      curG = G;
      const size_t per = (size_t)(mb * 1e6 / 8 / G) & ~(size_t)511;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        reset();
        hipEventRecord(a);
        for (int i = 0; i < N; i++)
          hipLaunchKernelGGL(k_stream, dim3(G), dim3(512), 0, 0, src + ((size_t)i % 8) * (pool / 8 / 8), per, sink, stamps, i);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
      }
      char what[96];
      snprintf(what, sizeof what, "stream %5.0f MB on %4d workgroups", mb, G);
      report(what, best);
    }
  for (int G : {49, 256, 2048}) {
    curG = G;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      reset();
      hipEventRecord(a);
      for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(G), dim3(256), 0, 0, sink, stamps, i);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      best = std::min(best, ms);
    }
    char what[96];
    snprintf(what, sizeof what, "tiny kernel (one store per thread), %4d wgs", G);
    report(what, best);
  }
  return 0;
}
