// What a grid-wide barrier costs on this GPU (8 XCDs, one L2 each): a persistent grid of G workgroups x 512 threads
// runs N barriers (agent-scope release / acquire on one counter); variant "data" also hands 4 KB per workgroup to a
// workgroup on another XCD across every barrier and checks it (the L2 write-back / invalidate a hand-off needs).
// For comparison: the same number of empty dependent kernel launches.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned target, int *err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 20000000) { *err = 1; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

template <bool DATA>
__global__ __launch_bounds__(512) void k_barriers(unsigned *ctr, int n, double *buf, int *err) {
  const int G = gridDim.x, b = blockIdx.x;
  for (int i = 0; i < n; i++) {
    if (DATA) buf[(size_t)b * 512 + threadIdx.x] = (double)(i * 1000003 + b);
    if (!grid_barrier(ctr, (unsigned)(i + 1) * G, err)) return;
    if (DATA) {
      const int src = (b + 37) % G;   // lands on another XCD (round-robin dispatch)
      const double v = buf[(size_t)src * 512 + threadIdx.x];
      if (v != (double)(i * 1000003 + src)) *err = 2;
      // a second barrier keeps the next round's writes from overtaking this round's reads
      if (!grid_barrier(ctr + 64, (unsigned)(i + 1) * G, err)) return;
    }
  }
}
__global__ void k_empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
  unsigned *ctr; double *buf; int *err;
  hipMalloc(&ctr, 1024); hipMalloc(&buf, 4096 * 512 * 8); hipMalloc(&err, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int N = 2000;
  for (int G : {8, 64, 256, 512, 768}) {
    for (int data = 0; data < 2; data++) {
      float best = 1e30f; int herr = 0;
      for (int rep = 0; rep < 3; rep++) {
        hipMemset(ctr, 0, 1024); hipMemset(err, 0, 4);
        hipEventRecord(a);
        if (data) hipLaunchKernelGGL(k_barriers<true>, dim3(G), dim3(512), 0, 0, ctr, N, buf, err);
        else hipLaunchKernelGGL(k_barriers<false>, dim3(G), dim3(512), 0, 0, ctr, N, buf, err);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
        hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      }
      printf("G %4d %-22s %7.2f us per %s  (err %d)\n", G, data ? "barrier + 4 KB hand-off" : "barrier only", best * 1e3 / N,
             data ? "round (2 barriers)" : "barrier", herr);
    }
  }
  for (int G : {256, 768, 4096}) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a);
      for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(G), dim3(512), 0, 0, (int *)nullptr);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
    }
    printf("G %4d empty dependent launches  %7.2f us per launch\n", G, best * 1e3 / N);
  }
  return 0;
}
