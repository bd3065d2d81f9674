// What a LEVEL-sized launch of the solve can reach: total bytes of one tree level (50 .. 240 MB), cut into per-wave
// tiles of 24 .. 96 KB (64 rows x K columns of doubles, read 512 B per wave load, HB..2HB loads in flight per lane as
// in stream_rest), with the solve's dependent chain in front of the stream switched on step by step:
//   chain 0: tile address from the wave index (no dependent load)
//   chain 1: + a 64-byte descriptor per tile (address and length read from memory)
//   chain 2: + an index list (K ints) and a gather of K 24-byte entries from 96-byte records into LDS before the stream
// and two launch shapes: one wave per tile (grid = tiles / 8 workgroups) or a persistent grid whose waves loop over
// tiles (stride = waves in the grid), optionally with the NEXT tile's chain issued before the current tile's stream.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_probe.hip -o /tmp/tile_probe && /tmp/tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Desc { long long off; int k; int idx_off; int pad[12]; };   // 64 bytes
static_assert(sizeof(Desc) == 64, "");

constexpr int HB = 8, D = 3;

__device__ __forceinline__ void stream(const double *wp, int kn, const double *fw, double (&acc)[D]) {
  double a[HB], b[HB];
  int kk = 0;
  bool have = HB <= kn;
  if (have) {
#pragma unroll
    for (int q = 0; q < HB; q++) a[q] = __builtin_nontemporal_load(wp + (size_t)q * 64);
  }
  while (have) {
    int k2 = kk + HB;
    bool more = k2 + HB <= kn;
    if (more) {
#pragma unroll
      for (int q = 0; q < HB; q++) b[q] = __builtin_nontemporal_load(wp + (size_t)(k2 + q) * 64);
    }
#pragma unroll
    for (int q = 0; q < HB; q++)
#pragma unroll
      for (int c = 0; c < D; c++) acc[c] = fma(a[q], fw[(kk + q) * D + c], acc[c]);
    kk = k2; have = more;
    if (!have) break;
    k2 = kk + HB;
    more = k2 + HB <= kn;
    if (more) {
#pragma unroll
      for (int q = 0; q < HB; q++) a[q] = __builtin_nontemporal_load(wp + (size_t)(k2 + q) * 64);
    }
#pragma unroll
    for (int q = 0; q < HB; q++)
#pragma unroll
      for (int c = 0; c < D; c++) acc[c] = fma(b[q], fw[(kk + q) * D + c], acc[c]);
    kk = k2; have = more;
  }
}

template <int CHAIN, bool PERSIST>
__global__ __launch_bounds__(512, 6) void k_tiles(const double *panels, const Desc *desc, const int *idx, const double *vec,
                                                  int ntiles, int kfix, double *out) {
  __shared__ double f[8][192 * D];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *fw = f[wv];
  const int wave = blockIdx.x * 8 + wv, nwaves = gridDim.x * 8;
  double acc[D] = {0, 0, 0};
  for (int t = wave; t < ntiles; t += nwaves) {
    long long off;
    int k, io;
    if (CHAIN >= 1) {
      const Desc dsc = desc[t];
      off = dsc.off; k = dsc.k; io = dsc.idx_off;
    } else {
      off = (long long)t * kfix * 64; k = kfix; io = t * kfix;
    }
    if (CHAIN >= 2) {
      for (int kk = lane; kk < k; kk += 64) {
        const double *src = vec + (size_t)idx[io + kk] * 12;
#pragma unroll
        for (int c = 0; c < D; c++) fw[kk * D + c] = src[c];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    stream(panels + off + lane, k, fw, acc);
    __builtin_amdgcn_wave_barrier();
    if (!PERSIST) break;
  }
  if (acc[0] + acc[1] + acc[2] == 12345.678) out[wave] = acc[0];
}

int main() {
  const size_t cap = (size_t)2 << 30;   // bytes of panels: every repetition reads a different (cold) part
  double *panels, *vec, *out;
  Desc *desc;
  int *idx;
  hipMalloc(&panels, cap);
  hipMemset(panels, 0, cap);
  const int nvec = 300000;
  hipMalloc(&vec, (size_t)nvec * 12 * 8);
  hipMemset(vec, 0, (size_t)nvec * 12 * 8);
  hipMalloc(&out, 1 << 22);
  const int max_tiles = 1 << 16;
  hipMalloc(&desc, sizeof(Desc) * max_tiles);
  hipMalloc(&idx, sizeof(int) * max_tiles * 512);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (double total_mb : {50.0, 100.0, 240.0})
    for (int k : {48, 96, 192}) {
      const int ntiles = (int)(total_mb * 1e6 / (k * 512.0));
      if (ntiles > max_tiles) continue;
      std::vector<Desc> hd(ntiles);
      std::vector<int> hi((size_t)ntiles * k);
      srand(1);
      for (int t = 0; t < ntiles; t++) {
        hd[t].off = (long long)t * k * 64; hd[t].k = k; hd[t].idx_off = t * k;
        const int base = rand() % (nvec - 4 * k);
        for (int q = 0; q < k; q++) hi[(size_t)t * k + q] = base + (q * 37) % (4 * k);   // scattered inside a neighbourhood
      }
      hipMemcpy(desc, hd.data(), sizeof(Desc) * ntiles, hipMemcpyHostToDevice);
      hipMemcpy(idx, hi.data(), sizeof(int) * hi.size(), hipMemcpyHostToDevice);
      auto time = [&](auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; rep++) {
          hipEventRecord(a);
          launch(panels + (size_t)rep * 32000000);
          hipEventRecord(b);
          hipEventSynchronize(b);
          float ms;
          hipEventElapsedTime(&ms, a, b);
          if (rep) best = ms < best ? ms : best;
        }
        return best * 1e3;
      };
      const double bytes = (double)ntiles * k * 512;
      const int g1 = (ntiles + 7) / 8;
      float t;
#define RUN(CH, PS, GRID, NAME)                                                                                         \
  t = time([&](const double *pp) { hipLaunchKernelGGL((k_tiles<CH, PS>), dim3(GRID), dim3(512), 0, 0, pp, desc, idx, vec, ntiles, k, out); }); \
  printf("%5.0f MB  K %3d (%5.1f KB/tile, %5d tiles)  %-34s %6.1f us %6.0f GB/s\n", total_mb, k, k * 0.5, ntiles, NAME, t, bytes / t * 1e-3);
      RUN(0, false, g1, "wave per tile, no chain")
      RUN(1, false, g1, "wave per tile, descriptor")
      RUN(2, false, g1, "wave per tile, descriptor + gather")
      RUN(0, true, 768, "persistent 768 wg, no chain")
      RUN(2, true, 768, "persistent 768 wg, desc + gather")
      RUN(2, true, 384, "persistent 384 wg, desc + gather")
#undef RUN
    }
  return 0;
}
