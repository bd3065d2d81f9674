// Harness around the REFERENCE's AVX2 nearest-rotation kernels.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/ref_so3/Makefile; the definitions of DPGO::internal::project_to_SO3 /
// project_to_SO2 (__m256d) come from the reference's C++/DPGO/src/internal/project_to_SOd.cpp:7-33, 97-196, cut
// out at build time into oracle/_ref/so_double_definitions.inc (never committed); the macros they expand are the
// reference's headers, included where they lie.  The batching below is that of project_to_SO3_d / project_to_SO2_d
// (C++/DPGO/include/DPGO/internal/project_to_SOd.h:44-101): four matrices per call, element (i, j) of matrix k in
// lane k of register 3 i + j.
//   so_ref 3 < in.bin > out.bin     in: int64 n, then n x 9 doubles (row-major A); out: n x 9 doubles (row-major U V^T)
//   so_ref 2 < in.bin > out.bin     in: int64 n, then n x 4 doubles;               out: n x 4 doubles
#include <x86intrin.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <DPGO/internal/traits.h>
#include <DPGO/internal/project_to_SO2.h>
#include <DPGO/internal/project_to_SO3.h>

namespace DPGO {
namespace internal {
#include "so_double_definitions.inc"
}  // namespace internal
}  // namespace DPGO

int main(int argc, char **argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 3;
  int64_t n = 0;
  if (fread(&n, 8, 1, stdin) != 1 || n < 0) return 1;
  const int q = d * d;
  std::vector<double> A((size_t)(n + 4) * q, 0.0), U((size_t)(n + 4) * q, 0.0);
  if (n && fread(A.data(), 8, (size_t)n * q, stdin) != (size_t)n * q) return 1;
  for (int64_t g = 0; g < n; g += 4) {
    double temp[9][4];
    for (int k = 0; k < 4; k++)
      for (int e = 0; e < q; e++) temp[e][k] = A[(size_t)(g + k) * q + e];
    __m256d a[9], u[9];
    for (int e = 0; e < q; e++) a[e] = _mm256_loadu_pd(temp[e]);
    if (d == 3) {
      DPGO::internal::project_to_SO3(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], u[0], u[1], u[2], u[3], u[4], u[5], u[6],
                                     u[7], u[8]);
      for (int e = 0; e < 9; e++) _mm256_storeu_pd(temp[e], u[e]);
    } else {
      // project_to_SO2_d (project_to_SOd.h:44-53): U = [[u11, -u21], [u21, u11]]
      __m256d u11, u21;
      DPGO::internal::project_to_SO2(a[0], a[1], a[2], a[3], u11, u21);
      double c[4], s[4];
      _mm256_storeu_pd(c, u11);
      _mm256_storeu_pd(s, u21);
      for (int k = 0; k < 4; k++) { temp[0][k] = c[k]; temp[1][k] = -s[k]; temp[2][k] = s[k]; temp[3][k] = c[k]; }
    }
    for (int k = 0; k < 4; k++)
      for (int e = 0; e < q; e++) U[(size_t)(g + k) * q + e] = temp[e][k];
  }
  fwrite(U.data(), 8, (size_t)n * q, stdout);
  return 0;
}
