// The REFERENCE's header-only TNT / STPCG driven by DPGO-shaped operators.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/ref_tnt/Makefile against the reference's headers where they lie under
// /root/reference (never copied), output oracle/_ref/tnt_pgo_ref.  It reads one node's surrogate problem -- the dense
// data matrix G, the linear term g, the constant f, G_tt^-1, (G_RR + lambda I)^-1 and a start point, written by
// tests/golden/make_tnt_pgo_golden.py -- and runs Optimization::Riemannian::TNT (TNT.h:242-693) exactly the way
// DPGOHash instantiates it (C++/DPGO/src/DPGOHash.cpp:270-349):
//   f(Y)       = tr(Y^T (g + 1/2 G Y)) + f                          DPGOProblem.cpp:180-205
//   grad       = Proj_R(g_R + (G Y)_R)                              DPGOProblem.h:380-406
//   Hess[Rdot] = Proj_R(G_Rt tdot + G_RR Rdot - SBD(Rdot, R, nabla)), tdot = -G_tt^-1 G_tR Rdot     DPGOProblem.cpp:552-577
//   precon     = Proj_R((G_RR + lambda I)^-1 V)                     DPGOProblem.cpp:579-598
//   retraction = [ -G_tt^-1 (g_t + G_tR R+) ; R+ = proj_SO(d)(R + V) ]                               DPGOProblem.cpp:127-143
//   metric     = sum V1 .* V2                                       DPGOHash.cpp:307-310
// with these operators restated here on dense row-major arrays (own code).  Output: one JSON object per case with the
// solver's trace (objective values, radii, inner iterations, gain ratios, step norms), the status and the final point.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <optional>
#include <string>
#include <vector>

#include "Optimization/Riemannian/TNT.h"

struct Mat {   // rows x d, row-major
  int rows = 0, d = 0;
  std::vector<double> v;
  Mat() {}
  Mat(int r, int dd) : rows(r), d(dd), v((size_t)r * dd, 0.0) {}
  double &operator()(int i, int j) { return v[(size_t)i * d + j]; }
  double operator()(int i, int j) const { return v[(size_t)i * d + j]; }
  Mat &operator+=(const Mat &o) { for (size_t i = 0; i < v.size(); i++) v[i] += o.v[i]; return *this; }
  Mat &operator-=(const Mat &o) { for (size_t i = 0; i < v.size(); i++) v[i] -= o.v[i]; return *this; }
  Mat &operator*=(double a) { for (auto &x : v) x *= a; return *this; }
};
static Mat operator+(Mat a, const Mat &b) { a += b; return a; }
static Mat operator-(Mat a, const Mat &b) { a -= b; return a; }
static Mat operator*(double a, Mat b) { b *= a; return b; }
static Mat operator-(Mat a) { a *= -1.0; return a; }

struct Problem {
  int d = 0, n = 0;               // n own poses; rows of Y: [n translations ; d n rotation rows]
  std::vector<double> G;          // (d+1)n x (d+1)n
  Mat g;                          // (d+1)n x d
  double f = 0;
  std::vector<double> GttInv;     // n x n
  std::vector<double> Minv;       // dn x dn (empty: no preconditioner)
  Mat x0;
  int N() const { return (d + 1) * n; }
};

// C (r x d) = A[r0 .. r0+r, c0 .. c0+c] (dense, leading dimension ld) * B (c x d)
static Mat mul(const std::vector<double> &A, int ld, int r0, int r, int c0, int c, const Mat &B, int brow0 = 0) {
  Mat C(r, B.d);
  for (int i = 0; i < r; i++)
    for (int k = 0; k < c; k++) {
      const double a = A[(size_t)(r0 + i) * ld + c0 + k];
      if (a == 0.0) continue;
      for (int j = 0; j < B.d; j++) C(i, j) += a * B(brow0 + k, j);
    }
  return C;
}
static Mat rows(const Mat &A, int r0, int r) {
  Mat B(r, A.d);
  for (int i = 0; i < r; i++)
    for (int j = 0; j < A.d; j++) B(i, j) = A(r0 + i, j);
  return B;
}
// SOdProduct::SymBlockDiagProduct (SOdProduct.h:64-89): P_i = sym(C_i B_i^T) A_i
static Mat sbd(const Mat &A, const Mat &B, const Mat &C, int d) {
  Mat P(A.rows, d);
  for (int i = 0; i < A.rows / d; i++) {
    double Gm[9], S[9];
    for (int r = 0; r < d; r++)
      for (int c = 0; c < d; c++) {
        double a = 0;
        for (int k = 0; k < d; k++) a += C(i * d + r, k) * B(i * d + c, k);
        Gm[r * d + c] = a;
      }
    for (int r = 0; r < d; r++)
      for (int c = 0; c < d; c++) S[r * d + c] = 0.5 * (Gm[r * d + c] + Gm[c * d + r]);
    for (int r = 0; r < d; r++)
      for (int c = 0; c < d; c++) {
        double a = 0;
        for (int k = 0; k < d; k++) a += S[r * d + k] * A(i * d + k, c);
        P(i * d + r, c) = a;
      }
  }
  return P;
}
static Mat proj_tangent(const Mat &R, const Mat &V, int d) { return V - sbd(R, R, V, d); }   // SOdProduct::Proj (:96-103)

// nearest rotation of a 3 x 3 matrix M: U diag(1, 1, det(U V^T)) V^T from the eigen-decomposition of M^T M (cyclic Jacobi)
static void nearest_rotation3(const double *M, double *R) {
  double S[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double a = 0;
      for (int k = 0; k < 3; k++) a += M[k * 3 + r] * M[k * 3 + c];
      S[r * 3 + c] = a;
    }
  for (int sweep = 0; sweep < 30; sweep++)
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (std::fabs(S[p * 3 + q]) < 1e-300) continue;
        const double th = 0.5 * (S[q * 3 + q] - S[p * 3 + p]) / S[p * 3 + q];
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {   // S <- J^T S J, V <- V J
          const double a = S[k * 3 + p], b = S[k * 3 + q];
          S[k * 3 + p] = c * a - s * b;
          S[k * 3 + q] = s * a + c * b;
        }
        for (int k = 0; k < 3; k++) {
          const double a = S[p * 3 + k], b = S[q * 3 + k];
          S[p * 3 + k] = c * a - s * b;
          S[q * 3 + k] = s * a + c * b;
        }
        for (int k = 0; k < 3; k++) {
          const double a = V[k * 3 + p], b = V[k * 3 + q];
          V[k * 3 + p] = c * a - s * b;
          V[k * 3 + q] = s * a + c * b;
        }
      }
  // order the eigenvalues decreasingly (keeps det V = +-1 irrelevant: the sign fix below sees U V^T)
  int idx[3] = {0, 1, 2};
  for (int a = 0; a < 3; a++)
    for (int b = a + 1; b < 3; b++)
      if (S[idx[b] * 3 + idx[b]] > S[idx[a] * 3 + idx[a]]) std::swap(idx[a], idx[b]);
  double Vs[9], U[9];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) Vs[r * 3 + c] = V[r * 3 + idx[c]];
  for (int c = 0; c < 3; c++) {   // U = M V Sigma^-1 for the two leading columns, the third by the cross product
    double col[3];
    for (int r = 0; r < 3; r++) col[r] = M[r * 3 + 0] * Vs[0 * 3 + c] + M[r * 3 + 1] * Vs[1 * 3 + c] + M[r * 3 + 2] * Vs[2 * 3 + c];
    for (int r = 0; r < 3; r++) U[r * 3 + c] = col[r];
  }
  auto normalise = [&](int c) { const double nn = std::sqrt(U[c] * U[c] + U[3 + c] * U[3 + c] + U[6 + c] * U[6 + c]); for (int r = 0; r < 3; r++) U[r * 3 + c] /= nn; };
  normalise(0);
  { const double dp = U[0] * U[1] + U[3] * U[4] + U[6] * U[7]; for (int r = 0; r < 3; r++) U[r * 3 + 1] -= dp * U[r * 3 + 0]; }
  normalise(1);
  // third column: +-(u1 x u2), the sign such that det(U V^T) = +1, i.e. det(U) = det(V)
  const double cx = U[3] * U[7] - U[6] * U[4], cy = U[6] * U[1] - U[0] * U[7], cz = U[0] * U[4] - U[3] * U[1];
  const double detV = Vs[0] * (Vs[4] * Vs[8] - Vs[5] * Vs[7]) - Vs[1] * (Vs[3] * Vs[8] - Vs[5] * Vs[6]) + Vs[2] * (Vs[3] * Vs[7] - Vs[4] * Vs[6]);
  const double sg = detV >= 0 ? 1.0 : -1.0;
  U[2] = sg * cx; U[5] = sg * cy; U[8] = sg * cz;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r * 3 + c] = U[r * 3 + 0] * Vs[c * 3 + 0] + U[r * 3 + 1] * Vs[c * 3 + 1] + U[r * 3 + 2] * Vs[c * 3 + 2];
}
static Mat project_rotations(const Mat &A, int d) {
  if (d != 3) { fprintf(stderr, "harness_pgo: d = 3 only\n"); exit(2); }
  Mat R(A.rows, d);
  for (int i = 0; i < A.rows / 3; i++) nearest_rotation3(&A.v[(size_t)i * 9], &R.v[(size_t)i * 9]);
  return R;
}

static bool read_problem(const char *path, Problem &P, std::vector<double> &prm) {
  FILE *fh = fopen(path, "rb");
  if (!fh) return false;
  int hdr[4];
  if (fread(hdr, sizeof(int), 4, fh) != 4) return false;
  P.d = hdr[0]; P.n = hdr[1];
  const int has_precon = hdr[2], nprm = hdr[3];
  const int N = P.N(), dn = P.d * P.n;
  auto rd = [&](std::vector<double> &v, size_t cnt) { v.resize(cnt); return fread(v.data(), sizeof(double), cnt, fh) == cnt; };
  std::vector<double> tmp;
  bool ok = rd(P.G, (size_t)N * N);
  P.g = Mat(N, P.d); ok = ok && fread(P.g.v.data(), 8, P.g.v.size(), fh) == P.g.v.size();
  ok = ok && fread(&P.f, 8, 1, fh) == 1;
  ok = ok && rd(P.GttInv, (size_t)P.n * P.n);
  if (has_precon) ok = ok && rd(P.Minv, (size_t)dn * dn);
  P.x0 = Mat(N, P.d); ok = ok && fread(P.x0.v.data(), 8, P.x0.v.size(), fh) == P.x0.v.size();
  ok = ok && rd(prm, nprm);
  fclose(fh);
  return ok;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: tnt_pgo_ref <case name> <problem.bin>\n"); return 2; }
  Problem P;
  std::vector<double> prm_v;
  if (!read_problem(argv[2], P, prm_v) || prm_v.size() < 9) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
  namespace R = Optimization::Riemannian;
  const int d = P.d, n = P.n, N = P.N(), dn = d * n;
  auto recover_t = [&](const Mat &Rr) {   // t = -G_tt^-1 (g_t + G_tR R)   (DPGOProblem.h:275-294)
    Mat rhs = rows(P.g, 0, n) + mul(P.G, N, 0, n, n, dn, Rr);
    return -mul(P.GttInv, n, 0, n, 0, n, rhs);
  };
  auto stack = [&](const Mat &t, const Mat &Rr) {
    Mat Y(N, d);
    for (int i = 0; i < n; i++) for (int j = 0; j < d; j++) Y(i, j) = t(i, j);
    for (int i = 0; i < dn; i++) for (int j = 0; j < d; j++) Y(n + i, j) = Rr(i, j);
    return Y;
  };
  Optimization::Objective<Mat, double, Mat> F = [&](const Mat &Y, Mat &) {
    Mat t = P.g + 0.5 * mul(P.G, N, 0, N, 0, N, Y);
    double s = 0;
    for (size_t i = 0; i < Y.v.size(); i++) s += Y.v[i] * t.v[i];
    return s + P.f;
  };
  R::QuadraticModel<Mat, Mat, Mat> QM = [&](const Mat &Y, Mat &grad, R::LinearOperator<Mat, Mat, Mat> &Hess, Mat &nablaF) {
    nablaF = rows(P.g, n, dn) + mul(P.G, N, n, dn, 0, N, Y);     // reduced Euclidean gradient (DPGOProblem.h:380-393)
    grad = proj_tangent(rows(Y, n, dn), nablaF, d);
    Hess = [&](const Mat &Yc, const Mat &Ydot, Mat &nab) {        // DPGOProblem.cpp:552-577
      Mat tdot = -mul(P.GttInv, n, 0, n, 0, n, mul(P.G, N, 0, n, n, dn, Ydot));
      Mat E = mul(P.G, N, n, dn, 0, n, tdot) + mul(P.G, N, n, dn, n, dn, Ydot);
      const Mat Rc = rows(Yc, n, dn);
      E -= sbd(Ydot, Rc, nab, d);
      return proj_tangent(Rc, E, d);
    };
  };
  R::RiemannianMetric<Mat, Mat, double, Mat> metric = [](const Mat &, const Mat &a, const Mat &b, Mat &) {
    double s = 0;
    for (size_t i = 0; i < a.v.size(); i++) s += a.v[i] * b.v[i];
    return s;
  };
  R::Retraction<Mat, Mat, Mat> retract = [&](const Mat &Y, const Mat &V, Mat &) {   // DPGOProblem.cpp:127-143
    Mat Rp = project_rotations(rows(Y, n, dn) + V, d);
    return stack(recover_t(Rp), Rp);
  };
  std::optional<R::LinearOperator<Mat, Mat, Mat>> precon;
  if (!P.Minv.empty())
    precon = R::LinearOperator<Mat, Mat, Mat>([&](const Mat &Y, const Mat &V, Mat &) {   // DPGOProblem.cpp:579-598
      return proj_tangent(rows(Y, n, dn), mul(P.Minv, dn, 0, dn, 0, dn, V), d);
    });
  R::TNTParams<double> prm;
  prm.max_iterations = (size_t)prm_v[0];
  prm.max_iterations_accepted = (int)prm_v[1];
  prm.gradient_tolerance = prm_v[2];
  prm.preconditioned_gradient_tolerance = prm_v[3];
  prm.relative_decrease_tolerance = prm_v[4];
  prm.stepsize_tolerance = prm_v[5];
  prm.kappa_fgr = prm_v[6];
  prm.theta = prm_v[7];
  prm.max_TPCG_iterations = (size_t)prm_v[8];
  Mat nabla(dn, d);
  auto res = R::TNT<Mat, Mat, double, Mat>(F, QM, metric, retract, P.x0, nabla, precon, prm);
  auto list = [](const char *name, const auto &x) {
    printf("\"%s\": [", name);
    size_t i = 0;
    for (const auto &e : x) printf("%s%.17g", i++ ? ", " : "", (double)e);
    printf("]");
  };
  printf("{\"case\": \"%s\", \"d\": %d, \"n\": %d, \"precon\": %d, \"f\": %.17g, \"status\": %d, \"gradfx_norm\": %.17g, \"pgradfx_norm\": %.17g, ",
         argv[1], d, n, (int)!P.Minv.empty(), res.f, (int)res.status, res.gradfx_norm, res.preconditioned_grad_f_x_norm);
  list("objective_values", res.objective_values); printf(", ");
  list("trust_region_radius", res.trust_region_radius); printf(", ");
  list("inner_iterations", res.inner_iterations); printf(", ");
  list("gain_ratios", res.gain_ratios); printf(", ");
  list("update_step_M_norms", res.update_step_M_norms); printf(", ");
  list("x", res.x.v);
  printf("}\n");
  return 0;
}
