// Harness around the REFERENCE's header-only TNT / STPCG solvers.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/ref_tnt/Makefile against the
// headers where they lie under /root/reference (never copied), output
// oracle/_ref/tnt_ref.  It runs the known-answer problems of the reference's
// unit tests (C++/Optimization/tests/IterativeSolvers_unit_test.cpp:79-247,
// TNT_unit_test.cpp:63-187) plus a few parameter variations, and prints one
// JSON object per case.  tests/golden/tnt_ref.json is its committed output;
// tests/test_oracle_tnt.py checks oracle/tnt.py against it.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <limits>
#include <vector>

#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/Riemannian/TNT.h"

struct Vec {
  std::vector<double> v;
  Vec() {}
  explicit Vec(size_t n) : v(n, 0.0) {}
  Vec(std::initializer_list<double> l) : v(l) {}
  size_t size() const { return v.size(); }
  double &operator[](size_t i) { return v[i]; }
  double operator[](size_t i) const { return v[i]; }
  Vec &operator+=(const Vec &o) { for (size_t i = 0; i < v.size(); i++) v[i] += o.v[i]; return *this; }
  Vec &operator-=(const Vec &o) { for (size_t i = 0; i < v.size(); i++) v[i] -= o.v[i]; return *this; }
  Vec &operator*=(double a) { for (auto &x : v) x *= a; return *this; }
};
static Vec operator+(Vec a, const Vec &b) { a += b; return a; }
static Vec operator-(Vec a, const Vec &b) { a -= b; return a; }
static Vec operator*(double a, Vec b) { b *= a; return b; }
static Vec operator-(Vec a) { a *= -1.0; return a; }
static double dot(const Vec &a, const Vec &b) { double s = 0; for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i]; return s; }
static Vec cwise(const Vec &d, const Vec &x) { Vec r(x.size()); for (size_t i = 0; i < x.size(); i++) r[i] = d[i] * x[i]; return r; }

static void print_vec(const char *name, const Vec &x) {
  printf("\"%s\": [", name);
  for (size_t i = 0; i < x.size(); i++) printf("%s%.17g", i ? ", " : "", x[i]);
  printf("]");
}
template <class T> static void print_list(const char *name, const std::vector<T> &x) {
  printf("\"%s\": [", name);
  for (size_t i = 0; i < x.size(); i++) printf("%s%.17g", i ? ", " : "", (double)x[i]);
  printf("]");
}

using namespace Optimization;
using namespace Optimization::LinearAlgebra;

static void run_stpcg(const char *name, const Vec &g, const Vec &Hdiag, const Vec *Mdiag, double Delta,
                      size_t max_it, double kappa, double theta) {
  InnerProduct<Vec> ip = [](const Vec &a, const Vec &b) { return dot(a, b); };
  SymmetricLinearOperator<Vec> H = [&](const Vec &x) { return cwise(Hdiag, x); };
  std::optional<STPCGPreconditioner<Vec, Vec>> P;
  Vec Minv;
  if (Mdiag) {
    Minv = *Mdiag;
    for (auto &x : Minv.v) x = 1.0 / x;
    P = STPCGPreconditioner<Vec, Vec>([&](const Vec &x) { return std::make_pair(cwise(Minv, x), Vec()); });
  }
  double step_norm = 0;
  size_t nit = 0;
  Vec s = STPCG<Vec, Vec>(g, H, ip, step_norm, nit, Delta, max_it, kappa, theta, P);
  printf("{\"case\": \"%s\", \"kind\": \"stpcg\", ", name);
  print_vec("g", g); printf(", "); print_vec("Hdiag", Hdiag); printf(", ");
  if (Mdiag) { print_vec("Mdiag", *Mdiag); printf(", "); }
  printf("\"Delta\": %.17g, \"max_it\": %zu, \"kappa\": %.17g, \"theta\": %.17g, ", Delta, max_it, kappa, theta);
  print_vec("s", s);
  printf(", \"step_norm\": %.17g, \"num_iterations\": %zu}\n", step_norm, nit);
}

// TNT on the sphere S^2: f(X) = |X - P|^2  (TNT_unit_test.cpp:63-124)
static void run_tnt(const char *name, bool use_precon, const Vec &X0, size_t max_it, int max_acc,
                    double gtol, double pgtol, double reltol, double steptol, double kappa, double theta) {
  namespace R = Optimization::Riemannian;
  using R::TNTParams; using R::TNT; using R::QuadraticModel; using R::RiemannianMetric; using R::Retraction;
  Vec Pt{0.0, 0.0, 1.0};
  auto project = [](const Vec &X, const Vec &V) { return V - dot(X, V) * X; };
  Objective<Vec, double, Vec> F = [](const Vec &X, Vec &P) { Vec dlt = X - P; return dot(dlt, dlt); };
  auto gradF = [project](const Vec &X, const Vec &P) { return project(X, 2.0 * (X - P)); };
  QuadraticModel<Vec, Vec, Vec> QM = [project, gradF](const Vec &X, Vec &grad, R::LinearOperator<Vec, Vec, Vec> &Hess, Vec &P) {
    grad = gradF(X, P);
    Hess = [project, gradF](const Vec &X, const Vec &Xdot, Vec &P) {
      return project(X, 2.0 * Xdot) - dot(X, gradF(X, P)) * Xdot;
    };
  };
  RiemannianMetric<Vec, Vec, double, Vec> metric = [](const Vec &X, const Vec &a, const Vec &b, Vec &P) { return dot(a, b); };
  Retraction<Vec, Vec, Vec> retract = [](const Vec &X, const Vec &V, Vec &P) {
    Vec y = X + V; return (1.0 / std::sqrt(dot(y, y))) * y; };
  std::optional<R::LinearOperator<Vec, Vec, Vec>> precon;
  if (use_precon)
    precon = R::LinearOperator<Vec, Vec, Vec>([](const Vec &X, const Vec &V, Vec &P) { return cwise(Vec{1.0, 2.0, 3.0}, V); });
  TNTParams<double> prm;
  prm.max_iterations = max_it;
  prm.max_iterations_accepted = max_acc;
  prm.gradient_tolerance = gtol;
  prm.preconditioned_gradient_tolerance = pgtol;
  prm.relative_decrease_tolerance = reltol;
  prm.stepsize_tolerance = steptol;
  prm.kappa_fgr = kappa;
  prm.theta = theta;
  prm.max_TPCG_iterations = 10000;
  auto res = TNT<Vec, Vec, double, Vec>(F, QM, metric, retract, X0, Pt, precon, prm);
  printf("{\"case\": \"%s\", \"kind\": \"tnt\", \"precon\": %d, ", name, (int)use_precon);
  print_vec("x0", X0);
  printf(", \"max_it\": %zu, \"max_acc\": %d, \"gtol\": %.17g, \"pgtol\": %.17g, \"reltol\": %.17g, \"steptol\": %.17g, "
         "\"kappa\": %.17g, \"theta\": %.17g, ", max_it, max_acc, gtol, pgtol, reltol, steptol, kappa, theta);
  print_vec("x", res.x);
  printf(", \"f\": %.17g, \"status\": %d, \"gradfx_norm\": %.17g, \"pgradfx_norm\": %.17g, ", res.f, (int)res.status,
         res.gradfx_norm, res.preconditioned_grad_f_x_norm);
  print_list("objective_values", res.objective_values); printf(", ");
  print_list("trust_region_radius", res.trust_region_radius); printf(", ");
  print_list("inner_iterations", res.inner_iterations); printf(", ");
  print_list("gain_ratios", res.gain_ratios); printf(", ");
  print_list("update_step_M_norms", res.update_step_M_norms);
  printf("}\n");
}

int main() {
  const double inf = std::numeric_limits<double>::max();
  Vec g{21, -.4, 19}, Pd{1000, 100, 1}, Nd{-1000, -100, -1}, Md{100, 10, 1};
  // IterativeSolvers_unit_test.cpp:138-247
  run_stpcg("ExactSTPCG", g, Pd, nullptr, inf, 3, 1e-8, .999);
  run_stpcg("ExactSTPCGwithNegativeCurvature", g, Nd, nullptr, 1000, 3, 1e-8, .999);
  run_stpcg("ExactSTPCGwithPreconditioning", g, Pd, &Md, inf, 3, 1e-8, .999);
  run_stpcg("ExactSTPCGwithNegativeCurvatureAndPreconditioning", g, Nd, &Md, 1000, 3, 1e-8, .999);
  // radius-limited and truncated variants on the same fixture
  run_stpcg("RadiusLimited", g, Pd, nullptr, 5.0, 3, 1e-8, .999);
  run_stpcg("RadiusLimitedPrecon", g, Pd, &Md, 5.0, 3, 1e-8, .999);
  run_stpcg("Truncated", g, Pd, nullptr, inf, 1000, .05, .9);
  run_stpcg("TruncatedPrecon", g, Pd, &Md, inf, 1000, .05, .9);
  {  // a deterministic 40-dimensional instance
    Vec gg(40), hh(40), mm(40);
    for (int i = 0; i < 40; i++) { gg[i] = std::sin(1.0 + i) * (1 + i % 5); hh[i] = 1.0 + 0.37 * i + 3 * std::cos(0.3 * i) * std::cos(0.3 * i); mm[i] = 0.5 + 0.25 * i; }
    run_stpcg("Large40", gg, hh, nullptr, inf, 1000, .05, .9);
    run_stpcg("Large40Precon", gg, hh, &mm, inf, 1000, .05, .9);
    run_stpcg("Large40PreconRadius", gg, hh, &mm, 0.3, 1000, .05, .9);
  }
  Vec X0{-0.5, -0.5, -0.707107};
  // TNT_unit_test.cpp:126-187
  run_tnt("RiemannianTNTSphere", false, X0, 100, 100, 1e-8, 0, 0, 0, .1, .5);
  run_tnt("RiemannianTNTSphereWithPrecon", true, X0, 100, 100, 1e-8, 0, 0, 0, .1, .5);
  // the parameter set DPGO uses (dist_pgo.cpp:110-119, DPGOHash.cpp:337-349)
  run_tnt("DPGOParams", false, X0, 10, 1, 1e-3, 1e-4, 1e-6, 1e-4, .05, .9);
  run_tnt("DPGOParamsPrecon", true, X0, 10, 1, 1e-3, 1e-4, 1e-6, 1e-4, .05, .9);
  run_tnt("DPGOParams3Accepted", true, Vec{0.6, -0.64, 0.48}, 10, 3, 1e-3, 1e-4, 1e-6, 1e-4, .05, .9);
  return 0;
}
