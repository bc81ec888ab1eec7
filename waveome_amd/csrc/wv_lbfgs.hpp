// Reverse-communication L-BFGS-B for unbounded problems (host C++).
//
// Replaces gpflow.optimizers.Scipy().minimize(method="L-BFGS-B") as called at
// waveome/model_classes.py:325-334.  Restates the published algorithm of
// Byrd-Lu-Nocedal-Zhu (L-BFGS-B 3.0) for the case nbd = 0 everywhere:
//   * first direction  d = -g, first trial step 1/||d||   (lnsrlb, iter 0, not boxed)
//   * later directions d = -H_k g with H_k the limited-memory BFGS inverse with
//     scaling 1/theta, theta = y'y / s'y (matupd); the compact form L-BFGS-B
//     uses and the two-loop recursion used here are the same matrix
//   * More'-Thuente line search dcsrch/dcstep, ftol 1e-3, gtol 0.9, xtol 0.1,
//     stpmin 0, stpmax 1e10, at most maxls trial points
//   * update skipped when s'y <= epsmch * (-g_old'd * stp)
//   * stops: max|g_i| <= pgtol; (f_old - f) <= factr*epsmch*max(|f_old|,|f|,1);
//     SciPy's driver limits maxiter (after an iteration) and maxfun
//   * line-search failure: restore the previous iterate, drop the memory and
//     restart; with empty memory -> ABNORMAL_TERMINATION_IN_LNSRCH
// One deviation, documented in DESIGN.md: a non-finite f (Gram matrix not
// positive definite) shrinks the trial step by bisection toward the last good
// point instead of raising (the reference lets TensorFlow raise and retries the
// whole minimise, waveome/model_classes.py:323-340).
#pragma once
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/waveome_amd.h"

namespace wv {

struct LineSearch {  // dcsrch state (MINPACK-2)
  bool brackt = false;
  int stage = 1;
  double ginit = 0, gtest = 0, gx = 0, gy = 0, finit = 0, fx = 0, fy = 0;
  double stx = 0, sty = 0, stmin = 0, stmax = 0, width = 0, width1 = 0;
  enum Task { START, FG, CONVERGENCE, WARNING, ERROR };
  Task task = START;
  // one call of dcsrch: updates stp, returns the new task
  Task step(double& stp, double f, double g, double ftol, double gtol, double xtol, double stpmin,
            double stpmax);
};

class Lbfgs {
 public:
  Lbfgs(int n, const wv_lbfgs_opts& o, const double* x0);
  const double* x() const { return x_.data(); }
  // feed f(x), g(x); returns WV_LB_FG while another evaluation is needed
  int feed(double f, const double* g);
  int n_iter() const { return iter_; }
  int n_eval() const { return nfev_; }
  double f() const { return f_; }
  int n() const { return n_; }
  // float32 contexts only (the float64 path keeps SciPy's rule exactly): how many consecutive hits of the
  // relative-decrease test are passed over before one counts as "converged"
  void set_ftol_patience(int k) { ftol_patience_ = k; }

 private:
  void direction();        // d = -H g
  int begin_linesearch();  // sets up stp and the first trial point
  int finish_iteration();  // after the line search converged
  void reset_memory();

  int n_, m_;
  wv_lbfgs_opts o_;
  std::vector<double> x_, g_, d_, xold_, gold_, z_;
  std::vector<double> S_, Y_, rho_;  // ring buffers, m x n
  int col_ = 0, head_ = 0;           // number of stored pairs, index of the oldest
  double theta_ = 1.0;
  // f_ starts at +inf: a fit whose very first evaluation is non-finite must not report a finite loss
  double f_ = std::numeric_limits<double>::infinity(), fold_ = 0, gd_ = 0, gdold_ = 0, stp_ = 0, dnorm_ = 0, sbgnrm_ = 0;
  int iter_ = 0, nfev_ = 0, ifun_ = 0, iback_ = 0;
  int ftol_patience_ = 0, ftol_strikes_ = 0;
  bool first_ = true, in_ls_ = false;
  double bis_lo_ = 0;  // non-finite handling: last finite step on the ray
  bool had_nonfinite_ = false;
  LineSearch ls_;
};

}  // namespace wv
