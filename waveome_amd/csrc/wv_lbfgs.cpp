// See wv_lbfgs.hpp for what this restates and the reference call site it replaces.
#include "wv_lbfgs.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace wv {

namespace {
constexpr double kEps = DBL_EPSILON;  // epsmch
constexpr double kBig = 1.0e10;       // stpmx for unconstrained problems
constexpr double kFtol = 1.0e-3, kGtol = 0.9, kXtol = 0.1, kStpMin = 0.0;

inline double dot(const double* a, const double* b, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

// dcstep: safeguarded cubic / quadratic step of More' and Thuente.
void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
            double fp, double dp, bool& brackt, double stpmin, double stpmax) {
  double sgnd = dp * (dx / std::fabs(dx));
  double stpf, stpc, stpq, theta, s, gamma, p, q, r;
  if (fp > fx) {  // case 1: higher function value, minimum is bracketed
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
    gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    p = (gamma - dx) + theta;
    q = ((gamma - dx) + gamma) + dp;
    r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (std::fabs(stpc - stx) < std::fabs(stpq - stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {  // case 2: derivatives of opposite sign
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
    gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + dx;
    r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (std::fabs(stpc - stp) > std::fabs(stpq - stp))
      stpf = stpc;
    else
      stpf = stpq;
    brackt = true;
  } else if (std::fabs(dp) < std::fabs(dx)) {  // case 3: derivative magnitude decreases
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
    gamma = s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (dx - dp)) + gamma;
    r = p / q;
    if (r < 0.0 && gamma != 0.0)
      stpc = stp + r * (stx - stp);
    else if (stp > stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (std::fabs(stpc - stp) < std::fabs(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      if (stp > stx)
        stpf = std::min(stp + 0.66 * (sty - stp), stpf);
      else
        stpf = std::max(stp + 0.66 * (sty - stp), stpf);
    } else {
      if (std::fabs(stpc - stp) > std::fabs(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      stpf = std::min(stpmax, stpf);
      stpf = std::max(stpmin, stpf);
    }
  } else {  // case 4: derivative does not decrease
    if (brackt) {
      theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      s = std::max({std::fabs(theta), std::fabs(dy), std::fabs(dp)});
      gamma = s * std::sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + dy;
      r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  // update the interval that contains a minimiser
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stp = stpf;
}
}  // namespace

LineSearch::Task LineSearch::step(double& stp, double f, double g, double ftol, double gtol,
                                  double xtol, double stpmin, double stpmax) {
  constexpr double xtrapl = 1.1, xtrapu = 4.0;
  if (task == START) {
    if (stp < stpmin || stp > stpmax || g >= 0.0 || ftol < 0 || gtol < 0 || xtol < 0 ||
        stpmin < 0 || stpmax < stpmin) {
      task = ERROR;
      return task;
    }
    brackt = false;
    stage = 1;
    finit = f;
    ginit = g;
    gtest = ftol * ginit;
    width = stpmax - stpmin;
    width1 = width / 0.5;
    stx = 0;
    fx = finit;
    gx = ginit;
    sty = 0;
    fy = finit;
    gy = ginit;
    stmin = 0;
    stmax = stp + xtrapu * stp;
    task = FG;
    return task;
  }
  double ftest = finit + stp * gtest;
  if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
  // warnings
  if (brackt && (stp <= stmin || stp >= stmax)) task = WARNING;
  if (brackt && stmax - stmin <= xtol * stmax) task = WARNING;
  if (stp == stpmax && f <= ftest && g <= gtest) task = WARNING;
  if (stp == stpmin && (f > ftest || g >= gtest)) task = WARNING;
  // convergence
  if (f <= ftest && std::fabs(g) <= gtol * (-ginit)) task = CONVERGENCE;
  if (task == WARNING || task == CONVERGENCE) return task;

  if (stage == 1 && f <= fx && f > ftest) {
    double fm = f - stp * gtest, fxm = fx - stx * gtest, fym = fy - sty * gtest;
    double gm = g - gtest, gxm = gx - gtest, gym = gy - gtest;
    dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax);
    fx = fxm + stx * gtest;
    fy = fym + sty * gtest;
    gx = gxm + gtest;
    gy = gym + gtest;
  } else {
    dcstep(stx, fx, gx, sty, fy, gy, stp, f, g, brackt, stmin, stmax);
  }
  if (brackt) {
    if (std::fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
    width1 = width;
    width = std::fabs(sty - stx);
  }
  if (brackt) {
    stmin = std::min(stx, sty);
    stmax = std::max(stx, sty);
  } else {
    stmin = stp + xtrapl * (stp - stx);
    stmax = stp + xtrapu * (stp - stx);
  }
  stp = std::max(stp, stpmin);
  stp = std::min(stp, stpmax);
  if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax))
    stp = stx;
  task = FG;
  return task;
}

Lbfgs::Lbfgs(int n, const wv_lbfgs_opts& o, const double* x0)
    : n_(n), m_(std::max(1, o.m)), o_(o), x_(x0, x0 + n), g_(n), d_(n), xold_(n), gold_(n), z_(n),
      S_((size_t)std::max(1, o.m) * n), Y_((size_t)std::max(1, o.m) * n), rho_(std::max(1, o.m)) {}

void Lbfgs::reset_memory() {
  col_ = 0;
  head_ = 0;
  theta_ = 1.0;
}

void Lbfgs::direction() {
  // two-loop recursion, H0 = I/theta
  std::vector<double> a(col_);
  for (int i = 0; i < n_; ++i) d_[i] = g_[i];
  for (int k = col_ - 1; k >= 0; --k) {
    int j = (head_ + k) % m_;
    a[k] = rho_[j] * dot(&S_[(size_t)j * n_], d_.data(), n_);
    const double* y = &Y_[(size_t)j * n_];
    for (int i = 0; i < n_; ++i) d_[i] -= a[k] * y[i];
  }
  for (int i = 0; i < n_; ++i) d_[i] /= theta_;
  for (int k = 0; k < col_; ++k) {
    int j = (head_ + k) % m_;
    double b = rho_[j] * dot(&Y_[(size_t)j * n_], d_.data(), n_);
    const double* s = &S_[(size_t)j * n_];
    for (int i = 0; i < n_; ++i) d_[i] += s[i] * (a[k] - b);
  }
  for (int i = 0; i < n_; ++i) d_[i] = -d_[i];
}

int Lbfgs::begin_linesearch() {
  for (;;) {
    direction();
    dnorm_ = std::sqrt(dot(d_.data(), d_.data(), n_));
    stp_ = (iter_ == 0) ? std::min(1.0 / dnorm_, kBig) : 1.0;
    xold_ = x_;
    gold_ = g_;
    fold_ = f_;
    for (int i = 0; i < n_; ++i) z_[i] = x_[i] + d_[i];
    ifun_ = 0;
    iback_ = 0;
    had_nonfinite_ = false;
    ls_ = LineSearch();
    gd_ = dot(g_.data(), d_.data(), n_);
    gdold_ = gd_;
    if (!(gd_ < 0.0)) {  // ascent direction in projection (info = -4)
      if (col_ == 0) return WV_LB_ABNORMAL;
      reset_memory();
      continue;
    }
    ls_.step(stp_, f_, gd_, kFtol, kGtol, kXtol, kStpMin, kBig);
    if (ls_.task == LineSearch::ERROR) return WV_LB_ABNORMAL;
    ifun_ = 1;
    iback_ = 0;
    if (stp_ == 1.0)
      x_ = z_;
    else
      for (int i = 0; i < n_; ++i) x_[i] = stp_ * d_[i] + xold_[i];
    in_ls_ = true;
    return WV_LB_FG;
  }
}

int Lbfgs::feed(double f, const double* g) {
  ++nfev_;
  bool finite = std::isfinite(f);
  if (finite)
    for (int i = 0; i < n_; ++i)
      if (!std::isfinite(g[i])) {
        finite = false;
        break;
      }
  if (first_) {
    first_ = false;
    if (!finite) return WV_LB_ABNORMAL;
    f_ = f;
    std::memcpy(g_.data(), g, sizeof(double) * n_);
    sbgnrm_ = 0;
    for (int i = 0; i < n_; ++i) sbgnrm_ = std::max(sbgnrm_, std::fabs(g_[i]));
    if (sbgnrm_ <= o_.gtol) return WV_LB_CONV_GTOL;
    return begin_linesearch();
  }
  // ---- inside a line search: (f, g) belong to the trial point x_ ----
  bool ls_done = false;
  if (!finite) {
    // deviation from the reference (see header): bisect toward the last good step
    had_nonfinite_ = true;
    ls_.sty = stp_;
    ls_.fy = std::max(std::fabs(ls_.finit), 1.0) * 1e30;
    ls_.gy = std::fabs(ls_.ginit) * 1e30;
    ls_.brackt = true;
    stp_ = 0.5 * (ls_.stx + stp_);
    ls_.stmin = std::min(ls_.stx, ls_.sty);
    ls_.stmax = std::max(ls_.stx, ls_.sty);
    ls_.width = std::fabs(ls_.sty - ls_.stx);
    ls_.width1 = 2.0 * ls_.width;
  } else {
    f_ = f;
    std::memcpy(g_.data(), g, sizeof(double) * n_);
    gd_ = dot(g_.data(), d_.data(), n_);
    LineSearch::Task t = ls_.step(stp_, f_, gd_, kFtol, kGtol, kXtol, kStpMin, kBig);
    ls_done = (t == LineSearch::CONVERGENCE || t == LineSearch::WARNING);
  }
  if (!ls_done) {
    ++ifun_;
    iback_ = ifun_ - 1;
    if (iback_ >= o_.maxls || ls_.task == LineSearch::ERROR) {
      // line search failed: restore the previous iterate
      x_ = xold_;
      g_ = gold_;
      f_ = fold_;
      in_ls_ = false;
      if (col_ == 0) {
        ++iter_;
        return WV_LB_ABNORMAL;
      }
      reset_memory();
      return begin_linesearch();
    }
    if (stp_ == 1.0)
      x_ = z_;
    else
      for (int i = 0; i < n_; ++i) x_[i] = stp_ * d_[i] + xold_[i];
    return WV_LB_FG;
  }
  in_ls_ = false;
  return finish_iteration();
}

int Lbfgs::finish_iteration() {
  // NEW_X
  ++iter_;
  sbgnrm_ = 0;
  for (int i = 0; i < n_; ++i) sbgnrm_ = std::max(sbgnrm_, std::fabs(g_[i]));
  // SciPy driver checks come first (scipy/optimize/_lbfgsb_py.py, task NEW_X)
  if (iter_ >= o_.maxiter) return WV_LB_STOP_MAXITER;
  if (nfev_ > o_.maxfun) return WV_LB_STOP_MAXFUN;
  if (sbgnrm_ <= o_.gtol) return WV_LB_CONV_GTOL;
  double ddum = std::max({std::fabs(fold_), std::fabs(f_), 1.0});
  if ((fold_ - f_) <= o_.ftol * ddum) {
    // A noisy objective (float32 contexts) can fake this test far from the optimum: only the
    // (ftol_patience_ + 1)-th hit in a row stops; until then the iteration goes on as if the test had passed.
    if (ftol_strikes_ >= ftol_patience_) return WV_LB_CONV_FTOL;
    ++ftol_strikes_;
  } else {
    ftol_strikes_ = 0;
  }
  // BFGS pair: y = g - gold, s = stp * d
  double dr, dd;
  if (stp_ == 1.0) {
    dr = gd_ - gdold_;
    dd = -gdold_;
  } else {
    dr = (gd_ - gdold_) * stp_;
    for (int i = 0; i < n_; ++i) d_[i] *= stp_;
    dd = -gdold_ * stp_;
  }
  if (dr > kEps * dd) {
    int j;
    if (col_ < m_) {
      j = (head_ + col_) % m_;
      ++col_;
    } else {
      j = head_;
      head_ = (head_ + 1) % m_;
    }
    double* s = &S_[(size_t)j * n_];
    double* y = &Y_[(size_t)j * n_];
    double rr = 0;
    for (int i = 0; i < n_; ++i) {
      s[i] = d_[i];
      y[i] = g_[i] - gold_[i];
      rr += y[i] * y[i];
    }
    rho_[j] = 1.0 / dr;
    theta_ = rr / dr;
  }
  return begin_linesearch();
}

}  // namespace wv
