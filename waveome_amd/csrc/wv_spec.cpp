#include "wv_spec.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <utility>

#include "wv_lbfgs.hpp"

namespace wv {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
// A launcher that cannot serve a request (an argument combination no kernel instantiation covers) records it
// here instead of aborting the host process; the C-ABI entry point that enqueued it reports WV_E_UNSUPPORTED.
static thread_local bool g_launch_failed = false;
void fail_launch(const std::string& msg) { g_err = msg; g_launch_failed = true; }
bool take_launch_failure() { bool f = g_launch_failed; g_launch_failed = false; return f; }

int n_factor_params(int type) {
  switch (type) {
    case WV_K_SE:
    case WV_K_M12:
    case WV_K_M32:
    case WV_K_M52:
    case WV_K_POLY:
      return 2;
    case WV_K_PERIODIC:
      return 3;
    case WV_K_LIN:
    case WV_K_CONST:
    case WV_K_CAT:
      return 1;
    case WV_K_EMPTY:
      return 0;
    default:
      return -1;
  }
}

// positive parameters saturate at softplus(-340) ~ 2e-148: never exactly 0, where the Horseshoe
// log-density is +inf (gpflow exposes the same idea as default_positive_minimum)
constexpr double kUFloor = -340.0;  // theta >= ~2e-148 so (x/s)^2 stays a normal double
double softplus(double u) {
  if (u < kUFloor) u = kUFloor;
  return u > 0 ? u + std::log1p(std::exp(-u)) : std::log1p(std::exp(u));
}
double inv_softplus(double x) { return x + std::log(-std::expm1(-x)); }
// d softplus/du without cancellation for u << 0 (pruned variances sit at u ~ -34)
double sigmoid(double u) {
  if (u < kUFloor) return 0.0;
  const double e = std::exp(-std::fabs(u));
  return u >= 0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
}

// tfd.Horseshoe._log_prob restated (see oracle/gp_oracle.py: horseshoe_logpdf).
double horseshoe_logpdf(double x, double scale, double* dlogp) {
  const double g = 0.5614594835668851, b = 1.0420764938351215, h_inf = 1.0801359952503342;
  const double pw = 1.0919284281983377;
  double xx = (x / scale) * (x / scale) / 2.0;
  double q = 20.0 / 47.0 * std::pow(xx, pw);
  double x15 = std::pow(xx, 1.5);
  double h = 1.0 / (1.0 + x15) + h_inf * q / (1.0 + q);
  double c = -0.5 * std::log(2.0 * M_PI * M_PI * M_PI) - std::log(g * scale);
  double z = std::log1p(-g) - std::log(g);
  double t = z - xx / (1.0 - g);
  double den = h + b * xx;
  double inner = g / xx - (1.0 - g) / (den * den);
  double l1p = std::log1p(inner);
  if (dlogp) {
    double dxx = x / (scale * scale);
    double dq = 20.0 / 47.0 * pw * std::pow(xx, pw - 1.0);
    double dh = -1.5 * std::sqrt(xx) / ((1.0 + x15) * (1.0 + x15)) + h_inf * dq / ((1.0 + q) * (1.0 + q));
    double d1 = sigmoid(t) / (1.0 - g);
    // d2 * dxx with numerator and denominator multiplied by xx^2 (xx^2 underflows for x < 1e-77)
    double c1 = 2.0 * (1.0 - g) * (dh + b) / (den * den * den);
    double d2dxx = (-g + xx * xx * c1) / ((xx + g - xx * (1.0 - g) / (den * den)) * l1p) * (2.0 / x);
    *dlogp = d1 * dxx + d2dxx;
  }
  return -softplus(t) + std::log(l1p) + c;
}

std::string Spec::parse(const wv_kernel_spec* s, Spec& out) {
  if (!s) return "spec is NULL";
  if (s->n_dims < 1 || s->n_dims > WV_MAX_DIMS) return "n_dims out of range [1, 32]";
  if (s->n_params < 1 || s->n_params > WV_MAX_PARAMS) return "n_params out of range (1 .. WV_MAX_PARAMS = 256)";
  if (!s->params || !s->term_start || (s->n_terms > 0 && !s->factors)) return "null table in spec";
  if (s->n_terms < 0) return "n_terms < 0";
  out = Spec();
  out.n_dims = s->n_dims;
  out.params.assign(s->params, s->params + s->n_params);
  out.term_start.assign(s->term_start, s->term_start + s->n_terms + 1);
  if (out.term_start[0] != 0) return "term_start[0] != 0";
  for (int t = 0; t < s->n_terms; ++t)
    if (out.term_start[t + 1] <= out.term_start[t]) return "empty or unordered term";
  int nf = out.term_start[s->n_terms];
  out.factors.assign(s->factors, s->factors + nf);
  uint32_t all = s->n_dims == 32 ? 0xffffffffu : ((1u << s->n_dims) - 1u);
  for (const auto& f : out.factors) {
    int np = n_factor_params(f.type);
    if (np < 0) return "unknown factor type";
    if (f.type != WV_K_EMPTY && f.type != WV_K_CONST) {
      if (f.dim_mask == 0 || (f.dim_mask & ~all)) return "factor dim_mask outside the D columns";
    }
    if (f.type == WV_K_CAT && __builtin_popcount(f.dim_mask) != 1)
      return "categorical factor needs exactly one column";
    if (f.type == WV_K_POLY && (f.degree < 1 || f.degree > 16)) return "poly degree out of range";
    for (int k = 0; k < np; ++k)
      if (f.param[k] < 0 || f.param[k] >= s->n_params) return "factor parameter index out of range";
  }
  if (s->noise_param < 0 || s->noise_param >= s->n_params) return "noise_param out of range";
  if (s->mean_param < -1 || s->mean_param >= s->n_params) return "mean_param out of range";
  out.noise_param = s->noise_param;
  out.mean_param = s->mean_param;
  for (int i = 0; i < s->n_params; ++i) {
    const auto& p = out.params[i];
    if (p.transform < WV_T_IDENTITY || p.transform > WV_T_SOFTPLUS_SHIFT) return "unknown transform";
    if (p.transform == WV_T_SOFTPLUS && !(p.init > 0)) return "positive parameter with init <= 0";
    // (a fit can end with the noise exactly ON its floor - softplus(u) below the last bit of 1e-6 for u < -50 - and
    // the fitted model must stay usable: the floor itself is a valid value, unconstrained at the saturation point)
    if (p.transform == WV_T_SOFTPLUS_SHIFT && !(p.init >= kNoiseFloor))
      return "noise variance init < 1e-6";
    if (p.trainable) out.trainable.push_back(i);
  }
  return "";
}

void Spec::constrain(const double* u, double* theta) const {
  for (int i = 0; i < n_params(); ++i) theta[i] = params[i].init;
  for (int k = 0; k < n_trainable(); ++k) {
    int i = trainable[k];
    switch (params[i].transform) {
      case WV_T_SOFTPLUS:
        theta[i] = softplus(u[k]);
        break;
      case WV_T_SOFTPLUS_SHIFT:
        theta[i] = kNoiseFloor + softplus(u[k]);
        break;
      default:
        theta[i] = u[k];
    }
  }
}

void Spec::unconstrain(const double* theta, double* u) const {
  for (int k = 0; k < n_trainable(); ++k) {
    int i = trainable[k];
    switch (params[i].transform) {
      case WV_T_SOFTPLUS:
        u[k] = std::max(kUFloor, inv_softplus(theta[i]));
        break;
      case WV_T_SOFTPLUS_SHIFT: {
        const double x = theta[i] - kNoiseFloor;
        u[k] = x > 0 ? std::max(kUFloor, inv_softplus(x)) : kUFloor;
        break;
      }
      default:
        u[k] = theta[i];
    }
  }
}

double Spec::dconstrain(int k, double u) const {
  int t = params[trainable[k]].transform;
  return (t == WV_T_IDENTITY) ? 1.0 : sigmoid(u);
}

double Spec::log_prior(const double* theta, double* grad, double scale_override) const {
  double tot = 0;
  if (grad) std::memset(grad, 0, sizeof(double) * n_params());
  for (int i = 0; i < n_params(); ++i) {
    const auto& p = params[i];
    if (p.trainable && p.prior_scale > 0) {
      double d = 0;
      tot += horseshoe_logpdf(theta[i], scale_override > 0 ? scale_override : p.prior_scale, grad ? &d : nullptr);
      if (grad) grad[i] = d;
    }
  }
  return tot;
}

}  // namespace wv

// ---------------- C-ABI: host-only entry points ----------------
using namespace wv;


namespace wv {
int block_column(const Spec& s) {
  uint32_t common = ~0u;
  bool any = false;
  for (int t = 0; t < s.n_terms(); ++t) {
    uint32_t cats = 0;
    bool zero = false;
    for (int f = s.term_start[t]; f < s.term_start[t + 1]; ++f) {
      if (s.factors[f].type == WV_K_EMPTY) zero = true;
      if (s.factors[f].type == WV_K_CAT) cats |= s.factors[f].dim_mask;
    }
    if (zero) continue;  // the term contributes nothing
    common &= cats;
    any = true;
  }
  if (!any || common == 0) return -1;
  return __builtin_ctz(common);
}

void block_spans(const double* X, int64_t N, int D, int col, int64_t Npad, std::vector<int32_t>& spans) {
  // labels as the device compares them: rint(x); NaN never equals anything, so such a row pairs with itself only
  std::unordered_map<long long, std::pair<int32_t, int32_t>> range;  // label -> first, last row
  std::vector<long long> label(N);
  std::vector<char> lone(N, 0);
  for (int64_t i = 0; i < N; ++i) {
    const double x = X[i * D + col];
    if (!(std::fabs(x) < 9e15)) {
      lone[i] = 1;
      continue;
    }
    label[i] = std::llrint(x);
    auto it = range.find(label[i]);
    if (it == range.end()) range.emplace(label[i], std::make_pair((int32_t)i, (int32_t)i));
    else it->second.second = (int32_t)i;
  }
  const int64_t ng = Npad / 64;
  spans.assign(2 * ng, 0);
  for (int64_t g = 0; g < ng; ++g) {
    int32_t lo = INT32_MAX, hi = 0;
    for (int64_t i = 64 * g; i < 64 * (g + 1); ++i) {
      int32_t a = (int32_t)i, b = (int32_t)i;
      if (i < N && !lone[i]) {
        const auto& r = range[label[i]];
        a = r.first;
        b = r.second;
      }
      lo = std::min(lo, a);
      hi = std::max(hi, b + 1);
    }
    spans[2 * g] = lo / 16 * 16;
    spans[2 * g + 1] = (int32_t)std::min<int64_t>(Npad, (hi + 15) / 16 * 16);
  }
}
}  // namespace wv

extern "C" {

int32_t wv_abi_version(void) { return WV_ABI_VERSION; }
const char* wv_last_error(void) { return g_err.c_str(); }

void wv_lbfgs_default_opts(wv_lbfgs_opts* o) {
  if (!o) return;
  o->m = 10;
  o->maxiter = 50000;  // waveome/model_classes.py:310-313
  o->maxfun = 50000;
  o->maxls = 20;
  o->ftol = 2.220446049250313e-09;
  o->gtol = 1e-05;
}

int32_t wv_spec_block_column(const wv_kernel_spec* spec, int32_t* column_out) {
  Spec s;
  std::string e = Spec::parse(spec, s);
  if (!e.empty()) {
    set_error(e);
    return WV_E_SPEC;
  }
  if (!column_out) {
    set_error("null argument");
    return WV_E_BADARG;
  }
  *column_out = block_column(s);
  return WV_OK;
}

int32_t wv_block_spans(const double* X, int64_t N, int32_t D, int32_t column, int32_t* spans_out) {
  if (!X || !spans_out || N < 1 || D < 1 || column < 0 || column >= D) {
    set_error("wv_block_spans: bad argument");
    return WV_E_BADARG;
  }
  std::vector<int32_t> sp;
  block_spans(X, N, D, column, (N + 63) / 64 * 64, sp);
  std::memcpy(spans_out, sp.data(), sizeof(int32_t) * sp.size());
  return WV_OK;
}

int32_t wv_spec_check(const wv_kernel_spec* spec, int32_t* n_trainable_out) {
  Spec s;
  std::string e = Spec::parse(spec, s);
  if (!e.empty()) {
    set_error(e);
    return WV_E_SPEC;
  }
  if (n_trainable_out) *n_trainable_out = s.n_trainable();
  return WV_OK;
}

int32_t wv_spec_constrain(const wv_kernel_spec* spec, const double* u, double* theta_out) {
  Spec s;
  std::string e = Spec::parse(spec, s);
  if (!e.empty()) {
    set_error(e);
    return WV_E_SPEC;
  }
  if (!u || !theta_out) {
    set_error("null argument");
    return WV_E_BADARG;
  }
  s.constrain(u, theta_out);
  return WV_OK;
}

int32_t wv_spec_unconstrain(const wv_kernel_spec* spec, const double* theta, double* u_out) {
  Spec s;
  std::string e = Spec::parse(spec, s);
  if (!e.empty()) {
    set_error(e);
    return WV_E_SPEC;
  }
  if (!theta || !u_out) {
    set_error("null argument");
    return WV_E_BADARG;
  }
  s.unconstrain(theta, u_out);
  return WV_OK;
}

int32_t wv_spec_log_prior(const wv_kernel_spec* spec, const double* theta, double* value_out,
                          double* grad_out) {
  Spec s;
  std::string e = Spec::parse(spec, s);
  if (!e.empty()) {
    set_error(e);
    return WV_E_SPEC;
  }
  if (!theta || !value_out) {
    set_error("null argument");
    return WV_E_BADARG;
  }
  *value_out = s.log_prior(theta, grad_out);
  return WV_OK;
}

struct wv_lbfgs {
  Lbfgs impl;
  wv_lbfgs(int n, const wv_lbfgs_opts& o, const double* x0) : impl(n, o, x0) {}
};

int32_t wv_lbfgs_create(int32_t n, const wv_lbfgs_opts* opts, const double* x0, wv_lbfgs** out) {
  if (n < 1 || !opts || !x0 || !out) {
    set_error("wv_lbfgs_create: bad argument");
    return WV_E_BADARG;
  }
  *out = new wv_lbfgs(n, *opts, x0);
  return WV_OK;
}
const double* wv_lbfgs_x(const wv_lbfgs* s) { return s->impl.x(); }
int32_t wv_lbfgs_feed(wv_lbfgs* s, double f, const double* g) { return s->impl.feed(f, g); }
int32_t wv_lbfgs_n_iter(const wv_lbfgs* s) { return s->impl.n_iter(); }
int32_t wv_lbfgs_n_eval(const wv_lbfgs* s) { return s->impl.n_eval(); }
double wv_lbfgs_f(const wv_lbfgs* s) { return s->impl.f(); }
void wv_lbfgs_destroy(wv_lbfgs* s) { delete s; }

}  // extern "C"
