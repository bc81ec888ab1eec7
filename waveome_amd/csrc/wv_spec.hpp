// Host-side kernel-spec handling: validation, parameter transforms, Horseshoe prior.
// Mirrors what GPflow's Parameter / prior machinery does for the reference's
// training loss (waveome/model_classes.py:817-864); formulas restated in
// oracle/gp_oracle.py and SURVEY.md Appendix A.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/waveome_amd.h"

namespace wv {

constexpr double kNoiseFloor = 1e-6;  // gpflow Gaussian likelihood variance lower bound

struct Spec {
  int n_dims = 0;
  std::vector<wv_param> params;
  std::vector<int32_t> term_start;
  std::vector<wv_factor> factors;
  int noise_param = -1, mean_param = -1;
  std::vector<int> trainable;  // indices of trainable params, table order

  int n_params() const { return (int)params.size(); }
  int n_terms() const { return (int)term_start.size() - 1; }
  int n_trainable() const { return (int)trainable.size(); }

  // returns empty string when valid
  static std::string parse(const wv_kernel_spec* s, Spec& out);
  void constrain(const double* u, double* theta) const;
  void unconstrain(const double* theta, double* u) const;
  // d theta_p / d u for trainable k at unconstrained value u_k
  double dconstrain(int k, double u) const;
  // scale_override > 0 replaces every prior's scale (per-task penalisation factors)
  double log_prior(const double* theta, double* grad /*n_params or null*/, double scale_override = -1.0) const;
};

int n_factor_params(int type);
double softplus(double u);
double inv_softplus(double x);
double sigmoid(double u);
double horseshoe_logpdf(double x, double scale, double* dlogp);

void set_error(const std::string& msg);
void fail_launch(const std::string& msg);
bool take_launch_failure();

}  // namespace wv
