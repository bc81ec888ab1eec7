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

// ---- block structure (designs in which every additive term carries the same Categorical factor) ----
// The column c such that every term that is not identically zero has a Categorical factor on column c, or -1.  Then
// K_y[i][j] = 0 whenever rint(x_ic) != rint(x_jc): K_y, its Cholesky factor, the factor's inverse and K_y^-1 are all
// zero outside the blocks of equal labels, and the O(N^3) products only need the k inside them (wv_gemm.hip).
int block_column(const Spec& s);
// spans[2 g], spans[2 g + 1]: for the 64-row granule g of the padded index range, the half-open range [lo, hi) of row
// indices that can share a label of column `col` with any row of the granule (first to last occurrence of each of its
// labels), rounded outwards to multiples of 16; padding rows (identity rows of K_y) only pair with themselves.  Rows
// sorted by the column give tight ranges, scattered labels wide ones - correct either way.
void block_spans(const double* X, int64_t N, int D, int col, int64_t Npad, std::vector<int32_t>& spans);

int n_factor_params(int type);
double softplus(double u);
double inv_softplus(double x);
double sigmoid(double u);
double horseshoe_logpdf(double x, double scale, double* dlogp);

void set_error(const std::string& msg);
void fail_launch(const std::string& msg);
bool take_launch_failure();

}  // namespace wv
