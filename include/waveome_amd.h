/* waveome_amd.h - C-ABI of the MI355X-native GP fitting core.
 *
 * Drop-in boundary for ONE path of omicsEye/waveome: the per-output Gaussian
 * marginal-likelihood optimisation that `GPSearch.penalized_optimization()`
 * runs per Y column (reference waveome/model_search.py:197-517).  The reference
 * has no FFI of its own (it is pure Python over GPflow/TensorFlow); each entry
 * point below names the reference interface it replaces.  INTEGRATION.md shows
 * the ctypes stub a waveome maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy types.
 *   - matrices are row-major, contiguous, float64.
 *   - X is N x D.  Y is P x N ("one contiguous row per output column").
 *   - the caller owns every host buffer; the library copies X / Y to the device
 *     at problem creation and owns all device memory.
 *   - return value 0 = WV_OK, negative = WV_E_*; wv_last_error() gives a
 *     thread-local message.  Numerical failure of ONE fit (non-PD Gram matrix)
 *     is not an error: it is reported per task in status arrays and the
 *     objective is +inf (the reference lets TF raise InvalidArgumentError and
 *     retries, waveome/model_classes.py:323-340).
 *   - a wv_ctx is re-entrant against other contexts, not thread-safe within
 *     one context.  No device allocation happens inside wv_value_grad / wv_fit.
 */
#ifndef WAVEOME_AMD_H
#define WAVEOME_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WV_ABI_VERSION 2

typedef struct wv_ctx wv_ctx;
typedef struct wv_problem wv_problem;
typedef struct wv_lbfgs wv_lbfgs;

enum {
  WV_OK = 0,
  WV_E_BADARG = -1,   /* null pointer, size out of range */
  WV_E_SPEC = -2,     /* malformed kernel spec */
  WV_E_NODEVICE = -3, /* no gfx950 device visible */
  WV_E_HIP = -4,      /* HIP runtime error (message in wv_last_error) */
  WV_E_OOM = -5,      /* device allocation failed */
  WV_E_UNSUPPORTED = -6
};

/* Factor types.  Formulas: SURVEY.md Appendix A.
 * SE/M12/M32/M52/PERIODIC/CONST replace gpflow.kernels.{SquaredExponential,
 * Matern12,Matern32,Matern52,Periodic(SquaredExponential),Constant}.K;
 * LIN/POLY/CAT/EMPTY replace waveome/kernels.py:19-32, :56-73, :95-117, :136-139. */
enum {
  WV_K_SE = 0,
  WV_K_M12 = 1,
  WV_K_M32 = 2,
  WV_K_M52 = 3,
  WV_K_PERIODIC = 4, /* params: variance, lengthscales, period */
  WV_K_LIN = 5,      /* params: variance */
  WV_K_POLY = 6,     /* params: variance, offset; integer degree */
  WV_K_CONST = 7,    /* params: variance */
  WV_K_CAT = 8,      /* params: variance; 1[rint(x_i) == rint(x_j)] on one column */
  WV_K_EMPTY = 9     /* no params; identically zero */
};

/* Parameter transforms (gpflow.utilities.positive() = Softplus; Gaussian
 * likelihood variance = Softplus + Shift(1e-6); Constant mean = identity). */
enum { WV_T_IDENTITY = 0, WV_T_SOFTPLUS = 1, WV_T_SOFTPLUS_SHIFT = 2 };

#define WV_MAX_DIMS 32
#define WV_MAX_PARAMS 256

typedef struct {
  int32_t type;      /* WV_K_* */
  uint32_t dim_mask; /* bit d set <=> column d of X is active (gpflow active_dims) */
  int32_t param[3];  /* indices into wv_kernel_spec.params; -1 = unused */
  int32_t degree;    /* WV_K_POLY only */
} wv_factor;

typedef struct {
  double init;         /* constrained initial value (GPflow default 1.0; mean 0.0) */
  double prior_scale;  /* Horseshoe scale = 1/penalization_factor
                          (waveome/model_classes.py:857-859); <= 0: no prior */
  int32_t transform;   /* WV_T_* */
  int32_t trainable;   /* 0 = frozen (e.g. Categorical variance inside a Product,
                          waveome/regularization.py:109,132) */
} wv_param;

/* K(x, x') = sum_t prod_{f in term t} k_f(x, x')   (gpflow Sum of Products,
 * waveome/regularization.py:112,139,166,181).  The objective maximised is
 *   LML(y | c, K + sigma^2 I) + sum_{trainable p with prior} log p_HS(theta_p)
 * (waveome/model_classes.py:817-864). */
typedef struct {
  int32_t n_dims;   /* D */
  int32_t n_params; /* <= WV_MAX_PARAMS */
  const wv_param* params;
  int32_t n_terms;
  const int32_t* term_start; /* n_terms + 1 offsets into factors */
  const wv_factor* factors;
  int32_t noise_param; /* index of the likelihood variance */
  int32_t mean_param;  /* index of the Constant mean c, or -1 (zero mean) */
} wv_kernel_spec;

/* SciPy L-BFGS-B options as gpflow.optimizers.Scipy passes them
 * (waveome/model_classes.py:310-315, 325-334).  Zero-initialise then call
 * wv_lbfgs_default_opts for SciPy's defaults with the reference's
 * maxiter = maxfun = 50000. */
typedef struct {
  int32_t m;       /* maxcor, 10 */
  int32_t maxiter; /* 50000 */
  int32_t maxfun;  /* 50000 */
  int32_t maxls;   /* 20 */
  double ftol;     /* 2.220446049250313e-09 */
  double gtol;     /* 1e-05 */
} wv_lbfgs_opts;

/* Fit termination codes (wv_fit status_out, wv_lbfgs_feed return value). */
enum {
  WV_LB_FG = 1,            /* evaluate f,g at wv_lbfgs_x() and feed again */
  WV_LB_CONV_GTOL = 0,     /* CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL */
  WV_LB_CONV_FTOL = 2,     /* CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH */
  WV_LB_STOP_MAXITER = 3,  /* STOP: TOTAL NO. of ITERATIONS REACHED LIMIT */
  WV_LB_STOP_MAXFUN = 4,   /* STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT */
  WV_LB_ABNORMAL = 5       /* ABNORMAL_TERMINATION_IN_LNSRCH */
};

int32_t wv_abi_version(void);
const char* wv_last_error(void);
void wv_lbfgs_default_opts(wv_lbfgs_opts* opts);

/* ---- host-only helpers (usable without a GPU) ------------------------- */

/* Validate a spec; n_trainable_out receives the length of the unconstrained
 * vector (trainable parameters in table order, the order
 * gpflow `trainable_variables` would be flattened in is irrelevant to L-BFGS). */
int32_t wv_spec_check(const wv_kernel_spec* spec, int32_t* n_trainable_out);
/* Block structure of longitudinal designs (the structures come from
 * waveome/regularization.py:103-149: unit x numeric and categorical x numeric terms).
 * column_out: the column c such that every additive term that is not identically zero
 * carries a Categorical factor on c, or -1.  For such a structure K_y vanishes outside
 * the blocks of equal labels of column c, and so do its Cholesky factor, the factor's
 * inverse and K_y^-1: the device products then only walk the k inside the blocks
 * (rows sorted by c give the saving; any order gives the same numbers). */
int32_t wv_spec_block_column(const wv_kernel_spec* spec, int32_t* column_out);
/* spans_out[2 g], spans_out[2 g + 1] for each 64-row granule g of ceil(N / 64) * 64 rows:
 * the half-open row range that can share a label of `column` with a row of the granule,
 * rounded outwards to multiples of 16 (what wv_problem_create uploads for such a structure). */
int32_t wv_block_spans(const double* X, int64_t N, int32_t D, int32_t column, int32_t* spans_out);
/* unconstrained trainables -> constrained full table (frozen entries = init). */
int32_t wv_spec_constrain(const wv_kernel_spec* spec, const double* u, double* theta_out);
/* constrained full table -> unconstrained trainables. */
int32_t wv_spec_unconstrain(const wv_kernel_spec* spec, const double* theta, double* u_out);
/* sum of Horseshoe log-densities over trainable params with a prior, and its
 * gradient w.r.t. the constrained values (grad_out may be NULL). Restates
 * tfd.Horseshoe.log_prob, call site waveome/model_classes.py:857-864. */
int32_t wv_spec_log_prior(const wv_kernel_spec* spec, const double* theta, double* value_out,
                          double* grad_out);

/* Reverse-communication L-BFGS-B (no bounds) - replaces
 * gpflow.optimizers.Scipy().minimize(method="L-BFGS-B")
 * (waveome/model_classes.py:325-334).  create -> x() -> feed(f,g) until the
 * return value is not WV_LB_FG. */
int32_t wv_lbfgs_create(int32_t n, const wv_lbfgs_opts* opts, const double* x0, wv_lbfgs** out);
const double* wv_lbfgs_x(const wv_lbfgs* s);
int32_t wv_lbfgs_feed(wv_lbfgs* s, double f, const double* g);
int32_t wv_lbfgs_n_iter(const wv_lbfgs* s);
int32_t wv_lbfgs_n_eval(const wv_lbfgs* s);
double wv_lbfgs_f(const wv_lbfgs* s);
void wv_lbfgs_destroy(wv_lbfgs* s);

/* ---- device path ------------------------------------------------------ */

int32_t wv_device_count(int32_t* count_out);
/* One context per GPU, process and precision.  dtype: 0 = float64; 1 = float32 storage and MFMA products
 * for the N x N work matrices of problems created on this context - the reference's
 * set_precision("float32") (waveome/utilities.py:34-49).  Parameters, X, Y, vectors, reductions and
 * every number crossing this ABI stay float64.  The float32 products accumulate through float64 sums and every value
 * (wv_lml, wv_value_grad, the fits) is corrected back to float64 through E = M K_y M' - I with K_y recomputed in
 * float64 (wv_chol.hip, reduce_kernel): 1e-7 relative to the float64 path at N = 16384 for cond(K_y) ~ 5e4; wv_lml
 * therefore costs a gradient evaluation on such a context.  Post-fit calls run on a float64 twin of the problem. */
int32_t wv_ctx_create(int32_t device_ordinal, int32_t dtype, wv_ctx** out);
void wv_ctx_destroy(wv_ctx* ctx);
/* Block until all work queued on this context's GPU has finished (hipDeviceSynchronize). */
int32_t wv_ctx_synchronize(wv_ctx* ctx);

/* Copies X (N x D) and Y (P x N) to HBM once, uploads the spec and sizes the
 * workspace for `max_batch` fits in flight (two N x N matrices per slot).
 * Replaces PSVGP.__init__ data staging, waveome/model_classes.py:123-140.
 * x_is_device / y_is_device != 0: the pointers are device pointers on this
 * context's GPU (e.g. buffers an RCCL broadcast filled); they are copied D2D.  * Limits of one structure on the device (WV_E_UNSUPPORTED otherwise): D <= 32 input columns, <= 64 additive
 * terms, <= 128 factors in total, <= 4 factors per product, <= 64 staged features (a Periodic factor uses
 * three per active column).  The reference's full_kernel_build and kernel search stay far inside them
 * (two-way products only, waveome/model_search.py:2463-2466). */
int32_t wv_problem_create(wv_ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y,
                          int32_t P, const wv_kernel_spec* spec, int32_t max_batch,
                          int32_t x_is_device, int32_t y_is_device, wv_problem** out);
void wv_problem_destroy(wv_problem* prob);
int64_t wv_problem_workspace_bytes(const wv_problem* prob);

/* Bring-up / parity: K_y = K_theta(X,X) + sigma^2 I for ONE constrained
 * parameter table, copied back as a full symmetric N x N host matrix.
 * Replaces kernel(X) + add_likelihood_noise_cov, waveome/model_types_DEPR.py:50-51. */
int32_t wv_gram(wv_problem* prob, const double* theta, double* K_out);

/* Exact LML at given constrained tables (no prior, no gradient) for B tasks.
 * Replaces PGPR.log_marginal_likelihood(penalize=False),
 * waveome/model_types_DEPR.py:49-56 (= gpflow GPR.log_marginal_likelihood). */
int32_t wv_lml(wv_problem* prob, const int32_t* col_idx, int32_t B, const double* theta,
               double* lml_out, int32_t* status_out);

/* Training loss -(LML + log prior) and its gradient in UNCONSTRAINED
 * coordinates for B tasks; theta_u is B x n_trainable.  Replaces the compiled
 * `training_loss_closure` + tape.gradient the SciPy wrapper calls
 * (waveome/model_classes.py:325-334).  status: 0 ok, 1 = Gram matrix not PD
 * (loss = +inf, gradient zeros). B <= max_batch. */
int32_t wv_value_grad(wv_problem* prob, const int32_t* col_idx, int32_t B, const double* theta_u,
                      double* loss_out, double* grad_out, int32_t* status_out);

/* Fit n_tasks independent (column, start point) tasks with L-BFGS-B, keeping up
 * to max_batch of them in flight on the GPU.  theta_u_inout is n_tasks x
 * n_trainable (start points in, optima out).  Replaces the per-output
 * `mod.optimize_params(optimizer="scipy")` calls that
 * GPSearch.penalized_optimization farms out through Ray
 * (waveome/model_search.py:250-329, 474-489).  Any out pointer may be NULL. */
int32_t wv_fit(wv_problem* prob, const int32_t* col_idx, int32_t n_tasks,
               const wv_lbfgs_opts* opts, double* theta_u_inout, double* loss_out, double* lml_out,
               int32_t* n_iter_out, int32_t* n_eval_out, int32_t* status_out);

/* wv_fit with a per-task Horseshoe scale (= 1 / penalization_factor of that output) overriding the
 * spec's prior scales; prior_scale = NULL is wv_fit.  Used by the iterated analytic penalisation
 * factor, which differs per output (waveome/model_search.py:273-299, 332-375). */
int32_t wv_fit_penalized(wv_problem* prob, const int32_t* col_idx, int32_t n_tasks,
                         const wv_lbfgs_opts* opts, const double* prior_scale, double* theta_u_inout,
                         double* loss_out, double* lml_out, int32_t* n_iter_out, int32_t* n_eval_out,
                         int32_t* status_out);

/* Several kernel structures on the same (X, Y): one workspace and ONE work queue for tasks that differ in
 * structure as well as in column - the "Y columns x candidate kernel structures" sweep that the reference
 * runs as one kernel_test per candidate (waveome/model_search.py:2239-2334, 2347-2558) and one Ray task
 * per output (:1200-1222).  Every batch slot carries its task's structure index, so a launch mixes
 * structures freely; only the Gram assembly and the gradient trace read the structure.
 * wv_fit_multi: spec_idx[t] selects the structure of task t; theta_u_inout is n_tasks x u_stride
 * (u_stride >= the largest number of trainable parameters; a task uses the first n_trainable of its row).
 * A multi-structure problem is served by wv_fit_multi only (the single-structure entry points answer
 * WV_E_UNSUPPORTED / WV_E_BADARG). */
int32_t wv_problem_create_multi(wv_ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y, int32_t P,
                                const wv_kernel_spec* const* specs, int32_t n_specs, int32_t max_batch,
                                int32_t x_is_device, int32_t y_is_device, wv_problem** out);
int32_t wv_fit_multi(wv_problem* prob, const int32_t* col_idx, const int32_t* spec_idx, int32_t n_tasks,
                     const wv_lbfgs_opts* opts, const double* prior_scale, int32_t u_stride,
                     double* theta_u_inout, double* loss_out, double* lml_out, int32_t* n_iter_out,
                     int32_t* n_eval_out, int32_t* status_out);

/* ---- a work queue that accepts tasks while it runs ------------------------------------------------------------
 * The reference's stepwise kernel search runs one Ray task per output column, each walking its own search tree
 * and calling kernel_test once per candidate (waveome/model_search.py:1200-1222, 2987-3273): columns never wait
 * for each other.  wv_fit_multi takes a FIXED task list, so a caller pooling the candidates of all columns has to
 * advance them level by level.  wv_queue_* expose the same work queue incrementally: submit (column, structure,
 * start point) tasks at any time, call wv_queue_advance to run one scheduling step (one batched evaluation round of
 * one batch group), fetch finished fits, submit what they unlock.  New kernel structures are added to the problem
 * with wv_problem_add_spec; the problem must have been created with room for them (wv_problem_create_ex).
 * Fits are bit-identical to wv_fit_multi's: a batch slot's arithmetic does not depend on its neighbours.
 * One queue per problem at a time, and enforced: while a queue is open the groups' staging buffers and slots are
 * the queue's - a second wv_queue_create and every other device entry point on that problem (wv_fit*, wv_lml,
 * wv_value_grad, wv_gram, wv_predict, wv_whitened_means, wv_whitened_state) answer WV_E_BADARG until wv_queue_destroy; wv_problem_add_spec
 * stays allowed.  Not thread-safe (like the problem itself). */
typedef struct wv_queue wv_queue;
typedef struct {
  int32_t max_specs;   /* kernel structures the problem can hold */
  int32_t max_params;  /* parameters of one structure (kernel parameters + noise + mean) */
  int32_t max_feat;    /* staged feature slots of one structure (<= 64) */
  int32_t max_factors; /* kernel factors of one structure (<= 128) */
  int32_t max_red;     /* gradient sums of one structure: 1-3 per trainable factor, + 1 */
  int32_t nan_is_missing; /* != 0: a NaN in column p of Y marks that row as missing FOR OUTPUT p: its fits, LML and
                             gradients are those of the complete cases of that output (what the reference's search
                             does per output, waveome/model_search.py:3034-3043), while all outputs keep sharing one
                             batch.  0 (default): NaN is data and makes every evaluation of that column non-finite.
                             wv_predict and wv_whitened_means work on the complete cases of the column they are asked
                             about (see there); wv_gram and wv_whitened_state answer WV_E_UNSUPPORTED on such a problem. */
} wv_problem_caps;
/* wv_problem_create_multi with explicit capacities (each at least what `specs` need; caps = NULL: exactly that). */
int32_t wv_problem_create_ex(wv_ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y, int32_t P,
                             const wv_kernel_spec* const* specs, int32_t n_specs, const wv_problem_caps* caps,
                             int32_t max_batch, int32_t x_is_device, int32_t y_is_device, wv_problem** out);
/* Adds a structure to a problem; *index_out is its spec_idx.  WV_E_UNSUPPORTED when it exceeds a capacity. */
int32_t wv_problem_add_spec(wv_problem* prob, const wv_kernel_spec* spec, int32_t* index_out);
int32_t wv_queue_create(wv_problem* prob, const wv_lbfgs_opts* opts, wv_queue** out);
/* theta_u0 is n x u_stride (a task uses the first n_trainable of its row); prior_scale may be NULL; tags are the
 * caller's handles, returned by wv_queue_fetch. */
int32_t wv_queue_submit(wv_queue* q, int32_t n, const int32_t* col_idx, const int32_t* spec_idx,
                        const double* prior_scale, int32_t u_stride, const double* theta_u0, const int64_t* tags);
/* One scheduling step (returns at once when nothing is queued or in flight).  n_done_out: finished fits waiting to
 * be fetched; n_in_system_out: tasks queued or being fitted. */
int32_t wv_queue_advance(wv_queue* q, int32_t* n_done_out, int32_t* n_in_system_out);
/* Up to cap finished fits, oldest first; theta_u_out is cap x u_stride.  If a fit is wider than u_stride the call
 * returns WV_E_BADARG with *n_out = the fits delivered before it (they are the caller's); the wide one stays queued. */
int32_t wv_queue_fetch(wv_queue* q, int32_t cap, int32_t u_stride, int64_t* tags_out, double* theta_u_out,
                       double* loss_out, double* lml_out, int32_t* n_iter_out, int32_t* n_eval_out,
                       int32_t* status_out, int32_t* n_out);
void wv_queue_destroy(wv_queue* q);

/* Posterior prediction for ONE task at constrained table theta: mean and variance of f (and of
 * y = f + noise when include_noise != 0).  Replaces predict_f / predict_y
 * (waveome/model_classes.py:526-544) for the exact GP.  Xnew = NULL: the N training inputs
 * (model_search.py:340, utilities.py:629) via mean_f = y - sigma^2 alpha, var_f = sigma^2 -
 * sigma^4 diag(K_y^-1); otherwise Xnew is a host M x D matrix and mean_out / var_out hold M doubles
 * (var_f = k(x*,x*) - |L^-1 k(X, x*)|^2, full_cov = False).  Either output may be NULL.  This call
 * allocates temporary device memory (it is not on the optimisation loop).
 * Post-fit calls are float64, complete-case computations whatever the problem: on a float32 problem, or one created
 * with nan_is_missing, wv_predict / wv_whitened_means (and wv_whitened_state, float32 only) run on a float64
 * single-column twin that the problem builds from its own device copies on first use and keeps (the workspace of up to
 * 8 batch slots, freed with the problem; a change of column re-uses it, or rebuilds it when rows are missing).  With
 * missing outputs the training data of column col are the rows observed for it; Xnew = NULL then still answers all N
 * rows - a row missing for this column gets the posterior at its input like any new input. */
int32_t wv_predict(wv_problem* prob, int32_t col, const double* theta, const double* Xnew, int64_t M,
                   int32_t include_noise, double* mean_out, double* var_out);

/* Means the reference's whitened SVGP (inducing points = X, q_mu at its optimum) predicts at the
 * training inputs when the kernel is replaced by the sub-sum of additive terms in term_masks[m]
 * (bit t = term t of the spec) while q_mu is kept:  mean_m = chol(K_m + jitter I) q_mu + c,
 * q_mu = chol(K_all + jitter I)' alpha.  This is what `model_copy.kernel.kernels.pop(k)` followed by
 * `predict_y` computes in calc_feature_importance_components (waveome/utilities.py:614-707);
 * jitter = gpflow's default 1e-6.  means_out is n_masks x N (rows missing for this column: NaN). */
int32_t wv_whitened_means(wv_problem* prob, int32_t col, const double* theta, const uint64_t* term_masks,
                          int32_t n_masks, double jitter, double* means_out);

/* The variational state the reference's model objects carry next to theta (PSVGP.q_mu / PSVGP.q_sqrt, whitened,
 * inducing points = X: waveome/model_classes.py:99-100, 149-155), at its optimum for the given theta - what the
 * reference's optimiser converges to and what code downstream of a fit reads off the model:
 *   q_mu   = L' alpha,                         L = chol(K + jitter I),  alpha = K_y^-1 (y - c)
 *   q_sqrt = chol((I + L'L / sigma^2)^-1)      (= chol(I - L' K_y^-1 L))
 * q_mu_out holds N doubles; q_sqrt_out N x N doubles, row-major, lower triangular (strictly upper part zero), or NULL to
 * skip the two extra factorisations it costs.  Single-structure problems without missing outputs (float32 ones: on their float64 twin); not on the
 * optimisation loop. */
int32_t wv_whitened_state(wv_problem* prob, int32_t col, const double* theta, double jitter, double* q_mu_out,
                          double* q_sqrt_out);

/* Timing of the device stages of the most recent wv_value_grad / wv_fit on this
 * problem, measured with HIP events on the stream the kernels ran on.
 * names_out receives up to `cap` pointers to static strings; ms_out and
 * launches_out the accumulated milliseconds / launch counts.  Returns the
 * number of stages. */
int32_t wv_stage_times(wv_problem* prob, int32_t cap, const char** names_out, double* ms_out,
                       int64_t* launches_out, double* flops_out, double* bytes_out);
void wv_stage_times_reset(wv_problem* prob);
/* on = 0: off; 1: every stage; otherwise a bitmask with bit (s + 1) enabling stage s
 * (stage order as returned by wv_stage_times). Event pairs cost ~2 us per timed launch. */
void wv_stage_timing_enable(wv_problem* prob, int32_t on);

#ifdef __cplusplus
}
#endif
#endif /* WAVEOME_AMD_H */
