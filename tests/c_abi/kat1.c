/* The drop-in boundary used from plain C: include/waveome_amd.h + libwaveome_amd.so, nothing else.
 *
 *   kat1 --host   host-only entry points (no GPU needed): ABI version, spec validation and transforms, the
 *                 reverse-communication L-BFGS-B on a quadratic, and wv_ctx_create failing loudly without a device
 *   kat1          KAT-1 (kernel_learning/gpflow_basic_colab.ipynb cells 3-7, 13): SquaredExponential GPR on the ten
 *                 seeded points, wv_fit from the GPflow defaults with maxiter = 100; the fitted parameters are the
 *                 notebook's at full precision, the LML its stored -9.914289155637
 *
 * Built and run by tests/test_c_abi.py (gcc -std=c99 -pedantic -Wall -Werror: the header is valid C, not only C++). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "waveome_amd.h"

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int32_t rc_ = (call);                                                               \
    if (rc_ != WV_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, wv_last_error());              \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

/* the ten seeded points of the notebook (np.random.seed(9102); X = uniform(0, 10, 10); Y = sin(X) + normal(0, 0.5, 10)):
 * read from the text file tests/test_c_abi.py writes out of tests/golden/kat.json - ten X values, then ten Y values */
static double X1[10];
static double Y1[10];

static int fill_spec(wv_param* pr, wv_factor* fa, int32_t* ts, wv_kernel_spec* sp) {
  memset(pr, 0, 3 * sizeof(*pr));
  pr[0].init = 1.0; pr[0].transform = WV_T_SOFTPLUS; pr[0].trainable = 1;        /* variance */
  pr[1].init = 1.0; pr[1].transform = WV_T_SOFTPLUS; pr[1].trainable = 1;        /* lengthscales */
  pr[2].init = 1.0; pr[2].transform = WV_T_SOFTPLUS_SHIFT; pr[2].trainable = 1;  /* likelihood variance */
  fa->type = WV_K_SE; fa->dim_mask = 1u; fa->param[0] = 0; fa->param[1] = 1; fa->param[2] = -1; fa->degree = 0;
  ts[0] = 0; ts[1] = 1;
  sp->n_dims = 1; sp->n_params = 3; sp->params = pr; sp->n_terms = 1; sp->term_start = ts; sp->factors = fa;
  sp->noise_param = 2; sp->mean_param = -1;
  return 0;
}

static int host_only(void) {
  wv_param pr[3]; wv_factor fa; int32_t ts[2]; wv_kernel_spec sp;
  int32_t ntr = 0;
  double u[3], th[3];
  const double theta[3] = {0.5, 2.0, 0.1};
  wv_lbfgs_opts op;
  wv_lbfgs* s = NULL;
  double x0[2] = {3.0, -2.0};
  int32_t task;
  wv_ctx* ctx = NULL;
  int32_t rc;

  if (wv_abi_version() != WV_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  fill_spec(pr, &fa, ts, &sp);
  CHECK(wv_spec_check(&sp, &ntr));
  if (ntr != 3) return 1;
  CHECK(wv_spec_unconstrain(&sp, theta, u));
  CHECK(wv_spec_constrain(&sp, u, th));
  if (fabs(th[0] - 0.5) > 1e-15 || fabs(th[1] - 2.0) > 1e-14 || fabs(th[2] - 0.1) > 1e-15) return 1;
  sp.noise_param = 7;                                   /* malformed: must be refused with a message */
  if (wv_spec_check(&sp, &ntr) != WV_E_SPEC || strlen(wv_last_error()) == 0) return 1;
  /* reverse-communication L-BFGS-B (what wv_fit runs per task) on f = (x - 1)^2 + 10 (y + 0.5)^2 */
  memset(&op, 0, sizeof(op));
  wv_lbfgs_default_opts(&op);
  if (op.m != 10 || op.maxiter != 50000 || op.maxls != 20) return 1;
  CHECK(wv_lbfgs_create(2, &op, x0, &s));
  task = WV_LB_FG;
  while (task == WV_LB_FG) {
    const double* x = wv_lbfgs_x(s);
    double g[2], f;
    f = (x[0] - 1.0) * (x[0] - 1.0) + 10.0 * (x[1] + 0.5) * (x[1] + 0.5);
    g[0] = 2.0 * (x[0] - 1.0); g[1] = 20.0 * (x[1] + 0.5);
    task = wv_lbfgs_feed(s, f, g);
  }
  if ((task != WV_LB_CONV_GTOL && task != WV_LB_CONV_FTOL) || fabs(wv_lbfgs_x(s)[0] - 1.0) > 1e-5 ||
      fabs(wv_lbfgs_x(s)[1] + 0.5) > 1e-5) return 1;
  printf("host L-BFGS-B: %d iterations, %d evaluations\n", (int)wv_lbfgs_n_iter(s), (int)wv_lbfgs_n_eval(s));
  wv_lbfgs_destroy(s);
  rc = wv_ctx_create(0, 0, &ctx);
  if (rc == WV_OK) { wv_ctx_destroy(ctx); printf("a device is present\n"); }
  else if (rc != WV_E_NODEVICE || strlen(wv_last_error()) == 0) return 1;
  else printf("no device: %s\n", wv_last_error());
  return 0;
}

int main(int argc, char** argv) {
  /* the notebook's outputs: cell 13 (parameters at full precision), cell 7 (LML) */
  const double want[3] = {0.7498722077802612, 0.7401937019825038, 0.03797644835355126};
  const double want_lml = -9.914289155637;
  wv_param pr[3]; wv_factor fa; int32_t ts[2]; wv_kernel_spec sp;
  wv_ctx* ctx = NULL;
  wv_problem* prob = NULL;
  wv_lbfgs_opts op;
  const int32_t col = 0;
  double u[3], th[3], loss, lml, init[3] = {1.0, 1.0, 1.0};
  int32_t nit, nfev, status, i;
  FILE* fh;

  if (argc > 1 && strcmp(argv[1], "--host") == 0) return host_only();
  fh = fopen(argc > 1 ? argv[1] : "kat1_xy.txt", "r");
  if (!fh) { fprintf(stderr, "usage: kat1 <file with the ten X values, then the ten Y values>\n"); return 2; }
  for (i = 0; i < 10; ++i) if (fscanf(fh, "%lf", &X1[i]) != 1) return 2;
  for (i = 0; i < 10; ++i) if (fscanf(fh, "%lf", &Y1[i]) != 1) return 2;
  fclose(fh);
  fill_spec(pr, &fa, ts, &sp);
  CHECK(wv_ctx_create(0, 0, &ctx));
  CHECK(wv_problem_create(ctx, X1, 10, 1, Y1, 1, &sp, 1, 0, 0, &prob));
  memset(&op, 0, sizeof(op));
  wv_lbfgs_default_opts(&op);
  op.maxiter = 100;                                       /* Scipy().minimize(..., options=dict(maxiter=100)) */
  CHECK(wv_spec_unconstrain(&sp, init, u));
  CHECK(wv_fit(prob, &col, 1, &op, u, &loss, &lml, &nit, &nfev, &status));
  CHECK(wv_spec_constrain(&sp, u, th));
  printf("KAT-1 from C: variance %.16g lengthscales %.16g noise %.16g  LML %.13g  (%d iterations, %d evaluations, status %d)\n",
         th[0], th[1], th[2], lml, (int)nit, (int)nfev, (int)status);
  wv_problem_destroy(prob);
  wv_ctx_destroy(ctx);
  for (i = 0; i < 3; ++i)
    if (fabs(th[i] / want[i] - 1.0) > 1e-9) { fprintf(stderr, "parameter %d off\n", (int)i); return 1; }
  if (fabs(lml - want_lml) > 5e-12) { fprintf(stderr, "LML off\n"); return 1; }
  return 0;
}
