// C-ABI device entry points (include/waveome_amd.h): contexts, problems, batched evaluation
// and the batched L-BFGS-B fit driver.  Host logic only - the kernels live in wv_gram.hip,
// wv_gemm.hip and wv_chol.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>

#include "wv_device.hpp"
#include "wv_lbfgs.hpp"

namespace wv {

// The environment switches of the release library (see wv_device.hpp).  Every one selects between paths that give the
// same bits (tests/test_gpu_parity.py::test_every_switch_of_the_release_library_gives_the_same_bits) except where noted.
static const char* const kTuningSwitches[] = {
    "WV_GROUPS",         // batch groups (streams) the fits in flight are split over; default 1
    "WV_BN_FILL",        // tiles a launch needs for the 128 x 128 tiling (default 4096)
    "WV_SMALL_FILL",     // ... for the 128 x 64 tiling before the small 64 x 64 tile (default 8192)
    "WV_PAIR",           // 0: trailing update and T as two launches instead of one (few fits in flight)
    "WV_TRI_SPLIT",      // 0: lower-only products as one launch (no diagonal-tile launch)
    "WV_TRI_SPLIT_MIN",  // diagonal tiles a launch needs to get its own diagonal-tile launch (default 256)
    "WV_BLOCKS",         // 0: no zero-block skipping (dense products on block-diagonal structures)
    "WV_VALUE_FULL",     // value-only evaluations through factor + inverse (same LML to rounding, not to the bit)
    "WV_FIT_TRACE",      // host-side timing of a fit loop on stderr (no effect on results)
    "WV_TRIM",           // 0: no partial-chunk streams beside a triangular operand's diagonal (full chunks over its zeros)
    "WV_F32_ORDER",      // float32 contexts: order of the value correction in E = Mt K_y Mt' - I (1 default; 2, 4: + probe passes)
};
const char* tuning_env(const char* name) {
  for (const char* s : kTuningSwitches)
    if (std::strcmp(s, name) == 0) return std::getenv(name);
  return nullptr;
}

const char* const kStageNames[ST_COUNT] = {
    "gram_assembly", "potrf_diag", "chol_trsm", "chol_panel_update", "chol_trailing_syrk",
    "trtri_tri_gemm", "trtri_diag_gemm", "lauum_gemm", "solve_matvec", "grad_trace", "reduce",
    "lauum_gemm_few_fits"};

// ---------------- stage timer ----------------
void StageTimer::begin(hipStream_t s, int stage) {
  cur = enabled && (mask >> stage & 1u);
  if (!cur) return;
  Rec r;
  auto get = [&]() {
    hipEvent_t e;
    if (!pool.empty()) {
      e = pool.back();
      pool.pop_back();
    } else {
      hipEventCreate(&e);
    }
    return e;
  };
  r.a = get();
  r.b = get();
  r.stage = stage;
  hipEventRecord(r.a, s);
  pending.push_back(r);
}
void StageTimer::end(hipStream_t s) {
  if (!cur || pending.empty()) return;
  cur = false;
  hipEventRecord(pending.back().b, s);
}
void StageTimer::collect() {
  for (auto& r : pending) {
    hipEventSynchronize(r.b);
    float t = 0;
    hipEventElapsedTime(&t, r.a, r.b);
    ms[r.stage] += t;
    launches[r.stage] += 1;
    pool.push_back(r.a);
    pool.push_back(r.b);
  }
  pending.clear();
}
void StageTimer::reset() {
  collect();
  for (int i = 0; i < ST_COUNT; ++i) ms[i] = 0, launches[i] = 0, flops[i] = 0, bytes[i] = 0;
}
StageTimer::~StageTimer() {
  for (auto& r : pending) {
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  for (auto e : pool) hipEventDestroy(e);
}

// ---------------- device spec ----------------
std::string build_devspec(const Spec& s, DevSpec& d) {
  std::memset(&d, 0, sizeof(d));
  if (s.n_terms() > 64) return "more than 64 additive terms";
  if ((int)s.factors.size() > 128) return "more than 128 factors";
  d.n_terms = s.n_terms();
  d.n_factors = (int)s.factors.size();
  d.n_params = s.n_params();
  d.noise_param = s.noise_param;
  d.mean_param = s.mean_param;
  d.n_dims = s.n_dims;
  for (int t = 0; t <= s.n_terms(); ++t) d.term_start[t] = s.term_start[t];
  for (int t = 0; t < s.n_terms(); ++t)
    if (s.term_start[t + 1] - s.term_start[t] > MAX_TERM_F) return "more than 4 factors in a product";
  int nfeat = 0, nslot = 0, nred = 0;
  auto feat = [&](int kind, int dim, int param, int sparam) -> int {
    for (int i = 0; i < nfeat; ++i)
      if (d.feat[i].kind == kind && d.feat[i].dim == dim && d.feat[i].param == param && d.feat[i].sparam == sparam)
        return i;
    if (nfeat >= MAX_FEAT) return -1;
    d.feat[nfeat] = FeatOp{kind, dim, param, sparam};
    return nfeat++;
  };
  for (size_t fi = 0; fi < s.factors.size(); ++fi) {
    const wv_factor& f = s.factors[fi];
    DevFactor& o = d.factors[fi];
    o.type = f.type;
    o.degree = f.degree;
    o.slot0 = nslot;
    o.ndims = 0;
    int np = n_factor_params(f.type);
    for (int k = 0; k < 3; ++k) {
      o.p[k] = k < np ? f.param[k] : 0;
      if (k < np && s.params[f.param[k]].trainable) o.trainable |= (1 << k);
    }
    o.nsum = f.type == WV_K_PERIODIC ? 3 : ((f.type <= WV_K_M52 || f.type == WV_K_POLY) ? 2 : (np > 0 ? 1 : 0));
    o.sbase = -1;
    if (o.trainable) {
      o.sbase = nred;
      nred += o.nsum;
    }
    if (f.type == WV_K_EMPTY || f.type == WV_K_CONST) continue;
    for (int dim = 0; dim < s.n_dims; ++dim) {
      if (!(f.dim_mask & (1u << dim))) continue;
      int need = f.type == WV_K_PERIODIC ? 3 : 1;
      if (nslot + need > 512) return "feature slot table overflow";
      if (f.type == WV_K_PERIODIC) {
        // sin / cos carry the factor's lengthscale on the row side; the raw column is shared with every other factor
        int a = feat(FEAT_SINPI, dim, f.param[2], f.param[1]), b = feat(FEAT_COSPI, dim, f.param[2], f.param[1]),
            c = feat(FEAT_RAW, dim, -1, -1);
        if (a < 0 || b < 0 || c < 0) return "more than 64 staged features (too many periodic factors)";
        d.fslots[nslot++] = (int16_t)a;
        d.fslots[nslot++] = (int16_t)b;
        d.fslots[nslot++] = (int16_t)c;
      } else {
        int a = feat(f.type == WV_K_CAT ? FEAT_RINT : FEAT_RAW, dim, -1, -1);
        if (a < 0) return "more than 64 staged features";
        d.fslots[nslot++] = (int16_t)a;
      }
      o.ndims++;
    }
  }
  d.n_feat = nfeat;
  d.n_red = nred + 1;  // last slot: the noise variance
  return "";
}

// ---------------- evaluation pipeline ----------------
// Enqueue one batched evaluation for the first `nb` slots of group g; results land in the
// group's pinned h_out / h_status when g.done fires.
static int enqueue_eval(Problem& p, Group& g, int nb, bool with_grad) {
  hipStream_t st = g.stream;
  const int np = p.np_stride;
  WV_HIP_TRY(hipMemcpyAsync(p.dtheta + (int64_t)g.slot0 * np, g.h_theta, sizeof(double) * np * nb,
                            hipMemcpyHostToDevice, st));
  WV_HIP_TRY(hipMemcpyAsync(p.dcol + g.slot0, g.h_col, sizeof(int32_t) * nb, hipMemcpyHostToDevice, st));
  if (p.specs.size() > 1)  // (single-structure problems keep the all-zero table written at creation)
    WV_HIP_TRY(hipMemcpyAsync(p.dslotspec + g.slot0, g.h_spec, sizeof(int32_t) * nb, hipMemcpyHostToDevice, st));
  launch_resid(p, st, g.slot0, nb);
  launch_gram(p, st, g.slot0, nb, 0);
  static const bool old_sched = experiment_env("WV_SCHED_OLD") != nullptr;  // (round 1's right-looking schedule)
  static const bool full_value = tuning_env("WV_VALUE_FULL") != nullptr;   // value-only through factor + inverse
  // float32 contexts: the value is only a float64-grade number with the first-order correction that the gradient
  // pipeline's trace pass accumulates (reduce_kernel), so value-only evaluations run that pipeline too
  if (p.f32) with_grad = true;
  if (old_sched) {
    cholesky(p, st, g.slot0, nb);
    trtri(p, st, g.slot0, nb);
    launch_solve(p, st, g.slot0, nb, with_grad);
  } else if (with_grad || full_value) {
    factor_inverse(p, st, g.slot0, nb);
    launch_solve(p, st, g.slot0, nb, with_grad);
  } else {
    factor_value_only(p, st, g.slot0, nb);  // z comes out of the factorisation itself
  }
  if (with_grad) {
    lauum(p, st, g.slot0, nb);
    if (p.f32) launch_f32_refine(p, st, g.slot0, nb);   // (before the trace pass: it reads the float64 diagonal of Mt'Mt)
    launch_trace(p, st, g.slot0, nb);
  }
  launch_reduce(p, st, g.slot0, nb, with_grad);
  WV_HIP_TRY(hipMemcpyAsync(g.h_out, p.dout + (int64_t)g.slot0 * p.out_stride,
                            sizeof(double) * p.out_stride * nb, hipMemcpyDeviceToHost, st));
  WV_HIP_TRY(hipMemcpyAsync(g.h_status, p.dstatus + g.slot0, sizeof(int32_t) * nb,
                            hipMemcpyDeviceToHost, st));
  WV_HIP_TRY(hipEventRecord(g.done, st));
  WV_HIP_TRY(hipGetLastError());
  if (take_launch_failure()) return WV_E_UNSUPPORTED;  // message already set by the launcher
  return WV_OK;
}

static int wait_group(Problem& p, Group& g) {
  WV_HIP_TRY(hipEventSynchronize(g.done));
  if (p.timer.enabled) p.timer.collect();
  return WV_OK;
}

// Every C-ABI call that uses the groups' staging buffers or enqueues kernels starts here: a launcher's refusal left
// over from an earlier call must not fail this one, and while a work queue is open on the problem (wv_queue_*) the
// staging buffers and slots are the queue's.
static int enter_device_call(Problem& p, const char* who) {
  (void)take_launch_failure();
  if (p.open_queue) {
    set_error(std::string(who) + ": the problem has an open work queue (wv_queue_*); destroy the queue first");
    return WV_E_BADARG;
  }
  return WV_OK;
}
// ... and a launcher that refused a request (fail_launch: message already set) turns WV_OK into WV_E_UNSUPPORTED
// instead of handing back buffers that were never computed.
static int leave_device_call(int rc) {
  const bool refused = take_launch_failure();
  return (rc == WV_OK && refused) ? WV_E_UNSUPPORTED : rc;
}

// loss = -(LML + log prior), gradient w.r.t. unconstrained trainables, from a result record
static void loss_from_record(const Spec& s, const double* u, const double* theta, const double* rec,
                             int status, double* loss, double* grad_u, double* lml_out,
                             double prior_scale = -1.0) {
  const int nt = s.n_trainable();
  double lml = rec[0];
  bool ok = status == 0 && std::isfinite(lml);
  if (lml_out) *lml_out = ok ? lml : -std::numeric_limits<double>::infinity();
  if (!ok) {
    *loss = std::numeric_limits<double>::infinity();
    if (grad_u) std::fill(grad_u, grad_u + nt, 0.0);
    return;
  }
  std::vector<double> gp(s.n_params());
  double lp = s.log_prior(theta, grad_u ? gp.data() : nullptr, prior_scale);
  *loss = -(lml + lp);
  if (grad_u) {
    for (int k = 0; k < nt; ++k) {
      int i = s.trainable[k];
      grad_u[k] = -(rec[3 + i] + gp[i]) * s.dconstrain(k, u[k]);
    }
  }
}

}  // namespace wv

using namespace wv;

struct wv_ctx {
  Ctx c;
};
struct wv_problem {
  Problem p;
};

extern "C" {

int32_t wv_device_count(int32_t* count_out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    n = 0;
    (void)hipGetLastError();
  }
  if (count_out) *count_out = n;
  return WV_OK;
}

int32_t wv_ctx_create(int32_t device_ordinal, int32_t dtype, wv_ctx** out) {
  if (!out) {
    set_error("wv_ctx_create: out is NULL");
    return WV_E_BADARG;
  }
  if (dtype != 0 && dtype != 1) {
    set_error("wv_ctx_create: dtype must be 0 (float64) or 1 (float32 matrices)");
    return WV_E_UNSUPPORTED;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    (void)hipGetLastError();
    set_error("no HIP device visible (this library has no CPU fallback)");
    return WV_E_NODEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= n) {
    set_error("device ordinal out of range");
    return WV_E_BADARG;
  }
  WV_HIP_TRY(hipSetDevice(device_ordinal));
  auto* c = new wv_ctx();
  c->c.device = device_ordinal;
  c->c.dtype = dtype;
  if (hipGetDeviceProperties(&c->c.prop, device_ordinal) != hipSuccess) {
    delete c;
    set_error("hipGetDeviceProperties failed");
    return WV_E_HIP;
  }
  if (std::string(c->c.prop.gcnArchName).rfind("gfx950", 0) != 0) {
    std::string arch = c->c.prop.gcnArchName;
    delete c;
    set_error("device is " + arch + ", this library is built for gfx950 only");
    return WV_E_NODEVICE;
  }
  *out = c;
  return WV_OK;
}

void wv_ctx_destroy(wv_ctx* ctx) { delete ctx; }

int32_t wv_ctx_synchronize(wv_ctx* ctx) {
  if (!ctx) return WV_E_BADARG;
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  WV_HIP_TRY(hipDeviceSynchronize());
  return WV_OK;
}

void wv_problem_destroy(wv_problem* h) {
  if (!h) return;
  Problem& p = h->p;
  if (p.twin) wv_problem_destroy(p.twin);
  if (p.ctx) (void)hipSetDevice(p.ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& g : p.groups) {
    if (g.stream) hipStreamDestroy(g.stream);
    if (g.done) hipEventDestroy(g.done);
    if (g.h_theta) hipHostFree(g.h_theta);
    if (g.h_col) hipHostFree(g.h_col);
    if (g.h_out) hipHostFree(g.h_out);
    if (g.h_status) hipHostFree(g.h_status);
    if (g.h_spec) hipHostFree(g.h_spec);
  }
  void* ptrs[] = {p.dX, p.dY, p.dspec, p.Kbuf, p.Mbuf, p.MTbuf, p.Pbuf, p.dtheta, p.dr, p.dz, p.dalpha, p.dfeat, p.dfconst,
                  p.dlogdet, p.dpart, p.dout, p.dcol, p.dstatus, p.dsolve_part, p.dslotspec, p.dspans, p.dspec_span, p.dmask,
                  p.dnobs, p.Kbuf32, p.Mbuf32, p.MTbuf32,
                  p.Pbuf32, p.dquad, p.df32r};
  for (void* q : ptrs)
    if (q) hipFree(q);
  delete h;
}

int64_t wv_problem_workspace_bytes(const wv_problem* h) { return h ? h->p.workspace_bytes : 0; }

// Block structure of structure `si`: if every non-zero term carries a Categorical factor on one column,
// point the structure at that column's span table (built from the host copy of X on first use).  WV_BLOCKS=0 at
// problem creation keeps everything dense (the two paths give bit-identical results: tests/test_gpu_blocks.py).
static int register_block_structure(Problem& p, int si) {
  int32_t table = -1;
  const int col = p.hX.empty() ? -1 : block_column(p.specs[si]);
  if (col >= 0) {
    if (p.span_table_of_col[col] < 0) {
      std::vector<int32_t> sp;
      block_spans(p.hX.data(), p.N, p.D, col, p.Npad, sp);
      WV_HIP_TRY(hipMemcpy(p.dspans + (int64_t)p.n_span_tables * sp.size(), sp.data(), sizeof(int32_t) * sp.size(),
                           hipMemcpyHostToDevice));
      p.span_table_of_col[col] = p.n_span_tables++;
      p.hspan_tabs.push_back(sp);
    }
    table = p.span_table_of_col[col];
  }
  // (a new entry is referenced by no slot yet, so kernels in flight do not read it)
  WV_HIP_TRY(hipMemcpy(p.dspec_span + si, &table, sizeof(int32_t), hipMemcpyHostToDevice));
  if ((int)p.spec_table.size() <= si) p.spec_table.resize(si + 1, -1);
  p.spec_table[si] = table;
  // boundaries at which every structure of the problem decouples (Problem::decoupled)
  const int64_t ng = p.Npad / 64;
  bool all_block = true;
  for (int t : p.spec_table) all_block &= t >= 0;
  if (!all_block) {
    // a dense structure: every product runs again (and rewrites the M21 blocks) for good
    std::fill(p.decoupled.begin(), p.decoupled.end(), 0);
    if (p.decoupled.empty()) p.decoupled.assign(ng + 1, 0);
    p.ever_dense = true;
  } else if (!p.ever_dense) {
    p.decoupled.assign(ng + 1, 1);
    p.decoupled[0] = p.decoupled[ng] = 0;
    for (int t : p.spec_table) {
      const std::vector<int32_t>& sp = p.hspan_tabs[t];
      int32_t hi_below = 0;
      for (int64_t g = 1; g < ng; ++g) {   // rows below 64 g reach up to hi_below; rows from 64 g on reach down to lo_above
        hi_below = std::max(hi_below, sp[2 * (g - 1) + 1]);
        if (hi_below > 64 * g) p.decoupled[g] = 0;
      }
      int32_t lo_above = INT32_MAX;
      for (int64_t g = ng - 1; g >= 1; --g) {
        lo_above = std::min(lo_above, sp[2 * g]);
        if (lo_above < 64 * g) p.decoupled[g] = 0;
      }
    }
  }
  return WV_OK;
}

int32_t wv_problem_create(wv_ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y,
                          int32_t P, const wv_kernel_spec* spec, int32_t max_batch,
                          int32_t x_is_device, int32_t y_is_device, wv_problem** out) {
  return wv_problem_create_multi(ctx, X, N, D, Y, P, &spec, 1, max_batch, x_is_device, y_is_device, out);
}

int32_t wv_problem_create_multi(wv_ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y,
                                int32_t P, const wv_kernel_spec* const* spec_list, int32_t n_specs,
                                int32_t max_batch, int32_t x_is_device, int32_t y_is_device,
                                wv_problem** out) {
  return wv_problem_create_ex(ctx, X, N, D, Y, P, spec_list, n_specs, nullptr, max_batch, x_is_device, y_is_device, out);
}

static int32_t create_from_specs(Ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y, int32_t P,
                                 std::vector<Spec> parsed, const wv_problem_caps* caps, int32_t max_batch,
                                 int32_t x_is_device, int32_t y_is_device, wv_problem** out);

int32_t wv_problem_create_ex(wv_ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y, int32_t P,
                             const wv_kernel_spec* const* spec_list, int32_t n_specs, const wv_problem_caps* caps,
                             int32_t max_batch, int32_t x_is_device, int32_t y_is_device, wv_problem** out) {
  if (!ctx || !X || !Y || !spec_list || n_specs < 1 || !out || N < 1 || D < 1 || P < 1 || max_batch < 1) {
    set_error("wv_problem_create: bad argument");
    return WV_E_BADARG;
  }
  if (N > (1 << 20)) {
    set_error("N too large");
    return WV_E_BADARG;
  }
  std::vector<Spec> parsed(n_specs);
  for (int si = 0; si < n_specs; ++si) {
    if (!spec_list[si]) {
      set_error("wv_problem_create: null spec");
      return WV_E_BADARG;
    }
    std::string e = Spec::parse(spec_list[si], parsed[si]);
    if (!e.empty()) {
      set_error(e);
      return WV_E_SPEC;
    }
    if (parsed[si].n_dims != D) {
      set_error("spec.n_dims != D");
      return WV_E_SPEC;
    }
  }
  return create_from_specs(&ctx->c, X, N, D, Y, P, std::move(parsed), caps, max_batch, x_is_device, y_is_device, out);
}

// The problem proper, from parsed structures (the float64 twin of a float32 / missing-output problem is made here from
// the parent's own tables and device copies: post_fit_twin).
static int32_t create_from_specs(Ctx* ctx, const double* X, int64_t N, int32_t D, const double* Y, int32_t P,
                                 std::vector<Spec> parsed, const wv_problem_caps* caps, int32_t max_batch,
                                 int32_t x_is_device, int32_t y_is_device, wv_problem** out) {
  const int n_specs = (int)parsed.size();
  std::unique_ptr<wv_problem, void (*)(wv_problem*)> h(new wv_problem(), wv_problem_destroy);
  Problem& p = h->p;
  p.ctx = ctx;
  p.f32 = ctx->dtype == 1;
  p.specs = std::move(parsed);
  std::vector<DevSpec> hspecs(n_specs);
  for (int si = 0; si < n_specs; ++si) {
    std::string e = build_devspec(p.specs[si], hspecs[si]);
    if (!e.empty()) {
      set_error(e);
      return WV_E_UNSUPPORTED;
    }
    p.np_stride = std::max(p.np_stride, p.specs[si].n_params());
    p.max_feat = std::max(p.max_feat, (int)hspecs[si].n_feat);
    p.max_fact = std::max(p.max_fact, (int)hspecs[si].n_factors);
    p.max_red = std::max(p.max_red, (int)hspecs[si].n_red);
    for (int fi = 0; fi < hspecs[si].n_factors; ++fi) p.any_multidim |= hspecs[si].factors[fi].ndims > 1;
    p.any_product |= hspecs[si].n_factors > hspecs[si].n_terms;
  }
  p.spec_cap = n_specs;
  const int ksum = p.f32 ? 1 : 0;   // float32 contexts: one more sum slot and record column in the trace pass (KSUM)
  if (caps) {  // room for structures added later (wv_problem_add_spec)
    if (caps->max_feat > MAX_FEAT || caps->max_factors > 128 || caps->max_specs > (1 << 20)) {
      set_error("wv_problem_create_ex: capacity beyond the device limits (64 features, 128 factors)");
      return WV_E_UNSUPPORTED;
    }
    p.spec_cap = std::max(p.spec_cap, (int)caps->max_specs);
    p.np_stride = std::max(p.np_stride, (int)caps->max_params);
    p.max_feat = std::max(p.max_feat, (int)caps->max_feat);
    p.max_fact = std::max(p.max_fact, (int)caps->max_factors);
    p.max_red = std::max(p.max_red, (int)caps->max_red);
    if (caps->max_specs > n_specs) p.any_multidim = p.any_product = true;  // (later structures may act on several columns or carry products: the general kernels)
  }
  p.max_red += ksum;
  p.specs.reserve(p.spec_cap);
  for (auto& d : hspecs) d.n_params = p.np_stride;  // the device uses this field as the table stride
  p.spec = p.specs[0];
  p.hspec = hspecs[0];
  WV_HIP_TRY(hipSetDevice(ctx->device));
  p.N = N;
  p.Npad = (N + NB - 1) / NB * NB;
  p.D = D;
  p.P = P;
  p.max_batch = max_batch;
  p.out_stride = out_stride_for(p.np_stride);
  const int64_t Np = p.Npad, mat = Np * Np;
  const int np = p.np_stride;
  int64_t total = 0;
  auto alloc = [&](void** ptr, int64_t bytes) -> bool {
    if (hipMalloc(ptr, (size_t)bytes) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    total += bytes;
    return true;
  };
  bool ok = alloc((void**)&p.dX, sizeof(double) * Np * D) && alloc((void**)&p.dY, sizeof(double) * Np * P) &&
            alloc((void**)&p.dspec, sizeof(DevSpec) * p.spec_cap) &&
            alloc((void**)&p.dslotspec, sizeof(int32_t) * max_batch) &&
            alloc((void**)&p.dspans, sizeof(int32_t) * 2 * (Np / 64) * D) &&
            alloc((void**)&p.dspec_span, sizeof(int32_t) * p.spec_cap) &&
            (p.f32 ? alloc((void**)&p.Kbuf32, sizeof(float) * mat * max_batch) &&
                         alloc((void**)&p.Mbuf32, sizeof(float) * mat * max_batch) &&
                         alloc((void**)&p.MTbuf32, sizeof(float) * mat * max_batch) &&
                         alloc((void**)&p.Pbuf32, sizeof(float) * p.pbuf_elems() * max_batch)
                   : alloc((void**)&p.Kbuf, sizeof(double) * mat * max_batch) &&
                         alloc((void**)&p.Mbuf, sizeof(double) * mat * max_batch) &&
                         alloc((void**)&p.MTbuf, sizeof(double) * mat * max_batch) &&
                         alloc((void**)&p.Pbuf, sizeof(double) * p.pbuf_elems() * max_batch)) &&
            alloc((void**)&p.dtheta, sizeof(double) * np * max_batch) &&
            alloc((void**)&p.dr, sizeof(double) * Np * max_batch) &&
            alloc((void**)&p.dz, sizeof(double) * Np * max_batch) &&
            alloc((void**)&p.dalpha, sizeof(double) * Np * max_batch) &&
            alloc((void**)&p.dfeat, sizeof(double) * 2 * std::max(1, p.max_feat) * Np * max_batch) &&
            alloc((void**)&p.dfconst, sizeof(double) * 4 * std::max(1, p.max_fact) * max_batch) &&
            alloc((void**)&p.dlogdet, sizeof(double) * p.nblk() * max_batch) &&
            alloc((void**)&p.dpart, sizeof(double) * trace_part_stride(p) * (np + ksum) * max_batch) &&
            alloc((void**)&p.dout, sizeof(double) * p.out_stride * max_batch) &&
            alloc((void**)&p.dcol, sizeof(int32_t) * max_batch) &&
            alloc((void**)&p.dstatus, sizeof(int32_t) * max_batch);
  if (ok && Np <= 2048 && !p.f32) ok = alloc((void**)&p.dsolve_part, sizeof(double) * (Np / 64) * Np * max_batch);
  else if (ok && !p.f32) ok = alloc((void**)&p.dsolve_part, sizeof(double) * 8 * Np * max_batch);  // (TRMV_SLABS partial alphas)
  if (ok && p.f32) ok = alloc((void**)&p.df32r, sizeof(double) * F32R_VECS * Np * max_batch);
  if (!ok) {
    set_error("device allocation failed (workspace for max_batch fits does not fit in HBM)");
    return WV_E_OOM;
  }
  p.workspace_bytes = total;
  const int nwg = trace_part_stride(p);
  WV_HIP_TRY(hipMemset(p.dX, 0, sizeof(double) * Np * D));
  WV_HIP_TRY(hipMemset(p.dY, 0, sizeof(double) * Np * P));
  if (p.f32) {
    WV_HIP_TRY(hipMemset(p.Kbuf32, 0, sizeof(float) * mat * max_batch));
    WV_HIP_TRY(hipMemset(p.Mbuf32, 0, sizeof(float) * mat * max_batch));
    WV_HIP_TRY(hipMemset(p.MTbuf32, 0, sizeof(float) * mat * max_batch));
  } else {
    WV_HIP_TRY(hipMemset(p.Kbuf, 0, sizeof(double) * mat * max_batch));
    WV_HIP_TRY(hipMemset(p.Mbuf, 0, sizeof(double) * mat * max_batch));
    WV_HIP_TRY(hipMemset(p.MTbuf, 0, sizeof(double) * mat * max_batch));
  }
  WV_HIP_TRY(hipMemset(p.dalpha, 0, sizeof(double) * Np * max_batch));
  WV_HIP_TRY(hipMemset(p.dpart, 0, sizeof(double) * nwg * (np + ksum) * max_batch));
  if (p.f32) {
    // the probe vector of the second-order term (Rademacher, fixed: the objective stays a deterministic function)
    std::vector<double> hv((size_t)F32R_VECS * Np * max_batch, 0.0);
    uint64_t sd = 0x9E3779B97F4A7C15ull;
    std::vector<double> v(Np);
    for (int64_t i = 0; i < Np; ++i) {
      sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17;
      v[i] = (sd >> 33) & 1 ? 1.0 : -1.0;
    }
    for (int b = 0; b < max_batch; ++b) std::copy(v.begin(), v.end(), hv.begin() + ((size_t)b * F32R_VECS + F32R_V) * Np);
    WV_HIP_TRY(hipMemcpy(p.df32r, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice));
  }
  WV_HIP_TRY(hipMemcpy(p.dX, X, sizeof(double) * N * D, x_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  bool y_uploaded = false;
  if (caps && caps->nan_is_missing) {
    // NaN in column p of Y = row missing for output p (complete-case fits per output, all outputs in one batch):
    // those rows become padding rows of that output's fits.  Nothing is allocated when no NaN is found.
    std::vector<double> hY((size_t)N * P);
    WV_HIP_TRY(hipMemcpy(hY.data(), Y, sizeof(double) * N * P, y_is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
    std::vector<uint8_t> mask((size_t)Np * P, 0);
    std::vector<int32_t> nobs(P, 0);
    bool any = false;
    for (int c = 0; c < P; ++c)
      for (int64_t i = 0; i < N; ++i) {
        double& y = hY[(size_t)c * N + i];
        if (y != y) {
          y = 0.0;
          any = true;
        } else {
          mask[(size_t)c * Np + i] = 1;
          ++nobs[c];
        }
      }
    if (any) {
      for (int c = 0; c < P; ++c)
        if (nobs[c] == 0) {
          set_error("wv_problem_create_ex: an output column has no observed value");
          return WV_E_BADARG;
        }
      if (hipMalloc((void**)&p.dmask, mask.size()) != hipSuccess || hipMalloc((void**)&p.dnobs, sizeof(int32_t) * P) != hipSuccess) {
        (void)hipGetLastError();
        set_error("device allocation failed (missing-value masks)");
        return WV_E_OOM;
      }
      WV_HIP_TRY(hipMemcpy(p.dmask, mask.data(), mask.size(), hipMemcpyHostToDevice));
      WV_HIP_TRY(hipMemcpy(p.dnobs, nobs.data(), sizeof(int32_t) * P, hipMemcpyHostToDevice));
      WV_HIP_TRY(hipMemcpy2D(p.dY, sizeof(double) * Np, hY.data(), sizeof(double) * N, sizeof(double) * N, P,
                             hipMemcpyHostToDevice));
      y_uploaded = true;
    }
  }
  if (!y_uploaded)
    WV_HIP_TRY(hipMemcpy2D(p.dY, sizeof(double) * Np, Y, sizeof(double) * N, sizeof(double) * N, P,
                           y_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemcpy(p.dspec, hspecs.data(), sizeof(DevSpec) * n_specs, hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemset(p.dslotspec, 0, sizeof(int32_t) * max_batch));
  WV_HIP_TRY(hipMemset(p.dspec_span, 0xff, sizeof(int32_t) * p.spec_cap));  // -1: dense
  {
    const char* e = tuning_env("WV_BLOCKS");
    if (!(e && std::atoi(e) == 0)) {
      p.hX.resize((size_t)N * D);
      WV_HIP_TRY(hipMemcpy(p.hX.data(), X, sizeof(double) * N * D, x_is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
      p.span_table_of_col.assign(D, -1);
      for (int si = 0; si < n_specs; ++si) {
        const int rc = register_block_structure(p, si);
        if (rc != WV_OK) return rc;
      }
    }
  }
  // batch groups: independent streams, each advancing its own set of fits.  Measured on MI355X
  // (N = 2048, 1024 tasks): ONE group of 256 slots gives 3720 evaluations/s, two groups of 128 give
  // 3500 - large launches amortise the serial small kernels better than cross-stream overlap does -
  // so the default is a single group; WV_GROUPS overrides.
  int ng = 1;
  if (const char* s = tuning_env("WV_GROUPS")) ng = std::max(1, std::atoi(s));
  ng = std::min(ng, max_batch);
  int base = max_batch / ng, extra = max_batch % ng, at = 0;
  p.groups.resize(ng);
  for (int i = 0; i < ng; ++i) {
    Group& g = p.groups[i];
    g.slot0 = at;
    g.nslots = base + (i < extra ? 1 : 0);
    at += g.nslots;
    WV_HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    WV_HIP_TRY(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    WV_HIP_TRY(hipHostMalloc((void**)&g.h_theta, sizeof(double) * np * g.nslots));
    WV_HIP_TRY(hipHostMalloc((void**)&g.h_col, sizeof(int32_t) * g.nslots));
    WV_HIP_TRY(hipHostMalloc((void**)&g.h_out, sizeof(double) * p.out_stride * g.nslots));
    WV_HIP_TRY(hipHostMalloc((void**)&g.h_status, sizeof(int32_t) * g.nslots));
    WV_HIP_TRY(hipHostMalloc((void**)&g.h_spec, sizeof(int32_t) * g.nslots));
    std::memset(g.h_spec, 0, sizeof(int32_t) * g.nslots);
  }
  {
    // per-quad gradient accumulators of the few-fits regime (wv_gram.hip, trace_group_kernel): optional - without the
    // buffer few fits are traced like many (four tiles per workgroup), same numbers, a little more latency
    p.quad_stride = 64 * (int64_t)std::max(1, p.max_red);
    p.quad_fits = std::min(FEW_FITS, base + (extra ? 1 : 0));
    const size_t bytes = sizeof(double) * (size_t)ng * p.quad_fits * p.ntiles() * p.quad_stride;
    // (an optional buffer must not be what exhausts the device: at most an eighth of what is still free - at
    //  N ~ 98304 it would be ~4 GB on top of a workspace that nearly fills the 288 GB)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    if (bytes > ((size_t)8 << 30) || bytes > free_b / 8 || hipMalloc((void**)&p.dquad, bytes) != hipSuccess) {
      (void)hipGetLastError();
      p.dquad = nullptr;
    } else {
      p.workspace_bytes += (int64_t)bytes;
    }
  }
  *out = h.release();
  return WV_OK;
}

static int check_cols(const Problem& p, const int32_t* col_idx, int n) {
  for (int i = 0; i < n; ++i)
    if (col_idx[i] < 0 || col_idx[i] >= p.P) {
      set_error("column index out of range");
      return WV_E_BADARG;
    }
  return WV_OK;
}

int32_t wv_gram(wv_problem* h, const double* theta, double* K_out) {
  if (!h || !theta || !K_out) {
    set_error("wv_gram: null argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if (p.dmask) {
    set_error("wv_gram: the problem has missing outputs (nan_is_missing): fits, LML and gradients only");
    return WV_E_UNSUPPORTED;
  }
  if (p.specs.size() != 1) {
    set_error("wv_gram: the problem holds several structures (only wv_fit_multi serves those)");
    return WV_E_UNSUPPORTED;
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  if (int rc = enter_device_call(p, "wv_gram")) return rc;
  Group& g = p.groups[0];
  const int np = p.spec.n_params();
  std::memcpy(g.h_theta, theta, sizeof(double) * np);
  WV_HIP_TRY(hipMemcpyAsync(p.dtheta + (int64_t)g.slot0 * np, g.h_theta, sizeof(double) * np,
                            hipMemcpyHostToDevice, g.stream));
  launch_gram(p, g.stream, g.slot0, 1, 0);
  WV_HIP_TRY(hipGetLastError());
  if (int rc = leave_device_call(WV_OK)) return rc;
  if (p.f32) {  // float32 storage: fetch as stored, widen on the host
    std::vector<float> tmp((size_t)p.N * p.N);
    WV_HIP_TRY(hipMemcpy2DAsync(tmp.data(), sizeof(float) * p.N, p.Kbuf32 + (int64_t)g.slot0 * p.Npad * p.Npad,
                                sizeof(float) * p.Npad, sizeof(float) * p.N, p.N, hipMemcpyDeviceToHost, g.stream));
    WV_HIP_TRY(hipStreamSynchronize(g.stream));
    for (int64_t q = 0; q < p.N * p.N; ++q) K_out[q] = (double)tmp[q];
  } else {
    WV_HIP_TRY(hipMemcpy2DAsync(K_out, sizeof(double) * p.N, p.Kbuf + (int64_t)g.slot0 * p.Npad * p.Npad,
                                sizeof(double) * p.Npad, sizeof(double) * p.N, p.N,
                                hipMemcpyDeviceToHost, g.stream));
    WV_HIP_TRY(hipStreamSynchronize(g.stream));
  }
  // the device kernel fills lower tiles only; mirror on the host so the caller sees K_y = K_y'
  for (int64_t i = 0; i < p.N; ++i)
    for (int64_t j = i + 1; j < p.N; ++j) K_out[i * p.N + j] = K_out[j * p.N + i];
  return WV_OK;
}

// shared by wv_lml / wv_value_grad: evaluate B tasks given constrained tables
static int eval_batch(Problem& p, const int32_t* col_idx, int B, const double* theta_rows, bool with_grad,
                      std::vector<double>& recs, std::vector<int32_t>& status) {
  const int np = p.spec.n_params();
  recs.resize((size_t)B * p.out_stride);
  status.resize(B);
  int done = 0;
  // round-robin chunks over the groups
  struct Inflight { int g; int first; int n; };
  std::vector<Inflight> fl;
  size_t gi = 0;
  while (done < B || !fl.empty()) {
    while (done < B && fl.size() < p.groups.size()) {
      Group& g = p.groups[gi % p.groups.size()];
      int n = std::min(g.nslots, B - done);
      std::memcpy(g.h_theta, theta_rows + (size_t)done * np, sizeof(double) * np * n);
      std::memcpy(g.h_col, col_idx + done, sizeof(int32_t) * n);
      int rc = enqueue_eval(p, g, n, with_grad);
      if (rc != WV_OK) return rc;
      fl.push_back({(int)(gi % p.groups.size()), done, n});
      done += n;
      ++gi;
    }
    Inflight f = fl.front();
    fl.erase(fl.begin());
    Group& g = p.groups[f.g];
    int rc = wait_group(p, g);
    if (rc != WV_OK) return rc;
    std::memcpy(&recs[(size_t)f.first * p.out_stride], g.h_out, sizeof(double) * p.out_stride * f.n);
    std::memcpy(&status[f.first], g.h_status, sizeof(int32_t) * f.n);
  }
  return WV_OK;
}

int32_t wv_lml(wv_problem* h, const int32_t* col_idx, int32_t B, const double* theta, double* lml_out,
               int32_t* status_out) {
  if (!h || !col_idx || !theta || !lml_out || B < 1) {
    set_error("wv_lml: bad argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if (p.specs.size() != 1) {
    set_error("wv_lml: the problem holds several structures (only wv_fit_multi serves those)");
    return WV_E_UNSUPPORTED;
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  int rc = check_cols(p, col_idx, B);
  if (rc != WV_OK) return rc;
  if ((rc = enter_device_call(p, "wv_lml")) != WV_OK) return rc;
  std::vector<double> recs;
  std::vector<int32_t> st;
  rc = eval_batch(p, col_idx, B, theta, false, recs, st);
  if (rc != WV_OK) return rc;
  for (int i = 0; i < B; ++i) {
    double v = recs[(size_t)i * p.out_stride];
    bool ok = st[i] == 0 && std::isfinite(v);
    lml_out[i] = ok ? v : -std::numeric_limits<double>::infinity();
    if (status_out) status_out[i] = ok ? 0 : 1;
  }
  return WV_OK;
}

int32_t wv_value_grad(wv_problem* h, const int32_t* col_idx, int32_t B, const double* theta_u,
                      double* loss_out, double* grad_out, int32_t* status_out) {
  if (!h || !col_idx || !theta_u || !loss_out || B < 1) {
    set_error("wv_value_grad: bad argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if (p.specs.size() != 1) {
    set_error("wv_value_grad: the problem holds several structures (only wv_fit_multi serves those)");
    return WV_E_UNSUPPORTED;
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  int rc = check_cols(p, col_idx, B);
  if (rc != WV_OK) return rc;
  if ((rc = enter_device_call(p, "wv_value_grad")) != WV_OK) return rc;
  const int np = p.spec.n_params(), nt = p.spec.n_trainable();
  std::vector<double> theta((size_t)B * np);
  for (int i = 0; i < B; ++i) p.spec.constrain(theta_u + (size_t)i * nt, &theta[(size_t)i * np]);
  std::vector<double> recs;
  std::vector<int32_t> st;
  rc = eval_batch(p, col_idx, B, theta.data(), true, recs, st);
  if (rc != WV_OK) return rc;
  for (int i = 0; i < B; ++i) {
    loss_from_record(p.spec, theta_u + (size_t)i * nt, &theta[(size_t)i * np], &recs[(size_t)i * p.out_stride],
                     st[i], &loss_out[i], grad_out ? grad_out + (size_t)i * nt : nullptr, nullptr);
    if (status_out) status_out[i] = std::isfinite(loss_out[i]) ? 0 : 1;
  }
  return WV_OK;
}

int32_t wv_fit(wv_problem* h, const int32_t* col_idx, int32_t n_tasks, const wv_lbfgs_opts* opts,
               double* theta_u_inout, double* loss_out, double* lml_out, int32_t* n_iter_out,
               int32_t* n_eval_out, int32_t* status_out) {
  return wv_fit_penalized(h, col_idx, n_tasks, opts, nullptr, theta_u_inout, loss_out, lml_out, n_iter_out,
                          n_eval_out, status_out);
}

// ---------------------------------------------------------------------------------------------------------------
// The work queue over (column, structure) tasks.  wv_fit / wv_fit_penalized / wv_fit_multi submit a fixed list and run
// it dry; wv_queue_* expose it so that a caller can add tasks while others are still being fitted (the kernel search:
// an output column whose candidates are all scored moves on to its next level without waiting for the other columns).
// A slot's arithmetic does not depend on what else is in the batch, so both uses give bit-identical fits.
// ---------------------------------------------------------------------------------------------------------------
namespace wv {

struct QTask {
  int32_t col = 0, spec = 0;
  double prior_scale = -1.0;  // < 0: the spec's own prior scales
  int64_t tag = 0;
  std::vector<double> u0;     // n_trainable of the task's structure
};
struct QResult {
  int64_t tag = 0;
  std::vector<double> u;
  double loss = 0, lml = 0;
  int32_t nit = 0, nfev = 0, status = 0;
};

class FitQueue {
 public:
  FitQueue(Problem& p, const wv_lbfgs_opts& o) : p_(p), opts_(o), slots_(p.groups.size()), inflight_(p.groups.size(), 0) {}
  void submit(QTask&& t) { queue_.push_back(std::move(t)); }
  size_t in_system() const {
    size_t n = queue_.size() + held_.size();
    for (auto& sl : slots_)
      for (auto& s : sl) n += s.live;
    return n;
  }
  std::deque<QResult> done;
  // One scheduling step: groups without a launch in flight are (re)filled and launched; then the next group in turn
  // is waited for, its results are fed to the optimisers, finished fits move to `done`, the group is refilled and
  // relaunched.  Returns WV_OK (also when there was nothing to do).
  int advance();
  void trace(int n_tasks, double wall) const {
    fprintf(stderr, "[wv_fit] %d tasks, %ld group-rounds, %ld evals (%.1f per round), wall %.3f s: wait %.3f host %.3f launch %.3f\n",
            n_tasks, n_rounds_, n_evals_, (double)n_evals_ / std::max(1L, n_rounds_), wall, t_wait_, t_host_, t_launch_);
  }

 private:
  struct Slot {
    int live = 0;
    QTask task;
    std::unique_ptr<Lbfgs> opt;
  };
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  int refill_and_launch(int gi);
  Problem& p_;
  wv_lbfgs_opts opts_;
  std::deque<QTask> queue_;
  std::vector<std::vector<Slot>> slots_;
  std::vector<int> inflight_;  // number of slots evaluated by the group's pending launch
  std::vector<Slot> held_;     // drain phase: tasks parked until they can join another group's batch
  int gi_ = 0;
  double t_wait_ = 0, t_host_ = 0, t_launch_ = 0;
  long n_rounds_ = 0, n_evals_ = 0;
};

int FitQueue::refill_and_launch(int gi) {
  Group& g = p_.groups[gi];
  auto& sl = slots_[gi];
  const int np = p_.np_stride;
  // drop finished slots, pull new tasks, keep active tasks packed at the front
  std::vector<Slot> act;
  for (auto& s : sl)
    if (s.live) act.push_back(std::move(s));
  // fits parked by the drain fold come back first: they already own an optimiser state and, when the queue is fed
  // incrementally (wv_queue_submit after a fold), must not sit idle until the queue runs dry again
  while ((int)act.size() < g.nslots && !held_.empty()) {
    act.push_back(std::move(held_.back()));
    held_.pop_back();
  }
  while ((int)act.size() < g.nslots && !queue_.empty()) {
    Slot s;
    s.live = 1;
    s.task = std::move(queue_.front());
    queue_.pop_front();
    s.opt.reset(new Lbfgs(p_.specs[s.task.spec].n_trainable(), opts_, s.task.u0.data()));
    if (p_.f32) s.opt->set_ftol_patience(2);  // float32 objective noise (~1e-5 relative) fakes SciPy's ftol test
    act.push_back(std::move(s));
  }
  sl = std::move(act);
  inflight_[gi] = (int)sl.size();
  if (sl.empty()) return WV_OK;
  for (size_t i = 0; i < sl.size(); ++i) {
    const Spec& sp = p_.specs[sl[i].task.spec];
    if (sp.n_params() < np) std::fill(g.h_theta + i * np + sp.n_params(), g.h_theta + (i + 1) * np, 0.0);
    sp.constrain(sl[i].opt->x(), g.h_theta + i * np);
    g.h_col[i] = sl[i].task.col;
    g.h_spec[i] = sl[i].task.spec;
  }
  return enqueue_eval(p_, g, (int)sl.size(), true);
}

int FitQueue::advance() {
  const int ng = (int)p_.groups.size();
  const int np = p_.np_stride;
  // idle groups pick up queued work
  for (int g = 0; g < ng; ++g)
    if (inflight_[g] == 0 && !queue_.empty()) {  // (refill takes parked fits first, then queued tasks)
      double t2 = now();
      int rc = refill_and_launch(g);
      if (rc != WV_OK) return rc;
      t_launch_ += now() - t2;
    }
  int busy = 0;
  for (int g = 0; g < ng; ++g) busy += inflight_[g] > 0;
  if (busy == 0) return WV_OK;
  while (inflight_[gi_] == 0) gi_ = (gi_ + 1) % ng;
  const int gi = gi_;
  Group& g = p_.groups[gi];
  double t0 = now();
  int rc = wait_group(p_, g);
  if (rc != WV_OK) return rc;
  double t1 = now();
  t_wait_ += t1 - t0;
  auto& sl = slots_[gi];
  ++n_rounds_;
  n_evals_ += (long)sl.size();
  std::vector<double> gu, theta(np);
  for (size_t i = 0; i < sl.size(); ++i) {
    Slot& s = sl[i];
    double loss, lml;
    // h_theta still holds the constrained table this evaluation used
    const double ps = s.task.prior_scale;
    const Spec& sp = p_.specs[s.task.spec];
    gu.resize(sp.n_trainable());
    loss_from_record(sp, s.opt->x(), g.h_theta + i * np, g.h_out + i * p_.out_stride, g.h_status[i], &loss, gu.data(),
                     &lml, ps);
    int task = s.opt->feed(loss, gu.data());
    if (task != WV_LB_FG) {
      // a fit that never saw a finite evaluation (NaN in the column, K_y not positive definite at the
      // start point) reports loss = +inf, lml = -inf and WV_LB_ABNORMAL - never a finite score
      QResult r;
      r.tag = s.task.tag;
      r.u.assign(s.opt->x(), s.opt->x() + sp.n_trainable());
      const double f_end = s.opt->f();
      const bool f_ok = std::isfinite(f_end);
      r.loss = f_ok ? f_end : std::numeric_limits<double>::infinity();
      r.nit = s.opt->n_iter();
      r.nfev = s.opt->n_eval();
      r.status = f_ok ? task : WV_LB_ABNORMAL;
      // LML at the returned point = -(loss) - log prior
      sp.constrain(s.opt->x(), theta.data());
      r.lml = f_ok ? -f_end - sp.log_prior(theta.data(), nullptr, ps) : -std::numeric_limits<double>::infinity();
      done.push_back(std::move(r));
      s.live = 0;
    }
  }
  // Drain phase (queue empty): once everything still running fits in one group, fold the groups
  // together - half-empty batches cost almost as much as full ones (latency-bound small kernels).
  if (ng > 1 && queue_.empty()) {
    int mine = 0, others_inflight = 0, busy_groups = 0;
    for (auto& s : sl) mine += s.live;
    for (int o = 0; o < ng; ++o)
      if (o != gi && inflight_[o] > 0) {
        others_inflight += inflight_[o];
        ++busy_groups;
      }
    if (busy_groups > 0 && mine + (int)held_.size() + others_inflight <= g.nslots) {
      for (auto& s : sl)
        if (s.live) held_.push_back(std::move(s));
      sl.clear();
      inflight_[gi] = 0;
      gi_ = (gi + 1) % ng;
      return WV_OK;  // these tasks join the next group that finishes
    }
    if (!held_.empty() && mine + (int)held_.size() <= g.nslots) {
      for (auto& s : held_) sl.push_back(std::move(s));
      held_.clear();
    }
  }
  double t2 = now();
  t_host_ += t2 - t1;
  rc = refill_and_launch(gi);
  if (rc != WV_OK) return rc;
  t_launch_ += now() - t2;
  gi_ = (gi + 1) % ng;
  return WV_OK;
}

}  // namespace wv

// u_stride = doubles between consecutive tasks' start points in theta_u_inout.
static int32_t fit_impl(wv_problem* h, const int32_t* col_idx, const int32_t* spec_idx, int32_t n_tasks,
                        const wv_lbfgs_opts* opts, const double* prior_scale, int32_t u_stride,
                        double* theta_u_inout, double* loss_out, double* lml_out, int32_t* n_iter_out,
                        int32_t* n_eval_out, int32_t* status_out) {
  Problem& p = h->p;
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  int rc = check_cols(p, col_idx, n_tasks);
  if (rc != WV_OK) return rc;
  if ((rc = enter_device_call(p, "wv_fit")) != WV_OK) return rc;
  for (int t = 0; t < n_tasks; ++t) {
    const int si = spec_idx ? spec_idx[t] : 0;
    if (si < 0 || si >= (int)p.specs.size()) {
      set_error("wv_fit: structure index out of range");
      return WV_E_BADARG;
    }
    const int nt = p.specs[si].n_trainable();
    if (nt < 1) {
      set_error("wv_fit: no trainable parameters");
      return WV_E_SPEC;
    }
    if (nt > u_stride) {
      set_error("wv_fit: u_stride smaller than a structure's number of trainable parameters");
      return WV_E_BADARG;
    }
  }
  FitQueue q(p, *opts);
  for (int t = 0; t < n_tasks; ++t) {
    QTask task;
    task.col = col_idx[t];
    task.spec = spec_idx ? spec_idx[t] : 0;
    task.prior_scale = prior_scale ? prior_scale[t] : -1.0;
    task.tag = t;
    const double* u0 = theta_u_inout + (size_t)t * u_stride;
    task.u0.assign(u0, u0 + p.specs[task.spec].n_trainable());
    q.submit(std::move(task));
  }
  const bool trace = tuning_env("WV_FIT_TRACE") != nullptr;
  const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  while (q.in_system() > 0) {
    rc = q.advance();
    if (rc != WV_OK) return rc;
  }
  for (const QResult& r : q.done) {
    const int t = (int)r.tag;
    std::memcpy(theta_u_inout + (size_t)t * u_stride, r.u.data(), sizeof(double) * r.u.size());
    if (loss_out) loss_out[t] = r.loss;
    if (lml_out) lml_out[t] = r.lml;
    if (n_iter_out) n_iter_out[t] = r.nit;
    if (n_eval_out) n_eval_out[t] = r.nfev;
    if (status_out) status_out[t] = r.status;
  }
  if (trace)
    q.trace(n_tasks, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_begin);
  return WV_OK;
}

int32_t wv_fit_penalized(wv_problem* h, const int32_t* col_idx, int32_t n_tasks, const wv_lbfgs_opts* opts,
                         const double* prior_scale, double* theta_u_inout, double* loss_out,
                         double* lml_out, int32_t* n_iter_out, int32_t* n_eval_out, int32_t* status_out) {
  if (!h || !col_idx || !opts || !theta_u_inout || n_tasks < 1) {
    set_error("wv_fit: bad argument");
    return WV_E_BADARG;
  }
  if (h->p.specs.size() != 1) {
    set_error("wv_fit: the problem holds several structures, use wv_fit_multi");
    return WV_E_BADARG;
  }
  return fit_impl(h, col_idx, nullptr, n_tasks, opts, prior_scale, h->p.spec.n_trainable(), theta_u_inout, loss_out,
                  lml_out, n_iter_out, n_eval_out, status_out);
}

int32_t wv_fit_multi(wv_problem* h, const int32_t* col_idx, const int32_t* spec_idx, int32_t n_tasks,
                     const wv_lbfgs_opts* opts, const double* prior_scale, int32_t u_stride,
                     double* theta_u_inout, double* loss_out, double* lml_out, int32_t* n_iter_out,
                     int32_t* n_eval_out, int32_t* status_out) {
  if (!h || !col_idx || !spec_idx || !opts || !theta_u_inout || n_tasks < 1 || u_stride < 1) {
    set_error("wv_fit_multi: bad argument");
    return WV_E_BADARG;
  }
  return fit_impl(h, col_idx, spec_idx, n_tasks, opts, prior_scale, u_stride, theta_u_inout, loss_out, lml_out,
                  n_iter_out, n_eval_out, status_out);
}

// Runs the value+gradient pipeline for one task in slot 0 of group 0 and leaves alpha (dalpha),
// K_y^-1 (Kbuf lower) and the result record on the device / in g.h_out.
static int eval_one(Problem& p, int32_t col, const double* theta) {
  Group& g = p.groups[0];
  const int np = p.spec.n_params();
  std::memcpy(g.h_theta, theta, sizeof(double) * np);
  g.h_col[0] = col;
  int rc = enqueue_eval(p, g, 1, true);
  if (rc != WV_OK) return rc;
  return wait_group(p, g);
}

__global__ void kinv_diag_kernel(const double* __restrict__ Kinv, double* __restrict__ out, int64_t Npad,
                                 int64_t N) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = Kinv[i * Npad + i];
}

// mean[m] = sum_j Ks[m][j] alpha[j]   (one wave per row of the cross-Gram)
__global__ __launch_bounds__(256) void cross_mean_kernel(const double* __restrict__ Ks,
                                                         const double* __restrict__ alpha,
                                                         double* __restrict__ mean, int64_t Mrow, int64_t Npad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  if (m >= Mrow) return;
  double s = 0.0;
  for (int64_t j = 2 * lane; j < Npad; j += 128) {
    double2 k = *reinterpret_cast<const double2*>(Ks + m * Npad + j);
    double2 a = *reinterpret_cast<const double2*>(alpha + j);
    s += k.x * a.x + k.y * a.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) mean[m] = s;
}

// q[m] = sum_i V[i][m]^2   (V = M Ks', N x Mpad): threads along m, rows split over blockIdx.y
__global__ __launch_bounds__(256) void colsumsq_kernel(const double* __restrict__ V, double* __restrict__ part,
                                                       int64_t Npad, int64_t Mpad, int rows_per_block) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= Mpad) return;
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_block, i1 = min(Npad, i0 + rows_per_block);
  double s = 0.0;
  for (int64_t i = i0; i < i1; ++i) {
    double v = V[i * Mpad + m];
    s += v * v;
  }
  part[(int64_t)blockIdx.y * Mpad + m] = s;
}

// k(x, x) of the flattened Sum-of-Products at one point (host; used for predictive variances)
static double kernel_diag_at(const Spec& s, const double* theta, const double* x) {
  double tot = 0.0;
  for (int t = 0; t < s.n_terms(); ++t) {
    double prod = 1.0;
    for (int f = s.term_start[t]; f < s.term_start[t + 1]; ++f) {
      const wv_factor& fa = s.factors[f];
      double xx = 0.0;
      for (int d = 0; d < s.n_dims; ++d)
        if (fa.dim_mask & (1u << d)) xx += x[d] * x[d];
      double v;
      switch (fa.type) {
        case WV_K_LIN: v = theta[fa.param[0]] * xx; break;
        case WV_K_POLY: v = std::pow(theta[fa.param[0]] * xx + theta[fa.param[1]], fa.degree); break;
        case WV_K_EMPTY: v = 0.0; break;
        default: v = theta[fa.param[0]];  // stationary, periodic, constant, categorical: the variance
      }
      prod *= v;
    }
    tot += prod;
  }
  return tot;
}

// The float64, complete-case twin that serves the post-fit calls of a float32 problem or of one with missing outputs
// (Problem::twin).  One column at a time: without missing values a change of column is one device copy of Y's column;
// with them the observed rows differ from column to column and the twin is rebuilt (7 ms at N = 2048; these calls are
// one evaluation each, after the fits) from the host copies of X and the masks kept beside it.
// `want_slots`: batch slots the caller can use - 1 for wv_predict / wv_whitened_state; wv_whitened_means factors its
// leave-one-out Gram matrices as a batch (up to 8, within 8 GiB).  The twin is created with what the first caller needs
// and re-created larger only when a later call asks for more: a predict alone pins 4 N^2 doubles, not 32 N^2.
static int32_t post_fit_twin(Problem& p, int col, const char* who, int want_slots) {
  if (p.specs.size() != 1) {
    set_error(std::string(who) + ": the problem holds several structures (only wv_fit_multi serves those)");
    return WV_E_UNSUPPORTED;
  }
  if (p.open_queue) {
    set_error(std::string(who) + ": a wv_queue is open on this problem");
    return WV_E_BADARG;
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  const int64_t per_slot = 4 * (int64_t)sizeof(double) * p.Npad * p.Npad;
  const int slots = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(8, std::max(1, want_slots)), ((int64_t)8 << 30) / per_slot));
  const bool big_enough = p.twin && p.twin_slots >= slots;
  if (big_enough && p.twin_col == col) return WV_OK;
  if (big_enough && !p.dmask) {
    WV_HIP_TRY(hipMemcpy(p.twin->p.dY, p.dY + (int64_t)col * p.Npad, sizeof(double) * p.Npad, hipMemcpyDeviceToDevice));
    p.twin_col = col;
    return WV_OK;
  }
  if (p.twin) {
    wv_problem_destroy(p.twin);
    p.twin = nullptr;
    p.twin_col = -1;
    p.twin_slots = 0;
  }
  p.ctx64 = *p.ctx;
  p.ctx64.dtype = 0;
  int32_t rc;
  p.twin_rows.clear();
  if (!p.dmask) {
    rc = create_from_specs(&p.ctx64, p.dX, p.N, p.D, p.dY + (int64_t)col * p.Npad, 1, {p.spec}, nullptr, slots, 1, 1, &p.twin);
  } else {
    if (p.twin_hostX.empty()) {   // X does not change over the life of a problem: one download
      p.twin_hostX.resize((size_t)p.N * p.D);
      WV_HIP_TRY(hipMemcpy(p.twin_hostX.data(), p.dX, sizeof(double) * p.N * p.D, hipMemcpyDeviceToHost));
    }
    std::vector<uint8_t> mask(p.N);
    std::vector<double> X((size_t)p.N * p.D), y(p.N);
    WV_HIP_TRY(hipMemcpy(mask.data(), p.dmask + (int64_t)col * p.Npad, p.N, hipMemcpyDeviceToHost));
    WV_HIP_TRY(hipMemcpy(y.data(), p.dY + (int64_t)col * p.Npad, sizeof(double) * p.N, hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (int64_t i = 0; i < p.N; ++i)
      if (mask[i]) {
        std::copy(p.twin_hostX.begin() + i * p.D, p.twin_hostX.begin() + (i + 1) * p.D, X.begin() + n * p.D);
        y[n++] = y[i];
        p.twin_rows.push_back(i);
      }
    if (n == p.N) p.twin_rows.clear();  // (this column is complete)
    rc = create_from_specs(&p.ctx64, X.data(), n, p.D, y.data(), 1, {p.spec}, nullptr, slots, 0, 0, &p.twin);
  }
  if (rc != WV_OK) {
    p.twin = nullptr;
    p.twin_rows.clear();
    return rc;
  }
  p.twin_col = col;
  p.twin_slots = slots;
  return WV_OK;
}

int32_t wv_predict(wv_problem* h, int32_t col, const double* theta, const double* Xnew, int64_t M,
                   int32_t include_noise, double* mean_out, double* var_out) {
  if (!h || !theta || col < 0 || col >= h->p.P || (Xnew && M < 1)) {
    set_error("wv_predict: bad argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if (p.f32 || p.dmask) {
    // float64 on the complete cases of this column (Problem::twin).  Training inputs of a problem with missing outputs:
    // all N rows - the observed ones by the training-input identities, the others as new inputs.
    if (int rc = post_fit_twin(p, col, "wv_predict", 1)) return rc;
    if (Xnew || p.twin_rows.empty()) return wv_predict(p.twin, 0, theta, Xnew, M, include_noise, mean_out, var_out);
    const int64_t n = (int64_t)p.twin_rows.size(), nmiss = p.N - n;
    std::vector<double> m(std::max(n, nmiss)), v(std::max(n, nmiss));
    if (int rc = wv_predict(p.twin, 0, theta, nullptr, 0, include_noise, mean_out ? m.data() : nullptr, var_out ? v.data() : nullptr))
      return rc;
    std::vector<uint8_t> seen(p.N, 0);
    for (int64_t k = 0; k < n; ++k) {
      const int64_t i = p.twin_rows[k];
      seen[i] = 1;
      if (mean_out) mean_out[i] = m[k];
      if (var_out) var_out[i] = v[k];
    }
    const std::vector<double>& Xall = p.twin_hostX;   // (kept by post_fit_twin: no second download)
    std::vector<double> Xm((size_t)nmiss * p.D);
    std::vector<int64_t> miss;
    for (int64_t i = 0; i < p.N; ++i)
      if (!seen[i]) {
        std::copy(Xall.begin() + i * p.D, Xall.begin() + (i + 1) * p.D, Xm.begin() + (int64_t)miss.size() * p.D);
        miss.push_back(i);
      }
    if (int rc = wv_predict(p.twin, 0, theta, Xm.data(), nmiss, include_noise, mean_out ? m.data() : nullptr,
                            var_out ? v.data() : nullptr))
      return rc;
    for (int64_t k = 0; k < nmiss; ++k) {
      if (mean_out) mean_out[miss[k]] = m[k];
      if (var_out) var_out[miss[k]] = v[k];
    }
    return WV_OK;
  }
  if (p.specs.size() != 1) {
    set_error("wv_predict: the problem holds several structures (only wv_fit_multi serves those)");
    return WV_E_UNSUPPORTED;
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  int rc = enter_device_call(p, "wv_predict");
  if (rc != WV_OK) return rc;
  rc = eval_one(p, col, theta);
  if (rc != WV_OK) return rc;
  Group& g = p.groups[0];
  if (g.h_status[0] != 0) {
    set_error("wv_predict: Gram matrix not positive definite at theta");
    return WV_E_BADARG;
  }
  const double s2 = theta[p.spec.noise_param];
  const double c = p.spec.mean_param >= 0 ? theta[p.spec.mean_param] : 0.0;
  hipStream_t st = g.stream;
  const int64_t mat = p.Npad * p.Npad;
  if (Xnew == nullptr) {
    // training inputs: K alpha = r - sigma^2 alpha  =>  mean_f = y - sigma^2 alpha;
    // K - K K_y^-1 K = sigma^2 I - sigma^4 K_y^-1  =>  var_f = sigma^2 - sigma^4 diag(K_y^-1)
    std::vector<double> alpha(p.N), kd(p.N), y(p.N);
    hipLaunchKernelGGL(kinv_diag_kernel, dim3((unsigned)((p.N + 255) / 256)), dim3(256), 0, st,
                       p.Kbuf + (int64_t)g.slot0 * mat, p.dz + (int64_t)g.slot0 * p.Npad, p.Npad, p.N);
    WV_HIP_TRY(hipMemcpyAsync(kd.data(), p.dz + (int64_t)g.slot0 * p.Npad, sizeof(double) * p.N, hipMemcpyDeviceToHost, st));
    WV_HIP_TRY(hipMemcpyAsync(alpha.data(), p.dalpha + (int64_t)g.slot0 * p.Npad, sizeof(double) * p.N, hipMemcpyDeviceToHost, st));
    WV_HIP_TRY(hipMemcpyAsync(y.data(), p.dY + (int64_t)col * p.Npad, sizeof(double) * p.N, hipMemcpyDeviceToHost, st));
    WV_HIP_TRY(hipStreamSynchronize(st));
    for (int64_t i = 0; i < p.N; ++i) {
      if (mean_out) mean_out[i] = y[i] - s2 * alpha[i];
      if (var_out) var_out[i] = s2 - s2 * s2 * kd[i] + (include_noise ? s2 : 0.0);
    }
    return WV_OK;
  }
  // new inputs, in chunks of <= 4096 rows: Ks = K(Xnew, X); mean = Ks alpha + c;
  // V = M Ks' (M = L^-1 from the factorisation); var_f = k(x*, x*) - colsum(V .* V)
  const int64_t CH = 4096;
  const int64_t chunk_pad = (std::min(M, CH) + GT - 1) / GT * GT;
  const int nsplit = 16, rows_per = (int)((p.Npad + nsplit - 1) / nsplit);
  double *dXn = nullptr, *dKs = nullptr, *dV = nullptr, *dmean = nullptr, *dpart = nullptr;
  auto cleanup = [&]() {
    for (double* q : {dXn, dKs, dV, dmean, dpart})
      if (q) hipFree(q);
  };
  if (hipMalloc(&dXn, sizeof(double) * chunk_pad * p.D) != hipSuccess ||
      hipMalloc(&dKs, sizeof(double) * chunk_pad * p.Npad) != hipSuccess ||
      hipMalloc(&dV, sizeof(double) * p.Npad * chunk_pad) != hipSuccess ||
      hipMalloc(&dmean, sizeof(double) * chunk_pad) != hipSuccess ||
      hipMalloc(&dpart, sizeof(double) * nsplit * chunk_pad) != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    set_error("wv_predict: device allocation failed");
    return WV_E_OOM;
  }
  std::vector<double> hm(chunk_pad), hp((size_t)nsplit * chunk_pad);
  const double* dth = p.dtheta + (int64_t)g.slot0 * p.spec.n_params();
  for (int64_t m0 = 0; m0 < M; m0 += CH) {
    const int64_t mc = std::min(CH, M - m0), mpad = (mc + GT - 1) / GT * GT;
    hipMemsetAsync(dXn, 0, sizeof(double) * mpad * p.D, st);
    hipMemcpyAsync(dXn, Xnew + m0 * p.D, sizeof(double) * mc * p.D, hipMemcpyHostToDevice, st);
    launch_gram_rect(p, st, dXn, mc, mpad, dth, dKs);
    hipLaunchKernelGGL(cross_mean_kernel, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, st, dKs,
                       p.dalpha + (int64_t)g.slot0 * p.Npad, dmean, mc, p.Npad);
    if (var_out) {
      GemmArgs a{};
      a.A = p.Mbuf + (int64_t)g.slot0 * mat;  // A(i, k) = M[i][k], lower triangular
      a.B = dKs;                              // B(m, k) = Ks[m][k]
      a.C = dV;
      a.lda = (int)p.Npad; a.ldb = (int)p.Npad; a.ldc = (int)mpad;
      a.M = (int)p.Npad; a.N = (int)mpad; a.kb = 0; a.ke = (int)p.Npad; a.khi_mode = 1;
      a.alpha = 1.0; a.beta = 0.0; a.nbatch = 1;
      launch_gemm(st, a, true, true, mpad > 64 ? 128 : 64);
      hipLaunchKernelGGL(colsumsq_kernel, dim3((unsigned)((mpad + 255) / 256), nsplit), dim3(256), 0, st, dV, dpart,
                         p.Npad, mpad, rows_per);
      hipMemcpyAsync(hp.data(), dpart, sizeof(double) * nsplit * mpad, hipMemcpyDeviceToHost, st);
    }
    hipMemcpyAsync(hm.data(), dmean, sizeof(double) * mc, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
      cleanup();
      set_error("wv_predict: HIP error in the prediction kernels");
      return WV_E_HIP;
    }
    if (leave_device_call(WV_OK) != WV_OK) {  // a launcher refused (its message stands): nothing was computed
      cleanup();
      return WV_E_UNSUPPORTED;
    }
    for (int64_t i = 0; i < mc; ++i) {
      if (mean_out) mean_out[m0 + i] = hm[i] + c;
      if (var_out) {
        double q = 0.0;
        for (int sblk = 0; sblk < nsplit; ++sblk) q += hp[(size_t)sblk * mpad + i];
        var_out[m0 + i] = kernel_diag_at(p.spec, theta, Xnew + (m0 + i) * p.D) - q + (include_noise ? s2 : 0.0);
      }
    }
  }
  cleanup();
  return WV_OK;
}

int32_t wv_whitened_means(wv_problem* h, int32_t col, const double* theta, const uint64_t* term_masks,
                          int32_t n_masks, double jitter, double* means_out) {
  if (!h || !theta || !term_masks || !means_out || n_masks < 1 || col < 0 || col >= h->p.P || !(jitter >= 0)) {
    set_error("wv_whitened_means: bad argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if (p.f32 || p.dmask) {
    // float64 on the complete cases of this column (Problem::twin); rows missing for this column answer NaN
    if (int rc = post_fit_twin(p, col, "wv_whitened_means", n_masks)) return rc;
    if (p.twin_rows.empty()) return wv_whitened_means(p.twin, 0, theta, term_masks, n_masks, jitter, means_out);
    const int64_t n = (int64_t)p.twin_rows.size();
    std::vector<double> mm((size_t)n_masks * n);
    if (int rc = wv_whitened_means(p.twin, 0, theta, term_masks, n_masks, jitter, mm.data())) return rc;
    std::fill(means_out, means_out + (int64_t)n_masks * p.N, std::numeric_limits<double>::quiet_NaN());
    for (int m = 0; m < n_masks; ++m)
      for (int64_t k = 0; k < n; ++k) means_out[(int64_t)m * p.N + p.twin_rows[k]] = mm[(size_t)m * n + k];
    return WV_OK;
  }
  if (p.specs.size() != 1) {
    set_error("wv_whitened_means: the problem holds several structures (only wv_fit_multi serves those)");
    return WV_E_UNSUPPORTED;
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  int rc = enter_device_call(p, "wv_whitened_means");
  if (rc != WV_OK) return rc;
  rc = eval_one(p, col, theta);  // alpha of the full model
  if (rc != WV_OK) return rc;
  Group& g = p.groups[0];
  if (g.h_status[0] != 0) {
    set_error("wv_whitened_means: Gram matrix not positive definite at theta");
    return WV_E_BADARG;
  }
  hipStream_t st = g.stream;
  const int64_t mat = p.Npad * p.Npad;
  double* K0 = p.Kbuf + (int64_t)g.slot0 * mat;
  double* q_mu = p.dr + (int64_t)g.slot0 * p.Npad;   // re-use r
  const uint64_t all = p.spec.n_terms() >= 64 ? ~0ull : ((1ull << p.spec.n_terms()) - 1ull);
  const double c = p.spec.mean_param >= 0 ? theta[p.spec.mean_param] : 0.0;
  // q_mu = chol(K + jitter I)' alpha   (whitened optimum of the SVGP with Z = X)
  WV_HIP_TRY(hipMemsetAsync(p.dstatus + g.slot0, 0, sizeof(int32_t), st));
  launch_gram_masked(p, st, g.slot0, 1, all, jitter);
  cholesky(p, st, g.slot0, 1);
  launch_trmv(p, st, p.Kbuf, p.dalpha, p.dr, g.slot0, 1, true);
  // The masked Gram matrices are independent given q_mu: factor them as ONE batch over the group's slots
  // (chunks of nslots masks) instead of one latency-bound factorisation after the other.
  const int np_ = p.np_stride;
  const int cap = g.nslots;
  for (int s = 1; s < std::min(cap, (int)n_masks); ++s) {  // every slot gets the parameter row and q_mu
    std::memcpy(g.h_theta + (size_t)s * np_, g.h_theta, sizeof(double) * np_);
    WV_HIP_TRY(hipMemcpyAsync(p.dr + (int64_t)(g.slot0 + s) * p.Npad, p.dr + (int64_t)g.slot0 * p.Npad,
                              sizeof(double) * p.Npad, hipMemcpyDeviceToDevice, st));
  }
  if (cap > 1 && n_masks > 1)
    WV_HIP_TRY(hipMemcpyAsync(p.dtheta + (int64_t)g.slot0 * np_, g.h_theta,
                              sizeof(double) * np_ * std::min(cap, (int)n_masks), hipMemcpyHostToDevice, st));
  // status words only ever get set: slot 0 still carries the verdict on the q_mu factor above
  if (cap > 1) WV_HIP_TRY(hipMemsetAsync(p.dstatus + g.slot0 + 1, 0, sizeof(int32_t) * (cap - 1), st));
  std::vector<double> host((size_t)cap * p.Npad);
  int32_t bad_any = 0;
  for (int m0 = 0; m0 < n_masks; m0 += cap) {
    const int nbm = std::min(cap, n_masks - m0);
    for (int s = 0; s < nbm; ++s) launch_gram_masked(p, st, g.slot0 + s, 1, term_masks[m0 + s] & all, jitter);
    cholesky(p, st, g.slot0, nbm);
    launch_trmv(p, st, p.Kbuf, p.dr, p.dz, g.slot0, nbm, false);
    WV_HIP_TRY(hipMemcpyAsync(host.data(), p.dz + (int64_t)g.slot0 * p.Npad, sizeof(double) * p.Npad * nbm,
                              hipMemcpyDeviceToHost, st));
    WV_HIP_TRY(hipMemcpyAsync(g.h_status, p.dstatus + g.slot0, sizeof(int32_t) * nbm, hipMemcpyDeviceToHost, st));
    WV_HIP_TRY(hipStreamSynchronize(st));
    for (int s = 0; s < nbm; ++s) {
      bad_any |= g.h_status[s];
      for (int64_t i = 0; i < p.N; ++i) means_out[(int64_t)(m0 + s) * p.N + i] = host[(size_t)s * p.Npad + i] + c;
    }
  }
  const int32_t bad = bad_any;
  WV_HIP_TRY(hipGetLastError());
  (void)K0; (void)q_mu;
  if (int rc2 = leave_device_call(WV_OK)) return rc2;
  if (bad) {
    set_error("wv_whitened_means: a jittered component Gram matrix was not positive definite");
    return WV_E_BADARG;
  }
  return WV_OK;
}

// Kbuf(lower) = I + Mbuf(lower) * inv_s2   (the whitened posterior precision I + L'L / sigma^2 from L'L in Mbuf)
__global__ __launch_bounds__(256) void precision_kernel(double* __restrict__ Kmat, const double* __restrict__ LtL,
                                                        int64_t Npad, double inv_s2) {
  // (rows on grid.x: grid.y stops at 65535 and a problem may have more rows than that)
  const int64_t i = blockIdx.x;
  for (int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x; j <= i; j += (int64_t)gridDim.y * 256)
    Kmat[i * Npad + j] = LtL[i * Npad + j] * inv_s2 + (i == j ? 1.0 : 0.0);
}

int32_t wv_whitened_state(wv_problem* h, int32_t col, const double* theta, double jitter, double* q_mu_out,
                          double* q_sqrt_out) {
  if (!h || !theta || !q_mu_out || col < 0 || col >= h->p.P || !(jitter >= 0)) {
    set_error("wv_whitened_state: bad argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if (p.dmask) {
    set_error("wv_whitened_state: the problem has missing outputs (nan_is_missing): the variational state of a column lives on its observed rows only - build a problem on those rows");
    return WV_E_UNSUPPORTED;
  }
  if (p.specs.size() != 1) {
    set_error("wv_whitened_state: the problem holds several structures (only wv_fit_multi serves those)");
    return WV_E_UNSUPPORTED;
  }
  if (p.f32) {  // float64 on a twin of this problem (Problem::twin)
    if (int rc = post_fit_twin(p, col, "wv_whitened_state", 1)) return rc;
    return wv_whitened_state(p.twin, 0, theta, jitter, q_mu_out, q_sqrt_out);
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  int rc = enter_device_call(p, "wv_whitened_state");
  if (rc != WV_OK) return rc;
  rc = eval_one(p, col, theta);  // alpha
  if (rc != WV_OK) return rc;
  Group& g = p.groups[0];
  if (g.h_status[0] != 0) {
    set_error("wv_whitened_state: Gram matrix not positive definite at theta");
    return WV_E_BADARG;
  }
  hipStream_t st = g.stream;
  const int64_t Np = p.Npad, mat = Np * Np;
  double* K0 = p.Kbuf + (int64_t)g.slot0 * mat;
  double* M0 = p.Mbuf + (int64_t)g.slot0 * mat;
  const uint64_t all = p.spec.n_terms() >= 64 ? ~0ull : ((1ull << p.spec.n_terms()) - 1ull);
  const double s2 = theta[p.spec.noise_param];
  // L = chol(K + jitter I) in Kbuf; q_mu = L' alpha
  WV_HIP_TRY(hipMemsetAsync(p.dstatus + g.slot0, 0, sizeof(int32_t), st));
  launch_gram_masked(p, st, g.slot0, 1, all, jitter);
  cholesky(p, st, g.slot0, 1);
  launch_trmv(p, st, p.Kbuf, p.dalpha, p.dr, g.slot0, 1, true);
  WV_HIP_TRY(hipMemcpyAsync(q_mu_out, p.dr + (int64_t)g.slot0 * Np, sizeof(double) * p.N, hipMemcpyDeviceToHost, st));
  if (q_sqrt_out) {
    // L'L (lower) into Mbuf - the product K_y^-1 = M'M runs on, with L in M's place (the blocked factorisation's
    // diagonal-block inverses in Mbuf are dead once L is complete)
    GemmArgs a{};
    a.A = K0; a.B = K0; a.C = M0;
    a.lda = a.ldb = a.ldc = (int)Np;
    a.sA = a.sB = a.sC = mat;
    a.M = (int)Np; a.N = (int)Np; a.kb = 0; a.ke = (int)Np; a.klo_mode = 1; a.koff = 0;
    a.alpha = 1.0; a.beta = 0.0; a.nbatch = 1;
    a.tri = 1; a.tri_off = 0;
    launch_gemm(st, a, false, false, 64);
    hipLaunchKernelGGL(precision_kernel, dim3((unsigned)Np, (unsigned)std::min<int64_t>((Np + 255) / 256, 256)), dim3(256), 0,
                       st, K0, M0, Np, 1.0 / s2);
    // (I + L'L / sigma^2)^-1 by the factor-and-invert pipeline of the fits, then its Cholesky factor
    factor_inverse(p, st, g.slot0, 1);
    lauum(p, st, g.slot0, 1);
    cholesky(p, st, g.slot0, 1);
    WV_HIP_TRY(hipMemcpy2DAsync(q_sqrt_out, sizeof(double) * p.N, K0, sizeof(double) * Np, sizeof(double) * p.N, p.N,
                                hipMemcpyDeviceToHost, st));
  }
  WV_HIP_TRY(hipMemcpyAsync(g.h_status, p.dstatus + g.slot0, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  WV_HIP_TRY(hipStreamSynchronize(st));
  WV_HIP_TRY(hipGetLastError());
  const int32_t bad = g.h_status[0];
  if (int rc2 = leave_device_call(WV_OK)) return rc2;
  if (bad) {
    set_error("wv_whitened_state: a factorisation met a non-positive pivot");
    return WV_E_BADARG;
  }
  if (q_sqrt_out)
    for (int64_t i = 0; i < p.N; ++i)
      for (int64_t j = i + 1; j < p.N; ++j) q_sqrt_out[i * p.N + j] = 0.0;
  return WV_OK;
}

int32_t wv_problem_add_spec(wv_problem* h, const wv_kernel_spec* spec, int32_t* index_out) {
  if (!h || !spec || !index_out) {
    set_error("wv_problem_add_spec: null argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if ((int)p.specs.size() >= p.spec_cap) {
    set_error("wv_problem_add_spec: the problem's structure table is full (wv_problem_create_ex caps.max_specs)");
    return WV_E_UNSUPPORTED;
  }
  Spec sp;
  std::string e = Spec::parse(spec, sp);
  if (!e.empty()) {
    set_error(e);
    return WV_E_SPEC;
  }
  if (sp.n_dims != p.D) {
    set_error("spec.n_dims != D");
    return WV_E_SPEC;
  }
  DevSpec d;
  e = build_devspec(sp, d);
  if (!e.empty()) {
    set_error(e);
    return WV_E_UNSUPPORTED;
  }
  if (sp.n_params() > p.np_stride || d.n_feat > p.max_feat || d.n_factors > p.max_fact || d.n_red + (p.f32 ? 1 : 0) > p.max_red) {
    set_error("wv_problem_add_spec: the structure exceeds a capacity the problem was created with");
    return WV_E_UNSUPPORTED;
  }
  d.n_params = p.np_stride;
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  // the new table entry is referenced by no slot yet, so kernels in flight do not read it
  WV_HIP_TRY(hipMemcpy(p.dspec + p.specs.size(), &d, sizeof(DevSpec), hipMemcpyHostToDevice));
  *index_out = (int32_t)p.specs.size();
  p.specs.push_back(std::move(sp));
  return register_block_structure(p, *index_out);
}

struct wv_queue {
  wv_problem* prob;
  std::unique_ptr<FitQueue> q;
};

int32_t wv_queue_create(wv_problem* h, const wv_lbfgs_opts* opts, wv_queue** out) {
  if (!h || !opts || !out) {
    set_error("wv_queue_create: null argument");
    return WV_E_BADARG;
  }
  if (h->p.open_queue) {  // one queue per problem: the groups' staging buffers belong to it
    set_error("wv_queue_create: the problem already has an open work queue");
    return WV_E_BADARG;
  }
  h->p.open_queue = true;
  auto* w = new wv_queue();
  w->prob = h;
  w->q.reset(new FitQueue(h->p, *opts));
  *out = w;
  return WV_OK;
}

int32_t wv_queue_submit(wv_queue* w, int32_t n, const int32_t* col_idx, const int32_t* spec_idx,
                        const double* prior_scale, int32_t u_stride, const double* theta_u0, const int64_t* tags) {
  if (!w || n < 0 || (n > 0 && (!col_idx || !spec_idx || !theta_u0 || !tags)) || u_stride < 1) {
    set_error("wv_queue_submit: bad argument");
    return WV_E_BADARG;
  }
  Problem& p = w->prob->p;
  int rc = check_cols(p, col_idx, n);
  if (rc != WV_OK) return rc;
  for (int t = 0; t < n; ++t) {
    if (spec_idx[t] < 0 || spec_idx[t] >= (int)p.specs.size()) {
      set_error("wv_queue_submit: structure index out of range");
      return WV_E_BADARG;
    }
    const int nt = p.specs[spec_idx[t]].n_trainable();
    if (nt < 1 || nt > u_stride) {
      set_error("wv_queue_submit: structure without trainable parameters, or u_stride too small");
      return nt < 1 ? WV_E_SPEC : WV_E_BADARG;
    }
  }
  for (int t = 0; t < n; ++t) {
    QTask task;
    task.col = col_idx[t];
    task.spec = spec_idx[t];
    task.prior_scale = prior_scale ? prior_scale[t] : -1.0;
    task.tag = tags[t];
    const double* u0 = theta_u0 + (size_t)t * u_stride;
    task.u0.assign(u0, u0 + p.specs[task.spec].n_trainable());
    w->q->submit(std::move(task));
  }
  return WV_OK;
}

int32_t wv_queue_advance(wv_queue* w, int32_t* n_done_out, int32_t* n_in_system_out) {
  if (!w) {
    set_error("wv_queue_advance: null queue");
    return WV_E_BADARG;
  }
  WV_HIP_TRY(hipSetDevice(w->prob->p.ctx->device));
  int rc = w->q->advance();
  if (rc != WV_OK) return rc;
  if (n_done_out) *n_done_out = (int32_t)w->q->done.size();
  if (n_in_system_out) *n_in_system_out = (int32_t)w->q->in_system();
  return WV_OK;
}

int32_t wv_queue_fetch(wv_queue* w, int32_t cap, int32_t u_stride, int64_t* tags_out, double* theta_u_out,
                       double* loss_out, double* lml_out, int32_t* n_iter_out, int32_t* n_eval_out,
                       int32_t* status_out, int32_t* n_out) {
  if (!w || cap < 0 || !n_out || (cap > 0 && (!tags_out || !theta_u_out)) || u_stride < 1) {
    set_error("wv_queue_fetch: bad argument");
    return WV_E_BADARG;
  }
  int n = 0;
  while (n < cap && !w->q->done.empty()) {
    const QResult& r = w->q->done.front();
    if ((int)r.u.size() > u_stride) {
      // the results popped so far stay the caller's (n_out says how many); the wide one stays queued
      *n_out = n;
      set_error("wv_queue_fetch: u_stride smaller than a fit's number of trainable parameters");
      return WV_E_BADARG;
    }
    tags_out[n] = r.tag;
    std::fill(theta_u_out + (size_t)n * u_stride, theta_u_out + (size_t)(n + 1) * u_stride, 0.0);
    std::memcpy(theta_u_out + (size_t)n * u_stride, r.u.data(), sizeof(double) * r.u.size());
    if (loss_out) loss_out[n] = r.loss;
    if (lml_out) lml_out[n] = r.lml;
    if (n_iter_out) n_iter_out[n] = r.nit;
    if (n_eval_out) n_eval_out[n] = r.nfev;
    if (status_out) status_out[n] = r.status;
    w->q->done.pop_front();
    ++n;
  }
  *n_out = n;
  return WV_OK;
}

void wv_queue_destroy(wv_queue* w) {
  if (!w) return;
  // let launches in flight finish before the staging buffers are reused
  if (w->prob && w->prob->p.ctx) {
    (void)hipSetDevice(w->prob->p.ctx->device);
    for (auto& g : w->prob->p.groups)
      if (g.stream) (void)hipStreamSynchronize(g.stream);
  }
  if (w->prob) w->prob->p.open_queue = false;
  delete w;
}

int32_t wv_stage_times(wv_problem* h, int32_t cap, const char** names_out, double* ms_out,
                       int64_t* launches_out, double* flops_out, double* bytes_out) {
  if (!h) return 0;
  Problem& p = h->p;
  p.timer.collect();
  int n = std::min<int>(cap, ST_COUNT);
  for (int i = 0; i < n; ++i) {
    if (names_out) names_out[i] = kStageNames[i];
    if (ms_out) ms_out[i] = p.timer.ms[i];
    if (launches_out) launches_out[i] = p.timer.launches[i];
    if (flops_out) flops_out[i] = p.timer.flops[i];
    if (bytes_out) bytes_out[i] = p.timer.bytes[i];
  }
  return ST_COUNT;
}
void wv_stage_times_reset(wv_problem* h) {
  if (h) h->p.timer.reset();
}
void wv_stage_timing_enable(wv_problem* h, int32_t on) {
  if (h) {
    h->p.timer.collect();
    h->p.timer.enabled = on != 0;
    h->p.timer.mask = (on == 0 || on == 1) ? 0xffffffffu : ((uint32_t)on >> 1);
  }
}

}  // extern "C"
