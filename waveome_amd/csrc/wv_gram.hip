// Gram-matrix assembly and the fused hyper-parameter gradient ("trace") pass.
//
// gram_kernel:  K_y[i][j] = sum_t prod_f k_f(x_i, x_j) + sigma^2 [i==j]   (lower 64x64 tiles)
//   replaces kernel(X) + add_likelihood_noise_cov (waveome/model_types_DEPR.py:50-51), i.e. the
//   Sum/Product tree over waveome/kernels.py (Lin :19-32, Poly :56-73, Categorical :95-117,
//   Empty :136-139) and gpflow SquaredExponential / Matern / Periodic / Constant.
// trace_kernel: g_p = 1/2 sum_ij W_ij dK_ij/dtheta_p,  W = alpha alpha' - K_y^-1
//   replaces TensorFlow's reverse-mode pass through the kernel tree; every dK/dtheta_p is
//   recomputed per tile from LDS-staged X features, so HBM traffic is one read of the lower
//   triangle of K_y^-1 however many hyper-parameters there are.
//
// Both kernels are fp64-VALU-bound (PMC, profiles/r02_eval512_pmc_valu_*.csv: the VALU issues an fp64
// instruction in 71-82 % of all SIMD cycles, HBM traffic is 10-12 % of the 8 TB/s roofline), so the design
// goal is the instruction count per matrix element and term:
//   * per-row "features" - raw x_d, rint(x_d) for Categorical, sinpi / cospi(x_d / period) for Periodic so that
//     sin(pi (x_i - x_j)/p) = s_i c_j - c_i s_j needs no per-element trigonometry - are computed ONCE per fit and
//     evaluation by feature_kernel into a small HBM table and copied into LDS per tile (round 1 recomputed them
//     for each of the 528 tiles of a matrix);
//   * exponentials go through exp2_64: the argument is produced directly in units of ln 2 / 64 (the factor's
//     lengthscale is folded into one scalar, for Periodic into its row-side features), so that
//     rint / subtract / 5 FMAs / one conflict-free LDS table look-up / ldexp replace the 20-instruction
//     Cody-Waite + degree-13 polynomial of round 1;
//   * gradient sums are accumulated as sum w e, sum w e t, sum w e q2 and scaled by the parameter-dependent
//     constants once per tile instead of per element.
// Thread layout: 256 threads as 16 x 16, each owning rows ty + 16a (a < 4) and 4 contiguous
// columns 4 tx .. 4 tx + 3 (32-byte row segments -> 512 B contiguous per 16 threads).
#include <cstdlib>

#include <type_traits>

#include "wv_device.hpp"

namespace wv {

constexpr double kSqrt3 = 1.7320508075688772;
constexpr double kSqrt5 = 2.23606797749979;
constexpr double kPi = 3.141592653589793;
constexpr double kLn2_64 = 0.010830424696249145;    // ln 2 / 64
constexpr double k64_Ln2 = 92.33248261689366;       // 64 / ln 2
constexpr double kCse2 = 46.16624130844683;         // 32 / ln 2: exp(-q / (2 l^2)) = 2^(-(q * kCse2 / l^2) / 64)
constexpr double kCse = 6.7945744023041525;         // sqrt(32 / ln 2)

// 2^(j/64), j = 0..63, correctly rounded
__device__ const double kExp2Tab[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};

constexpr int EXPTAB_DOUBLES = 64 * 32;  // LDS image: [j][lane & 31], one private copy per lane of a half-wave

struct TileCtx {
  const double* fi;   // [n_feat][GT] row features (Periodic sin / cos already scaled by kCse / lengthscale)
  const double* fj;   // [n_feat][GT] column features
  const double* tab;  // exp table image in LDS, offset by this thread's lane & 31
  const double* fc;   // [n_factors][4] constants derived from theta once per evaluation (factor_constants)
  int ty, tx;
};

// exp2_64 forms its table address by OR: the table must start at offset 0 of the workgroup's LDS, i.e. the kernel must
// have no static __shared__ data.  Checked once per kernel symbol on the host; a kernel that fails is not launched.
template <typename Kern>
static bool exp_table_at_lds_zero(Kern kernel) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel)) != hipSuccess) {
    (void)hipGetLastError();
    return true;  // (cannot tell: the kernels in this file are written without static LDS)
  }
  if (a.sharedSizeBytes != 0) {
    fail_launch("Gram / trace kernel has static LDS: the exp table is not at LDS offset 0");
    return false;
  }
  return true;
}
template <auto Kernel>
static bool exp_table_ok() {
  static const bool ok = exp_table_at_lds_zero(Kernel);
  return ok;
}

// Per-factor constants that depend on theta only - computed once per fit and evaluation (feature_kernel), not
// per tile and thread: c[0] = element scale, c[1] = variance, c[2] / c[3] = what the lengthscale / period sums
// of the gradient pass are multiplied by.
//   SE:       t = q c0,  c0 = -(32/ln2) / l^2;        dk/dl = var e q / l^3      = c2 sum(w e t),  c2 = -var (ln2/32) / l
//   Matern:   u = sqrt(q) c0, c0 = sqrt(nu2) (64/ln2) / l;  dk/dl = var dl / l  -> c2 = var / l
//   Periodic: (lengthscale folded into the row features)   c2 as SE;  dk/dp = c3 sum(w e q2), c3 = var pi (ln2/32) / p^2
__device__ inline void factor_constants(const DevFactor& f, const double* __restrict__ th, double* __restrict__ c) {
  const int type = f.type;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
  if (type != WV_K_EMPTY) c1 = th[f.p[0]];
  if (type == WV_K_SE) {
    const double ell = th[f.p[1]];
    c0 = -kCse2 / (ell * ell);
    c2 = -c1 * (2.0 * kLn2_64) / ell;
  } else if (type == WV_K_M12 || type == WV_K_M32 || type == WV_K_M52) {
    const double ell = th[f.p[1]];
    const double rt = type == WV_K_M12 ? 1.0 : (type == WV_K_M32 ? kSqrt3 : kSqrt5);
    c0 = rt * k64_Ln2 / ell;
    c2 = c1 / ell;
  } else if (type == WV_K_PERIODIC) {
    const double ell = th[f.p[1]], per = th[f.p[2]];
    c2 = -c1 * (2.0 * kLn2_64) / ell;
    c3 = c1 * kPi * (2.0 * kLn2_64) / (per * per);
  } else if (type == WV_K_POLY) {
    c0 = th[f.p[1]];  // offset
  }
  c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
}

// 2^(t / 64) for t <= 0 (every kernel function here exponentiates a non-positive argument).
// k = rint(t), r = t - k exactly, |r| <= 1/2: 2^(t/64) = 2^floor(k/64) * 2^((k mod 64)/64) * exp(r ln2/64); the last
// factor by its degree-5 Taylor polynomial (|r ln2/64| <= 0.0054, truncation 3.5e-17), the middle one from the LDS
// table - stored [j][lane & 31], so lane l of each 32-lane group reads dwords 2l, 2l+1 (mod 64) whatever its j:
// conflict-free ds_read_b64.  ~2 ulp.  Arguments below ~ -2^31 saturate the integer conversion and underflow to 0.
// `tab` = this lane's slice of the table (table base + (lane & 31) doubles).  The table is the first 16 KB of the
// workgroup's LDS (offset 0: the kernels here have no static __shared__ data - launch_* check it once), so the entry's
// address is (j << 8) OR-ed into the lane's: one v_and_or_b32 instead of and + add.
// (Forms that were measured against this one and not kept - microbenchmark tools/probe_exp_rate.py and the kernels
//  themselves, profiles/r04_exp_rate.txt, profiles/r04_gram_trace_ab.txt: rint / convert by the 1.5 * 2^52 shift, the Horner
//  steps as 3-address v_fma_f64 with SGPR coefficients, four chains of a row phase by phase between scheduling barriers.
//  Bit-identical all of them, 14 % faster in the microbenchmark, and within +-1 % - or slower, with the range clamp the
//  shift needs - inside the Gram and trace kernels.)
__device__ inline double exp2_64(double t, const double* __restrict__ tab) {
  const double kd = rint(t);
  const double r = t - kd;
  const int k = (int)kd;
  double p = 1.2417843701716925e-12;           // (ln2/64)^5 / 5!
  p = fma(p, r, 5.732851688640402e-10);        // ^4 / 4!
  p = fma(p, r, 2.1173137155464776e-07);       // ^3 / 3!
  p = fma(p, r, 5.86490495505617e-05);         // ^2 / 2!
  p = fma(p, r, 0.010830424696249145);         // ln2/64
  p = fma(p, r, 1.0);
  typedef const __attribute__((address_space(3))) double* lds_cdp;
  const uint32_t lane_addr = (uint32_t)(size_t)(lds_cdp)tab;  // < 256: the lane's 8 bytes inside a table row
  const uint32_t addr = (((uint32_t)k << 8) & 0x3f00u) | lane_addr;
  return ldexp(*(lds_cdp)(size_t)addr * p, k >> 6);
}

__device__ inline double ipow(double b, int e) {
  double r = 1.0;
  for (int i = 0; i < e; ++i) r *= b;
  return r;
}

__device__ inline double feature_value(const FeatOp op, double x, const double* __restrict__ th, bool row_side) {
  double v;
  switch (op.kind) {
    case FEAT_RINT: v = rint(x); break;
    case FEAT_SINPI: v = sinpi(x / th[op.param]); break;
    case FEAT_COSPI: v = cospi(x / th[op.param]); break;
    default: v = x;
  }
  if (row_side && op.sparam >= 0) v *= kCse / th[op.sparam];
  return v;
}

// Per fit and evaluation: F[b][side][slot][row], side 0 = row features, 1 = column features.
__global__ __launch_bounds__(256) void feature_kernel(const DevSpec* __restrict__ specs,
                                                      const int32_t* __restrict__ slot_spec,
                                                      const double* __restrict__ X,
                                                      const double* __restrict__ theta_all,
                                                      double* __restrict__ F, double* __restrict__ FC,
                                                      int64_t N, int64_t Npad, int D, int max_feat, int max_fact,
                                                      int slot0) {
  const int b = slot0 + blockIdx.y;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const double* th = theta_all + (int64_t)b * sp->n_params;
  if (blockIdx.x == 0 && (int)threadIdx.x < sp->n_factors)
    factor_constants(sp->factors[threadIdx.x], th, FC + ((int64_t)b * max_fact + threadIdx.x) * 4);
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= Npad) return;
  double* Fr = F + (int64_t)b * 2 * max_feat * Npad;
  double* Fc = Fr + (int64_t)max_feat * Npad;
  for (int s = 0; s < sp->n_feat; ++s) {
    const FeatOp op = sp->feat[s];
    const double x = (row < N) ? X[row * D + op.dim] : 0.0;
    const double base = feature_value(op, x, th, false);
    Fc[(int64_t)s * Npad + row] = base;
    Fr[(int64_t)s * Npad + row] = (op.sparam >= 0) ? base * (kCse / th[op.sparam]) : base;
  }
}

// tile staging: copy 64 rows of every feature slot from the table into LDS
__device__ inline void stage_from_table(double* dst, const double* __restrict__ Fside, int n_feat, int64_t Npad,
                                        int row0, int tid) {
  for (int q = tid; q < n_feat * GT; q += 256) dst[q] = Fside[(int64_t)(q / GT) * Npad + row0 + (q % GT)];
}
// ... or compute them in place (rectangular cross-Gram of the prediction path: new rows have no table)
__device__ inline void stage_features(double* dst, const DevSpec* __restrict__ sp, const double* __restrict__ X,
                                      const double* __restrict__ th, int row0, int64_t N, int D, int tid,
                                      bool row_side) {
  const int total = sp->n_feat * GT;
  for (int q = tid; q < total; q += 256) {
    const int s = q / GT, r = q % GT;
    const FeatOp op = sp->feat[s];
    const int64_t row = row0 + r;
    dst[q] = feature_value(op, (row < N) ? X[row * D + op.dim] : 0.0, th, row_side);
  }
}
__device__ inline void stage_exp_table(double* tab, int tid) {
  for (int q = tid; q < EXPTAB_DOUBLES; q += 256) tab[q] = kExp2Tab[q >> 5];
}

#ifdef WV_EXPERIMENTS
// ---- exp2_64 variants side by side (tools/probe_exp_rate.py): rate of each and bit identity against variant 0 ----
//  1: the five Horner steps as 3-address v_fma_f64 with the coefficients in SGPRs (inside the Gram kernel hipcc emits
//     v_fmac + a 64-bit move of the coefficient per step: 862 v_mov_b64 beside 2320 fp64 arithmetic instructions)
//  2: + rint / convert by the 1.5 * 2^52 shift (two adds; the integer is the low word of the shifted sum)
//  3: + the power of two added into the exponent field instead of v_ldexp_f64
template <int V>
__device__ inline double exp2_64_variant(double t, const double* __restrict__ tab) {
  if (V == 0) return exp2_64(t, tab);
  double kd, r;
  int k;
  if (V >= 2) {
    const double big = 6755399441055744.0;  // 1.5 * 2^52
    const double sh = t + big;
    kd = sh - big;
    k = __double2loint(sh);
  } else {
    kd = rint(t);
    k = (int)kd;
  }
  r = t - kd;
  double p;
  const double c5 = 1.2417843701716925e-12, c4 = 5.732851688640402e-10, c3 = 2.1173137155464776e-07,
               c2 = 5.86490495505617e-05, c1 = 0.010830424696249145;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(c5), "v"(r), "v"(c4));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "s"(c3));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "s"(c2));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "s"(c1));
  p = fma(p, r, 1.0);
  typedef const __attribute__((address_space(3))) double* lds_cdp;
  const uint32_t lane_addr = (uint32_t)(size_t)(lds_cdp)tab;
  const uint32_t addr = (((uint32_t)k << 8) & 0x3f00u) | lane_addr;
  const double v = *(lds_cdp)(size_t)addr * p;
  if (V >= 3) {
    const int hi = __double2hiint(v) + ((k >> 6) << 20);
    return __hiloint2double(hi, __double2loint(v));
  }
  return ldexp(v, k >> 6);
}

template <int V>
__global__ __launch_bounds__(256) void exp_rate_kernel(double* __restrict__ out, int iters, double t0, double dt) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  stage_exp_table(smem, threadIdx.x);
  __syncthreads();
  const double* tab = smem + (threadIdx.x & 31);
  double t[8], acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    t[q] = t0 - 1e-3 * (double)(threadIdx.x + 256 * (blockIdx.x & 63)) - 7.3 * q;
    acc[q] = 0.0;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc[q] += exp2_64_variant<V>(t[q], tab);
      t[q] -= dt;
    }
  }
  double a = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) a += acc[q];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = a;
}

template <int V>
__global__ __launch_bounds__(256) void exp_check_kernel(const double* __restrict__ t, int64_t n, unsigned long long* __restrict__ bad,
                                                        double* __restrict__ first_bad) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  stage_exp_table(smem, threadIdx.x);
  __syncthreads();
  const double* tab = smem + (threadIdx.x & 31);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double a = exp2_64(t[i], tab), b = exp2_64_variant<V>(t[i], tab);
    if (__double_as_longlong(a) != __double_as_longlong(b))
      if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = t[i]; first_bad[1] = a; first_bad[2] = b; }
  }
}

}  // namespace wv
extern "C" int32_t wv_dbg_exp_rate(int32_t variant, int32_t iters, int32_t waves_per_simd, double t0, double dt,
                                   double* gexp_per_s_out) {
  using namespace wv;
  hipDeviceProp_t prop;
  WV_HIP_TRY(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * waves_per_simd;
  double* d;
  WV_HIP_TRY(hipMalloc(&d, sizeof(double) * blocks * 256));
  hipEvent_t a, b;
  WV_HIP_TRY(hipEventCreate(&a));
  WV_HIP_TRY(hipEventCreate(&b));
  const size_t sh = sizeof(double) * EXPTAB_DOUBLES;
  auto go = [&](int n) {
    switch (variant) {
      case 0: hipLaunchKernelGGL(exp_rate_kernel<0>, dim3(blocks), dim3(256), sh, nullptr, d, n, t0, dt); break;
      case 1: hipLaunchKernelGGL(exp_rate_kernel<1>, dim3(blocks), dim3(256), sh, nullptr, d, n, t0, dt); break;
      case 2: hipLaunchKernelGGL(exp_rate_kernel<2>, dim3(blocks), dim3(256), sh, nullptr, d, n, t0, dt); break;
      default: hipLaunchKernelGGL(exp_rate_kernel<3>, dim3(blocks), dim3(256), sh, nullptr, d, n, t0, dt); break;
    }
  };
  go(iters / 8 + 1);
  WV_HIP_TRY(hipEventRecord(a, nullptr));
  go(iters);
  WV_HIP_TRY(hipEventRecord(b, nullptr));
  WV_HIP_TRY(hipEventSynchronize(b));
  float ms = 0;
  WV_HIP_TRY(hipEventElapsedTime(&ms, a, b));
  if (gexp_per_s_out) *gexp_per_s_out = (double)blocks * 256.0 * 8.0 * iters / (ms * 1e-3) / 1e9;
  hipEventDestroy(a); hipEventDestroy(b); hipFree(d);
  return WV_OK;
}
extern "C" int32_t wv_dbg_exp_check(int32_t variant, const double* t_host, int64_t n, int64_t* n_bad_out, double* first_bad3_out) {
  using namespace wv;
  double *dt_, *fb;
  unsigned long long* bad;
  WV_HIP_TRY(hipMalloc(&dt_, sizeof(double) * n));
  WV_HIP_TRY(hipMalloc(&fb, sizeof(double) * 3));
  WV_HIP_TRY(hipMalloc(&bad, sizeof(unsigned long long)));
  WV_HIP_TRY(hipMemcpy(dt_, t_host, sizeof(double) * n, hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemset(bad, 0, sizeof(unsigned long long)));
  WV_HIP_TRY(hipMemset(fb, 0, sizeof(double) * 3));
  const size_t sh = sizeof(double) * EXPTAB_DOUBLES;
  switch (variant) {
    case 1: hipLaunchKernelGGL(exp_check_kernel<1>, dim3(1024), dim3(256), sh, nullptr, dt_, n, bad, fb); break;
    case 2: hipLaunchKernelGGL(exp_check_kernel<2>, dim3(1024), dim3(256), sh, nullptr, dt_, n, bad, fb); break;
    default: hipLaunchKernelGGL(exp_check_kernel<3>, dim3(1024), dim3(256), sh, nullptr, dt_, n, bad, fb); break;
  }
  WV_HIP_TRY(hipDeviceSynchronize());
  unsigned long long nb = 0;
  WV_HIP_TRY(hipMemcpy(&nb, bad, sizeof(nb), hipMemcpyDeviceToHost));
  if (first_bad3_out) WV_HIP_TRY(hipMemcpy(first_bad3_out, fb, sizeof(double) * 3, hipMemcpyDeviceToHost));
  if (n_bad_out) *n_bad_out = (int64_t)nb;
  hipFree(dt_); hipFree(fb); hipFree(bad);
  return WV_OK;
}
namespace wv {
#endif  // WV_EXPERIMENTS

// One factor on the thread's 4 x 4 elements.
//   MODE 0: k[a][c]  = factor value                        (a factor of a product)
//   MODE 1: k[a][c] += factor value                        (a single-factor term, accumulated into the sum)
//   MODE 2: s[m] = sum_ac wo[a][c] dk[a][c]/dparam_m, m < 3  (gradient contraction; k is not written)
//   MODE 3: k[a][c] *= factor value                        (further factors of a product, in place)
// Single-column factors (what full_kernel_build produces) take fused paths that never materialise a 4 x 4
// distance array; WV_ROW_FENCE keeps the scheduler from interleaving all 16 exponential chains at once
// (4 in flight per wave are enough beside 2-3 resident waves, 16 cost ~60 more VGPRs and spill).
#define WV_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE, bool MULTI = true>
__device__ __forceinline__ void factor_tile(const DevFactor& f, const double* __restrict__ fc,
                                   const int16_t* __restrict__ slots, const TileCtx& t, double (&k)[4][4],
                                   const double (&wo)[4][4], double (&s)[3]) {
  const int type = f.type;
  const double c0 = fc[0], var = fc[1];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  auto emit = [&](int a, int c, double scale, double v) {  // factor value = scale * v
    if (MODE == 1) k[a][c] = fma(scale, v, k[a][c]);
    else if (MODE == 0) k[a][c] = scale * v;
    else if (MODE == 3) k[a][c] *= scale * v;
  };
  if (type == WV_K_SE || type == WV_K_M12 || type == WV_K_M32 || type == WV_K_M52) {
    // The kernel type is a compile-time tag of the row loops (one copy of them per stationary type): with the type tested
    // per element hipcc kept the tests (scalar branches) inside the loops.  Measured at 512 fits, N = 2048, same box
    // (profiles/r04_gram_trace_ab.txt): trace pass 6.34 -> 5.78 ms, Gram pass 5.39 -> 5.29 (4 waves) / 5.14 ms (3 waves).
    auto rows = [&](auto tag) __attribute__((always_inline)) {
      constexpr int TY = decltype(tag)::value;
      // a row of four elements given their squared distances q on the factor's columns
      auto row = [&](int a, const double (&q)[4]) __attribute__((always_inline)) {
        double tt[4], e[4];
        if constexpr (TY == WV_K_SE) {
#pragma unroll
          for (int c = 0; c < 4; ++c) tt[c] = q[c] * c0;
#pragma unroll
          for (int c = 0; c < 4; ++c) e[c] = exp2_64(tt[c], t.tab);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (MODE == 2) { const double we = wo[a][c] * e[c]; s0 += we; s1 = fma(we, tt[c], s1); }
            else emit(a, c, var, e[c]);
          }
        } else {
          double u[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { u[c] = sqrt(q[c]) * c0; tt[c] = -u[c]; }
#pragma unroll
          for (int c = 0; c < 4; ++c) e[c] = exp2_64(tt[c], t.tab);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double z = u[c] * kLn2_64;
            double gg, dl;  // k = var gg, dk/dl = var dl / l
            if constexpr (TY == WV_K_M12) { gg = e[c]; dl = z * e[c]; }
            else if constexpr (TY == WV_K_M32) { gg = (1.0 + z) * e[c]; dl = z * z * e[c]; }
            else { const double z23 = z * z * (1.0 / 3.0); gg = (1.0 + z + z23) * e[c]; dl = z23 * (1.0 + z) * e[c]; }
            if (MODE == 2) { s0 = fma(wo[a][c], gg, s0); s1 = fma(wo[a][c], dl, s1); }
            else emit(a, c, var, gg);
          }
        }
      };
      if (!MULTI || f.ndims == 1) {
        const int sl = slots[f.slot0];
        double xj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xj[c] = t.fj[sl * GT + 4 * t.tx + c];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const double xi = t.fi[sl * GT + t.ty + 16 * a];
          double q[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double df = xi - xj[c];  // differences first: no cancellation for short lengthscales
            q[c] = df * df;
          }
          row(a, q);
          WV_ROW_FENCE();
        }
      } else {
        double q[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) q[a][c] = 0.0;
        for (int d = 0; d < f.ndims; ++d) {
          const int sl = slots[f.slot0 + d];
          double xj[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) xj[c] = t.fj[sl * GT + 4 * t.tx + c];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const double xi = t.fi[sl * GT + t.ty + 16 * a];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const double df = xi - xj[c];
              q[a][c] = fma(df, df, q[a][c]);
            }
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          row(a, q[a]);
          WV_ROW_FENCE();
        }
      }
    };
    if (type == WV_K_SE) rows(std::integral_constant<int, WV_K_SE>{});
    else if (type == WV_K_M12) rows(std::integral_constant<int, WV_K_M12>{});
    else if (type == WV_K_M32) rows(std::integral_constant<int, WV_K_M32>{});
    else rows(std::integral_constant<int, WV_K_M52>{});
    s1 *= fc[2];
  } else if (type == WV_K_PERIODIC) {
    // t = -sum_d (b sin(pi (x_i - x_j)/p))^2 with b = kCse / l folded into the ROW-side sin / cos features
    auto row = [&](int a, const double (&tt)[4], const double (&q2)[4]) __attribute__((always_inline)) {
      double e[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) e[c] = exp2_64(tt[c], t.tab);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (MODE == 2) {
          const double we = wo[a][c] * e[c];
          s0 += we;
          s1 = fma(we, tt[c], s1);
          s2 = fma(we, q2[c], s2);
        } else emit(a, c, var, e[c]);
      }
    };
    if (!MULTI || f.ndims == 1) {
      const int ss = slots[f.slot0], sc = slots[f.slot0 + 1], sr = slots[f.slot0 + 2];
      double sj[4], cj[4], xj[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sj[c] = t.fj[ss * GT + 4 * t.tx + c];
        cj[c] = t.fj[sc * GT + 4 * t.tx + c];
        xj[c] = MODE == 2 ? t.fj[sr * GT + 4 * t.tx + c] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double si = t.fi[ss * GT + t.ty + 16 * a], ci = t.fi[sc * GT + t.ty + 16 * a];
        const double xi = MODE == 2 ? t.fi[sr * GT + t.ty + 16 * a] : 0.0;
        double tr[4], q2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double sn = fma(si, cj[c], -(ci * sj[c]));
          q2[c] = 0.0;
          if (MODE == 2) q2[c] = (xi - xj[c]) * sn * fma(ci, cj[c], si * sj[c]);
          tr[c] = -(sn * sn);
        }
        row(a, tr, q2);
        WV_ROW_FENCE();
      }
    } else {
      double tt[4][4], q2[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) { tt[a][c] = 0.0; q2[a][c] = 0.0; }
      for (int d = 0; d < f.ndims; ++d) {
        const int ss = slots[f.slot0 + 3 * d], sc = slots[f.slot0 + 3 * d + 1], sr = slots[f.slot0 + 3 * d + 2];
        double sj[4], cj[4], xj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sj[c] = t.fj[ss * GT + 4 * t.tx + c];
          cj[c] = t.fj[sc * GT + 4 * t.tx + c];
          xj[c] = MODE == 2 ? t.fj[sr * GT + 4 * t.tx + c] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const double si = t.fi[ss * GT + t.ty + 16 * a], ci = t.fi[sc * GT + t.ty + 16 * a];
          const double xi = MODE == 2 ? t.fi[sr * GT + t.ty + 16 * a] : 0.0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double sn = fma(si, cj[c], -(ci * sj[c]));
            tt[a][c] = fma(-sn, sn, tt[a][c]);
            if (MODE == 2) q2[a][c] = fma((xi - xj[c]) * sn, fma(ci, cj[c], si * sj[c]), q2[a][c]);
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        row(a, tt[a], q2[a]);
        WV_ROW_FENCE();
      }
    }
    s1 *= fc[2];
    s2 *= fc[3];
  } else if (type == WV_K_LIN || type == WV_K_POLY || type == WV_K_CAT) {
    double q[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) q[a][c] = 0.0;
    if (type == WV_K_CAT) {
      const int sl = slots[f.slot0];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double ri = t.fi[sl * GT + t.ty + 16 * a];
#pragma unroll
        for (int c = 0; c < 4; ++c) q[a][c] = (ri == t.fj[sl * GT + 4 * t.tx + c]) ? 1.0 : 0.0;
      }
    } else {
      for (int d = 0; d < f.ndims; ++d) {
        const int sl = slots[f.slot0 + d];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const double xi = t.fi[sl * GT + t.ty + 16 * a];
#pragma unroll
          for (int c = 0; c < 4; ++c) q[a][c] = fma(xi, t.fj[sl * GT + 4 * t.tx + c], q[a][c]);
        }
      }
    }
    if (type == WV_K_POLY) {
      const double off = c0;
      const int deg = f.degree;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double base = var * q[a][c] + off, pm1 = ipow(base, deg - 1);
          if (MODE == 2) { const double dk = deg * pm1 * wo[a][c]; s0 = fma(dk, q[a][c], s0); s1 += dk; }
          else emit(a, c, pm1, base);
        }
    } else {  // Lin, Categorical: k = var q, dk/dvar = q
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (MODE == 2) s0 = fma(wo[a][c], q[a][c], s0);
          else emit(a, c, var, q[a][c]);
        }
    }
  } else if (type == WV_K_CONST) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (MODE == 2) s0 += wo[a][c];
        else emit(a, c, var, 1.0);
      }
  } else {  // WV_K_EMPTY
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) emit(a, c, 0.0, 0.0);
  }
  if (MODE == 2) { s[0] = s0; s[1] = s1; s[2] = s2; }
}

// sum over (unmasked) terms of the product of their factors, on the thread's 4 x 4 elements
// (MULTI = false: every factor acts on one column - the paths that keep 4 x 4 distance arrays are not instantiated)
template <bool MULTI = true>
__device__ __forceinline__ void kernel_sum_tile(const DevSpec* __restrict__ sp, const TileCtx& t, uint64_t term_mask,
                                       double (&sum)[4][4]) {
  const int16_t* slots = sp->fslots;
  double prod[4][4], dummy_s[3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum[a][c] = 0.0;
  for (int term = 0; term < sp->n_terms; ++term) {
    if (!((term_mask >> term) & 1ull)) continue;  // leave-component-out Gram matrices
    const int f0 = sp->term_start[term], f1 = sp->term_start[term + 1];
    if (f1 - f0 == 1) {
      factor_tile<1, MULTI>(sp->factors[f0], t.fc + 4 * f0, slots, t, sum, sum, dummy_s);
      continue;
    }
    factor_tile<0, MULTI>(sp->factors[f0], t.fc + 4 * f0, slots, t, prod, prod, dummy_s);
    for (int f = f0 + 1; f < f1; ++f) factor_tile<3, MULTI>(sp->factors[f], t.fc + 4 * f, slots, t, prod, prod, dummy_s);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum[a][c] += prod[a][c];
  }
}

// lower-triangular tile index -> (ti, tj), ti >= tj
__device__ inline void tri_decode(int t, int& ti, int& tj) {
  int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  while (i * (i + 1) / 2 > t) --i;
  ti = i;
  tj = t - i * (i + 1) / 2;
}

// Block structure (wv_spec.hpp, block_spans): the span table of a slot's structure, or null; and "tile (ti, tj) of 64 x
// 64 holds only structural zeros" = the rows that can share a label with tile row ti miss tile column tj entirely.
__device__ inline const int32_t* block_span_table(const int32_t* __restrict__ spans, const int32_t* __restrict__ spec_span,
                                                  int spec, int64_t Npad) {
  if (!spans) return nullptr;
  const int t = spec_span[spec];
  return t < 0 ? nullptr : spans + (int64_t)t * (Npad / 64) * 2;
}
__device__ inline bool tile_is_zero(const int32_t* __restrict__ span, int ti, int tj) {
  return span && ti != tj && (span[2 * ti] >= (tj + 1) * GT || span[2 * ti + 1] <= tj * GT);
}

constexpr int GRAM_TPW = 2;  // tiles per workgroup (the exp table is staged once per workgroup) ...
// ... but ONE with at most FEW_FITS (wv_device.hpp) fits in flight: more, shorter workgroups (latency regime)

// MASKED: missing outputs (row masks per output column), see trace_kernel.  MULTI = false: every factor of every
// structure of the problem acts on ONE column (what full_kernel_build produces) - fewer live registers, more waves.
// W4: held to 128 VGPRs (4 waves per SIMD, 23 VGPRs spilled) instead of 3 waves - diagnostic build only (WV_GRAM_WAVES=4).
template <typename R, bool MASKED, bool MULTI, bool W4 = false>
__global__ __launch_bounds__(256, W4 ? 4 : 3) void gram_kernel(const DevSpec* __restrict__ specs,
                                                   const int32_t* __restrict__ slot_spec,
                                                   const double* __restrict__ F, const double* __restrict__ FC,
                                                   const double* __restrict__ theta_all,
                                                   R* __restrict__ Kall, int64_t N,
                                                   int64_t Npad, int max_feat, int max_fact, int slot0, int ntiles,
                                                   int tpw, int write_upper, uint64_t term_mask, double diag_override,
                                                   const int32_t* __restrict__ spans,
                                                   const int32_t* __restrict__ spec_span,
                                                   const int32_t* __restrict__ col,
                                                   const uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = slot0 + blockIdx.y;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const double* th = theta_all + (int64_t)b * sp->n_params;
  R* K = Kall + (int64_t)b * Npad * Npad;
  const uint8_t* seen = MASKED ? mask + (int64_t)col[b] * Npad : nullptr;  // observed rows of this fit's output
  const int32_t* span = block_span_table(spans, spec_span, slot_spec[b], Npad);
  const int tid = threadIdx.x;
  double* tab = smem;
  double* fi = smem + EXPTAB_DOUBLES;
  double* fj = fi + sp->n_feat * GT;
  const double* Fr = F + (int64_t)b * 2 * max_feat * Npad;
  const double* Fc = Fr + (int64_t)max_feat * Npad;
  stage_exp_table(tab, tid);
  TileCtx t{fi, fj, tab + (tid & 31), FC + (int64_t)b * max_fact * 4, tid >> 4, tid & 15};
  const double noise = diag_override >= 0.0 ? diag_override : th[sp->noise_param];
  for (int tt = 0; tt < tpw; ++tt) {
    const int tile = blockIdx.x * tpw + tt;
    if (tile >= ntiles) break;
    int ti, tj;
    tri_decode(tile, ti, tj);
    double sum[4][4];
    if (tile_is_zero(span, ti, tj)) {  // block structure: no row of tile row ti shares a label with tile column tj
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) sum[a][c] = 0.0;
    } else {
      __syncthreads();
      stage_from_table(fi, Fr, sp->n_feat, Npad, ti * GT, tid);
      stage_from_table(fj, Fc, sp->n_feat, Npad, tj * GT, tid);
      __syncthreads();
      kernel_sum_tile<MULTI>(sp, t, term_mask, sum);
    }
    bool live_j[4];  // (padding rows and rows whose output is missing are identity rows / columns of K_y)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t j = (int64_t)tj * GT + 4 * t.tx + c;
      live_j[c] = MASKED ? seen[j] != 0 : j < N;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t i = (int64_t)ti * GT + t.ty + 16 * a;
      const bool live_i = MASKED ? seen[i] != 0 : i < N;
      double v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t j = (int64_t)tj * GT + 4 * t.tx + c;
        double x = sum[a][c];
        if (i == j) x += noise;
        if (!live_i || !live_j[c]) x = (i == j) ? 1.0 : 0.0;  // identity padding
        v[c] = x;
      }
      store4(K + i * Npad + (int64_t)tj * GT + 4 * t.tx, v);
      if (write_upper && ti != tj) {
#pragma unroll
        for (int c = 0; c < 4; ++c) K[((int64_t)tj * GT + 4 * t.tx + c) * Npad + i] = (R)v[c];
      }
    }
  }
}

// Rectangular cross-Gram K(Xrow, Xcol) for prediction at new inputs: rows = Mrow new points,
// columns = the N training points, no noise, padded entries zero.  One 64 x 64 tile per workgroup; features
// and factor constants are computed in place (the new rows have no feature table).
__global__ __launch_bounds__(256) void gram_rect_kernel(const DevSpec* __restrict__ sp,
                                                        const double* __restrict__ Xrow,
                                                        const double* __restrict__ Xcol,
                                                        const double* __restrict__ th,
                                                        double* __restrict__ out, int64_t Mrow,
                                                        int64_t N, int64_t ldo, int D, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int tid = threadIdx.x;
  double* tab = smem;
  double* fi = smem + EXPTAB_DOUBLES;
  double* fj = fi + sp->n_feat * GT;
  double* fcl = fj + sp->n_feat * GT;  // [n_factors][4]
  stage_exp_table(tab, tid);
  stage_features(fi, sp, Xrow, th, ti * GT, Mrow, D, tid, true);
  stage_features(fj, sp, Xcol, th, tj * GT, N, D, tid, false);
  if (tid < sp->n_factors) factor_constants(sp->factors[tid], th, fcl + 4 * tid);
  __syncthreads();
  TileCtx t{fi, fj, tab + (tid & 31), fcl, tid >> 4, tid & 15};
  double sum[4][4];
  kernel_sum_tile(sp, t, ~0ull, sum);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = (int64_t)ti * GT + t.ty + 16 * a;
    double v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t j = (int64_t)tj * GT + 4 * t.tx + c;
      v[c] = (i < Mrow && j < N) ? sum[a][c] : 0.0;
    }
    double* dst = out + i * ldo + (int64_t)tj * GT + 4 * t.tx;
    *reinterpret_cast<double2*>(dst) = make_double2(v[0], v[1]);
    *reinterpret_cast<double2*>(dst + 2) = make_double2(v[2], v[3]);
  }
}

void launch_gram_rect(Problem& p, hipStream_t st, const double* dXnew, int64_t Mrow, int64_t Mpad,
                      const double* dtheta_row, double* out) {
  const int tiles_n = (int)(p.Npad / GT);
  dim3 grid((unsigned)((Mpad / GT) * tiles_n)), blk(256);
  size_t sh = ((size_t)2 * p.hspec.n_feat * GT + EXPTAB_DOUBLES + 4 * p.hspec.n_factors) * sizeof(double);  // (structure 0)
  if (!exp_table_ok<gram_rect_kernel>()) return;
  hipLaunchKernelGGL(gram_rect_kernel, grid, blk, sh, st, p.dspec, dXnew, p.dX, dtheta_row, out, Mrow, p.N,
                     p.Npad, p.D, tiles_n);
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the 4 lanes of a quad, result in every lane of it: two DPP quad permutes (register to register,
// no LDS round trip - the full 64-lane reductions of round 1 were a chain of six ds_bpermute latencies per sum)
__device__ inline double quad_sum(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false),
                        __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
  lo = __double2loint(v);
  hi = __double2hiint(v);
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false),
                        __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
  return v;
}

constexpr int TRACE_TPW = 4;  // tiles per workgroup

// part[b][wg][p] = sum over this workgroup's tiles of  w_ij W_ij dK_ij/dtheta_p
// (w = 1 below the diagonal, 1/2 on it, 0 above; noise_param gets 1/2 W_ii).
// Every (trainable factor, parameter) pair owns a "sum slot" (DevFactor::sbase); a thread's contribution is
// reduced over its quad in registers and added to the slot's 64 per-quad accumulators in LDS (ds_add_f64, fire
// and forget); one full reduction per slot happens at the very end of the workgroup instead of one per slot, factor
// and tile.
// MULTI = false: every factor of every structure acts on ONE column (what full_kernel_build produces): the paths that
// keep 4 x 4 distance arrays are not instantiated, 186 instead of 230 VGPRs, and the kernel is held to 168 (3 waves per
// SIMD instead of 2; 11 VGPRs spill) - the pass is latency-bound, not issue-bound (section 4.0 of DESIGN.md).
// MASKED: the problem has missing outputs (row masks per output column); its own instantiation so that the common case
// keeps the register budget it was tuned for.
// (Held to 128 VGPRs for a fourth wave per SIMD, as the Gram kernel is, it spills 54 VGPRs and takes 19.5 ms instead of
// 7.4 per 512 fits - measured in round 3 and not kept.)
// PROD = false: no structure of the problem has a product term (sums of single factors - the headline, C2, C5 and most
// kernel-search candidates): the 4 x 4 copy of the weights that products need is not instantiated.
// QUADOUT (few fits in flight: one tile per workgroup): the workgroup leaves its per-quad accumulators red[slot][64] in
// `quad` instead of reducing them - trace_group_kernel then adds the tiles of every group of TRACE_TPW in the order a
// TRACE_TPW-tile workgroup would have, so that a gradient's reduction tree - and with it a fit's whole trajectory -
// does not depend on how many fits share the launch.
// KSUM (float32 contexts): the workgroup also accumulates sum_ij w_ij W_ij K_y,ij - K_y recomputed in float64 on the tile
// like the Gram pass does - into one more sum slot (index n_red) and writes it as column n_params of its record (records
// are n_params + 1 wide then).  It is the first-order correction that makes the float32 factor's LML a float64 number
// (reduce_kernel).
template <typename R, bool MULTI, bool MASKED, bool PROD = true, bool W4 = false, bool QUADOUT = false, bool KSUM = false>
__global__ __launch_bounds__(256, MULTI ? 2 : (W4 ? 4 : 3)) void trace_kernel(const DevSpec* __restrict__ specs,
                                                    const int32_t* __restrict__ slot_spec,
                                                    const double* __restrict__ F, const double* __restrict__ FC,
                                                    const double* __restrict__ theta_all,
                                                    const R* __restrict__ Kinv_all,
                                                    const double* __restrict__ alpha_all,
                                                    double* __restrict__ part, int64_t N,
                                                    int64_t Npad, int max_feat, int max_fact, int slot0, int ntiles,
                                                    int tpw, int part_stride, const int32_t* __restrict__ spans,
                                                    const int32_t* __restrict__ spec_span,
                                                    const int32_t* __restrict__ col,
                                                    const uint8_t* __restrict__ mask,
                                                    double* __restrict__ quad_out = nullptr, int quad_stride = 0,
                                                    const double* __restrict__ pdiag_all = nullptr, int64_t pdiag_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = slot0 + blockIdx.y;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const int16_t* slots = sp->fslots;
  const int np = sp->n_params, nred = sp->n_red;
  const R* Kinv = Kinv_all + (int64_t)b * Npad * Npad;
  const double* alpha = alpha_all + (int64_t)b * Npad;
  const double* Fr = F + (int64_t)b * 2 * max_feat * Npad;
  const double* Fc = Fr + (int64_t)max_feat * Npad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, quad = tid >> 2;
  const bool quad_leader = (tid & 3) == 0;
  double* tab = smem;
  double* fi = smem + EXPTAB_DOUBLES;
  double* fj = fi + sp->n_feat * GT;
  double* red = fj + sp->n_feat * GT;  // [nred (+ 1: KSUM)][64 quads]
  constexpr int NX = KSUM ? 1 : 0;
  double* pacc = red + (nred + NX) * 64;      // [np]
  stage_exp_table(tab, tid);
  for (int q = tid; q < (nred + NX) * 64; q += 256) red[q] = 0.0;
  for (int q = tid; q < np; q += 256) pacc[q] = 0.0;
  TileCtx t{fi, fj, tab + (tid & 31), FC + (int64_t)b * max_fact * 4, tid >> 4, tid & 15};
  auto contribute = [&](int slot, double v) {
    v = quad_sum(v);
    if (quad_leader) __hip_atomic_fetch_add(&red[slot * 64 + quad], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  const int32_t* span = block_span_table(spans, spec_span, slot_spec[b], Npad);
  const uint8_t* seen = MASKED ? mask + (int64_t)col[b] * Npad : nullptr;  // observed rows of this fit's output
  for (int tt = 0; tt < tpw; ++tt) {
    const int tile = blockIdx.x * tpw + tt;
    if (tile >= ntiles) break;
    int ti, tj;
    tri_decode(tile, ti, tj);
    if (tile_is_zero(span, ti, tj)) continue;  // dK/dtheta vanishes on the whole tile
    __syncthreads();
    stage_from_table(fi, Fr, sp->n_feat, Npad, ti * GT, tid);
    stage_from_table(fj, Fc, sp->n_feat, Npad, tj * GT, tid);
    __syncthreads();

    // weights: W_ij = alpha_i alpha_j - Kinv_ij with the symmetry / padding mask folded in
    double w[4][4];
    double aj[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aj[c] = alpha[(int64_t)tj * GT + 4 * t.tx + c];
    double snoise = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t i = (int64_t)ti * GT + t.ty + 16 * a;
      const double ai = alpha[i];
      double kv[4];
      load4(Kinv + i * Npad + (int64_t)tj * GT + 4 * t.tx, kv);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t j = (int64_t)tj * GT + 4 * t.tx + c;
        double m = (i > j) ? 1.0 : ((i == j) ? 0.5 : 0.0);
        if (MASKED ? (seen[i] == 0 || seen[j] == 0) : (i >= N || j >= N)) m = 0.0;
        // (KSUM: the diagonal of Mt'Mt in float64 - the float32 product's diagonal carries a small systematic error that
        //  would enter the corrected value N-fold)
        if (KSUM && i == j && pdiag_all) kv[c] = pdiag_all[(int64_t)b * pdiag_stride + i];
        w[a][c] = m * (ai * aj[c] - kv[c]);
        if (i == j) snoise += w[a][c];
      }
    }
    if (ti == tj) contribute(nred - 1, snoise);  // (workgroup-uniform branch)
    double wk = 0.0;   // (KSUM) this thread's share of sum w W K_y over the tile
    if (KSUM && ti == tj) wk = snoise * theta_all[(int64_t)b * np + sp->noise_param];  // K_y's diagonal carries the noise

    for (int term = 0; term < sp->n_terms; ++term) {
      const int f0 = sp->term_start[term], f1 = sp->term_start[term + 1];
      // (KSUM) sum w k_term comes for free from the gradient sums when a factor g of the term has a trainable variance
      // and k_g = var_g x (what its variance sum weights): sum w k_term = var_g s0_g.  Otherwise the term is evaluated.
      bool term_done = !KSUM;
      for (int g = f0; g < f1; ++g) {
        const DevFactor fg = sp->factors[g];
        if (fg.trainable == 0) continue;
        double s[3], unused[4][4];
        if (!PROD || f1 - f0 == 1) {
          factor_tile<2, MULTI>(fg, t.fc + 4 * g, slots, t, unused, w, s);
        } else {
          double wo[4][4];
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) wo[a][c] = w[a][c];
          for (int f = f0; f < f1; ++f)
            if (f != g) factor_tile<3, MULTI>(sp->factors[f], t.fc + 4 * f, slots, t, wo, wo, s);
          factor_tile<2, MULTI>(fg, t.fc + 4 * g, slots, t, unused, wo, s);
        }
        contribute(fg.sbase, s[0]);
        if (fg.nsum > 1) contribute(fg.sbase + 1, s[1]);
        if (fg.nsum > 2) contribute(fg.sbase + 2, s[2]);
        if (KSUM && !term_done && (fg.trainable & 1) && fg.type != WV_K_POLY && fg.type != WV_K_EMPTY) {
          wk = fma(t.fc[4 * g + 1], s[0], wk);
          term_done = true;
        }
      }
      if (KSUM && !term_done) {   // (every variance of the term frozen, or only a Polynomial's trainable)
        double ks[4][4];
        kernel_sum_tile<MULTI>(sp, t, 1ull << term, ks);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) wk = fma(w[a][c], ks[a][c], wk);
      }
    }
    if constexpr (KSUM) contribute(nred, wk);
  }
  __syncthreads();
  if constexpr (QUADOUT) {  // (tpw == 1: this workgroup's tile is blockIdx.x; a skipped zero tile leaves zeros)
    double* q = quad_out + ((int64_t)blockIdx.y * ntiles + blockIdx.x) * quad_stride;
    for (int i = tid; i < (nred + NX) * 64; i += 256) q[i] = red[i];
    return;
  }
  // one 64-lane reduction per slot (wave w takes slots w, w + 4, ...); the total goes back into red[slot * 64]
  for (int sidx = wave; sidx < nred + NX; sidx += 4) {
    const double v = wave_sum(red[sidx * 64 + lane]);
    if (lane == 0) red[sidx * 64] = v;
  }
  __syncthreads();
  // slots -> parameters (a parameter shared by several factors collects all of them)
  for (int f = tid; f < sp->n_factors; f += 256) {
    const DevFactor ff = sp->factors[f];
    if (ff.sbase < 0) continue;
    for (int m = 0; m < ff.nsum; ++m)
      if (ff.trainable & (1 << m))
        __hip_atomic_fetch_add(&pacc[ff.p[m]], red[(ff.sbase + m) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (tid == 0) __hip_atomic_fetch_add(&pacc[sp->noise_param], red[(nred - 1) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  double* out = part + ((int64_t)b * part_stride + blockIdx.x) * (np + NX);
  for (int p = tid; p < np; p += 256) out[p] = pacc[p];
  if (KSUM && tid == 0) out[np] = red[nred * 64];
}

// Few fits in flight: the tiles were traced one per workgroup (trace_kernel<..., QUADOUT>); group g adds the per-quad
// accumulators of tiles TRACE_TPW g ... in tile order - exactly what one workgroup walking those tiles does with its LDS
// accumulators - then reduces over the 64 quads and maps slots to parameters like the tail of trace_kernel, and writes
// the record part[b][g] that a TRACE_TPW-tile workgroup would have written.
__global__ __launch_bounds__(256) void trace_group_kernel(const DevSpec* __restrict__ specs,
                                                          const int32_t* __restrict__ slot_spec,
                                                          const double* __restrict__ quad, double* __restrict__ part,
                                                          int slot0, int ntiles, int quad_stride, int part_stride,
                                                          int ksum) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = slot0 + blockIdx.y, g = blockIdx.x;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const int np = sp->n_params, nred = sp->n_red;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* red = smem;          // [nred (+ 1: float32 contexts, trace_kernel's KSUM slot)] totals
  double* pacc = smem + nred + ksum;  // [np]
  for (int q = tid; q < np; q += 256) pacc[q] = 0.0;
  const int t0 = g * TRACE_TPW, t1 = min(ntiles, t0 + TRACE_TPW);
  for (int sidx = wave; sidx < nred + ksum; sidx += 4) {
    double a = 0.0;
    for (int t = t0; t < t1; ++t) a += quad[((int64_t)blockIdx.y * ntiles + t) * quad_stride + sidx * 64 + lane];
    const double v = wave_sum(a);
    if (lane == 0) red[sidx] = v;
  }
  __syncthreads();
  for (int f = tid; f < sp->n_factors; f += 256) {
    const DevFactor ff = sp->factors[f];
    if (ff.sbase < 0) continue;
    for (int m = 0; m < ff.nsum; ++m)
      if (ff.trainable & (1 << m))
        __hip_atomic_fetch_add(&pacc[ff.p[m]], red[ff.sbase + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (tid == 0) __hip_atomic_fetch_add(&pacc[sp->noise_param], red[nred - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  double* out = part + ((int64_t)b * part_stride + g) * (np + ksum);
  for (int p = tid; p < np; p += 256) out[p] = pacc[p];
  if (ksum && tid == 0) out[np] = red[nred];
}

static void launch_features(Problem& p, hipStream_t st, int slot0, int nb) {
  dim3 grid((unsigned)((p.Npad + 255) / 256), nb), blk(256);
  hipLaunchKernelGGL(feature_kernel, grid, blk, 0, st, p.dspec, p.dslotspec, p.dX, p.dtheta, p.dfeat, p.dfconst, p.N,
                     p.Npad, p.D, p.max_feat, p.max_fact, slot0);
}

static size_t gram_lds_bytes(const Problem& p) { return ((size_t)2 * p.max_feat * GT + EXPTAB_DOUBLES) * sizeof(double); }
static int gram_tpw(int nb) { return nb <= FEW_FITS ? 1 : GRAM_TPW; }

template <typename R>
static void launch_gram_t(Problem& p, hipStream_t st, R* Kall, int slot0, int nb, int write_upper, uint64_t term_mask,
                          double diag) {
  const int tpw = gram_tpw(nb);
  dim3 grid((p.ntiles() + tpw - 1) / tpw, nb), blk(256);
#define WV_GRAM_LAUNCH(MA, MU)                                                                                          \
  if (exp_table_ok<gram_kernel<R, MA, MU>>())                                                                           \
  hipLaunchKernelGGL((gram_kernel<R, MA, MU>), grid, blk, gram_lds_bytes(p), st, p.dspec, p.dslotspec, p.dfeat, p.dfconst, \
                     p.dtheta, Kall, p.N, p.Npad, p.max_feat, p.max_fact, slot0, p.ntiles(), tpw, write_upper, term_mask, \
                     diag, p.n_span_tables ? p.dspans : nullptr, p.dspec_span, p.dcol, p.dmask)
  static const bool force_multi = [] { const char* e = experiment_env("WV_TRACE_MULTI"); return e && atoi(e) != 0; }();
  // (512 fits at N = 2048, same box, round 4: 5.14 ms at 3 waves per SIMD - 150 VGPRs, nothing spilled - and 5.29 at 4
  //  with 23 VGPRs spilled, which also cost 6.7 GB of scratch traffic per launch in round 3: three waves)
  static const int waves = [] { const char* e = experiment_env("WV_GRAM_WAVES"); return e ? atoi(e) : 3; }();
  const bool multi = p.any_multidim || force_multi;
  if (p.dmask) {
    if (multi) { WV_GRAM_LAUNCH(true, true); } else { WV_GRAM_LAUNCH(true, false); }
  } else if (multi) {
    WV_GRAM_LAUNCH(false, true);
  } else if (waves == 4) {
    if (exp_table_ok<gram_kernel<R, false, false, true>>())
      hipLaunchKernelGGL((gram_kernel<R, false, false, true>), grid, blk, gram_lds_bytes(p), st, p.dspec, p.dslotspec, p.dfeat,
                         p.dfconst, p.dtheta, Kall, p.N, p.Npad, p.max_feat, p.max_fact, slot0, p.ntiles(), tpw, write_upper,
                         term_mask, diag, p.n_span_tables ? p.dspans : nullptr, p.dspec_span, p.dcol, p.dmask);
  } else {
    WV_GRAM_LAUNCH(false, false);
  }
#undef WV_GRAM_LAUNCH
}

// y1 = K_y w1, y2 = K_y w2 with K_y recomputed in float64 tile by tile (never stored): one workgroup per 64-row tile row,
// walking all tile columns.  Rows that are padding (or whose output is missing) are identity rows / columns of K_y.  Used by
// the float32 contexts' second-order value correction (launch_f32_refine, wv_chol.hip): w1 = Mt'v, w2 = alpha.
template <bool MULTI, bool MASKED>
__global__ __launch_bounds__(256, 3) void gram_matvec2_kernel(const DevSpec* __restrict__ specs,
                                                              const int32_t* __restrict__ slot_spec,
                                                              const double* __restrict__ F, const double* __restrict__ FC,
                                                              const double* __restrict__ theta_all,
                                                              const double* __restrict__ w1_all, const double* __restrict__ w2_all,
                                                              double* __restrict__ y1_all, double* __restrict__ y2_all,
                                                              int64_t vec_stride, int64_t alpha_stride, int64_t N, int64_t Npad,
                                                              int max_feat, int max_fact, int slot0,
                                                              const int32_t* __restrict__ col, const uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = slot0 + blockIdx.y, ti = blockIdx.x;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const double* th = theta_all + (int64_t)b * sp->n_params;
  const uint8_t* seen = MASKED ? mask + (int64_t)col[b] * Npad : nullptr;
  const int tid = threadIdx.x;
  double* tab = smem;
  double* fi = smem + EXPTAB_DOUBLES;
  double* fj = fi + sp->n_feat * GT;
  const double* Fr = F + (int64_t)b * 2 * max_feat * Npad;
  const double* Fc = Fr + (int64_t)max_feat * Npad;
  const double* w1 = w1_all + (int64_t)b * vec_stride;
  const double* w2 = w2_all + (int64_t)b * alpha_stride;
  stage_exp_table(tab, tid);
  stage_from_table(fi, Fr, sp->n_feat, Npad, ti * GT, tid);
  TileCtx t{fi, fj, tab + (tid & 31), FC + (int64_t)b * max_fact * 4, tid >> 4, tid & 15};
  const double noise = th[sp->noise_param];
  double y1[4] = {0, 0, 0, 0}, y2[4] = {0, 0, 0, 0};
  bool live_i[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = (int64_t)ti * GT + t.ty + 16 * a;
    live_i[a] = MASKED ? seen[i] != 0 : i < N;
  }
  const int ntile = (int)(Npad / GT);
  for (int tj = 0; tj < ntile; ++tj) {
    __syncthreads();
    stage_from_table(fj, Fc, sp->n_feat, Npad, tj * GT, tid);
    __syncthreads();
    double sum[4][4];
    kernel_sum_tile<MULTI>(sp, t, ~0ull, sum);
    double a1[4], a2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t j = (int64_t)tj * GT + 4 * t.tx + c;
      const bool lj = MASKED ? seen[j] != 0 : j < N;
      a1[c] = lj ? w1[j] : 0.0;
      a2[c] = lj ? w2[j] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        y1[a] = fma(sum[a][c], a1[c], y1[a]);
        y2[a] = fma(sum[a][c], a2[c], y2[a]);
      }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {   // the sixteen threads of a row are sixteen consecutive lanes
      y1[a] += __shfl_xor(y1[a], o, 64);
      y2[a] += __shfl_xor(y2[a], o, 64);
    }
    if (t.tx == 0) {
      const int64_t i = (int64_t)ti * GT + t.ty + 16 * a;
      const double u1 = w1[i], u2 = w2[i];
      y1_all[(int64_t)b * vec_stride + i] = live_i[a] ? y1[a] + noise * u1 : u1;
      y2_all[(int64_t)b * vec_stride + i] = live_i[a] ? y2[a] + noise * u2 : u2;
    }
  }
}

void launch_gram_matvec2(Problem& p, hipStream_t st, int slot0, int nb, const double* w1, const double* w2, double* y1,
                         double* y2, int64_t vec_stride, int64_t alpha_stride) {
  const dim3 grid((unsigned)(p.Npad / GT), nb), blk(256);
  const size_t sh = gram_lds_bytes(p);
#define WV_GMV(MU, MA)                                                                                                   \
  if (exp_table_ok<gram_matvec2_kernel<MU, MA>>())                                                                       \
  hipLaunchKernelGGL((gram_matvec2_kernel<MU, MA>), grid, blk, sh, st, p.dspec, p.dslotspec, p.dfeat, p.dfconst, p.dtheta, \
                     w1, w2, y1, y2, vec_stride, alpha_stride, p.N, p.Npad, p.max_feat, p.max_fact, slot0, p.dcol, p.dmask)
  if (p.any_multidim) { if (p.dmask) { WV_GMV(true, true); } else { WV_GMV(true, false); } }
  else { if (p.dmask) { WV_GMV(false, true); } else { WV_GMV(false, false); } }
#undef WV_GMV
}

void launch_gram_masked(Problem& p, hipStream_t st, int slot0, int nb, uint64_t term_mask, double diag) {
  launch_features(p, st, slot0, nb);  // (the parameter rows of these slots may have changed since the last call)
  if (p.f32) launch_gram_t<float>(p, st, p.Kbuf32, slot0, nb, 0, term_mask, diag);
  else launch_gram_t<double>(p, st, p.Kbuf, slot0, nb, 0, term_mask, diag);
}

void launch_gram(Problem& p, hipStream_t st, int slot0, int nb, int write_upper) {
  p.timer.begin(st, ST_GRAM);
  launch_features(p, st, slot0, nb);  // the table stays valid for this evaluation's trace pass
  if (p.f32) launch_gram_t<float>(p, st, p.Kbuf32, slot0, nb, write_upper, ~0ull, -1.0);
  else launch_gram_t<double>(p, st, p.Kbuf, slot0, nb, write_upper, ~0ull, -1.0);
  p.timer.end(st);
  p.timer.work(ST_GRAM, 0.0, (double)nb * (p.f32 ? 4.0 : 8.0) * ((double)p.Npad * (p.Npad + GT) / 2.0 + (double)p.N * p.D));
}

// Workgroups per fit of the trace pass over nb fits (= partial-sum records per fit that the reduce pass adds up); a
// fit's records start at slot * trace_part_stride(p) whatever nb is, so groups of different size never overlap.
int trace_nwg(const Problem& p, int /*nb*/) { return (p.ntiles() + TRACE_TPW - 1) / TRACE_TPW; }  // records per fit: one
// per group of TRACE_TPW tiles in every regime (written by the workgroup that walked them, or by trace_group_kernel)
int trace_part_stride(const Problem& p) { return p.ntiles(); }

template <typename RT, bool MU, bool MA, bool PROD, bool W4>
static void trace_launch(Problem& p, hipStream_t st, const RT* kbuf, int slot0, int nb) {
  constexpr bool KS = std::is_same<RT, float>::value;   // float32 contexts: + the K_y-weighted sum (Problem::max_red counts its slot)
  // (WV_F32_ORDER >= 2: launch_f32_refine left the float64 diagonal of Mt'Mt there)
  const double* pdiag = (p.df32r && f32_order() >= 2) ? p.df32r + (int64_t)F32R_PD * p.Npad : nullptr;
  // few fits in flight: one tile per workgroup, per-quad accumulators to p.dquad, groups summed by trace_group_kernel
  double* quad = nullptr;
  if (nb <= FEW_FITS && nb <= p.quad_fits && p.dquad)
    for (size_t gi = 0; gi < p.groups.size(); ++gi)
      if (p.groups[gi].slot0 == slot0) quad = p.dquad + (int64_t)gi * p.quad_fits * p.ntiles() * p.quad_stride;
  const size_t sh = gram_lds_bytes(p) + ((size_t)64 * p.max_red + p.np_stride) * sizeof(double);
  const dim3 blk(256);
  if (quad) {
    if (!exp_table_ok<trace_kernel<RT, MU, MA, PROD, W4, true, KS>>()) return;
    hipLaunchKernelGGL((trace_kernel<RT, MU, MA, PROD, W4, true, KS>), dim3(p.ntiles(), nb), blk, sh, st, p.dspec, p.dslotspec,
                       p.dfeat, p.dfconst, p.dtheta, kbuf, p.dalpha, p.dpart, p.N, p.Npad, p.max_feat, p.max_fact, slot0,
                       p.ntiles(), 1, trace_part_stride(p), p.n_span_tables ? p.dspans : nullptr, p.dspec_span, p.dcol,
                       p.dmask, quad, (int)p.quad_stride, pdiag, (int64_t)F32R_VECS * p.Npad);
    hipLaunchKernelGGL(trace_group_kernel, dim3(trace_nwg(p, nb), nb), blk, sizeof(double) * (p.max_red + p.np_stride), st,
                       p.dspec, p.dslotspec, quad, p.dpart, slot0, p.ntiles(), (int)p.quad_stride, trace_part_stride(p), KS ? 1 : 0);
  } else {
    if (!exp_table_ok<trace_kernel<RT, MU, MA, PROD, W4, false, KS>>()) return;
    hipLaunchKernelGGL((trace_kernel<RT, MU, MA, PROD, W4, false, KS>), dim3(trace_nwg(p, nb), nb), blk, sh, st, p.dspec,
                       p.dslotspec, p.dfeat, p.dfconst, p.dtheta, kbuf, p.dalpha, p.dpart, p.N, p.Npad, p.max_feat, p.max_fact,
                       slot0, p.ntiles(), TRACE_TPW, trace_part_stride(p), p.n_span_tables ? p.dspans : nullptr,
                       p.dspec_span, p.dcol, p.dmask, nullptr, 0, pdiag, (int64_t)F32R_VECS * p.Npad);
  }
}

void launch_trace(Problem& p, hipStream_t st, int slot0, int nb) {
  p.timer.begin(st, ST_TRACE);
  static const bool force_multi = [] { const char* e = experiment_env("WV_TRACE_MULTI"); return e && atoi(e) != 0; }();
  const bool multi = p.any_multidim || force_multi;
  // (512 fits at N = 2048, same box, round 4: without the product paths 6.34 ms at 3 waves per SIMD - 131 VGPRs, nothing
  // spilled - and 5.88 at 4 waves - 128 VGPRs, 2 spilled; WV_TRACE_WAVES=3 selects the former in the diagnostic build)
  static const int waves = [] { const char* e = experiment_env("WV_TRACE_WAVES"); return e ? atoi(e) : 4; }();
  if (!p.f32 && !multi && !p.any_product && !p.dmask) {  // sums of single-column factors, float64, no missing outputs
    if (waves == 4) trace_launch<double, false, false, false, true>(p, st, p.Kbuf, slot0, nb);
    else trace_launch<double, false, false, false, false>(p, st, p.Kbuf, slot0, nb);
  } else if (p.f32) {
    if (multi) { if (p.dmask) trace_launch<float, true, true, true, false>(p, st, p.Kbuf32, slot0, nb); else trace_launch<float, true, false, true, false>(p, st, p.Kbuf32, slot0, nb); }
    else { if (p.dmask) trace_launch<float, false, true, true, false>(p, st, p.Kbuf32, slot0, nb); else trace_launch<float, false, false, true, false>(p, st, p.Kbuf32, slot0, nb); }
  } else {
    if (multi) { if (p.dmask) trace_launch<double, true, true, true, false>(p, st, p.Kbuf, slot0, nb); else trace_launch<double, true, false, true, false>(p, st, p.Kbuf, slot0, nb); }
    else { if (p.dmask) trace_launch<double, false, true, true, false>(p, st, p.Kbuf, slot0, nb); else trace_launch<double, false, false, true, false>(p, st, p.Kbuf, slot0, nb); }
  }
  p.timer.end(st);
  p.timer.work(ST_TRACE, 0.0, (double)nb * (p.f32 ? 4.0 : 8.0) * ((double)p.Npad * (p.Npad + GT) / 2.0 + (double)p.N * p.D));
}

}  // namespace wv
