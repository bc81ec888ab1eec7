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
// Both kernels stage per-row "features" of the tile's 64 rows and 64 columns in LDS:
// raw x_d, rint(x_d) for Categorical, and sinpi(x_d/p), cospi(x_d/p) for Periodic so that
// sin(pi (x_i - x_j)/p) = s_i c_j - c_i s_j needs no per-element trigonometry.
// Thread layout: 256 threads as 16 x 16, each owning rows ty + 16a (a < 4) and 4 contiguous
// columns 4 tx .. 4 tx + 3 (32-byte row segments -> 512 B contiguous per 16 threads).
#include "wv_device.hpp"

namespace wv {

constexpr double kSqrt3 = 1.7320508075688772;
constexpr double kSqrt5 = 2.23606797749979;
constexpr double kPi = 3.141592653589793;

struct TileCtx {
  const double* fi;  // [n_feat][GT] row features
  const double* fj;  // [n_feat][GT] column features
  int ty, tx;
};

// exp(x) for x <= 0 (every kernel function here exponentiates a non-positive argument): Cody-Waite reduction
// r = x - k ln 2, |r| <= ln 2 / 2, degree-13 Taylor polynomial (truncation 4e-18 relative), scaling by ldexp.
// The library exp spends a fifth of its instructions on overflow / special-case selects that cannot occur
// here; arguments below -800 are clamped (the result underflows to zero either way).
__device__ inline double exp_nonpos(double x) {
  x = fmax(x, -800.0);
  const double k = rint(x * 1.4426950408889634);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;  // 1/13!
  p = fma(p, r, 2.08767569878681e-09);
  p = fma(p, r, 2.505210838544172e-08);
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.7557319223985893e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 1.984126984126984e-04);
  p = fma(p, r, 1.388888888888889e-03);
  p = fma(p, r, 8.333333333333333e-03);
  p = fma(p, r, 4.1666666666666664e-02);
  p = fma(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

__device__ inline double ipow(double b, int e) {
  double r = 1.0;
  for (int i = 0; i < e; ++i) r *= b;
  return r;
}

// Stage features for `rows0..rows0+63` into dst[n_feat][GT].
__device__ inline void stage_features(double* dst, const DevSpec* __restrict__ sp,
                                      const double* __restrict__ X, const double* __restrict__ th,
                                      int row0, int64_t N, int D, int tid) {
  const int total = sp->n_feat * GT;
  for (int q = tid; q < total; q += 256) {
    int s = q / GT, r = q % GT;
    const FeatOp op = sp->feat[s];
    int64_t row = row0 + r;
    double x = (row < N) ? X[row * D + op.dim] : 0.0;
    double v;
    switch (op.kind) {
      case FEAT_RINT: v = rint(x); break;
      case FEAT_SINPI: v = sinpi(x / th[op.param]); break;
      case FEAT_COSPI: v = cospi(x / th[op.param]); break;
      default: v = x;
    }
    dst[q] = v;
  }
}

// Evaluate one factor on the thread's 4x4 elements.
// GRAD == false: k[a][c] = factor value.
// GRAD == true : additionally contracts wo[a][c] * dk/dparam_q into s[q] (q < 3).
template <bool GRAD>
__device__ inline void eval_factor(const DevFactor& f, const int16_t* __restrict__ slots,
                                   const double* __restrict__ th, const TileCtx& t,
                                   double (&k)[4][4], const double (&wo)[4][4], double (&s)[3]) {
  const int type = f.type;
  double q[4][4], q2[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) { q[a][c] = 0.0; q2[a][c] = 0.0; }

  if (type == WV_K_PERIODIC) {
    for (int d = 0; d < f.ndims; ++d) {
      const int ss = slots[f.slot0 + 3 * d], sc = slots[f.slot0 + 3 * d + 1], sr = slots[f.slot0 + 3 * d + 2];
      double si[4], ci[4], xi[4], sj[4], cj[4], xj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        si[a] = t.fi[ss * GT + t.ty + 16 * a];
        ci[a] = t.fi[sc * GT + t.ty + 16 * a];
        xi[a] = t.fi[sr * GT + t.ty + 16 * a];
        sj[a] = t.fj[ss * GT + 4 * t.tx + a];
        cj[a] = t.fj[sc * GT + 4 * t.tx + a];
        xj[a] = t.fj[sr * GT + 4 * t.tx + a];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double sn = si[a] * cj[c] - ci[a] * sj[c];  // sin(pi (xi-xj)/p)
          q[a][c] += sn * sn;
          if (GRAD) {
            double cs = ci[a] * cj[c] + si[a] * sj[c];  // cos(pi (xi-xj)/p)
            q2[a][c] += (xi[a] - xj[c]) * sn * cs;
          }
        }
    }
  } else if (type == WV_K_CAT) {
    const int sl = slots[f.slot0];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double ri = t.fi[sl * GT + t.ty + 16 * a];
#pragma unroll
      for (int c = 0; c < 4; ++c) q[a][c] = (ri == t.fj[sl * GT + 4 * t.tx + c]) ? 1.0 : 0.0;
    }
  } else if (type == WV_K_LIN || type == WV_K_POLY) {
    for (int d = 0; d < f.ndims; ++d) {
      const int sl = slots[f.slot0 + d];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double xi = t.fi[sl * GT + t.ty + 16 * a];
#pragma unroll
        for (int c = 0; c < 4; ++c) q[a][c] += xi * t.fj[sl * GT + 4 * t.tx + c];
      }
    }
  } else if (type <= WV_K_M52) {  // SE / Matern: squared distance
    for (int d = 0; d < f.ndims; ++d) {
      const int sl = slots[f.slot0 + d];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double xi = t.fi[sl * GT + t.ty + 16 * a];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double df = xi - t.fj[sl * GT + 4 * t.tx + c];
          q[a][c] += df * df;
        }
      }
    }
  }

  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  switch (type) {
    case WV_K_SE: {
      const double var = th[f.p[0]], ell = th[f.p[1]];
      const double hi = -0.5 / (ell * ell), i3 = 1.0 / (ell * ell * ell);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double e = exp_nonpos(q[a][c] * hi);
          k[a][c] = var * e;
          if (GRAD) { s0 += wo[a][c] * e; s1 += wo[a][c] * (var * e * q[a][c] * i3); }
        }
    } break;
    case WV_K_M12: {
      const double var = th[f.p[0]], ell = th[f.p[1]], iell = 1.0 / ell;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double sr = sqrt(q[a][c]) * iell, e = exp_nonpos(-sr);
          k[a][c] = var * e;
          if (GRAD) { s0 += wo[a][c] * e; s1 += wo[a][c] * (var * e * sr * iell); }
        }
    } break;
    case WV_K_M32: {
      const double var = th[f.p[0]], ell = th[f.p[1]], iell = 1.0 / ell;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double sr = sqrt(q[a][c]) * iell, e = exp_nonpos(-kSqrt3 * sr), gg = (1.0 + kSqrt3 * sr) * e;
          k[a][c] = var * gg;
          if (GRAD) { s0 += wo[a][c] * gg; s1 += wo[a][c] * (var * 3.0 * sr * sr * e * iell); }
        }
    } break;
    case WV_K_M52: {
      const double var = th[f.p[0]], ell = th[f.p[1]], iell = 1.0 / ell;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double sr = sqrt(q[a][c]) * iell, e = exp_nonpos(-kSqrt5 * sr);
          double gg = (1.0 + kSqrt5 * sr + (5.0 / 3.0) * sr * sr) * e;
          k[a][c] = var * gg;
          if (GRAD) {
            s0 += wo[a][c] * gg;
            s1 += wo[a][c] * (var * (5.0 / 3.0) * sr * sr * (1.0 + kSqrt5 * sr) * e * iell);
          }
        }
    } break;
    case WV_K_PERIODIC: {
      const double var = th[f.p[0]], ell = th[f.p[1]], per = th[f.p[2]];
      const double hi = -0.5 / (ell * ell), i3 = 1.0 / (ell * ell * ell);
      const double cp = kPi / (per * per * ell * ell);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double e = exp_nonpos(q[a][c] * hi), kk = var * e;
          k[a][c] = kk;
          if (GRAD) {
            s0 += wo[a][c] * e;
            s1 += wo[a][c] * (kk * q[a][c] * i3);
            s2 += wo[a][c] * (kk * cp * q2[a][c]);
          }
        }
    } break;
    case WV_K_LIN: {
      const double var = th[f.p[0]];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          k[a][c] = var * q[a][c];
          if (GRAD) s0 += wo[a][c] * q[a][c];
        }
    } break;
    case WV_K_POLY: {
      const double var = th[f.p[0]], off = th[f.p[1]];
      const int deg = f.degree;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double base = var * q[a][c] + off;
          double pm1 = ipow(base, deg - 1);
          k[a][c] = pm1 * base;
          if (GRAD) { double dk = deg * pm1; s0 += wo[a][c] * dk * q[a][c]; s1 += wo[a][c] * dk; }
        }
    } break;
    case WV_K_CONST: {
      const double var = th[f.p[0]];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          k[a][c] = var;
          if (GRAD) s0 += wo[a][c];
        }
    } break;
    case WV_K_CAT: {
      const double var = th[f.p[0]];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          k[a][c] = var * q[a][c];
          if (GRAD) s0 += wo[a][c] * q[a][c];
        }
    } break;
    default:  // WV_K_EMPTY
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) k[a][c] = 0.0;
  }
  if (GRAD) { s[0] = s0; s[1] = s1; s[2] = s2; }
}

// lower-triangular tile index -> (ti, tj), ti >= tj
__device__ inline void tri_decode(int t, int& ti, int& tj) {
  int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  while (i * (i + 1) / 2 > t) --i;
  ti = i;
  tj = t - i * (i + 1) / 2;
}

template <typename R>
__global__ __launch_bounds__(256, 3) void gram_kernel(const DevSpec* __restrict__ specs,
                                                   const int32_t* __restrict__ slot_spec,
                                                   const double* __restrict__ X,
                                                   const double* __restrict__ theta_all,
                                                   R* __restrict__ Kall, int64_t N,
                                                   int64_t Npad, int D, int slot0, int write_upper,
                                                   uint64_t term_mask, double diag_override) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = slot0 + blockIdx.y;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const int16_t* slots = sp->fslots;
  const double* th = theta_all + (int64_t)b * sp->n_params;
  R* K = Kall + (int64_t)b * Npad * Npad;
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  const int tid = threadIdx.x;
  double* fi = smem;
  double* fj = smem + sp->n_feat * GT;
  stage_features(fi, sp, X, th, ti * GT, N, D, tid);
  stage_features(fj, sp, X, th, tj * GT, N, D, tid);
  __syncthreads();

  TileCtx t{fi, fj, tid >> 4, tid & 15};
  double sum[4][4], k[4][4], prod[4][4], dummy_w[4][4], dummy_s[3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum[a][c] = 0.0;
  for (int term = 0; term < sp->n_terms; ++term) {
    if (!((term_mask >> term) & 1ull)) continue;  // leave-component-out Gram matrices
    const int f0 = sp->term_start[term], f1 = sp->term_start[term + 1];
    for (int f = f0; f < f1; ++f) {
      eval_factor<false>(sp->factors[f], slots, th, t, k, dummy_w, dummy_s);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) prod[a][c] = (f == f0) ? k[a][c] : prod[a][c] * k[a][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum[a][c] += prod[a][c];
  }
  const double noise = diag_override >= 0.0 ? diag_override : th[sp->noise_param];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = (int64_t)ti * GT + t.ty + 16 * a;
    double v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t j = (int64_t)tj * GT + 4 * t.tx + c;
      double x = sum[a][c];
      if (i == j) x += noise;
      if (i >= N || j >= N) x = (i == j) ? 1.0 : 0.0;  // identity padding
      v[c] = x;
    }
    store4(K + i * Npad + (int64_t)tj * GT + 4 * t.tx, v);
    if (write_upper && ti != tj) {
#pragma unroll
      for (int c = 0; c < 4; ++c) K[((int64_t)tj * GT + 4 * t.tx + c) * Npad + i] = (R)v[c];
    }
  }
}

// Rectangular cross-Gram K(Xrow, Xcol) for prediction at new inputs: rows = Mrow new points,
// columns = the N training points, no noise, padded entries zero.  One 64 x 64 tile per workgroup.
__global__ __launch_bounds__(256) void gram_rect_kernel(const DevSpec* __restrict__ sp,
                                                        const double* __restrict__ Xrow,
                                                        const double* __restrict__ Xcol,
                                                        const double* __restrict__ th,
                                                        double* __restrict__ out, int64_t Mrow,
                                                        int64_t N, int64_t ldo, int D, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int16_t* slots = sp->fslots;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int tid = threadIdx.x;
  double* fi = smem;
  double* fj = smem + sp->n_feat * GT;
  stage_features(fi, sp, Xrow, th, ti * GT, Mrow, D, tid);
  stage_features(fj, sp, Xcol, th, tj * GT, N, D, tid);
  __syncthreads();
  TileCtx t{fi, fj, tid >> 4, tid & 15};
  double sum[4][4], k[4][4], prod[4][4], dummy_w[4][4], dummy_s[3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum[a][c] = 0.0;
  for (int term = 0; term < sp->n_terms; ++term) {
    const int f0 = sp->term_start[term], f1 = sp->term_start[term + 1];
    for (int f = f0; f < f1; ++f) {
      eval_factor<false>(sp->factors[f], slots, th, t, k, dummy_w, dummy_s);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) prod[a][c] = (f == f0) ? k[a][c] : prod[a][c] * k[a][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum[a][c] += prod[a][c];
  }
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
  size_t sh = (size_t)2 * p.hspec.n_feat * GT * sizeof(double);  // (prediction: structure 0)
  hipLaunchKernelGGL(gram_rect_kernel, grid, blk, sh, st, p.dspec, dXnew, p.dX, dtheta_row, out, Mrow, p.N,
                     p.Npad, p.D, tiles_n);
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int TRACE_TPW = 4;  // tiles per workgroup

// part[b][wg][p] = sum over this workgroup's tiles of  w_ij W_ij dK_ij/dtheta_p
// (w = 1 below the diagonal, 1/2 on it, 0 above; noise_param gets 1/2 W_ii).
template <typename R>
__global__ __launch_bounds__(256) void trace_kernel(const DevSpec* __restrict__ specs,
                                                    const int32_t* __restrict__ slot_spec,
                                                    const double* __restrict__ X,
                                                    const double* __restrict__ theta_all,
                                                    const R* __restrict__ Kinv_all,
                                                    const double* __restrict__ alpha_all,
                                                    double* __restrict__ part, int64_t N,
                                                    int64_t Npad, int D, int slot0, int ntiles,
                                                    int nwg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = slot0 + blockIdx.y;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const int16_t* slots = sp->fslots;
  const int np = sp->n_params;
  const double* th = theta_all + (int64_t)b * np;
  const R* Kinv = Kinv_all + (int64_t)b * Npad * Npad;
  const double* alpha = alpha_all + (int64_t)b * Npad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* fi = smem;
  double* fj = fi + sp->n_feat * GT;
  double* acc = fj + sp->n_feat * GT;  // [4 waves][np]
  for (int q = tid; q < 4 * np; q += 256) acc[q] = 0.0;
  TileCtx t{fi, fj, tid >> 4, tid & 15};

  for (int tt = 0; tt < TRACE_TPW; ++tt) {
    const int tile = blockIdx.x * TRACE_TPW + tt;
    if (tile >= ntiles) break;
    int ti, tj;
    tri_decode(tile, ti, tj);
    __syncthreads();
    stage_features(fi, sp, X, th, ti * GT, N, D, tid);
    stage_features(fj, sp, X, th, tj * GT, N, D, tid);
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
        if (i >= N || j >= N) m = 0.0;
        w[a][c] = m * (ai * aj[c] - kv[c]);
        if (i == j) snoise += w[a][c];
      }
    }
    if (ti == tj) {  // wave-uniform
      double r = wave_sum(snoise);
      if (lane == 0) acc[wave * np + sp->noise_param] += r;
    }

    for (int term = 0; term < sp->n_terms; ++term) {
      const int f0 = sp->term_start[term], f1 = sp->term_start[term + 1];
      for (int g = f0; g < f1; ++g) {
        const DevFactor fg = sp->factors[g];
        if (fg.trainable == 0) continue;
        double wo[4][4], k[4][4], s[3];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) wo[a][c] = w[a][c];
        for (int f = f0; f < f1; ++f) {
          if (f == g) continue;
          eval_factor<false>(sp->factors[f], slots, th, t, k, wo, s);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) wo[a][c] *= k[a][c];
        }
        eval_factor<true>(fg, slots, th, t, k, wo, s);
        double r0 = wave_sum(s[0]), r1 = wave_sum(s[1]), r2 = wave_sum(s[2]);
        if (lane == 0) {
          if (fg.trainable & 1) acc[wave * np + fg.p[0]] += r0;
          if (fg.trainable & 2) acc[wave * np + fg.p[1]] += r1;
          if (fg.trainable & 4) acc[wave * np + fg.p[2]] += r2;
        }
      }
    }
  }
  __syncthreads();
  double* out = part + ((int64_t)b * nwg + blockIdx.x) * np;
  for (int p = tid; p < np; p += 256) out[p] = acc[p] + acc[np + p] + acc[2 * np + p] + acc[3 * np + p];
}

void launch_gram_masked(Problem& p, hipStream_t st, int slot0, int nb, uint64_t term_mask, double diag) {
  dim3 grid(p.ntiles(), nb), blk(256);
  size_t sh = (size_t)2 * p.max_feat * GT * sizeof(double);
  if (p.f32)
    hipLaunchKernelGGL(gram_kernel<float>, grid, blk, sh, st, p.dspec, p.dslotspec, p.dX, p.dtheta, p.Kbuf32, p.N, p.Npad, p.D,
                       slot0, 0, term_mask, diag);
  else
    hipLaunchKernelGGL(gram_kernel<double>, grid, blk, sh, st, p.dspec, p.dslotspec, p.dX, p.dtheta, p.Kbuf, p.N, p.Npad, p.D,
                       slot0, 0, term_mask, diag);
}

void launch_gram(Problem& p, hipStream_t st, int slot0, int nb, int write_upper) {
  dim3 grid(p.ntiles(), nb), blk(256);
  size_t sh = (size_t)2 * p.max_feat * GT * sizeof(double);
  p.timer.begin(st, ST_GRAM);
  if (p.f32)
    hipLaunchKernelGGL(gram_kernel<float>, grid, blk, sh, st, p.dspec, p.dslotspec, p.dX, p.dtheta, p.Kbuf32, p.N, p.Npad,
                       p.D, slot0, write_upper, ~0ull, -1.0);
  else
    hipLaunchKernelGGL(gram_kernel<double>, grid, blk, sh, st, p.dspec, p.dslotspec, p.dX, p.dtheta, p.Kbuf, p.N, p.Npad,
                       p.D, slot0, write_upper, ~0ull, -1.0);
  p.timer.end(st);
  p.timer.work(ST_GRAM, 0.0, (double)nb * (p.f32 ? 4.0 : 8.0) * ((double)p.Npad * (p.Npad + GT) / 2.0 + (double)p.N * p.D));
}

int trace_nwg(const Problem& p) { return (p.ntiles() + TRACE_TPW - 1) / TRACE_TPW; }

void launch_trace(Problem& p, hipStream_t st, int slot0, int nb) {
  int nwg = trace_nwg(p);
  dim3 grid(nwg, nb), blk(256);
  size_t sh = ((size_t)2 * p.max_feat * GT + 4 * p.np_stride) * sizeof(double);
  p.timer.begin(st, ST_TRACE);
  if (p.f32)
    hipLaunchKernelGGL(trace_kernel<float>, grid, blk, sh, st, p.dspec, p.dslotspec, p.dX, p.dtheta, p.Kbuf32, p.dalpha,
                       p.dpart, p.N, p.Npad, p.D, slot0, p.ntiles(), nwg);
  else
    hipLaunchKernelGGL(trace_kernel<double>, grid, blk, sh, st, p.dspec, p.dslotspec, p.dX, p.dtheta, p.Kbuf, p.dalpha,
                       p.dpart, p.N, p.Npad, p.D, slot0, p.ntiles(), nwg);
  p.timer.end(st);
  p.timer.work(ST_TRACE, 0.0, (double)nb * (p.f32 ? 4.0 : 8.0) * ((double)p.Npad * (p.Npad + GT) / 2.0 + (double)p.N * p.D));
}

}  // namespace wv
