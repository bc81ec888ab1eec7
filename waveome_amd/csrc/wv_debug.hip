// Test hooks of the release library (include/waveome_amd_debug.h): single kernels and intermediate buffers, so that
// tests/ can localise a parity failure.  They run the production kernels and return correct results.  The rate /
// timing probes live in wv_probes.hip and are NOT part of the release library (`make probes`).
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/waveome_amd_debug.h"
#include "wv_device.hpp"

using namespace wv;

struct wv_ctx { Ctx c; };
struct wv_problem { Problem p; };

// host double matrices -> device matrices of storage type R -> batched MFMA GEMM -> back
template <typename R>
static int32_t dbg_gemm_impl(wv_ctx* ctx, const double* A, const double* B, double* C, int32_t M, int32_t N,
                             int32_t K, int32_t a_kc, int32_t b_kc, int32_t bn, double alpha, double beta,
                             int32_t nbatch) {
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  const size_t na = (size_t)M * K * nbatch, nbb = (size_t)N * K * nbatch, nc = (size_t)M * N * nbatch;
  std::vector<R> hA(na), hB(nbb), hC(nc);
  for (size_t q = 0; q < na; ++q) hA[q] = (R)A[q];
  for (size_t q = 0; q < nbb; ++q) hB[q] = (R)B[q];
  for (size_t q = 0; q < nc; ++q) hC[q] = (R)C[q];
  R *dA, *dB, *dC;
  WV_HIP_TRY(hipMalloc(&dA, sizeof(R) * na));
  WV_HIP_TRY(hipMalloc(&dB, sizeof(R) * nbb));
  WV_HIP_TRY(hipMalloc(&dC, sizeof(R) * nc));
  WV_HIP_TRY(hipMemcpy(dA, hA.data(), sizeof(R) * na, hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemcpy(dB, hB.data(), sizeof(R) * nbb, hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemcpy(dC, hC.data(), sizeof(R) * nc, hipMemcpyHostToDevice));
  GemmArgsT<R> g{};
  g.A = dA; g.B = dB; g.C = dC;
  g.sA = (int64_t)M * K; g.sB = (int64_t)N * K; g.sC = (int64_t)M * N;
  g.lda = a_kc ? K : M; g.ldb = b_kc ? K : N; g.ldc = N;
  g.M = M; g.N = N; g.kb = 0; g.ke = K; g.alpha = alpha; g.beta = beta; g.nbatch = nbatch;
  launch_gemm(nullptr, g, a_kc != 0, b_kc != 0, bn);
  if (take_launch_failure()) return WV_E_UNSUPPORTED;
  WV_HIP_TRY(hipGetLastError());
  WV_HIP_TRY(hipDeviceSynchronize());
  WV_HIP_TRY(hipMemcpy(hC.data(), dC, sizeof(R) * nc, hipMemcpyDeviceToHost));
  for (size_t q = 0; q < nc; ++q) C[q] = (double)hC[q];
  hipFree(dA); hipFree(dB); hipFree(dC);
  return WV_OK;
}

extern "C" {

// Both GEMM probes run in the storage type of the context (float64, or float32 on a dtype-1 context,
// where B must be K-strided and the tile width is fixed at 128).
int32_t wv_dbg_gemm(wv_ctx* ctx, const double* A, const double* B, double* C, int32_t M, int32_t N,
                    int32_t K, int32_t a_kc, int32_t b_kc, int32_t bn, double alpha, double beta,
                    int32_t nbatch) {
  if (!ctx || !A || !B || !C || M < 2 || N < 2 || K < 16 || (K % 16) || (M % 2) || (N % 2) ||
      (bn != 32 && bn != 64 && bn != 128) || nbatch < 1) {  // (32: launch_gemm's code for the small 64 x 64 tile)
    set_error("wv_dbg_gemm: bad argument");
    return WV_E_BADARG;
  }
  if (ctx->c.dtype == 1) {
    if (b_kc || (K % 64) || (M % 64) || (N % 64)) {
      set_error("wv_dbg_gemm (float32): B must be K-strided and M, N, K multiples of 64");
      return WV_E_BADARG;
    }
    return dbg_gemm_impl<float>(ctx, A, B, C, M, N, K, a_kc, b_kc, bn, alpha, beta, nbatch);
  }
  return dbg_gemm_impl<double>(ctx, A, B, C, M, N, K, a_kc, b_kc, bn, alpha, beta, nbatch);
}

int64_t wv_dbg_npad(const wv_problem* h) { return h ? h->p.Npad : 0; }

int32_t wv_dbg_stage(wv_problem* h, int32_t col, const double* theta, int32_t what, double* out) {
  if (!h || !theta || !out || col < 0 || col >= h->p.P || what < 0 || what > 4) {
    set_error("wv_dbg_stage: bad argument");
    return WV_E_BADARG;
  }
  Problem& p = h->p;
  if (p.dmask) {
    set_error("wv_dbg_stage: the problem has missing outputs (nan_is_missing)");
    return WV_E_UNSUPPORTED;
  }
  WV_HIP_TRY(hipSetDevice(p.ctx->device));
  if (p.open_queue) {
    set_error("wv_dbg_stage: the problem has an open work queue");
    return WV_E_BADARG;
  }
  (void)take_launch_failure();
  Group& g = p.groups[0];
  hipStream_t st = g.stream;
  const int np = p.spec.n_params();
  std::memcpy(g.h_theta, theta, sizeof(double) * np);
  g.h_col[0] = col;
  WV_HIP_TRY(hipMemcpyAsync(p.dtheta + (int64_t)g.slot0 * np, g.h_theta, sizeof(double) * np, hipMemcpyHostToDevice, st));
  WV_HIP_TRY(hipMemcpyAsync(p.dcol + g.slot0, g.h_col, sizeof(int32_t), hipMemcpyHostToDevice, st));
  launch_resid(p, st, g.slot0, 1);
  if (p.f32 && what == 0) {
    set_error("wv_dbg_stage: the stand-alone Cholesky schedule (what = 0) is float64 only");
    return WV_E_UNSUPPORTED;
  }
  launch_gram(p, st, g.slot0, 1, 0);
  if (!p.f32) cholesky(p, st, g.slot0, 1);
  const int64_t mat = p.Npad * p.Npad;
  const double* src = nullptr;
  const float* src32 = nullptr;  // float32 problems: matrices are fetched as stored and widened
  size_t bytes = sizeof(double) * mat;
  if (what == 0) src = p.Kbuf + (int64_t)g.slot0 * mat;
  if (what >= 1) {
    // what == 1 exercises the production schedule (recursive factor + inverse) from a fresh Gram
    launch_gram(p, st, g.slot0, 1, 0);
    factor_inverse(p, st, g.slot0, 1);
    if (p.f32) src32 = p.Mbuf32 + (int64_t)g.slot0 * mat;
    else src = p.Mbuf + (int64_t)g.slot0 * mat;
  }
  if (what >= 2) {
    launch_solve(p, st, g.slot0, 1, true);
    lauum(p, st, g.slot0, 1);
    if (p.f32) src32 = p.Kbuf32 + (int64_t)g.slot0 * mat;
    else src = p.Kbuf + (int64_t)g.slot0 * mat;
  }
  if (what >= 3) src32 = nullptr;
  if (what == 3) { src = p.dz + (int64_t)g.slot0 * p.Npad; bytes = sizeof(double) * p.Npad; }
  if (what == 4) { src = p.dalpha + (int64_t)g.slot0 * p.Npad; bytes = sizeof(double) * p.Npad; }
  WV_HIP_TRY(hipGetLastError());
  WV_HIP_TRY(hipStreamSynchronize(st));
  if (take_launch_failure()) return WV_E_UNSUPPORTED;
  if (src32) {
    std::vector<float> tmp((size_t)mat);
    WV_HIP_TRY(hipMemcpy(tmp.data(), src32, sizeof(float) * mat, hipMemcpyDeviceToHost));
    for (int64_t q = 0; q < mat; ++q) out[q] = (double)tmp[q];
    return WV_OK;
  }
  WV_HIP_TRY(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  return WV_OK;
}
}
