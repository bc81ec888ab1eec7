// Rate / timing probes (include/waveome_amd_probes.h): MFMA and VALU issue rates, the 4x4x4 lane-map probe, a
// streaming copy, timed GEMM launches.  Built only by `make probes` (-DWV_EXPERIMENTS, tools/libwv_probes.so);
// the release library does not contain them.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/waveome_amd_probes.h"
#include "wv_device.hpp"

namespace wv {
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_rate_kernel(double* out, int iters) {
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// mode 0: pure v_fma_f64 (16 independent chains / lane); mode 1: MFMA f64 and v_fma_f64 interleaved
// in one wave; mode 2: even waves MFMA, odd waves VALU (separate waves on a SIMD).
__global__ __launch_bounds__(256) void valu_rate_kernel(double* out, int iters, int mode) {
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-9;
  const double a = 1.0000001, b = 1e-9;
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  const bool do_valu = (mode == 0) || (mode == 1) || (mode == 4) || (mode == 2 && ((blockIdx.x & 1) == 1));
  const bool do_mfma = (mode == 1) || (mode == 4) || (mode == 2 && ((blockIdx.x & 1) == 0));
  if (mode == 3) {
    double c4[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c4[i] = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) c4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4[i], 0, 0, 0);
    }
    double s4 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s4 += c4[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s4;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (do_valu) {
      if (mode == 4) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], a, b);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], a, b);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Lane-map probe for v_mfma_f64_4x4x4_4b_f64: for every (pa, pb) run D = mfma(e_pa, e_pb, 0) and
// record which D lane (if any) becomes 1.  table[pa*64 + pb] = lane or -1.
__global__ void mfma4_probe_kernel(int* table) {
  const int lane = threadIdx.x;
  for (int pa = 0; pa < 64; ++pa)
    for (int pb = 0; pb < 64; ++pb) {
      double a = lane == pa ? 1.0 : 0.0, b = lane == pb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) table[pa * 64 + pb] = m ? (__ffsll((long long)m) - 1) + 64 * (__popcll(m) - 1) : -1;
    }
}

__global__ void copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
}  // namespace wv

using namespace wv;

struct wv_ctx { Ctx c; };

template <typename R>
static int32_t dbg_gemm_bench_impl(wv_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t a_kc, int32_t b_kc,
                                   int32_t bn, int32_t mode, int32_t nbatch, int32_t iters, double* tflops_out,
                                   double* ms_out) {
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  R *dA, *dB, *dC;
  size_t na = (size_t)M * K * nbatch, nbb = (size_t)N * K * nbatch, nc = (size_t)M * N * nbatch;
  WV_HIP_TRY(hipMalloc(&dA, sizeof(R) * na));
  WV_HIP_TRY(hipMalloc(&dB, sizeof(R) * nbb));
  WV_HIP_TRY(hipMalloc(&dC, sizeof(R) * nc));
  {
    std::vector<R> h(std::max(na, nbb));
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (R)((double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    WV_HIP_TRY(hipMemcpy(dA, h.data(), sizeof(R) * na, hipMemcpyHostToDevice));
    WV_HIP_TRY(hipMemcpy(dB, h.data() + 7, sizeof(R) * (nbb - 7), hipMemcpyHostToDevice));
  }
  WV_HIP_TRY(hipMemset(dC, 0, sizeof(R) * nc));
  GemmArgsT<R> g{};
  g.A = dA; g.B = dB; g.C = dC;
  g.sA = (int64_t)M * K; g.sB = (int64_t)N * K; g.sC = (int64_t)M * N;
  g.lda = a_kc ? K : M; g.ldb = b_kc ? K : N; g.ldc = N;
  g.M = M; g.N = N; g.kb = 0; g.ke = K; g.alpha = 1.0; g.beta = 0.0; g.nbatch = nbatch;
  // K_y^-1 = M'M and the trailing update as the pipeline launches them: a matrix times its own transpose (B = A)
  if (mode == 1) { g.tri = 1; g.klo_mode = 1; g.tag = 1; g.B = g.A; g.sB = g.sA; g.ldb = g.lda; }
  if (mode == 2) { g.tri = 1; g.beta = 1.0; g.alpha = -1e-3; g.tag = 2; g.B = g.A; g.sB = g.sA; g.ldb = g.lda; }
  if (mode == 3) { g.khi_mode = 2; }
  if (mode == 4) { g.tri = 1; }
  if (mode == 6) { g.khi_mode = 1; }
  if (mode == 5) { g.beta = 1.0; g.alpha = -1e-3; }
  hipEvent_t a, b;
  WV_HIP_TRY(hipEventCreate(&a));
  WV_HIP_TRY(hipEventCreate(&b));
  launch_gemm(nullptr, g, a_kc != 0, b_kc != 0, bn);
  WV_HIP_TRY(hipEventRecord(a, nullptr));
  for (int i = 0; i < iters; ++i) launch_gemm(nullptr, g, a_kc != 0, b_kc != 0, bn);
  WV_HIP_TRY(hipEventRecord(b, nullptr));
  WV_HIP_TRY(hipEventSynchronize(b));
  WV_HIP_TRY(hipGetLastError());
  float ms = 0;
  WV_HIP_TRY(hipEventElapsedTime(&ms, a, b));
  double fl = 2.0 * M * N * (double)K;
  if (mode == 1) fl = (double)M * M * M / 3.0;
  if (mode == 2 || mode == 4) fl = (double)M * M * K;
  if (mode == 3 || mode == 6) fl = (double)M * N * K;
  if (tflops_out) *tflops_out = fl * nbatch * iters / (ms * 1e-3) / 1e12;
  if (ms_out) *ms_out = ms / iters;
  hipEventDestroy(a); hipEventDestroy(b); hipFree(dA); hipFree(dB); hipFree(dC);
  return WV_OK;
}

namespace wv { int read_gemm_stamps(unsigned long long* out); }

// One launch of the float64 GEMM with the structure switches of the pipeline's products (lower-only tiles, trimmed
// contraction ranges, transposed outputs), host matrices in and out: tools/check_gemm_asm.py compares the two K loops
// bit for bit on every such form.  A: [nb][K][M] (K-strided) or [nb][M][K]; B: [nb][K][N]; C: [nb][M][N]; CT: [nb][N][M].
extern "C" int32_t wv_dbg_gemm_ex(wv_ctx* ctx, const double* A, const double* B, double* C, double* CT, int32_t M, int32_t N,
                                  int32_t K, int32_t a_kc, double alpha, double beta, int32_t nbatch, int32_t tri,
                                  int32_t klo_mode, int32_t khi_mode, int32_t tout, int32_t tag, int32_t same_ab) {
  if (!ctx || !A || !B || !C || !CT || ctx->c.dtype == 1) return WV_E_BADARG;
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  const size_t na = (size_t)M * K * nbatch, nbb = (size_t)N * K * nbatch, nc = (size_t)M * N * nbatch;
  double *dA, *dB, *dC, *dCT;
  WV_HIP_TRY(hipMalloc(&dA, 8 * na));
  WV_HIP_TRY(hipMalloc(&dB, 8 * nbb));
  WV_HIP_TRY(hipMalloc(&dC, 8 * nc));
  WV_HIP_TRY(hipMalloc(&dCT, 8 * nc));
  WV_HIP_TRY(hipMemcpy(dA, A, 8 * na, hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemcpy(dB, B, 8 * nbb, hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemcpy(dC, C, 8 * nc, hipMemcpyHostToDevice));
  WV_HIP_TRY(hipMemcpy(dCT, CT, 8 * nc, hipMemcpyHostToDevice));
  GemmArgs g{};
  g.A = dA; g.B = same_ab ? dA : dB; g.C = dC; g.CT = dCT;
  g.sA = (int64_t)M * K; g.sB = same_ab ? g.sA : (int64_t)N * K; g.sC = (int64_t)M * N; g.sCT = (int64_t)M * N;
  g.lda = a_kc ? K : M; g.ldb = same_ab ? g.lda : N; g.ldc = N; g.ldct = M;
  g.M = M; g.N = N; g.kb = 0; g.ke = K; g.alpha = alpha; g.beta = beta; g.nbatch = nbatch;
  g.tri = tri; g.klo_mode = klo_mode; g.khi_mode = khi_mode; g.tout = tout; g.tag = tag;
  launch_gemm(nullptr, g, a_kc != 0, false, 128);
  if (take_launch_failure()) return WV_E_UNSUPPORTED;
  WV_HIP_TRY(hipGetLastError());
  WV_HIP_TRY(hipDeviceSynchronize());
  WV_HIP_TRY(hipMemcpy(C, dC, 8 * nc, hipMemcpyDeviceToHost));
  WV_HIP_TRY(hipMemcpy(CT, dCT, 8 * nc, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dCT);
  return WV_OK;
}

extern "C" {

/* shader-clock stamps of workgroup 0 of the last small-tile GEMM launch: entry, first chunk in LDS, end of the K loop, end */
int32_t wv_dbg_gemm_stamps(unsigned long long* out) { return wv::read_gemm_stamps(out); }

int32_t wv_dbg_gemm_bench(wv_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t a_kc, int32_t b_kc,
                          int32_t bn, int32_t nbatch, int32_t iters, double* tflops_out, double* ms_out) {
  // bn encodes a structure mode in its upper bits: bn + 1000*mode; mode 1 = LAUUM-like (lower tiles,
  // contraction from the tile row), 2 = SYRK-like (lower tiles, beta = 1), 3 = trimmed k < n0 + BN
  const int mode = bn / 1000;
  bn = bn % 1000;
  if (!ctx || M < 2 || N < 2 || K < 16 || (K % 16) || (M % 2) || (N % 2) || nbatch < 1 || iters < 1) {
    set_error("wv_dbg_gemm_bench: bad argument");
    return WV_E_BADARG;
  }
  if (ctx->c.dtype == 1) {
    if (b_kc || (K % 64) || (M % 64) || (N % 64)) {
      set_error("wv_dbg_gemm_bench (float32): B must be K-strided and M, N, K multiples of 64");
      return WV_E_BADARG;
    }
    return dbg_gemm_bench_impl<float>(ctx, M, N, K, a_kc, b_kc, bn, mode, nbatch, iters, tflops_out, ms_out);
  }
  return dbg_gemm_bench_impl<double>(ctx, M, N, K, a_kc, b_kc, bn, mode, nbatch, iters, tflops_out, ms_out);
}

int32_t wv_dbg_mfma_f64_rate(wv_ctx* ctx, int32_t iters, int32_t wps, double* tflops_out, double* ms_out) {
  if (!ctx || iters < 1 || wps < 1 || wps > 8) {
    set_error("wv_dbg_mfma_f64_rate: bad argument");
    return WV_E_BADARG;
  }
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  const int cus = ctx->c.prop.multiProcessorCount;
  const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
  double* d;
  WV_HIP_TRY(hipMalloc(&d, sizeof(double) * blocks * 256));
  hipEvent_t a, b;
  WV_HIP_TRY(hipEventCreate(&a));
  WV_HIP_TRY(hipEventCreate(&b));
  hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters / 8 + 1);  // warm
  WV_HIP_TRY(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters);
  WV_HIP_TRY(hipEventRecord(b, nullptr));
  WV_HIP_TRY(hipEventSynchronize(b));
  float ms = 0;
  WV_HIP_TRY(hipEventElapsedTime(&ms, a, b));
  double flops = (double)blocks * 4.0 * iters * 16.0 * 2.0 * 16 * 16 * 4;
  if (tflops_out) *tflops_out = flops / (ms * 1e-3) / 1e12;
  if (ms_out) *ms_out = ms;
  hipEventDestroy(a); hipEventDestroy(b); hipFree(d);
  return WV_OK;
}

int32_t wv_dbg_valu_f64_rate(wv_ctx* ctx, int32_t iters, int32_t waves_per_simd, int32_t mode,
                              double* valu_tflops_out, double* mfma_tflops_out) {
  if (!ctx || iters < 1 || waves_per_simd < 1 || waves_per_simd > 8 || mode < 0 || mode > 4) {
    set_error("wv_dbg_valu_f64_rate: bad argument");
    return WV_E_BADARG;
  }
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  const int blocks = ctx->c.prop.multiProcessorCount * waves_per_simd;
  double* d;
  WV_HIP_TRY(hipMalloc(&d, sizeof(double) * blocks * 256));
  hipEvent_t a, b;
  WV_HIP_TRY(hipEventCreate(&a));
  WV_HIP_TRY(hipEventCreate(&b));
  hipLaunchKernelGGL(valu_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters / 8 + 1, mode);
  WV_HIP_TRY(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL(valu_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters, mode);
  WV_HIP_TRY(hipEventRecord(b, nullptr));
  WV_HIP_TRY(hipEventSynchronize(b));
  float ms = 0;
  WV_HIP_TRY(hipEventElapsedTime(&ms, a, b));
  double frac_valu = mode == 2 ? 0.5 : 1.0, frac_mfma = mode == 0 ? 0.0 : (mode == 2 ? 0.5 : 1.0);
  double vf = (double)blocks * frac_valu * 256.0 * iters * (mode == 4 ? 128.0 : 64.0) * 2.0;
  double mf = (double)blocks * frac_mfma * 4.0 * iters * 8.0 * 2.0 * 16 * 16 * 4;
  if (mode == 3) { vf = 0; mf = (double)blocks * 4.0 * iters * 16.0 * 512.0; }
  if (valu_tflops_out) *valu_tflops_out = vf / (ms * 1e-3) / 1e12;
  if (mfma_tflops_out) *mfma_tflops_out = mf / (ms * 1e-3) / 1e12;
  hipEventDestroy(a); hipEventDestroy(b); hipFree(d);
  return WV_OK;
}

int32_t wv_dbg_mfma4_probe(wv_ctx* ctx, int32_t* table_out) {
  if (!ctx || !table_out) return WV_E_BADARG;
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  int* d;
  WV_HIP_TRY(hipMalloc(&d, sizeof(int) * 4096));
  hipLaunchKernelGGL(mfma4_probe_kernel, dim3(1), dim3(64), 0, nullptr, d);
  WV_HIP_TRY(hipDeviceSynchronize());
  WV_HIP_TRY(hipMemcpy(table_out, d, sizeof(int) * 4096, hipMemcpyDeviceToHost));
  hipFree(d);
  return WV_OK;
}

int32_t wv_dbg_copy_rate(wv_ctx* ctx, int64_t bytes, int32_t iters, double* gbps_out) {
  if (!ctx || bytes < 1024 || iters < 1) {
    set_error("wv_dbg_copy_rate: bad argument");
    return WV_E_BADARG;
  }
  WV_HIP_TRY(hipSetDevice(ctx->c.device));
  double2 *s, *d;
  WV_HIP_TRY(hipMalloc(&s, bytes));
  WV_HIP_TRY(hipMalloc(&d, bytes));
  WV_HIP_TRY(hipMemset(s, 1, bytes));
  hipEvent_t a, b;
  WV_HIP_TRY(hipEventCreate(&a));
  WV_HIP_TRY(hipEventCreate(&b));
  int64_t n = bytes / 16;
  hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, nullptr, s, d, n);
  WV_HIP_TRY(hipEventRecord(a, nullptr));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, nullptr, s, d, n);
  WV_HIP_TRY(hipEventRecord(b, nullptr));
  WV_HIP_TRY(hipEventSynchronize(b));
  float ms = 0;
  WV_HIP_TRY(hipEventElapsedTime(&ms, a, b));
  if (gbps_out) *gbps_out = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
  hipEventDestroy(a); hipEventDestroy(b); hipFree(s); hipFree(d);
  return WV_OK;
}
}
