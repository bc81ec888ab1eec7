/* waveome_amd_debug.h - bring-up / test hooks of libwaveome_amd.so.
 * Not part of the drop-in boundary: these expose single kernels and intermediate
 * buffers so tests/ can localise a parity failure (MFMA GEMM variants, the factor,
 * its inverse) and bench.py can measure the f64 MFMA issue rate the roofline is
 * priced against (the local hardware guide lists no f64 matrix rate). */
#ifndef WAVEOME_AMD_DEBUG_H
#define WAVEOME_AMD_DEBUG_H
#include <stdint.h>

#include "waveome_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* C[m][n] = beta*C + alpha * sum_k A(m,k) B(n,k) on host matrices via the batched MFMA kernel.
 * a_kcontig: A is stored [M][K] (1) or [K][M] (0); b_kcontig: B stored [N][K] (1) or [K][N] (0).
 * bn = 128 or 64 selects the block-tile width. All dims multiples of 16 (K) / 2 (M, N). */
int32_t wv_dbg_gemm(wv_ctx* ctx, const double* A, const double* B, double* C, int32_t M, int32_t N,
                    int32_t K, int32_t a_kcontig, int32_t b_kcontig, int32_t bn, double alpha,
                    double beta, int32_t nbatch);

/* Time `iters` launches of the batched GEMM on device-resident random operands. */
int32_t wv_dbg_gemm_bench(wv_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t a_kcontig,
                          int32_t b_kcontig, int32_t bn, int32_t nbatch, int32_t iters,
                          double* tflops_out, double* ms_out);

/* Run the evaluation pipeline for ONE task up to a stage and copy a buffer back.
 * what: 0 = L (Cholesky factor, Npad x Npad, lower), 1 = M = L^-1 (lower),
 *       2 = K_y^-1 (lower, after the full value+grad pipeline), 3 = z (Npad), 4 = alpha (Npad). */
int32_t wv_dbg_stage(wv_problem* prob, int32_t col, const double* theta, int32_t what, double* out);
int64_t wv_dbg_npad(const wv_problem* prob);

/* Back-to-back v_mfma_f64_16x16x4_f64 issue-rate probe: every CU, `waves_per_simd` waves per
 * SIMD, `iters` x 16 independent MFMAs per wave.  Returns TFLOP/s and the elapsed ms. */
int32_t wv_dbg_mfma_f64_rate(wv_ctx* ctx, int32_t iters, int32_t waves_per_simd, double* tflops_out,
                             double* ms_out);
/* v_fma_f64 issue-rate probe. mode 0: VALU only; 1: MFMA + VALU interleaved in each wave;
 * 2: alternate workgroups run MFMA-only / VALU-only. Reports both rates in TFLOP/s. */
int32_t wv_dbg_valu_f64_rate(wv_ctx* ctx, int32_t iters, int32_t waves_per_simd, int32_t mode,
                             double* valu_tflops_out, double* mfma_tflops_out);
/* Lane-map probe of v_mfma_f64_4x4x4_4b_f64: table_out[pa*64+pb] = D lane set by A lane pa x B lane pb. */
int32_t wv_dbg_mfma4_probe(wv_ctx* ctx, int32_t* table_out);
/* float4-style streaming copy of `bytes` bytes, GB/s (read + write counted). */
int32_t wv_dbg_copy_rate(wv_ctx* ctx, int64_t bytes, int32_t iters, double* gbps_out);

#ifdef __cplusplus
}
#endif
#endif
