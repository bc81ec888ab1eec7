/* waveome_amd_debug.h - test hooks of libwaveome_amd.so.
 * Not part of the drop-in boundary: these expose single production kernels and intermediate
 * buffers so tests/ can localise a parity failure (MFMA GEMM variants, the factor, its inverse).
 * They return correct results.  Rate / timing probes are declared in waveome_amd_probes.h and exist
 * only in the `make probes` build (tools/libwv_probes.so), not in the release library. */
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

/* Run the evaluation pipeline for ONE task up to a stage and copy a buffer back.
 * what: 0 = L (Cholesky factor, Npad x Npad, lower), 1 = M = L^-1 (lower),
 *       2 = K_y^-1 (lower, after the full value+grad pipeline), 3 = z (Npad), 4 = alpha (Npad). */
int32_t wv_dbg_stage(wv_problem* prob, int32_t col, const double* theta, int32_t what, double* out);
int64_t wv_dbg_npad(const wv_problem* prob);

#ifdef __cplusplus
}
#endif
#endif
