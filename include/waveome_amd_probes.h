/* waveome_amd_probes.h - rate / timing probes.  NOT in the release library: built by `make -C waveome_amd/csrc probes`
 * (-DWV_EXPERIMENTS) into tools/libwv_probes.so, which also carries the GEMM timing experiments (WV_GEMM_DBG,
 * WV_ROT_TRI, WV_GEMM_OCC, WV_GEMM_LATIN; wv_gemm.hip) - switches that can make results wrong and therefore have no
 * code path in libwaveome_amd.so.  tools/probe_*.py load it through WAVEOME_AMD_LIB.  Used to price the roofline:
 * the local hardware guide lists no f64 matrix rate, tools/probe_rates.py measures it. */
#ifndef WAVEOME_AMD_PROBES_H
#define WAVEOME_AMD_PROBES_H
#include <stdint.h>

#include "waveome_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Time `iters` launches of the batched GEMM on device-resident random operands. */
int32_t wv_dbg_gemm_bench(wv_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t a_kcontig,
                          int32_t b_kcontig, int32_t bn, int32_t nbatch, int32_t iters,
                          double* tflops_out, double* ms_out);

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

/* Shader-clock stamps (8 x uint64) of workgroup 0 / thread 0 of the last small-tile GEMM launch: [0] entry, [1] first
 * chunk in LDS, [2] end of the K loop, [3] end of the epilogue. */
int32_t wv_dbg_gemm_stamps(unsigned long long* out);

/* exp2_64 (wv_gram.hip: 2^(t / 64), t <= 0) and three cheaper forms of it (variant 1: 3-address v_fma_f64 with SGPR
 * coefficients; 2: + rint / convert by the 1.5 * 2^52 shift; 3: + ldexp as an add into the exponent field): rate in
 * 1e9 evaluations per second with `waves_per_simd` waves per SIMD on every CU, arguments t0 - k dt; and the number of
 * arguments of t_host[n] on which a variant's bits differ from variant 0's (first_bad3_out: t, shipped, variant).
 * tools/probe_exp_rate.py, profiles/r04_exp_rate.txt. */
int32_t wv_dbg_exp_rate(int32_t variant, int32_t iters, int32_t waves_per_simd, double t0, double dt,
                        double* gexp_per_s_out);
int32_t wv_dbg_exp_check(int32_t variant, const double* t_host, int64_t n, int64_t* n_bad_out, double* first_bad3_out);

#ifdef __cplusplus
}
#endif
#endif
