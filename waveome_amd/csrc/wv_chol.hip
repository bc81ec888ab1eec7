// Factorisation / inversion schedules and the small kernels around the MFMA GEMMs.
//
// Per evaluation and batch slot (reference: tf.linalg.cholesky + triangular_solve forward and
// reverse mode under gpflow's log marginal likelihood, waveome/model_types_DEPR.py:49-56):
//   factor_inverse     value + gradient: recursive K_y = L L' together with M = L^-1 (fi_node); leaves are
//                      potrf_diag (64 x 64) and potrf128 (a whole 128-node in one kernel)
//   factor_value_only  value only: the same recursion without the inverse of the whole matrix, forward
//                      substitution z = L^-1 r fused in (fv_node), 0.38 N^3 instead of 0.67 N^3
//   lauum              K_y^-1 = M' M, one launch, lower tiles, contraction over rows >= tile row
//   solve              z = M r, alpha = M' z (HBM-bound matvecs); reduce: LML and gradient assembly
//   cholesky / trtri   the first, right-looking schedule (outer 256, inner 64 panels); still used by the
//                      stand-alone factor of wv_whitened_means and by wv_dbg_stage(what = 0)
// Storage type R is double, or float on a float32 context; vectors and reductions are always double.
#include <type_traits>

#include "wv_device.hpp"

namespace wv {

constexpr int OUTER = 256;  // outer Cholesky block
// LDS row stride of the 64 x 64 blocks: 66 = 2 mod 32 doubles keeps an MFMA fragment read of a row-major operand
// (16 rows x 2 k per half-wave -> 2 row + k) free of bank conflicts; the column-wise reads are 2-way
constexpr int DS = NB + 2;

// 4 consecutive elements in their own type
__device__ inline void ld4(const double* __restrict__ p, double (&v)[4]) { load4(p, v); }
__device__ inline void st4(double* __restrict__ p, const double (&v)[4]) { store4(p, v); }
__device__ inline void ld4(const float* __restrict__ p, float (&v)[4]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ inline void st4(float* __restrict__ p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ inline double rsq(double x) { return rsqrt(x); }
__device__ inline float rsq(float x) { return rsqrtf(x); }

// Scratch of the triangular inversion inside the 64 x 64 block (inv_level_mma): NB/2 * 32 entries.
constexpr int INV_SCRATCH = NB * NB / 4;
template <typename R>
__global__ void potrf_diag_kernel(R* __restrict__ Kall, R* __restrict__ Mall, R* __restrict__ MTall,
                                  double* __restrict__ logdet, int32_t* __restrict__ status, int64_t Npad, int nblk,
                                  int slot0, int j0);  // (defined below, after the helpers it shares with potrf128)

void launch_potrf_diag(Problem& p, hipStream_t st, int slot0, int nb, int j0) {
  p.timer.begin(st, ST_POTRF_DIAG);
  if (p.f32)
    hipLaunchKernelGGL(potrf_diag_kernel<float>, dim3(nb), dim3(256), 0, st, p.Kbuf32, p.Mbuf32, p.MTbuf32,
                       p.dlogdet, p.dstatus, p.Npad, p.nblk(), slot0, j0);
  else
    hipLaunchKernelGGL(potrf_diag_kernel<double>, dim3(nb), dim3(256), 0, st, p.Kbuf, p.Mbuf, p.MTbuf, p.dlogdet,
                       p.dstatus, p.Npad, p.nblk(), slot0, j0);
  p.timer.end(st);
}

// ---------------------------------------------------------------------------------------
// potrf128: the whole recursion node of size 128 in ONE kernel - factor + invert the 64 x 64 block at (o, o),
// S = A21 M11', A22 -= S S', T = S M11, factor + invert the Schur complement, M21 = -M22 T - instead of two
// leaf launches and four 64^3 GEMM launches.  Writes M and M' of the 128 block and the two log-det partials; K's own
// blocks are dead after this node and are not written back.
//
// With one fit in flight the sixteen leaves of an N = 2048 evaluation are a dependent chain, so this kernel is
// built for the latency of ONE workgroup (round 2; shader-clock stamps of the phases: tools/probe_potrf_stamps.py):
//   * factor64c (round 5): a row per lane in every wave, the columns dealt to the four waves in blocks of four; the
//     owner's chain pivot -> rsqrt -> scale -> next pivot stays in its registers (v_readlane), one barrier and one
//     LDS round trip per four columns; the 4 x 4 inverses are formed afterwards, sixteen threads at once
//     (rounds 2 - 4: factor64b, 16 x 16 threads x (4 x 4) registers, two barriers and three round trips per step);
//   * the 64^3 products and the two upper levels of the triangular inversion run on v_mfma_f64_4x4x4_4b out of LDS
//     (wave_mma; the lane maps are those of wv_gemm.hip) instead of fp64 VALU dot products.
// ---------------------------------------------------------------------------------------
#ifdef WV_STAMPS  // diagnostic build only (tools/probe_potrf_stamps.py): cycle stamps of workgroup 0's thread 0
__device__ unsigned long long wv_stamps[32];
#define WV_STAMP(i)                                                                       \
  do {                                                                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0) wv_stamps[i] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define WV_STAMP(i)
#endif

// acc[ia][jc][r] += sum over k in [kbeg, kend) of A(16 ia + 4 blk + e, k) * B(k, 16 jc + 4 ((blk + r0 + r) & 3) + e)
// for this lane's (k, blk, e) = (lane >> 4, (lane >> 2) & 3, lane & 3); kbeg, kend multiples of 4.
//   A(row, k) = AT ? A[k * lda + row] : A[row * lda + k]      B(k, col) = BT ? B[col * ldb + k] : B[k * ldb + col]
// Afterwards element (i, j) = (lane >> 4, lane & 3) of block blk of acc[ia][jc][r] is the product's entry
// (16 ia + 4 blk + i, 16 jc + 4 ((blk + r0 + r) & 3) + j): see mma_rc.
template <int NA, int NC, int NR, bool AT, bool BT, typename R>
__device__ inline void wave_mma(double (&acc)[NA][NC][NR], const R* __restrict__ A, int lda, const R* __restrict__ B,
                                int ldb, int kbeg, int kend, int r0, int lane) {
  const int fk = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  for (int k0 = kbeg; k0 < kend; k0 += 4) {
    const int k = k0 + fk;
    double af[NA];
#pragma unroll
    for (int ia = 0; ia < NA; ++ia) {
      const int row = 16 * ia + 4 * blk + e;
      af[ia] = (double)(AT ? A[k * lda + row] : A[row * lda + k]);
    }
#pragma unroll
    for (int jc = 0; jc < NC; ++jc)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int col = 16 * jc + 4 * ((blk + r0 + r) & 3) + e;
        const double bf = (double)(BT ? B[col * ldb + k] : B[k * ldb + col]);
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) acc[ia][jc][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[ia], bf, acc[ia][jc][r], 0, 0, 0);
      }
  }
}
// row / column (relative to the wave's tile) of this lane's element of acc[ia][jc][r]
__device__ inline int mma_row(int ia, int lane) { return 16 * ia + 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ inline int mma_col(int jc, int r, int r0, int lane) {
  return 16 * jc + 4 * ((((lane >> 2) & 3) + r0 + r) & 3) + (lane & 3);
}
template <int NA, int NC, int NR>
__device__ inline void mma_zero(double (&acc)[NA][NC][NR]) {
#pragma unroll
  for (int ia = 0; ia < NA; ++ia)
#pragma unroll
    for (int jc = 0; jc < NC; ++jc)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[ia][jc][r] = 0.0;
}
// C(row, col) = sign * acc, C row-major with leading dimension ldc, pre-offset to the wave's tile
template <int NA, int NC, int NR, typename R>
__device__ inline void mma_store(const double (&acc)[NA][NC][NR], R* __restrict__ C, int ldc, double sign, int r0, int lane) {
#pragma unroll
  for (int ia = 0; ia < NA; ++ia)
#pragma unroll
    for (int jc = 0; jc < NC; ++jc)
#pragma unroll
      for (int r = 0; r < NR; ++r) C[mma_row(ia, lane) * ldc + mma_col(jc, r, r0, lane)] = (R)(sign * acc[ia][jc][r]);
}

// ---------------------------------------------------------------------------------------
// factor64c: the 64 x 64 Cholesky factor by the four waves of the workgroup, a ROW per lane in every wave, the COLUMNS
// dealt to the waves in blocks of four (wave w owns column blocks w, w + 4, w + 8, w + 12: sixteen doubles per lane),
// every register index a compile-time constant.  Step jb: the owner factors its four columns in registers - pivot out of
// lane j by v_readlane, scale, the block's later columns updated with lane j + 1..'s entries by v_readlane again: the
// chain pivot -> rsqrt -> scale -> next pivot never leaves the wave - and publishes them in LDS (all sixteen panels keep
// their own place: no buffer is ever reused); one barrier; every wave then subtracts the panel from the blocks it owns
// (own-row values + uniform-address broadcast reads).  The NEXT owner updates only the block it is about to factor and
// catches up on its other blocks after its own chain.  factor64b's step went through two barriers and three LDS round
// trips with the 4 x 4 diagonal factor, the panel solve and the rank-4 update in sequence (2 500 cycles per four columns,
// tools/probe_potrf_stamps.py); a single wave holding whole rows (factor64w, tried first) is bound by its own issue rate
// (6 200 instructions, 35 k cycles: no gain).
//   Lb: in = the block (lower triangle used), out = L with the upper triangle zeroed;  rinv[64]: reciprocal pivots;
//   Pan: 4096 doubles of scratch ([column][row]).  Called by all 256 threads; returns "this wave saw a non-positive / NaN pivot".
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) double lds_f64;  // (LDS pointers as such: constant offsets fold into ds_* offset fields)
// 1 / sqrt(x) for finite x > 0: v_rsq_f64 and the one correction step of the library's rsqrt (same bits) without its
// fix-up of zero / infinite arguments - four instructions off the pivot chain (the caller has excluded those arguments)
__device__ inline double rsq_pos(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
__device__ inline double lane_bcast(double v, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// a[q][t] -= sum_c panel(P)[row][c] * panel(P)[4 B + t][c] for this wave's block q (B = 4 q + W)
template <int W, int P, int Q>
__device__ __forceinline__ void factor64c_apply(double (&a)[4][4], const double (&own)[4], const lds_f64* __restrict__ Pan) {
  constexpr int B = 4 * Q + W;
  double bc[4][4];  // (all sixteen broadcast values in flight before the first product: one LDS latency, not eight)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) bc[c][t] = Pan[(4 * P + c) * NB + 4 * B + t];
  __builtin_amdgcn_sched_barrier(0);  // (hipcc's scheduler would keep two reads in flight and wait on each pair)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) a[Q][t] -= own[c] * bc[c][t];
}
// panel P subtracted from this wave's blocks FROM..3 (those that lie behind block AFTER)
template <int W, int P, int AFTER, bool ONLY_NEXT>
__device__ __forceinline__ void factor64c_update(double (&a)[4][4], const lds_f64* __restrict__ Pan, int lane) {
  constexpr bool any = 4 * 3 + W > AFTER;
  if constexpr (any) {
    double own[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) own[c] = Pan[(4 * P + c) * NB + lane];
    if constexpr (ONLY_NEXT) {
      factor64c_apply<W, P, (AFTER + 1) / 4>(a, own, Pan);  // (AFTER + 1 is this wave's block)
    } else {
      if constexpr (4 * 0 + W > AFTER) factor64c_apply<W, P, 0>(a, own, Pan);
      if constexpr (4 * 1 + W > AFTER) factor64c_apply<W, P, 1>(a, own, Pan);
      if constexpr (4 * 2 + W > AFTER) factor64c_apply<W, P, 2>(a, own, Pan);
      if constexpr (4 * 3 + W > AFTER) factor64c_apply<W, P, 3>(a, own, Pan);
    }
  }
}
template <int W, int JB>
__device__ __forceinline__ void factor64c_step(double (&a)[4][4], lds_f64* __restrict__ Pan, lds_f64* __restrict__ rinv,
                                               int lane, bool& anybad) {
  constexpr int Q = JB / 4;
  if constexpr (JB % 4 == W) {  // the owner: four columns in registers
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int J = 4 * JB + c;
      const double piv = lane_bcast(a[Q][c], J);
      const bool ok = piv > 0.0 && piv <= 1.7976931348623157e308;  // false for NaN; an overflowed pivot counts as bad too
      // (branch-free, and the test runs beside the rsqrt instead of in front of it: a bad pivot is replaced by 1 - the
      //  pipeline keeps running on finite numbers - by selecting this column's results at the end)
      const double r0 = rsq_pos(piv);
      const double alt = (lane == J) ? 1.0 : a[Q][c];
      const double lj = ok ? a[Q][c] * r0 : alt;
      const double ri = ok ? r0 : 1.0;
      anybad |= !ok;
      a[Q][c] = lj;
      rinv[J] = ri;  // (the same value from every lane)
#pragma unroll
      for (int c2 = c + 1; c2 < 4; ++c2) a[Q][c2] -= lj * lane_bcast(lj, 4 * JB + c2);
      Pan[J * NB + lane] = lj;
    }
  }
  __syncthreads();  // panel JB is published
  if constexpr (JB < 15) {
    if constexpr (JB % 4 == W) {
      // the owner catches up: the panel before its own (deferred while it was the next owner), then its own
      if constexpr (JB >= 1) factor64c_update<W, JB - 1, JB, false>(a, Pan, lane);
      factor64c_update<W, JB, JB, false>(a, Pan, lane);
    } else if constexpr ((JB + 1) % 4 == W) {
      factor64c_update<W, JB, JB, true>(a, Pan, lane);  // the next owner: only the block it factors next
    } else {
      factor64c_update<W, JB, JB, false>(a, Pan, lane);
    }
    factor64c_step<W, JB + 1>(a, Pan, rinv, lane, anybad);
  }
}
template <int W>
__device__ __forceinline__ bool factor64c_wave(lds_f64* __restrict__ Lb, lds_f64* __restrict__ Pan,
                                               lds_f64* __restrict__ rinv, int lane) {
  double a[4][4];  // a[q][c] = column 4 (4 q + W) + c of row `lane`
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[q][c] = Lb[lane * DS + 4 * (4 * q + W) + c];
  bool anybad = false;
  factor64c_step<W, 0>(a, Pan, rinv, lane, anybad);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = 4 * (4 * q + W) + c;
      Lb[lane * DS + col] = (col <= lane) ? a[q][c] : 0.0;
    }
  return anybad;
}
__device__ __noinline__ bool factor64c(lds_f64* __restrict__ Lb, lds_f64* __restrict__ Pan, lds_f64* __restrict__ rinv,
                                       int tid) {
  const int lane = tid & 63;
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: return factor64c_wave<0>(Lb, Pan, rinv, lane);
    case 1: return factor64c_wave<1>(Lb, Pan, rinv, lane);
    case 2: return factor64c_wave<2>(Lb, Pan, rinv, lane);
    default: return factor64c_wave<3>(Lb, Pan, rinv, lane);
  }
}

// The four 16 x 16 diagonal blocks of X = L^-1, all at once by ONE wave: lane = 16 blk + column, the column's sixteen
// entries in registers, forward substitution down the rows (x_k final -> its multiples into the rows below; the L entries
// are per-block broadcast reads), no barrier inside.  Replaces the sixteen 4 x 4 inverses and the two lowest levels of
// the block recursion (two barriers each, 0.9 + 1.1 + 1.9 k cycles; this: ~1.3 k beside the log-det sum of another wave).
// X's entries above the diagonal of the blocks are written as zeros; X outside the blocks is not touched.
__device__ inline void inv16x4(const lds_f64* __restrict__ L, const lds_f64* __restrict__ rinv, lds_f64* __restrict__ X,
                               int lane) {
  const int o = 16 * (lane >> 4), c = lane & 15;
  const lds_f64* Lb = L + o * DS + o;
  double x[16], acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double rk = rinv[o + k];
    x[k] = (c == k) ? rk : -rk * acc[k];  // (rows above the diagonal: acc is still zero)
#pragma unroll
    for (int i = k + 1; i < 16; ++i) acc[i] = fma(Lb[i * DS + k], x[k], acc[i]);
  }
  lds_f64* Xb = X + o * DS + o + c;
#pragma unroll
  for (int i = 0; i < 16; ++i) Xb[i * DS] = (i >= c) ? x[i] : 0.0;
}

// Upper levels of the triangular inversion on the MFMA: for every 2H diagonal block X21 = -X22 (L21 X11), H = 16
// (two blocks, two waves each) or 32 (one block, the four waves as 2 x 2).  T: H x H scratch per block, row-major.
template <int H, typename R>
__device__ inline void inv_level_mma(const R* __restrict__ L, R* __restrict__ X, R* __restrict__ T, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  if (H == 16) {
    const int o = 32 * (w >> 1), r0 = 2 * (w & 1);
    R* Tp = T + 256 * (w >> 1);
    double acc[1][1][2];
    mma_zero(acc);
    wave_mma<1, 1, 2, false, false>(acc, L + (o + 16) * DS + o, DS, X + o * DS + o, DS, 0, 16, r0, lane);
    mma_store(acc, Tp, 16, 1.0, r0, lane);
    __syncthreads();
    mma_zero(acc);
    wave_mma<1, 1, 2, false, false>(acc, X + (o + 16) * DS + o + 16, DS, Tp, 16, 0, 16, r0, lane);
    mma_store(acc, X + (o + 16) * DS + o, DS, -1.0, r0, lane);
  } else {
    const int ia = w >> 1, jc = w & 1;
    double acc[1][1][4];
    mma_zero(acc);  // X11 is lower triangular: rows k < 16 jc of its columns 16 jc.. are zero
    wave_mma<1, 1, 4, false, false>(acc, L + (32 + 16 * ia) * DS, DS, X + 16 * jc, DS, 16 * jc, 32, 0, lane);
    mma_store(acc, T + (16 * ia) * 32 + 16 * jc, 32, 1.0, 0, lane);
    __syncthreads();
    mma_zero(acc);  // X22 is lower triangular: columns k >= 16 (ia + 1) of its rows 16 ia.. are zero
    wave_mma<1, 1, 4, false, false>(acc, X + (32 + 16 * ia) * DS + 32, DS, T + 16 * jc, 32, 0, 16 * ia + 16, 0, lane);
    mma_store(acc, X + (32 + 16 * ia) * DS + 16 * jc, DS, -1.0, 0, lane);
  }
  __syncthreads();
}

// L (lower, LDS) -> X = L^-1 (LDS).  On entry X holds the inverses of the four 16 x 16 diagonal blocks (inv16x4) and
// zeros elsewhere.  Ends with a barrier.
template <typename R>
__device__ inline void invert64(const R* __restrict__ L, R* __restrict__ X, R* __restrict__ T, int tid) {
  WV_STAMP(18);
  inv_level_mma<16>(L, X, T, tid);
  WV_STAMP(19);
  inv_level_mma<32>(L, X, T, tid);
}

// potrf_diag: the 64 x 64 leaf (block counts that are not a multiple of two leave one per odd node; also the
// stand-alone right-looking schedule): factor the block at (j0, j0) of every slot's Kbuf, write L_jj back (upper part
// zeroed), inv(L_jj) into Mbuf's diagonal block (and its transpose into MTbuf), sum log L_ii.  A non-positive / NaN
// pivot sets status[slot] = 1 and is replaced by 1 so the pipeline keeps running on finite numbers (the host reports
// +inf for that task).  Same machinery as potrf128 (blocked factor, MFMA inversion levels, float64 inside).
template <typename R>
__global__ __launch_bounds__(256, 2) void potrf_diag_kernel(R* __restrict__ Kall, R* __restrict__ Mall,
                                                            R* __restrict__ MTall, double* __restrict__ logdet,
                                                            int32_t* __restrict__ status, int64_t Npad, int nblk,
                                                            int slot0, int j0) {
  using W = double;
  __shared__ W L[NB * DS];
  __shared__ W X[NB * DS];
  __shared__ W T[INV_SCRATCH];
  __shared__ W rinv[NB];
  __shared__ int bad;
  const int b = slot0 + blockIdx.x;
  R* K = Kall + (int64_t)b * Npad * Npad + (int64_t)j0 * Npad + j0;
  R* M = Mall + (int64_t)b * Npad * Npad + (int64_t)j0 * Npad + j0;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  if (tid == 0) bad = 0;
  W a[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    R v[4];
    ld4(K + (int64_t)(4 * ty + r) * Npad + 4 * tx, v);
#pragma unroll
    for (int c = 0; c < 4; ++c) a[r][c] = (W)v[c];
  }
  // stage the block, factor it in place (factor64c: panels in X), clear X
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(4 * ty + r) * DS + 4 * tx + c] = a[r][c];
  __syncthreads();
  if (factor64c((lds_f64*)L, (lds_f64*)X, (lds_f64*)rinv, tid)) bad = 1;  // (same value from every writer)
  __syncthreads();
  for (int q = tid; q < NB * DS; q += 256) X[q] = 0;
  // L (upper part zeroed by factor64c) back to HBM
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    R v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (R)L[(4 * ty + r) * DS + 4 * tx + c];
    st4(K + (int64_t)(4 * ty + r) * Npad + 4 * tx, v);
  }
  __syncthreads();
  if (tid < 64) {
    inv16x4((const lds_f64*)L, (const lds_f64*)rinv, (lds_f64*)X, tid);
  } else if (tid >= 192) {  // (the last wave: see potrf128_kernel)
    double v = log((double)L[(tid - 192) * DS + tid - 192]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (tid == 192) {
      logdet[(int64_t)b * nblk + j0 / NB] = v;
      if (bad) status[b] = 1;
    }
  }
  __syncthreads();
  invert64<W>(L, X, T, tid);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * ty + r;
    const R v[4] = {(R)X[i * DS + 4 * tx], (R)X[i * DS + 4 * tx + 1], (R)X[i * DS + 4 * tx + 2], (R)X[i * DS + 4 * tx + 3]};
    st4(M + (int64_t)i * Npad + 4 * tx, v);
  }
  if (MTall) {  // transposed image of the inverse block
    R* MT = MTall + (int64_t)b * Npad * Npad + (int64_t)j0 * Npad + j0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * ty + r;
      const R v[4] = {(R)X[(4 * tx) * DS + i], (R)X[(4 * tx + 1) * DS + i], (R)X[(4 * tx + 2) * DS + i],
                      (R)X[(4 * tx + 3) * DS + i]};
      st4(MT + (int64_t)i * Npad + 4 * tx, v);
    }
  }
}

template <typename R>
__global__ __launch_bounds__(256, 2) void potrf128_kernel(R* __restrict__ Kall, R* __restrict__ Mall, R* __restrict__ MTall,
                                                       double* __restrict__ logdet, int32_t* __restrict__ status,
                                                       int64_t Npad, int nblk, int slot0, int o) {
  // LDS: L and X (64 x 65 each) + the inversion scratch + the panel buffers: 77 KB, so TWO workgroups
  // share a CU (round 1 kept a third 64 x 65 array, 100 KB: one workgroup per CU, two passes over a 512-fit batch).
  // Everything in here is float64 whatever the storage type R (float on a float32 context: the leaves are a sliver
  // of the arithmetic, and the diagonal blocks are where a float32 factorisation loses its digits first).
  using W = double;
  __shared__ W L[NB * DS];
  __shared__ W X[NB * DS];
  __shared__ W T[INV_SCRATCH];
  __shared__ W rinv[NB];
  __shared__ int bad;
  const int b = slot0 + blockIdx.x;
  o += 2 * NB * (int)blockIdx.y;   // (a run of independent 128-nodes in one launch: block structures, fi_node)
  const int64_t mat = Npad * Npad;
  const R* K = Kall + (int64_t)b * mat;
  R* M = Mall + (int64_t)b * mat;
  R* MT = MTall + (int64_t)b * mat;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int lane = tid & 63, wr = tid >> 7, wc = (tid >> 6) & 1;  // the four waves as 2 x 2 quadrants of 32 x 32
  const int64_t o1 = o, o2 = o + NB;
  if (tid == 0) bad = 0;

  // factor the block staged in L (factor64c, its panels in X), clear X, 4 x 4 inverses, log-det partial
  auto factor_L_and_logdet = [&](W (&a)[4][4], int blk) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) L[(4 * ty + r) * DS + 4 * tx + c] = a[r][c];
    __syncthreads();
    if (factor64c((lds_f64*)L, (lds_f64*)X, (lds_f64*)rinv, tid)) bad = 1;  // (same value from every writer)
    __syncthreads();
    for (int q = tid; q < NB * DS; q += 256) X[q] = 0;  // (the panels lived in X)
    __syncthreads();
    if (blk & 1) WV_STAMP(8); else WV_STAMP(2);
    if (tid < 64) {  // the first wave: the four 16 x 16 diagonal inverses;  the last one: the log-det partial
      inv16x4((const lds_f64*)L, (const lds_f64*)rinv, (lds_f64*)X, tid);
    } else if (tid >= 192) {
      double v = log((double)L[(tid - 192) * DS + tid - 192]);
#pragma unroll
      for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w, 64);
      if (tid == 192) logdet[(int64_t)b * nblk + blk] = v;
    }
    __syncthreads();
  };
  auto write_inverse_block = [&](int64_t row0, int64_t col0) {  // X -> M[row0.., col0..] and its transpose into M'
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * ty + r;
      const R v[4] = {(R)X[i * DS + 4 * tx], (R)X[i * DS + 4 * tx + 1], (R)X[i * DS + 4 * tx + 2], (R)X[i * DS + 4 * tx + 3]};
      st4(M + (row0 + i) * Npad + col0 + 4 * tx, v);
      const R w[4] = {(R)X[(4 * tx) * DS + i], (R)X[(4 * tx + 1) * DS + i], (R)X[(4 * tx + 2) * DS + i],
                      (R)X[(4 * tx + 3) * DS + i]};
      st4(MT + (col0 + i) * Npad + row0 + 4 * tx, w);
    }
  };

  // ---- A11: factor, invert -> M11 ----
  auto load_block = [&](int64_t row0, int64_t col0, W (&dst)[4][4]) {  // this thread's 4 x 4 of K[row0.., col0..]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      R v[4];
      ld4(K + (row0 + 4 * ty + r) * Npad + col0 + 4 * tx, v);
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[r][c] = (W)v[c];
    }
  };
  W a[4][4];
  WV_STAMP(0);
  load_block(o1, o1, a);
  WV_STAMP(1);
  factor_L_and_logdet(a, o / NB);
  WV_STAMP(3);
  invert64<W>(L, X, T, tid);  // X = M11
  WV_STAMP(4);
  write_inverse_block(o1, o1);
  WV_STAMP(5);

  // ---- S = A21 M11'  (A21 staged in L - L11 is dead once M11 and its log-det exist; M11 lower: k <= column) ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    R v[4];
    ld4(K + (o2 + 4 * ty + r) * Npad + o1 + 4 * tx, v);
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(4 * ty + r) * DS + 4 * tx + c] = (W)v[c];
  }
  __syncthreads();
  double acc[2][2][4];
  mma_zero(acc);
  wave_mma<2, 2, 4, false, true>(acc, L + 32 * wr * DS, DS, X + 32 * wc * DS, DS, 0, 32 * wc + 32, 0, lane);
  __syncthreads();  // every read of L (A21) is done
  mma_store(acc, L + 32 * wr * DS + 32 * wc, DS, 1.0, 0, lane);  // L now holds S
  __syncthreads();
  WV_STAMP(6);

  // ---- T2 = S M11 (kept in registers; M11 lower: k >= column) and the Schur product S S' (lower quadrants) ----
  double t2[2][2][4];
  mma_zero(t2);
  wave_mma<2, 2, 4, false, false>(t2, L + 32 * wr * DS, DS, X + 32 * wc, DS, 32 * wc, NB, 0, lane);
  mma_zero(acc);
  load_block(o2, o2, a);  // A22 (its latency hides behind the product below)
  if (wr >= wc) wave_mma<2, 2, 4, false, true>(acc, L + 32 * wr * DS, DS, L + 32 * wc * DS, DS, 0, NB, 0, lane);
  __syncthreads();  // S and M11 are no longer needed in LDS
  if (wr >= wc) mma_store(acc, X + 32 * wr * DS + 32 * wc, DS, 1.0, 0, lane);  // staged in X for the 4 x 4 layout
  __syncthreads();
  if (ty >= tx) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) a[r][c] -= X[(4 * ty + r) * DS + 4 * tx + c];
  }
  __syncthreads();
  WV_STAMP(7);

  // ---- Schur complement: factor, invert -> M22 ----
  factor_L_and_logdet(a, o / NB + 1);
  WV_STAMP(9);
  invert64<W>(L, X, T, tid);  // X = M22
  WV_STAMP(10);
  write_inverse_block(o2, o2);
  if (tid == 0 && bad) status[b] = 1;
  WV_STAMP(11);

  // ---- M21 = -M22 T2  (T2 staged in L - L22 is dead after the inversion; M22 lower: k <= row) ----
  mma_store(t2, L + 32 * wr * DS + 32 * wc, DS, 1.0, 0, lane);
  __syncthreads();
  mma_zero(acc);
  wave_mma<2, 2, 4, false, false>(acc, X + 32 * wr * DS, DS, L + 32 * wc, DS, 0, 32 * wr + 32, 0, lane);
  __syncthreads();  // every read of T2 is done: L becomes the staging image of M21
  mma_store(acc, L + 32 * wr * DS + 32 * wc, DS, -1.0, 0, lane);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * ty + r;  // row of M21 for the direct store, row of M' block = column of M21 for the other
    const R v[4] = {(R)L[i * DS + 4 * tx], (R)L[i * DS + 4 * tx + 1], (R)L[i * DS + 4 * tx + 2], (R)L[i * DS + 4 * tx + 3]};
    st4(M + (o2 + i) * Npad + o1 + 4 * tx, v);
    const R w[4] = {(R)L[(4 * tx) * DS + i], (R)L[(4 * tx + 1) * DS + i], (R)L[(4 * tx + 2) * DS + i],
                    (R)L[(4 * tx + 3) * DS + i]};
    st4(MT + (o1 + i) * Npad + o2 + 4 * tx, w);
  }
  WV_STAMP(12);
}

#ifdef WV_STAMPS
extern "C" __attribute__((visibility("default"))) int wv_dbg_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wv_stamps), sizeof(wv_stamps));
}
#endif

void launch_potrf128(Problem& p, hipStream_t st, int slot0, int nb, int o, int count = 1) {
  p.timer.begin(st, ST_POTRF_DIAG);
  if (p.f32)
    hipLaunchKernelGGL(potrf128_kernel<float>, dim3(nb, count), dim3(256), 0, st, p.Kbuf32, p.Mbuf32, p.MTbuf32, p.dlogdet,
                       p.dstatus, p.Npad, p.nblk(), slot0, o);
  else
    hipLaunchKernelGGL(potrf128_kernel<double>, dim3(nb, count), dim3(256), 0, st, p.Kbuf, p.Mbuf, p.MTbuf, p.dlogdet,
                       p.dstatus, p.Npad, p.nblk(), slot0, o);
  p.timer.end(st);
}

// Block structures: are the two sides of a split of node (o, n) at o + h1 independent in every structure of the problem
// (Problem::decoupled)?  And is the whole node a run of independent 128-nodes?
static bool node_splits_free(const Problem& p, int64_t o, int64_t h1) { return p.decoupled_at(o + h1); }
static bool node_is_independent_leaves(const Problem& p, int64_t o, int64_t n) {
  if (n % (2 * NB) != 0 || n < 4 * NB) return false;
  for (int64_t b = o + 2 * NB; b < o + n; b += 2 * NB)
    if (!p.decoupled_at(b)) return false;
  return true;
}

// ---------------------------------------------------------------------------------------
// residual r = y[col] - c (zero on padded rows); also clears the slot's status word
// ---------------------------------------------------------------------------------------
__global__ void resid_kernel(const DevSpec* __restrict__ specs, const int32_t* __restrict__ slot_spec,
                             const double* __restrict__ Y,
                             const int32_t* __restrict__ col, const double* __restrict__ theta,
                             double* __restrict__ r, int32_t* __restrict__ status, int64_t N,
                             int64_t Npad, int slot0, const uint8_t* __restrict__ mask) {
  const int b = slot0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad) return;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const double c = sp->mean_param >= 0 ? theta[(int64_t)b * sp->n_params + sp->mean_param] : 0.0;
  const bool seen = mask ? mask[(int64_t)col[b] * Npad + i] != 0 : i < N;  // (a missing output is a padding row)
  r[(int64_t)b * Npad + i] = seen ? Y[(int64_t)col[b] * Npad + i] - c : 0.0;
  if (i == 0) status[b] = 0;
}

void launch_resid(Problem& p, hipStream_t st, int slot0, int nb) {
  dim3 grid((unsigned)((p.Npad + 255) / 256), nb);
  hipLaunchKernelGGL(resid_kernel, grid, dim3(256), 0, st, p.dspec, p.dslotspec, p.dY, p.dcol, p.dtheta, p.dr,
                     p.dstatus, p.N, p.Npad, slot0, p.dmask);
}

// z = M r  (M lower triangular). One wave per row, RPW rows per wave (4; 1 with few fits in flight: four times the
// workgroups, the same sums), 4 RPW rows per workgroup.
template <typename R, int RPW = 4>
__global__ __launch_bounds__(256) void trmv_kernel(const R* __restrict__ Mall,
                                                   const double* __restrict__ rall,
                                                   double* __restrict__ zall, int64_t Npad,
                                                   int slot0) {
  const int b = slot0 + blockIdx.y;
  const R* M = Mall + (int64_t)b * Npad * Npad;
  const double* r = rall + (int64_t)b * Npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = 0; rr < RPW; ++rr) {
    const int64_t i = (int64_t)blockIdx.x * (4 * RPW) + wave * RPW + rr;
    const R* row = M + i * Npad;
    double s = 0.0;
    const int64_t len = i + 1;  // columns 0..i
    for (int64_t j = 2 * lane; j < len; j += 128) {
      const double m0 = (double)row[j], m1 = (double)row[j + 1];
      double2 v = *reinterpret_cast<const double2*>(r + j);
      s += m0 * v.x;
      if (j + 1 < len) s += m1 * v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) zall[(int64_t)b * Npad + i] = s;
  }
}

// alpha = M' z. Workgroup = 64 columns; wave w takes rows j0 + w, j0 + w + 4, ...
template <typename R>
__global__ __launch_bounds__(256) void trmv_t_kernel(const R* __restrict__ Mall,
                                                     const double* __restrict__ zall,
                                                     double* __restrict__ aall, int64_t Npad,
                                                     int slot0) {
  __shared__ double red[4][64];
  const int b = slot0 + blockIdx.y;
  const R* M = Mall + (int64_t)b * Npad * Npad;
  const double* z = zall + (int64_t)b * Npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j0 = (int64_t)blockIdx.x * 64;
  double s = 0.0;
#pragma unroll 4
  for (int64_t i = j0 + wave; i < Npad; i += 4) s += (double)M[i * Npad + j0 + lane] * z[i];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) aall[(int64_t)b * Npad + j0 + lane] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// alpha = M' z above Npad = 2048 (where the one-pass kernel below has no registers for a row): the rows under a block of
// 64 columns in TRMV_SLABS slabs, one workgroup each - with ONE fit in flight trmv_t_kernel's 64 workgroups walked 4 096
// rows each (78 us at N = 4096); partial sums in a fixed place, summed in a fixed order (alpha_slab_reduce_kernel): the
// result does not depend on how many fits share the launch.
constexpr int TRMV_SLABS = 8;
__global__ __launch_bounds__(256) void trmv_t_slab_kernel(const double* __restrict__ Mall, const double* __restrict__ zall,
                                                          double* __restrict__ part, int64_t Npad, int slot0) {
  __shared__ double red[4][64];
  const int b = slot0 + blockIdx.y;
  const double* M = Mall + (int64_t)b * Npad * Npad;
  const double* z = zall + (int64_t)b * Npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cb = blockIdx.x / TRMV_SLABS, sl = blockIdx.x % TRMV_SLABS;
  const int64_t j0 = (int64_t)cb * 64;
  const int64_t rows = Npad - j0, per = (rows / TRMV_SLABS + 3) / 4 * 4;  // (a multiple of 4: every wave the same count)
  const int64_t i0 = j0 + sl * per, i1 = min(Npad, i0 + per);
  double s = 0.0;
#pragma unroll 4
  for (int64_t i = i0 + wave; i < i1; i += 4) s += M[i * Npad + j0 + lane] * z[i];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0)
    part[((int64_t)b * TRMV_SLABS + sl) * Npad + j0 + lane] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
__global__ void alpha_slab_reduce_kernel(const double* __restrict__ part, double* __restrict__ aall, int64_t Npad, int slot0) {
  const int b = slot0 + blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Npad) return;
  double s = 0.0;
#pragma unroll
  for (int sl = 0; sl < TRMV_SLABS; ++sl) s += part[((int64_t)b * TRMV_SLABS + sl) * Npad + j];
  aall[(int64_t)b * Npad + j] = s;
}

// One pass over M for both solves (Npad <= 2048): a wave owns 16 consecutive rows; for row i it
// keeps the row in registers, forms z_i = M[i,:] r (wave reduction) and immediately accumulates
// z_i M[i,:] into its slice of alpha = M' z.  Per-workgroup partial alphas (rows of a 64-row block)
// are summed across row blocks by alpha_reduce_kernel in a fixed order (deterministic).
constexpr int TRMV_MAXC = 16;  // 128-column chunks per row

__global__ __launch_bounds__(256) void trmv_fused_kernel(const double* __restrict__ Mall,
                                                         const double* __restrict__ rall,
                                                         double* __restrict__ zall,
                                                         double* __restrict__ part, int64_t Npad,
                                                         int nrb, int slot0) {
  extern __shared__ __attribute__((aligned(16))) double red[];  // [4][Npad]
  const int b = slot0 + blockIdx.y, rb = blockIdx.x;
  const double* M = Mall + (int64_t)b * Npad * Npad;
  const double* r = rall + (int64_t)b * Npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = (rb * 64 + 64 + 127) / 128;  // chunks that can hold non-zeros for this row block
  double2 rvec[TRMV_MAXC], acc[TRMV_MAXC];
#pragma unroll
  for (int c = 0; c < TRMV_MAXC; ++c) {
    acc[c] = make_double2(0.0, 0.0);
    rvec[c] = (c < nch && 128 * c + 2 * lane < Npad) ? *reinterpret_cast<const double2*>(r + 128 * c + 2 * lane)
                                                       : make_double2(0.0, 0.0);
  }
  for (int rr = 0; rr < 16; ++rr) {
    const int64_t i = (int64_t)rb * 64 + wave * 16 + rr;
    const double* row = M + i * Npad;
    double2 rv[TRMV_MAXC];
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < TRMV_MAXC; ++c) {
      // columns beyond i are zero in M (strictly upper part), so whole chunks can be skipped
      rv[c] = (128 * c <= i && 128 * c + 2 * lane < Npad) ? *reinterpret_cast<const double2*>(row + 128 * c + 2 * lane)
                                                           : make_double2(0.0, 0.0);
      s += rv[c].x * rvec[c].x + rv[c].y * rvec[c].y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) zall[(int64_t)b * Npad + i] = s;
#pragma unroll
    for (int c = 0; c < TRMV_MAXC; ++c) {
      acc[c].x += s * rv[c].x;
      acc[c].y += s * rv[c].y;
    }
  }
#pragma unroll
  for (int c = 0; c < TRMV_MAXC; ++c)
    if (c < nch && 128 * c + 2 * lane < Npad)
      *reinterpret_cast<double2*>(red + (int64_t)wave * Npad + 128 * c + 2 * lane) = acc[c];
  __syncthreads();
  double* out = part + ((int64_t)b * nrb + rb) * Npad;
  const int ncol = min(nch * 128, (int)Npad);
  for (int j = threadIdx.x; j < ncol; j += 256)
    out[j] = red[j] + red[Npad + j] + red[2 * Npad + j] + red[3 * Npad + j];
}

__global__ void alpha_reduce_kernel(const double* __restrict__ part, double* __restrict__ aall, int64_t Npad,
                                    int nrb, int slot0) {
  const int b = slot0 + blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Npad) return;
  double s = 0.0;
  // row block rb holds columns < 128 * ceil((64 rb + 64) / 128); column j first appears in block j / 64
  for (int rb = (int)(j / 64); rb < nrb; ++rb) s += part[((int64_t)b * nrb + rb) * Npad + j];
  aall[(int64_t)b * Npad + j] = s;
}

// ---- value-only path: forward substitution fused into the factorisation (see factor_value_only) ----
// z[off + i] = sum_{j <= i} M[off + i][off + j] r[off + j],  i < len: the diagonal block [off, off + len) of
// the lower-triangular M applied to the matching piece of r.  One wave per row, 16 rows per workgroup.
template <typename R>
__global__ __launch_bounds__(256) void trmv_sub_kernel(const R* __restrict__ Mall, const double* __restrict__ rall,
                                                       double* __restrict__ zall, int64_t Npad, int64_t off,
                                                       int64_t len, int slot0) {
  const int b = slot0 + blockIdx.y;
  const R* M = Mall + (int64_t)b * Npad * Npad + off * Npad + off;
  const double* r = rall + (int64_t)b * Npad + off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t i = (int64_t)blockIdx.x * 16 + wave * 4 + rr;
    if (i >= len) return;
    const R* row = M + i * Npad;
    double s = 0.0;
    for (int64_t j = lane; j <= i; j += 64) s += (double)row[j] * r[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) zall[(int64_t)b * Npad + off + i] = s;
  }
}

// r[roff + m] -= sum_{k < h1} St[k][m] z[zoff + k],  m < h2: the Schur-complement update of the right-hand
// side with S = L21 held transposed (St is h1 x h2, leading dimension h2, per-slot stride sS).
// Workgroup = 64 columns m; wave w takes rows k = w, w + 4, ...
template <typename R>
__global__ __launch_bounds__(256) void gemv_st_kernel(const R* __restrict__ Sall, int64_t sS, int64_t h1, int64_t h2,
                                                      const double* __restrict__ zall, double* __restrict__ rall,
                                                      int64_t Npad, int64_t zoff, int64_t roff, int slot0) {
  __shared__ double red[4][64];
  const int b = slot0 + blockIdx.y;
  const R* S = Sall + (int64_t)b * sS;
  const double* z = zall + (int64_t)b * Npad + zoff;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 64 + lane;
  double s = 0.0;
  if (m < h2)
#pragma unroll 4
    for (int64_t k = wave; k < h1; k += 4) s += (double)S[k * h2 + m] * z[k];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && m < h2) rall[(int64_t)b * Npad + roff + m] -= red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// generic lower-triangular products on any N x N buffer: out = T in (trans = false) or T' in
void launch_trmv(Problem& p, hipStream_t st, const double* Tmat, const double* in, double* out, int slot0,
                 int nb, bool trans) {
  if (!trans)
    hipLaunchKernelGGL(trmv_kernel<double>, dim3((unsigned)(p.Npad / 16), nb), dim3(256), 0, st, Tmat, in, out, p.Npad, slot0);
  else
    hipLaunchKernelGGL(trmv_t_kernel<double>, dim3((unsigned)(p.Npad / 64), nb), dim3(256), 0, st, Tmat, in, out, p.Npad, slot0);
}

void launch_solve(Problem& p, hipStream_t st, int slot0, int nb, bool need_alpha) {
  p.timer.begin(st, ST_SOLVE);
  if (p.f32) {  // float32 storage: the two plain passes over M (float reads, float64 accumulation)
    hipLaunchKernelGGL(trmv_kernel<float>, dim3((unsigned)(p.Npad / 16), nb), dim3(256), 0, st, p.Mbuf32, p.dr, p.dz,
                       p.Npad, slot0);
    if (need_alpha)
      hipLaunchKernelGGL(trmv_t_kernel<float>, dim3((unsigned)(p.Npad / 64), nb), dim3(256), 0, st, p.Mbuf32, p.dz,
                         p.dalpha, p.Npad, slot0);
    p.timer.end(st);
    p.timer.work(ST_SOLVE, 0.0, (double)nb * 4.0 * (double)p.Npad * (p.Npad + 1) / 2.0 * (need_alpha ? 2 : 1));
    return;
  }
  if (need_alpha && p.dsolve_part && p.Npad <= 128 * TRMV_MAXC) {
    const int nrb = (int)(p.Npad / 64);
    hipLaunchKernelGGL(trmv_fused_kernel, dim3(nrb, nb), dim3(256), (size_t)4 * p.Npad * sizeof(double), st, p.Mbuf,
                       p.dr, p.dz, p.dsolve_part, p.Npad, nrb, slot0);
    hipLaunchKernelGGL(alpha_reduce_kernel, dim3((unsigned)((p.Npad + 255) / 256), nb), dim3(256), 0, st,
                       p.dsolve_part, p.dalpha, p.Npad, nrb, slot0);
    p.timer.end(st);
    p.timer.work(ST_SOLVE, 0.0, (double)nb * 8.0 * ((double)p.Npad * (p.Npad + 1) / 2.0 + 2.0 * nrb * p.Npad / 2.0));
    return;
  }
  if (nb * (p.Npad / 16) < 2048)  // few fits: a row per wave (the sums of a row are the same ones)
    hipLaunchKernelGGL((trmv_kernel<double, 1>), dim3((unsigned)(p.Npad / 4), nb), dim3(256), 0, st, p.Mbuf, p.dr,
                       p.dz, p.Npad, slot0);
  else
    hipLaunchKernelGGL(trmv_kernel<double>, dim3((unsigned)(p.Npad / 16), nb), dim3(256), 0, st, p.Mbuf, p.dr,
                       p.dz, p.Npad, slot0);
  if (need_alpha && p.dsolve_part) {
    hipLaunchKernelGGL(trmv_t_slab_kernel, dim3((unsigned)(p.Npad / 64 * TRMV_SLABS), nb), dim3(256), 0, st, p.Mbuf, p.dz,
                       p.dsolve_part, p.Npad, slot0);
    hipLaunchKernelGGL(alpha_slab_reduce_kernel, dim3((unsigned)((p.Npad + 255) / 256), nb), dim3(256), 0, st,
                       p.dsolve_part, p.dalpha, p.Npad, slot0);
  } else if (need_alpha) {
    hipLaunchKernelGGL(trmv_t_kernel<double>, dim3((unsigned)(p.Npad / 64), nb), dim3(256), 0, st, p.Mbuf,
                       p.dz, p.dalpha, p.Npad, slot0);
  }
  p.timer.end(st);
  p.timer.work(ST_SOLVE, 0.0, (double)nb * 8.0 * (double)p.Npad * (p.Npad + 1) / 2.0 * (need_alpha ? 2 : 1));
}

// ---------------------------------------------------------------------------------------
// float32 contexts: what turns the float32 factor's value into a float64-grade number.
// With Mt = the STORED float32 inverse factor and E = Mt K_y Mt' - I (K_y in float64, recomputed - never stored):
//   log|K_y| = -2 sum log Mt_ii + tr E - 1/2 tr E^2 + ...      r'K_y^-1 r = zt'zt - zt'E zt + zt'E^2 zt - ...,  zt = Mt r
// First order (tr E, zt'E zt) comes from the trace pass (KSUM: sum W K_y); here: the float64 diagonal of Mt'Mt that pass
// uses, and the second-order terms -  zt'E^2 zt = |E zt|^2 exactly, E zt = Mt (K_y alpha) - zt, and tr E^2 = |E|_F^2 by ONE
// Rademacher probe v: |E v|^2, E v = Mt (K_y (Mt' v)) - v.  E of a float32 factor is a dense matrix of rounding errors
// (participation ratio ~1e3 measured, tools/probe_f32_residual.py): a single probe gives |E|_F^2 to 1 - 4 %, of a term that
// is itself ~2e-5 of the LML at N = 16384.  O(N^2) work: three passes over Mt, one Gram recomputation.
// ---------------------------------------------------------------------------------------
// WV_F32_ORDER (release switch, wv_api.hip): 1 (default) = the first-order correction only - all a float32 factor with
// float64-flushed accumulators needs at the sizes measured (8e-8 relative at N = 16384); 2 = + second order, 4 = + third and
// fourth (one more application of E): for ill-conditioned K_y (cond * 6e-8 no longer small).
int f32_order() {
  static const int order = [] { const char* e = tuning_env("WV_F32_ORDER"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
  return order;
}

// TWO right-hand sides per pass over the stored factor (vectors live in the [F32R_VECS][Npad] block of a slot, or - the
// `ext` ones: alpha, zt - in arrays of stride Npad):
//   TRANS:  out1 = M' in1, out2 = M' in2, and (pd != null) pd = diag(M'M) = the columns' sums of squares, in float64
//   else :  out1 = M in1 - sub1, out2 = M in2 - sub2
template <typename R, bool TRANS>
__global__ __launch_bounds__(256) void trmv2_kernel(const R* __restrict__ Mall, const double* __restrict__ in1_all,
                                                    const double* __restrict__ in2_all, int64_t in_stride,
                                                    const double* __restrict__ sub1_all, int64_t sub1_stride,
                                                    const double* __restrict__ sub2_all, int64_t sub2_stride,
                                                    double* __restrict__ out1_all, double* __restrict__ out2_all,
                                                    int64_t out_stride, double* __restrict__ pd_all, int64_t Npad, int slot0) {
  __shared__ double red[3][4][64];
  const int b = slot0 + blockIdx.y;
  const R* M = Mall + (int64_t)b * Npad * Npad;
  const double* x1 = in1_all + (int64_t)b * in_stride;
  const double* x2 = in2_all ? in2_all + (int64_t)b * in_stride : nullptr;
  double* y1 = out1_all + (int64_t)b * out_stride;
  double* y2 = out2_all ? out2_all + (int64_t)b * out_stride : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (TRANS) {   // 64 columns per workgroup, the waves stride over the rows (rows above the diagonal hold zeros)
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    double s1 = 0.0, s2 = 0.0, sq = 0.0;
#pragma unroll 8
    for (int64_t i = j0 + wave; i < Npad; i += 4) {
      const double m = (double)M[i * Npad + j0 + lane];
      s1 = fma(m, x1[i], s1);
      if (x2) s2 = fma(m, x2[i], s2);
      sq = fma(m, m, sq);
    }
    red[0][wave][lane] = s1; red[1][wave][lane] = s2; red[2][wave][lane] = sq;
    __syncthreads();
    if (wave == 0) {
      y1[j0 + lane] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
      if (y2) y2[j0 + lane] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
      if (pd_all) pd_all[(int64_t)b * out_stride + j0 + lane] = red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane];
    }
  } else {       // 64 rows per workgroup: a wave takes 16 of them, four at a time (independent loads in flight)
    const double* u1 = sub1_all + (int64_t)b * sub1_stride;
    const double* u2 = sub2_all + (int64_t)b * sub2_stride;
    for (int rr = 0; rr < 16; rr += 4) {
      const int64_t i0 = (int64_t)blockIdx.x * 64 + wave * 16 + rr;
      double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
      for (int64_t j = lane; j <= i0 + 3; j += 64) {
        const double a1 = x1[j], a2 = x2[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double m = j <= i0 + q ? (double)M[(i0 + q) * Npad + j] : 0.0;
          s1[q] = fma(m, a1, s1[q]);
          s2[q] = fma(m, a2, s2[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1[q] += __shfl_xor(s1[q], o, 64); s2[q] += __shfl_xor(s2[q], o, 64); }
        if (lane == 0) { y1[i0 + q] = s1[q] - u1[i0 + q]; y2[i0 + q] = s2[q] - u2[i0 + q]; }
      }
    }
  }
}

void launch_f32_refine(Problem& p, hipStream_t st, int slot0, int nb) {
  const int order = f32_order();
  if (!p.f32 || !p.df32r || order < 2) return;
  const int64_t Np = p.Npad, vs = (int64_t)F32R_VECS * Np;
  double* f = p.df32r;
  auto vec = [&](int k) { return f + (int64_t)k * Np; };
  const dim3 grid((unsigned)(Np / 64), nb), blk(256);
  p.timer.begin(st, ST_SOLVE);   // w = Mt'v, diag(Mt'Mt)
  hipLaunchKernelGGL((trmv2_kernel<float, true>), grid, blk, 0, st, p.Mbuf32, vec(F32R_V), nullptr, vs, nullptr, 0, nullptr, 0,
                     vec(F32R_W), nullptr, vs, vec(F32R_PD), Np, slot0);
  p.timer.end(st);
  p.timer.begin(st, ST_GRAM);    // K_y [w, alpha]
  launch_gram_matvec2(p, st, slot0, nb, vec(F32R_W), p.dalpha, vec(F32R_YV), vec(F32R_YA), vs, Np);
  p.timer.end(st);
  p.timer.begin(st, ST_SOLVE);   // E v = Mt K_y Mt'v - v,  E zt = Mt K_y alpha - zt;  then Mt' of both
  hipLaunchKernelGGL((trmv2_kernel<float, false>), grid, blk, 0, st, p.Mbuf32, vec(F32R_YV), vec(F32R_YA), vs, vec(F32R_V), vs,
                     p.dz, Np, vec(F32R_EV), vec(F32R_EA), vs, nullptr, Np, slot0);
  if (order < 3) { p.timer.end(st); return; }
  hipLaunchKernelGGL((trmv2_kernel<float, true>), grid, blk, 0, st, p.Mbuf32, vec(F32R_EV), vec(F32R_EA), vs, nullptr, 0, nullptr,
                     0, vec(F32R_W2V), vec(F32R_W2A), vs, nullptr, Np, slot0);
  p.timer.end(st);
  p.timer.begin(st, ST_GRAM);
  launch_gram_matvec2(p, st, slot0, nb, vec(F32R_W2V), vec(F32R_W2A), vec(F32R_Y2V), vec(F32R_Y2A), vs, vs);
  p.timer.end(st);
  p.timer.begin(st, ST_SOLVE);   // E^2 v, E^2 zt
  hipLaunchKernelGGL((trmv2_kernel<float, false>), grid, blk, 0, st, p.Mbuf32, vec(F32R_Y2V), vec(F32R_Y2A), vs, vec(F32R_EV), vs,
                     vec(F32R_EA), vs, vec(F32R_E2V), vec(F32R_E2A), vs, nullptr, Np, slot0);
  p.timer.end(st);
}

// ---------------------------------------------------------------------------------------
// reduce: out[b] = {lml, quad, logdet, dLML/dtheta[0..np)}
// ---------------------------------------------------------------------------------------
__device__ inline double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void reduce_kernel(const DevSpec* __restrict__ specs,
                                                     const int32_t* __restrict__ slot_spec,
                                                     const double* __restrict__ zall,
                                                     const double* __restrict__ aall,
                                                     const double* __restrict__ logdet,
                                                     const double* __restrict__ part,
                                                     double* __restrict__ out, int64_t N,
                                                     int64_t Npad, int nblk, int nwg, int part_stride,
                                                     int out_stride, int slot0, int with_grad,
                                                     const int32_t* __restrict__ col,
                                                     const int32_t* __restrict__ nobs,
                                                     const float* __restrict__ M32, const double* __restrict__ f32r,
                                                     int high_orders) {
  __shared__ double sh[4];
  const int b = slot0 + blockIdx.x, tid = threadIdx.x;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const int np = sp->n_params;
  // M32 (float32 contexts, gradient evaluations): the trace pass's records carry one more column, sum w W K_y (KSUM)
  const int ksum = (M32 && with_grad) ? 1 : 0;
  double q = 0.0, sa = 0.0, ld = 0.0, slm = 0.0, swk = 0.0, te2 = 0.0, ze2 = 0.0, te3 = 0.0, ze3 = 0.0, te4 = 0.0, ze4 = 0.0;
  for (int64_t i = tid; i < Npad; i += 256) {
    double zz = zall[(int64_t)b * Npad + i];
    q += zz * zz;
    if (with_grad) sa += aall[(int64_t)b * Npad + i];
    if (ksum) slm += log((double)M32[(int64_t)b * Npad * Npad + i * Npad + i]);   // the STORED inverse factor's diagonal
    if (ksum && f32r) {   // (WV_F32_ORDER >= 2: launch_f32_refine ran)
      const double* v = f32r + (int64_t)b * F32R_VECS * Npad;
      const double ev = v[F32R_EV * Npad + i], ez = v[F32R_EA * Npad + i];      // (E v)_i, (E zt)_i
      te2 = fma(ev, ev, te2);
      ze2 = fma(ez, ez, ze2);
      if (high_orders) {
        const double e2v = v[F32R_E2V * Npad + i], e2z = v[F32R_E2A * Npad + i];  // (E^2 v)_i, (E^2 zt)_i
        te3 = fma(ev, e2v, te3); te4 = fma(e2v, e2v, te4);
        ze3 = fma(ez, e2z, ze3); ze4 = fma(e2z, e2z, ze4);
      }
    }
  }
  if (ksum) {
    for (int w = tid; w < nwg; w += 256) swk += part[((int64_t)b * part_stride + w) * (np + 1) + np];
    slm = block_sum(slm, sh);
    swk = block_sum(swk, sh);
    te2 = block_sum(te2, sh); te3 = block_sum(te3, sh); te4 = block_sum(te4, sh);
    ze2 = block_sum(ze2, sh); ze3 = block_sum(ze3, sh); ze4 = block_sum(ze4, sh);
  }
  for (int i = tid; i < nblk; i += 256) ld += logdet[(int64_t)b * nblk + i];
  q = block_sum(q, sh);
  ld = block_sum(ld, sh);
  sa = block_sum(sa, sh);
  double* o = out + (int64_t)b * out_stride;
  if (tid == 0) {
    const double n_seen = nobs ? (double)nobs[col[b]] : (double)N;  // (complete cases of this output)
    o[0] = -0.5 * q - ld - 0.5 * n_seen * 1.8378770664093453;  // log(2 pi)
    o[1] = q;
    o[2] = ld;
    if (ksum) {
      // float32 context: Mt = the stored float32 inverse factor, zt = Mt r, at = Mt' zt, Wt = at at' - (Mt' Mt as the
      // float32 product left it).  With E = Mt K_y Mt' - I:  K_y^-1 = Mt' (I + E)^-1 Mt  and  log|K_y| = -2 log|Mt| +
      // log|I + E| EXACTLY (K_y in float64: the trace pass recomputes it); to first order in E
      //   r' K_y^-1 r = 2 zt'zt - at' K_y at,    log|K_y| = -2 sum log Mt_ii + tr(Mt' Mt K_y) - n
      // whose sum needs only sum_ij Wt_ij K_y,ij = 2 swk.  What is left is second order in E (~1e-9 relative at N = 16384)
      // plus the rounding of the float32 product Mt' Mt inside Wt.
      // ... orders two to four (launch_f32_refine): -1/2 log|I + E| -> + tr E^2 / 4 - tr E^3 / 6 + tr E^4 / 8 by the probe,
      // -1/2 zt'(I + E)^-1 zt -> - zt'E^2 zt / 2 + zt'E^3 zt / 2 - zt'E^4 zt / 2 exactly
      o[0] = -q + swk + slm + 0.5 * n_seen - 0.5 * n_seen * 1.8378770664093453
             + 0.25 * te2 - te3 / 6.0 + 0.125 * te4 - 0.5 * ze2 + 0.5 * ze3 - 0.5 * ze4;
      o[1] = te2;   // (diagnostics: |E v|^2 ~ |E|_F^2 and |E zt|^2)
      o[2] = ze2;
    }
  }
  if (with_grad) {
    // gradient = sum of the trace pass's per-workgroup partial sums: 16 parameters at a time, 16 threads per
    // parameter striding over the workgroups (a single thread per parameter walking all nwg partials was a chain of
    // dependent loads: 37 us at N = 2048, 131 us at N = 4096 with one fit in flight), then a fixed-order LDS sum
    __shared__ double gsh[16][17];
    const int pl = tid & 15, wl = tid >> 4;
    for (int p0 = 0; p0 < np; p0 += 16) {
      const int p = p0 + pl;
      double g = 0.0;
      if (p < np)
        for (int w = wl; w < nwg; w += 16) g += part[((int64_t)b * part_stride + w) * (np + ksum) + p];
      gsh[wl][pl] = g;
      __syncthreads();
      if (tid < 16 && p0 + tid < np) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += gsh[k][tid];
        o[3 + p0 + tid] = (p0 + tid == sp->mean_param) ? sa : t;
      }
      __syncthreads();
    }
  }
}

void launch_reduce(Problem& p, hipStream_t st, int slot0, int nb, bool with_grad) {
  p.timer.begin(st, ST_REDUCE);
  hipLaunchKernelGGL(reduce_kernel, dim3(nb), dim3(256), 0, st, p.dspec, p.dslotspec, p.dz, p.dalpha, p.dlogdet,
                     p.dpart, p.dout, p.N, p.Npad, p.nblk(), trace_nwg(p, nb), trace_part_stride(p), p.out_stride, slot0,
                     with_grad ? 1 : 0, p.dcol, p.dmask ? p.dnobs : nullptr, p.f32 ? p.Mbuf32 : nullptr,
                     f32_order() >= 2 ? p.df32r : nullptr, f32_order() >= 3 ? 1 : 0);
  p.timer.end(st);
}

// ---------------------------------------------------------------------------------------
// schedules
// ---------------------------------------------------------------------------------------
template <typename R = double>
static GemmArgsT<R> base_args(const Problem& p, int slot0, int nb) {
  GemmArgsT<R> a{};
  const int64_t mat = p.Npad * p.Npad;
  a.sA = a.sB = a.sC = mat;
  a.lda = a.ldb = a.ldc = (int)p.Npad;
  a.alpha = 1.0;
  a.beta = 0.0;
  a.nbatch = nb;
  (void)slot0;
  return a;
}
// Block structure: (r0, c0, k0) = the K_y index of the product's row 0, column 0 and k = 0 (see GemmArgsT::spans)
template <typename R>
static void with_spans(GemmArgsT<R>& a, const Problem& p, int slot0, int64_t r0, int64_t c0, int64_t k0) {
  if (p.n_span_tables == 0) return;
  a.spans = p.dspans;
  a.spec_span = p.dspec_span;
  a.slot_spec = p.dslotspec;
  a.span_nblk = (int)(p.Npad / 64);
  a.slot0 = slot0;
  a.sp_r0 = (int)r0; a.sp_c0 = (int)c0; a.sp_k0 = (int)k0;
}

void cholesky(Problem& p, hipStream_t st, int slot0, int nb) {
  const int64_t Np = p.Npad, mat = Np * Np;
  double* K0 = p.Kbuf + (int64_t)slot0 * mat;
  double* M0 = p.Mbuf + (int64_t)slot0 * mat;
  for (int64_t o0 = 0; o0 < Np; o0 += OUTER) {
    const int64_t W = std::min<int64_t>(OUTER, Np - o0);
    for (int64_t j0 = o0; j0 < o0 + W; j0 += NB) {
      launch_potrf_diag(p, st, slot0, nb, (int)j0);
      const int64_t r0 = j0 + NB, rem = Np - r0;
      if (rem <= 0) continue;
      {  // TRSM: L21 = A21 * inv(L_jj)'  (NT, K = 64), in place
        GemmArgs a = base_args(p, slot0, nb);
        a.A = K0 + r0 * Np + j0;
        a.B = M0 + j0 * Np + j0;  // inv(L_jj), row-major lower
        a.C = K0 + r0 * Np + j0;
        a.M = (int)rem; a.N = NB; a.kb = 0; a.ke = NB;
        p.timer.begin(st, ST_CHOL_TRSM);
        launch_gemm(st, a, true, true, 64);
        p.timer.end(st);
        p.timer.work(ST_CHOL_TRSM, 2.0 * nb * (double)rem * NB * NB, 0.0);
      }
      const int64_t wrem = o0 + W - r0;  // panel columns still to update
      if (wrem > 0) {  // A[r0:, r0:o0+W] -= L[r0:, j0:j0+64] L[r0:o0+W, j0:j0+64]'
        GemmArgs a = base_args(p, slot0, nb);
        a.A = K0 + r0 * Np + j0;
        a.B = K0 + r0 * Np + j0;
        a.C = K0 + r0 * Np + r0;
        a.M = (int)rem; a.N = (int)wrem; a.kb = 0; a.ke = NB;
        a.alpha = -1.0; a.beta = 1.0; a.tri = 1; a.tri_off = 0;
        p.timer.begin(st, ST_CHOL_PANEL);
        launch_gemm(st, a, true, true, wrem > 64 ? 128 : 64);
        p.timer.end(st);
        p.timer.work(ST_CHOL_PANEL, 2.0 * nb * NB * ((double)rem * wrem - 0.5 * (double)wrem * wrem), 0.0);
      }
    }
    const int64_t t0 = o0 + W, rem = Np - t0;
    if (rem > 0) {  // trailing SYRK: A[t0:, t0:] -= L[t0:, o0:t0] L[t0:, o0:t0]'
      GemmArgs a = base_args(p, slot0, nb);
      a.A = K0 + t0 * Np + o0;
      a.B = K0 + t0 * Np + o0;
      a.C = K0 + t0 * Np + t0;
      a.M = (int)rem; a.N = (int)rem; a.kb = 0; a.ke = (int)W;
      a.alpha = -1.0; a.beta = 1.0; a.tri = 1; a.tri_off = 0;
      p.timer.begin(st, ST_CHOL_TRAIL);
      launch_gemm(st, a, true, true, rem > 64 ? 128 : 64);
      p.timer.end(st);
      p.timer.work(ST_CHOL_TRAIL, nb * (double)W * (double)rem * (rem + 1), 0.0);
    }
  }
}

void trtri(Problem& p, hipStream_t st, int slot0, int nb) {
  const int64_t Np = p.Npad, mat = Np * Np;
  double* K0 = p.Kbuf + (int64_t)slot0 * mat;
  double* M0 = p.Mbuf + (int64_t)slot0 * mat;
  double* P0 = p.Pbuf + (int64_t)slot0 * p.pbuf_elems();
  for (int64_t j0 = Np - 2 * NB; j0 >= 0; j0 -= NB) {
    const int64_t r0 = j0 + NB, rem = Np - r0;
    {  // P = M22 * L21   (NN; M22 lower triangular -> K range ends at the tile's last row)
      GemmArgs a = base_args(p, slot0, nb);
      a.A = M0 + r0 * Np + r0;
      a.B = K0 + r0 * Np + j0;  // L21 stored [k][n]
      a.C = P0;
      a.sC = p.pbuf_elems(); a.ldc = NB;
      a.M = (int)rem; a.N = NB; a.kb = 0; a.ke = (int)rem; a.khi_mode = 1; a.koff = 0;
      p.timer.begin(st, ST_TRTRI_A);
      launch_gemm(st, a, true, false, 64);
      p.timer.end(st);
      p.timer.work(ST_TRTRI_A, nb * (double)NB * (double)rem * (rem + 1), 0.0);
    }
    {  // M21 = -P * M_jj   (NN, K = 64)
      GemmArgs a = base_args(p, slot0, nb);
      a.A = P0; a.sA = p.pbuf_elems(); a.lda = NB;
      a.B = M0 + j0 * Np + j0;  // M_jj stored [k][n]
      a.C = M0 + r0 * Np + j0;
      a.M = (int)rem; a.N = NB; a.kb = 0; a.ke = NB; a.alpha = -1.0;
      p.timer.begin(st, ST_TRTRI_B);
      launch_gemm(st, a, true, false, 64);
      p.timer.end(st);
      p.timer.work(ST_TRTRI_B, 2.0 * nb * (double)rem * NB * NB, 0.0);
    }
  }
}

// ---------------------------------------------------------------------------------------
// factor_inverse: recursive K_y = L L' together with M = L^-1, every O(n^3) piece a square-ish
// GEMM.  Node (o, n) splits n = h1 + h2 (multiples of 64):
//   node(o, h1)                       -> L11 (diag blocks), M11
//   S   = A21 M11'                    L21, kept in scratch S (contraction k < column + 1)
//   A22 -= S S'                       Schur complement, lower tiles
//   T   = S M11                       stored over the dead A21 block (contraction k >= column)
//   node(o + h1, h2)                  -> M22
//   M21 = -M22 T                      (contraction k <= row)
// L itself is never assembled: log|K_y| comes from the diagonal blocks, z = M r, alpha = M' z.
// ---------------------------------------------------------------------------------------
// Tile for a product with a rows x cols output over nb matrices (launch_gemm's tile code): 128 x 128 when the launch
// has at least 4096 such tiles (eight per resident workgroup slot), else the small 64 x 64 tile - four times the
// workgroups, which balance better over the 256 CUs when the tiles are few (one fit in flight: every product of the
// recursion) or the launch is only a few waves of workgroups deep (full batches: the two lowest levels of the
// recursion).  Measured (ms per round, thresholds 512 -> 4096): one fit N = 2048 3.72 -> 2.06 with the small tile at
// all; 16 fits 5.42 -> 5.08, 128 fits 27.5 -> 26.6, 512 fits 101.8 -> 100.6.  The 128 x 64 tile of round 1 has no
// window left (WV_SMALL_FILL = 2 x WV_BN_FILL); WV_BN_FILL=0 restores 128 x 128 everywhere, WV_SMALL_FILL=0 stops at
// 128 x 64.
static int pick_bn(int64_t rows, int64_t cols, int nb, bool lower_only) {
  static const int fill = [] { const char* e = tuning_env("WV_BN_FILL"); return e ? atoi(e) : 4096; }();
  static const int small_fill = [] { const char* e = tuning_env("WV_SMALL_FILL"); return e ? atoi(e) : 8192; }();
  int64_t tiles = ((rows + 127) / 128) * ((cols + 127) / 128);
  if (lower_only) tiles = (tiles + (rows + 127) / 128) / 2;
  if (tiles * nb >= fill) return cols <= 64 ? 64 : 128;
  if (2 * tiles * nb >= small_fill) return 64;
  return 32;
}

// Size of the left child of a recursion node of n rows (a multiple of 64): half of n in units of the fused 128-leaf
// when n is a multiple of 128 - so that e.g. 384 = 128 + 256 ends in three fused leaves instead of 192 + 192 =
// 2 x (64-leaf + 128-leaf) with their extra small products - else half in units of 64.
static int64_t split_point(int64_t n) {
  if (n % (2 * NB) == 0) return (n / (2 * NB) / 2) * (2 * NB);
  return (n / NB / 2) * NB;
}

template <typename R>
static void fi_node(Problem& p, hipStream_t st, int slot0, int nb, int64_t o, int64_t n) {
  if (n == NB) {
    launch_potrf_diag(p, st, slot0, nb, (int)o);
    return;
  }
  if (n == 2 * NB) {  // the whole node in one kernel
    launch_potrf128(p, st, slot0, nb, (int)o);
    return;
  }
  const int64_t Np = p.Npad, mat = Np * Np;
  const int64_t h1 = split_point(n), h2 = n - h1;
  // Block structures (Problem::decoupled): K_y(rows >= o + h1, columns < o + h1) is structurally zero in every structure
  // of the problem - S, the update, T and M21 vanish (M21 stays the zero block it was created as), the children are
  // independent; a node that is nothing but independent 128-nodes is ONE leaf launch.
  if (node_is_independent_leaves(p, o, n)) {
    launch_potrf128(p, st, slot0, nb, (int)o, (int)(n / (2 * NB)));
    return;
  }
  if (node_splits_free(p, o, h1)) {
    fi_node<R>(p, st, slot0, nb, o, h1);
    fi_node<R>(p, st, slot0, nb, o + h1, h2);
    return;
  }
  const MatBufs<R> mb = mat_bufs<R>(p);
  R* K0 = mb.K + (int64_t)slot0 * mat;
  R* M0 = mb.M + (int64_t)slot0 * mat;
  R* MT0 = mb.MT + (int64_t)slot0 * mat;
  R* S0 = mb.P + (int64_t)slot0 * p.pbuf_elems();  // holds S' (h1 x h2): K-strided image of S
  const int64_t sS = p.pbuf_elems();
  fi_node<R>(p, st, slot0, nb, o, h1);
  {  // S = A21 * M11'   written transposed;  B(n, k) = M11[n][k] = MT[k][n]  (K-strided)
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = K0 + (o + h1) * Np + o;
    a.B = MT0 + o * Np + o;
    a.C = nullptr;
    a.CT = S0; a.sCT = sS; a.ldct = (int)h2; a.tout = 1;
    a.M = (int)h2; a.N = (int)h1; a.kb = 0; a.ke = (int)h1; a.khi_mode = 2;
    with_spans(a, p, slot0, o + h1, o, o);
    p.timer.begin(st, ST_CHOL_TRSM);
    launch_gemm(st, a, true, false, pick_bn(h2, h1, nb, false));
    p.timer.end(st);
    p.timer.work(ST_CHOL_TRSM, nb * (double)h2 * h1 * (h1 + NB), 0.0);
  }
  // A22 -= S S'   with A(m, k) = S'[k][m], B(n, k) = S'[k][n]
  GemmArgsT<R> su = base_args<R>(p, slot0, nb);
  su.A = S0; su.B = S0; su.sA = su.sB = sS; su.lda = su.ldb = (int)h2;
  su.C = K0 + (o + h1) * Np + (o + h1);
  su.M = (int)h2; su.N = (int)h2; su.kb = 0; su.ke = (int)h1;
  su.alpha = -1.0; su.beta = 1.0; su.tri = 1; su.tag = 2;
  with_spans(su, p, slot0, o + h1, o + h1, o);
  // T = S * M11 over the dead A21 block;  A(m, k) = S'[k][m],  B(n, k) = M11[k][n].  Off the critical path: only
  // M21, after the whole right child, reads it.
  GemmArgsT<R> tp = base_args<R>(p, slot0, nb);
  tp.A = S0; tp.sA = sS; tp.lda = (int)h2;
  tp.B = M0 + o * Np + o;
  tp.C = K0 + (o + h1) * Np + o;
  tp.M = (int)h2; tp.N = (int)h1; tp.kb = 0; tp.ke = (int)h1; tp.klo_mode = 2;
  with_spans(tp, p, slot0, o + h1, o, o);
  const int tile_su = pick_bn(h2, h2, nb, true), tile_tp = pick_bn(h2, h1, nb, false);
  bool paired = false;
  if constexpr (std::is_same<R, double>::value) {
    // few fits in flight (both products on 64-wide or small tiles): ONE launch runs them side by side
    static const bool pair_ok = [] { const char* e = tuning_env("WV_PAIR"); return e ? atoi(e) != 0 : true; }();
    if (pair_ok && tile_su <= 64 && tile_tp <= 64) {
      p.timer.begin(st, ST_CHOL_TRAIL);
      launch_gemm_pair(st, su, tp, std::min(tile_su, tile_tp));
      p.timer.end(st);
      paired = true;
    }
  }
  if (!paired) {
    p.timer.begin(st, ST_CHOL_TRAIL);
    launch_gemm(st, su, false, false, tile_su);
    p.timer.end(st);
    p.timer.begin(st, ST_TRTRI_A);
    launch_gemm(st, tp, false, false, tile_tp);
    p.timer.end(st);
  }
  p.timer.work(ST_CHOL_TRAIL, nb * (double)h1 * (double)h2 * (h2 + NB), 0.0);
  p.timer.work(ST_TRTRI_A, nb * (double)h2 * h1 * (h1 + NB), 0.0);
  fi_node<R>(p, st, slot0, nb, o + h1, h2);
  {  // M21 = -M22 * T  (also written transposed into MT);  A(m, k) = M22[m][k] = MT[k][m]
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = MT0 + (o + h1) * Np + (o + h1);
    a.B = K0 + (o + h1) * Np + o;  // B(n, k) = T[k][n]
    a.C = M0 + (o + h1) * Np + o;
    a.CT = MT0 + o * Np + (o + h1); a.sCT = mat; a.ldct = (int)Np; a.tout = 2;
    a.M = (int)h2; a.N = (int)h1; a.kb = 0; a.ke = (int)h2; a.khi_mode = 1; a.alpha = -1.0;
    with_spans(a, p, slot0, o + h1, o, o + h1);
    p.timer.begin(st, ST_TRTRI_B);
    launch_gemm(st, a, false, false, pick_bn(h2, h1, nb, false));
    p.timer.end(st);
    p.timer.work(ST_TRTRI_B, nb * (double)h1 * (double)h2 * (h2 + NB), 0.0);
  }
}

// ---------------------------------------------------------------------------------------
// factor_value_only: log|K_y| and z = L^-1 r WITHOUT the inverse of the whole matrix (value-only evaluations:
// wv_lml).  Node (o, n), n = h1 + h2:
//   fi_node(o, h1)                full factor + inverse of the left block (its M11 makes the TRSM a GEMM)
//   z1  = M11 r1                  forward substitution on the left block
//   S   = A21 M11'                (as S')
//   r2 -= S z1                    right-hand side of the Schur complement
//   A22 -= S S'
//   node(o + h1, h2)              value-only recursion on the Schur complement
// T(n) = n^3/3 + T(n/2)  ->  0.38 N^3 instead of the 0.67 N^3 of factor + inverse; z lands in p.dz, r is consumed.
// ---------------------------------------------------------------------------------------
template <typename R>
static void fv_node(Problem& p, hipStream_t st, int slot0, int nb, int64_t o, int64_t n) {
  const int64_t Np = p.Npad, mat = Np * Np;
  const MatBufs<R> mb = mat_bufs<R>(p);
  if (n == NB) {
    launch_potrf_diag(p, st, slot0, nb, (int)o);
    p.timer.begin(st, ST_SOLVE);
    hipLaunchKernelGGL(trmv_sub_kernel<R>, dim3((unsigned)(NB / 16), nb), dim3(256), 0, st, mb.M, p.dr, p.dz, Np, o,
                       (int64_t)NB, slot0);
    p.timer.end(st);
    return;
  }
  if (n == 2 * NB) {  // fused node: M of the whole 128 block is available, one forward substitution
    launch_potrf128(p, st, slot0, nb, (int)o);
    p.timer.begin(st, ST_SOLVE);
    hipLaunchKernelGGL(trmv_sub_kernel<R>, dim3((unsigned)(2 * NB / 16), nb), dim3(256), 0, st, mb.M, p.dr, p.dz, Np, o,
                       (int64_t)(2 * NB), slot0);
    p.timer.end(st);
    return;
  }
  const int64_t h1 = split_point(n), h2 = n - h1;
  R* K0 = mb.K + (int64_t)slot0 * mat;
  R* MT0 = mb.MT + (int64_t)slot0 * mat;
  R* S0 = mb.P + (int64_t)slot0 * p.pbuf_elems();
  const int64_t sS = p.pbuf_elems();
  if (node_is_independent_leaves(p, o, n)) {   // block structures: independent 128-nodes - one leaf launch, one substitution
    launch_potrf128(p, st, slot0, nb, (int)o, (int)(n / (2 * NB)));
    p.timer.begin(st, ST_SOLVE);
    hipLaunchKernelGGL(trmv_sub_kernel<R>, dim3((unsigned)(n / 16), nb), dim3(256), 0, st, mb.M, p.dr, p.dz, Np, o, n, slot0);
    p.timer.end(st);
    return;
  }
  fi_node<R>(p, st, slot0, nb, o, h1);
  p.timer.begin(st, ST_SOLVE);
  hipLaunchKernelGGL(trmv_sub_kernel<R>, dim3((unsigned)(h1 / 16), nb), dim3(256), 0, st, mb.M, p.dr, p.dz, Np, o, h1,
                     slot0);
  p.timer.end(st);
  if (node_splits_free(p, o, h1)) {   // S = 0: nothing reaches the right-hand side or the Schur complement of the right block
    fv_node<R>(p, st, slot0, nb, o + h1, h2);
    return;
  }
  {  // S = A21 * M11'   written transposed (same product as in fi_node)
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = K0 + (o + h1) * Np + o;
    a.B = MT0 + o * Np + o;
    a.C = nullptr;
    a.CT = S0; a.sCT = sS; a.ldct = (int)h2; a.tout = 1;
    a.M = (int)h2; a.N = (int)h1; a.kb = 0; a.ke = (int)h1; a.khi_mode = 2;
    with_spans(a, p, slot0, o + h1, o, o);
    p.timer.begin(st, ST_CHOL_TRSM);
    launch_gemm(st, a, true, false, pick_bn(h2, h1, nb, false));
    p.timer.end(st);
    p.timer.work(ST_CHOL_TRSM, nb * (double)h2 * h1 * (h1 + NB), 0.0);
  }
  p.timer.begin(st, ST_SOLVE);
  hipLaunchKernelGGL(gemv_st_kernel<R>, dim3((unsigned)((h2 + 63) / 64), nb), dim3(256), 0, st, mb.P, sS, h1, h2, p.dz,
                     p.dr, Np, o, o + h1, slot0);
  p.timer.end(st);
  {  // A22 -= S S'
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = S0; a.B = S0; a.sA = a.sB = sS; a.lda = a.ldb = (int)h2;
    a.C = K0 + (o + h1) * Np + (o + h1);
    a.M = (int)h2; a.N = (int)h2; a.kb = 0; a.ke = (int)h1;
    a.alpha = -1.0; a.beta = 1.0; a.tri = 1;
    with_spans(a, p, slot0, o + h1, o + h1, o);
    p.timer.begin(st, ST_CHOL_TRAIL);
    launch_gemm(st, a, false, false, pick_bn(h2, h2, nb, true));
    p.timer.end(st);
    p.timer.work(ST_CHOL_TRAIL, nb * (double)h1 * (double)h2 * (h2 + NB), 0.0);
  }
  fv_node<R>(p, st, slot0, nb, o + h1, h2);
}

void factor_value_only(Problem& p, hipStream_t st, int slot0, int nb) {
  if (p.f32) fv_node<float>(p, st, slot0, nb, 0, p.Npad);
  else fv_node<double>(p, st, slot0, nb, 0, p.Npad);
}

void factor_inverse(Problem& p, hipStream_t st, int slot0, int nb) {
  if (p.f32) fi_node<float>(p, st, slot0, nb, 0, p.Npad);
  else fi_node<double>(p, st, slot0, nb, 0, p.Npad);
}

template <typename R>
static void lauum_t(Problem& p, hipStream_t st, int slot0, int nb) {
  const int64_t Np = p.Npad, mat = Np * Np;
  const MatBufs<R> mb = mat_bufs<R>(p);
  GemmArgsT<R> a = base_args<R>(p, slot0, nb);
  a.A = mb.M + (int64_t)slot0 * mat;  // A(m, l) = M[l][m]
  a.B = mb.M + (int64_t)slot0 * mat;  // B(n, l) = M[l][n]
  a.C = mb.K + (int64_t)slot0 * mat;
  a.M = (int)Np; a.N = (int)Np; a.kb = 0; a.ke = (int)Np; a.klo_mode = 1; a.koff = 0;
  a.tri = 1; a.tri_off = 0; a.tag = 1;
  with_spans(a, p, slot0, 0, 0, 0);
  // 128-wide tiles (its own kernel symbols <..., 0, 1> and, for the diagonal-tile launch, <..., 0, 5>) unless so few
  // fits are in flight that a few long tiles are the cost: then the small tile, one launch (timed as its own stage, so
  // that ST_LAUUM's launches are exactly those of the two named symbols)
  static const int forced = [] { const char* e = experiment_env("WV_LAUUM_TILE"); return e ? atoi(e) : 0; }();  // (experiments)
  const int tile = forced ? forced : pick_bn(Np, Np, nb, true);
  const int stage = tile == 128 ? ST_LAUUM : ST_LAUUM_FEW;
  p.timer.begin(st, stage);
  launch_gemm(st, a, false, false, tile);
  p.timer.end(st);
  p.timer.work(stage, nb * (double)Np * Np * Np / 3.0, 0.0);
}

void lauum(Problem& p, hipStream_t st, int slot0, int nb) {
  if (p.f32) lauum_t<float>(p, st, slot0, nb);
  else lauum_t<double>(p, st, slot0, nb);
}

}  // namespace wv
