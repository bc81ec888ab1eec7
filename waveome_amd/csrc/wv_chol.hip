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
#include "wv_device.hpp"

namespace wv {

constexpr int OUTER = 256;  // outer Cholesky block
constexpr int DS = NB + 1;  // LDS row stride of the 64x64 diagonal block

// ---------------------------------------------------------------------------------------
// potrf_diag: factor the 64x64 diagonal block at (j0, j0) of every slot's Kbuf in LDS, write
// L_jj back (upper part zeroed), inv(L_jj) into Mbuf's diagonal block, sum log L_ii.
// A non-positive / NaN pivot sets status[slot] = 1 and is replaced by 1 so the pipeline
// keeps running on finite numbers (the host reports +inf for that task).
// ---------------------------------------------------------------------------------------
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

// Triangular-inverse combine step of size H inside the 64x64 block: for every 2H diagonal block
// X21 = -X22 (L21 X11), all operands in LDS.
// T is a compact scratch of NB/2 * H <= 1024 entries: entry (pair blk, row i, column c) at (blk * H + i) * H + c.
constexpr int INV_SCRATCH = NB * NB / 4;
template <int H, typename R>
__device__ inline void inv_level(const R* __restrict__ L, R* __restrict__ X, R* __restrict__ T, int tid) {
  constexpr int NPAIR = NB / (2 * H);
  constexpr int PER = H * H;
  for (int q = tid; q < NPAIR * PER; q += 256) {
    const int blk = q / PER, e = q % PER, i = e / H, c = e % H, o = blk * 2 * H;
    R s = 0;
#pragma unroll 8
    for (int k = c; k < H; ++k) s += L[(o + H + i) * DS + o + k] * X[(o + k) * DS + o + c];
    T[q] = s;
  }
  __syncthreads();
  for (int q = tid; q < NPAIR * PER; q += 256) {
    const int blk = q / PER, e = q % PER, i = e / H, c = e % H, o = blk * 2 * H;
    R s = 0;
#pragma unroll 8
    for (int k = 0; k <= i; ++k) s += X[(o + H + i) * DS + o + H + k] * T[(blk * H + k) * H + c];
    X[(o + H + i) * DS + o + c] = -s;
  }
  __syncthreads();
}

// Register-tiled right-looking Cholesky of the 64x64 block: thread (ty, tx) of a 16 x 16 grid owns
// rows 4 ty.., columns 4 tx.. in registers; per column one LDS broadcast of the pivot column and
// ONE barrier (the column buffer is double-buffered).
template <typename R>
__global__ __launch_bounds__(256) void potrf_diag_kernel(R* __restrict__ Kall, R* __restrict__ Mall,
                                                         R* __restrict__ MTall,
                                                         double* __restrict__ logdet,
                                                         int32_t* __restrict__ status,
                                                         int64_t Npad, int nblk, int slot0, int j0) {
  __shared__ R L[NB * DS];
  __shared__ R X[NB * DS];
  __shared__ R T[INV_SCRATCH];
  __shared__ R colbuf[2][NB];
  __shared__ int bad;
  const int b = slot0 + blockIdx.x;
  R* K = Kall + (int64_t)b * Npad * Npad + (int64_t)j0 * Npad + j0;
  R* M = Mall + (int64_t)b * Npad * Npad + (int64_t)j0 * Npad + j0;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  if (tid == 0) bad = 0;
  R a[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ld4(K + (int64_t)(4 * ty + r) * Npad + 4 * tx, a[r]);
  for (int q = tid; q < NB * DS; q += 256) X[q] = 0;
  bool anybad = false;
  for (int jb = 0; jb < 16; ++jb) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * jb + jj;
      R* cb = colbuf[j & 1];
      if (tx == jb && ty >= jb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cb[4 * ty + r] = a[r][jj];
      }
      __syncthreads();
      R piv = cb[j];
      const bool ok = piv > R(0);  // false for NaN as well
      if (!ok) { piv = 1; anybad = true; }
      const R inv = rsq(piv);
      if (ty >= tx && tx >= jb) {
        R lr[4], lc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) lr[r] = cb[4 * ty + r] * inv;
#pragma unroll
        for (int c = 0; c < 4; ++c) lc[c] = cb[4 * tx + c] * inv;
        if (tx == jb) {
          // finalise column j (rows > j; the diagonal becomes piv * inv = sqrt(piv))
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r][jj] = (ty == jb && r == jj) ? piv * inv : lr[r];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c > jj) a[r][c] -= lr[r] * lc[c];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[r][c] -= lr[r] * lc[c];
        }
      }
    }
  }
  // L to LDS (upper part zeroed) and back to HBM
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = 4 * ty + r, jc = 4 * tx + c;
      const R v = (jc <= i) ? a[r][c] : R(0);
      a[r][c] = v;
      L[i * DS + jc] = v;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) st4(K + (int64_t)(4 * ty + r) * Npad + 4 * tx, a[r]);
  if (anybad && tid == 0) bad = 1;
  __syncthreads();
  if (tid < 64) {
    double v = log((double)L[tid * DS + tid]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (tid == 0) {
      logdet[(int64_t)b * nblk + j0 / NB] = v;
      if (bad) status[b] = 1;
    }
  }
  // ---- inverse of the lower-triangular block by recursive doubling ----
  if (tid < 64) {  // level 0: sixteen 4x4 diagonal blocks, one thread per column (a short dependent chain)
    const int blk = tid >> 2, c = tid & 3, o = blk * 4;
    X[(o + c) * DS + o + c] = R(1) / L[(o + c) * DS + o + c];
    for (int i = c + 1; i < 4; ++i) {
      R s = 0;
      for (int k = c; k < i; ++k) s += L[(o + i) * DS + o + k] * X[(o + k) * DS + o + c];
      X[(o + i) * DS + o + c] = -s / L[(o + i) * DS + o + i];
    }
  }
  __syncthreads();
  inv_level<4>(L, X, T, tid);
  inv_level<8>(L, X, T, tid);
  inv_level<16>(L, X, T, tid);
  inv_level<32>(L, X, T, tid);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * ty + r;
    const R v[4] = {X[i * DS + 4 * tx], X[i * DS + 4 * tx + 1], X[i * DS + 4 * tx + 2], X[i * DS + 4 * tx + 3]};
    st4(M + (int64_t)i * Npad + 4 * tx, v);
  }
  if (MTall) {  // transposed image of the inverse block
    R* MT = MTall + (int64_t)b * Npad * Npad + (int64_t)j0 * Npad + j0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * ty + r;
      const R v[4] = {X[(4 * tx) * DS + i], X[(4 * tx + 1) * DS + i], X[(4 * tx + 2) * DS + i], X[(4 * tx + 3) * DS + i]};
      st4(MT + (int64_t)i * Npad + 4 * tx, v);
    }
  }
}

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
// leaf launches and four 64^3 GEMM launches.  Same thread layout as potrf_diag_kernel (16 x 16 threads, 4 x 4
// elements each); the three 64^3 products run on the fp64 VALU out of LDS (2 MFLOP per matrix: noise next to the
// launches they replace, which are pure latency at this size).  Writes M and M' of the 128 block and the two
// log-det partials; K's own blocks are dead after this node and are not written back.
// ---------------------------------------------------------------------------------------
template <typename R>
__device__ inline bool factor64(R (&a)[4][4], R (*colbuf)[NB], int ty, int tx) {
  bool anybad = false;
  for (int jb = 0; jb < 16; ++jb) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * jb + jj;
      R* cb = colbuf[j & 1];
      if (tx == jb && ty >= jb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cb[4 * ty + r] = a[r][jj];
      }
      __syncthreads();
      R piv = cb[j];
      const bool ok = piv > R(0);
      if (!ok) { piv = 1; anybad = true; }
      const R inv = rsq(piv);
      if (ty >= tx && tx >= jb) {
        R lr[4], lc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) lr[r] = cb[4 * ty + r] * inv;
#pragma unroll
        for (int c = 0; c < 4; ++c) lc[c] = cb[4 * tx + c] * inv;
        if (tx == jb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r][jj] = (ty == jb && r == jj) ? piv * inv : lr[r];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c > jj) a[r][c] -= lr[r] * lc[c];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[r][c] -= lr[r] * lc[c];
        }
      }
    }
  }
  return anybad;
}

// L (lower, LDS) -> X = L^-1 (LDS; must be zero on entry), T scratch.  Ends with a barrier.
template <typename R>
__device__ inline void invert64(const R* __restrict__ L, R* __restrict__ X, R* __restrict__ T, int tid) {
  if (tid < 64) {
    const int blk = tid >> 2, c = tid & 3, o = blk * 4;
    X[(o + c) * DS + o + c] = R(1) / L[(o + c) * DS + o + c];
    for (int i = c + 1; i < 4; ++i) {
      R s = 0;
      for (int k = c; k < i; ++k) s += L[(o + i) * DS + o + k] * X[(o + k) * DS + o + c];
      X[(o + i) * DS + o + c] = -s / L[(o + i) * DS + o + i];
    }
  }
  __syncthreads();
  inv_level<4>(L, X, T, tid);
  inv_level<8>(L, X, T, tid);
  inv_level<16>(L, X, T, tid);
  inv_level<32>(L, X, T, tid);
}

template <typename R>
__global__ __launch_bounds__(256) void potrf128_kernel(R* __restrict__ Kall, R* __restrict__ Mall, R* __restrict__ MTall,
                                                       double* __restrict__ logdet, int32_t* __restrict__ status,
                                                       int64_t Npad, int nblk, int slot0, int o) {
  // LDS: L and X (64 x 65 each) + the compact inversion scratch: 76 KB in float64, so TWO workgroups share a CU
  // (round 1 kept a third 64 x 65 array, 100 KB: one workgroup per CU, two passes over a 512-fit batch).
  __shared__ R L[NB * DS];
  __shared__ R X[NB * DS];
  __shared__ R T[INV_SCRATCH];
  __shared__ R colbuf[2][NB];
  __shared__ int bad;
  const int b = slot0 + blockIdx.x;
  const int64_t mat = Npad * Npad;
  const R* K = Kall + (int64_t)b * mat;
  R* M = Mall + (int64_t)b * mat;
  R* MT = MTall + (int64_t)b * mat;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int64_t o1 = o, o2 = o + NB;
  if (tid == 0) bad = 0;

  auto stage_L_and_logdet = [&](R (&a)[4][4], bool anybad, int blk) {  // L -> LDS (upper zeroed), log-det partial
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = 4 * ty + r, jc = 4 * tx + c;
        L[i * DS + jc] = (jc <= i) ? a[r][c] : R(0);
      }
    if (anybad) bad = 1;  // (same value from every writer)
    __syncthreads();
    if (tid < 64) {
      double v = log((double)L[tid * DS + tid]);
#pragma unroll
      for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w, 64);
      if (tid == 0) logdet[(int64_t)b * nblk + blk] = v;
    }
  };
  auto write_inverse_block = [&](int64_t row0, int64_t col0) {  // X -> M[row0.., col0..] and its transpose into M'
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * ty + r;
      const R v[4] = {X[i * DS + 4 * tx], X[i * DS + 4 * tx + 1], X[i * DS + 4 * tx + 2], X[i * DS + 4 * tx + 3]};
      st4(M + (row0 + i) * Npad + col0 + 4 * tx, v);
      const R w[4] = {X[(4 * tx) * DS + i], X[(4 * tx + 1) * DS + i], X[(4 * tx + 2) * DS + i], X[(4 * tx + 3) * DS + i]};
      st4(MT + (col0 + i) * Npad + row0 + 4 * tx, w);
    }
  };

  // ---- A11: factor, invert -> M11 ----
  R a[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ld4(K + (o1 + 4 * ty + r) * Npad + o1 + 4 * tx, a[r]);
  for (int q = tid; q < NB * DS; q += 256) X[q] = 0;
  bool anybad = factor64<R>(a, colbuf, ty, tx);
  stage_L_and_logdet(a, anybad, o / NB);
  invert64<R>(L, X, T, tid);  // X = M11
  write_inverse_block(o1, o1);

  // ---- S = A21 M11'  (A21 staged in L - L11 is dead once M11 and its log-det exist; M11 lower: k <= column) ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    R v[4];
    ld4(K + (o2 + 4 * ty + r) * Npad + o1 + 4 * tx, v);
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(4 * ty + r) * DS + 4 * tx + c] = v[c];
  }
  __syncthreads();
  R sacc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sacc[r][c] = 0;
  for (int k = 0; k < 4 * tx + 4; ++k) {
    R ar[4], mc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ar[r] = L[(4 * ty + r) * DS + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) mc[c] = X[(4 * tx + c) * DS + k];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) sacc[r][c] += ar[r] * mc[c];
  }
  __syncthreads();  // every read of L (A21) is done
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(4 * ty + r) * DS + 4 * tx + c] = sacc[r][c];  // L now holds S
  __syncthreads();

  // ---- A22 -= S S'  and  T2 = S M11 (kept in registers) ----
#pragma unroll
  for (int r = 0; r < 4; ++r) ld4(K + (o2 + 4 * ty + r) * Npad + o2 + 4 * tx, a[r]);
  R t2[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) t2[r][c] = 0;
  for (int k = 0; k < NB; ++k) {
    R sr[4], sc[4], mk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sr[r] = L[(4 * ty + r) * DS + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) sc[c] = L[(4 * tx + c) * DS + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) mk[c] = X[k * DS + 4 * tx + c];  // M11[k][col] (zero above the diagonal)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a[r][c] -= sr[r] * sc[c];
        t2[r][c] += sr[r] * mk[c];
      }
  }
  __syncthreads();  // S and M11 are no longer needed in LDS

  // ---- Schur complement: factor, invert -> M22 ----
  for (int q = tid; q < NB * DS; q += 256) X[q] = 0;
  anybad = factor64<R>(a, colbuf, ty, tx);
  stage_L_and_logdet(a, anybad, o / NB + 1);
  invert64<R>(L, X, T, tid);  // X = M22
  write_inverse_block(o2, o2);
  if (tid == 0 && bad) status[b] = 1;

  // ---- M21 = -M22 T2  (T2 staged in L - L22 is dead after the inversion; M22 lower: k <= row) ----
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(4 * ty + r) * DS + 4 * tx + c] = t2[r][c];
  __syncthreads();
  R macc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) macc[r][c] = 0;
  for (int k = 0; k < 4 * ty + 4; ++k) {
    R mr[4], tc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mr[r] = X[(4 * ty + r) * DS + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) tc[c] = L[k * DS + 4 * tx + c];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) macc[r][c] -= mr[r] * tc[c];
  }
  // M21 -> M[(o2 + i)][o1 + j] and M'[(o1 + j)][o2 + i]
#pragma unroll
  for (int r = 0; r < 4; ++r) st4(M + (o2 + 4 * ty + r) * Npad + o1 + 4 * tx, macc[r]);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(4 * ty + r) * DS + 4 * tx + c] = macc[r][c];  // stage for the transposed store
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 4 * ty + r;  // row of M' block = column of M21
    const R w[4] = {L[(4 * tx) * DS + j], L[(4 * tx + 1) * DS + j], L[(4 * tx + 2) * DS + j], L[(4 * tx + 3) * DS + j]};
    st4(MT + (o1 + j) * Npad + o2 + 4 * tx, w);
  }
}

void launch_potrf128(Problem& p, hipStream_t st, int slot0, int nb, int o) {
  p.timer.begin(st, ST_POTRF_DIAG);
  if (p.f32)
    hipLaunchKernelGGL(potrf128_kernel<float>, dim3(nb), dim3(256), 0, st, p.Kbuf32, p.Mbuf32, p.MTbuf32, p.dlogdet,
                       p.dstatus, p.Npad, p.nblk(), slot0, o);
  else
    hipLaunchKernelGGL(potrf128_kernel<double>, dim3(nb), dim3(256), 0, st, p.Kbuf, p.Mbuf, p.MTbuf, p.dlogdet,
                       p.dstatus, p.Npad, p.nblk(), slot0, o);
  p.timer.end(st);
}

// ---------------------------------------------------------------------------------------
// residual r = y[col] - c (zero on padded rows); also clears the slot's status word
// ---------------------------------------------------------------------------------------
__global__ void resid_kernel(const DevSpec* __restrict__ specs, const int32_t* __restrict__ slot_spec,
                             const double* __restrict__ Y,
                             const int32_t* __restrict__ col, const double* __restrict__ theta,
                             double* __restrict__ r, int32_t* __restrict__ status, int64_t N,
                             int64_t Npad, int slot0) {
  const int b = slot0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad) return;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const double c = sp->mean_param >= 0 ? theta[(int64_t)b * sp->n_params + sp->mean_param] : 0.0;
  r[(int64_t)b * Npad + i] = (i < N) ? Y[(int64_t)col[b] * Npad + i] - c : 0.0;
  if (i == 0) status[b] = 0;
}

void launch_resid(Problem& p, hipStream_t st, int slot0, int nb) {
  dim3 grid((unsigned)((p.Npad + 255) / 256), nb);
  hipLaunchKernelGGL(resid_kernel, grid, dim3(256), 0, st, p.dspec, p.dslotspec, p.dY, p.dcol, p.dtheta, p.dr,
                     p.dstatus, p.N, p.Npad, slot0);
}

// z = M r  (M lower triangular). One wave per row, 4 rows per wave, 16 rows per workgroup.
template <typename R>
__global__ __launch_bounds__(256) void trmv_kernel(const R* __restrict__ Mall,
                                                   const double* __restrict__ rall,
                                                   double* __restrict__ zall, int64_t Npad,
                                                   int slot0) {
  const int b = slot0 + blockIdx.y;
  const R* M = Mall + (int64_t)b * Npad * Npad;
  const double* r = rall + (int64_t)b * Npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t i = (int64_t)blockIdx.x * 16 + wave * 4 + rr;
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
  hipLaunchKernelGGL(trmv_kernel<double>, dim3((unsigned)(p.Npad / 16), nb), dim3(256), 0, st, p.Mbuf, p.dr,
                     p.dz, p.Npad, slot0);
  if (need_alpha)
    hipLaunchKernelGGL(trmv_t_kernel<double>, dim3((unsigned)(p.Npad / 64), nb), dim3(256), 0, st, p.Mbuf,
                       p.dz, p.dalpha, p.Npad, slot0);
  p.timer.end(st);
  p.timer.work(ST_SOLVE, 0.0, (double)nb * 8.0 * (double)p.Npad * (p.Npad + 1) / 2.0 * (need_alpha ? 2 : 1));
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
                                                     int64_t Npad, int nblk, int nwg,
                                                     int out_stride, int slot0, int with_grad) {
  __shared__ double sh[4];
  const int b = slot0 + blockIdx.x, tid = threadIdx.x;
  const DevSpec* __restrict__ sp = specs + slot_spec[b];
  const int np = sp->n_params;
  double q = 0.0, sa = 0.0, ld = 0.0;
  for (int64_t i = tid; i < Npad; i += 256) {
    double zz = zall[(int64_t)b * Npad + i];
    q += zz * zz;
    if (with_grad) sa += aall[(int64_t)b * Npad + i];
  }
  for (int i = tid; i < nblk; i += 256) ld += logdet[(int64_t)b * nblk + i];
  q = block_sum(q, sh);
  ld = block_sum(ld, sh);
  sa = block_sum(sa, sh);
  double* o = out + (int64_t)b * out_stride;
  if (tid == 0) {
    o[0] = -0.5 * q - ld - 0.5 * (double)N * 1.8378770664093453;  // log(2 pi)
    o[1] = q;
    o[2] = ld;
  }
  if (with_grad) {
    for (int p = tid; p < np; p += 256) {
      double g = 0.0;
      for (int w = 0; w < nwg; ++w) g += part[((int64_t)b * nwg + w) * np + p];
      if (p == sp->mean_param) g = sa;
      o[3 + p] = g;
    }
  }
}

void launch_reduce(Problem& p, hipStream_t st, int slot0, int nb, bool with_grad) {
  p.timer.begin(st, ST_REDUCE);
  hipLaunchKernelGGL(reduce_kernel, dim3(nb), dim3(256), 0, st, p.dspec, p.dslotspec, p.dz, p.dalpha, p.dlogdet,
                     p.dpart, p.dout, p.N, p.Npad, p.nblk(), trace_nwg(p), p.out_stride, slot0,
                     with_grad ? 1 : 0);
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
// Tile width for a product with a rows x cols output over nb matrices: 128-wide tiles unless so few of them
// exist that most of the 512 resident workgroup slots would idle - then 64-wide tiles double the parallelism
// (the small-batch, large-N end: the tail of a job, a single fit).  WV_BN_FILL=0 restores 128 everywhere.
static int pick_bn(int64_t rows, int64_t cols, int nb, bool lower_only) {
  if (cols <= 64) return 64;
  static const int fill = [] { const char* e = getenv("WV_BN_FILL"); return e ? atoi(e) : 512; }();
  int64_t tiles = ((rows + 127) / 128) * ((cols + 127) / 128);
  if (lower_only) tiles = (tiles + (rows + 127) / 128) / 2;
  return tiles * nb < fill ? 64 : 128;
}

// ---- side lane: T = S M11 beside the critical path when few fits are in flight ----
static SideLane* side_lane(Problem& p, hipStream_t st, int nb) {
  static const int max_nb = [] { const char* e = getenv("WV_SIDE_T"); return e ? atoi(e) : 8; }();
  if (nb > max_nb || p.Npad < 512) return nullptr;  // full batches fill the GPU from one stream
  for (auto& l : p.lanes)
    if (l.main == st) return &l;
  SideLane l;
  l.main = st;
  if (hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (p.lanes.capacity() < 16) p.lanes.reserve(16);  // (pointers into the vector are held across the recursion)
  if (p.lanes.size() >= 16) {
    hipStreamDestroy(l.side);
    return nullptr;
  }
  p.lanes.push_back(l);
  return &p.lanes.back();
}
static hipEvent_t lane_event(SideLane& l) {
  if (l.next == l.events.size()) {
    hipEvent_t e;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    l.events.push_back(e);
  }
  return l.events[l.next++];
}
// the main stream must not overwrite the S scratch, or read T's output, before the side product is done
static void lane_join(SideLane* l) {
  if (l && l->pending) {
    hipStreamWaitEvent(l->main, l->pending, 0);
    l->pending = nullptr;
  }
}

template <typename R>
static void fi_node(Problem& p, hipStream_t st, int slot0, int nb, int64_t o, int64_t n) {
  if (n == NB) {
    launch_potrf_diag(p, st, slot0, nb, (int)o);
    return;
  }
  static const bool fuse128 = [] { const char* e = getenv("WV_LEAF128"); return e ? atoi(e) != 0 : true; }();
  if (n == 2 * NB && fuse128) {  // the whole node in one kernel
    launch_potrf128(p, st, slot0, nb, (int)o);
    return;
  }
  const int64_t Np = p.Npad, mat = Np * Np;
  const int64_t h1 = (n / NB / 2) * NB, h2 = n - h1;
  const MatBufs<R> mb = mat_bufs<R>(p);
  R* K0 = mb.K + (int64_t)slot0 * mat;
  R* M0 = mb.M + (int64_t)slot0 * mat;
  R* MT0 = mb.MT + (int64_t)slot0 * mat;
  R* S0 = mb.P + (int64_t)slot0 * p.pbuf_elems();  // holds S' (h1 x h2): K-strided image of S
  const int64_t sS = p.pbuf_elems();
  fi_node<R>(p, st, slot0, nb, o, h1);
  SideLane* lane = side_lane(p, st, nb);
  lane_join(lane);  // an ancestor's T may still be reading the S scratch this node is about to overwrite
  {  // S = A21 * M11'   written transposed;  B(n, k) = M11[n][k] = MT[k][n]  (K-strided)
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = K0 + (o + h1) * Np + o;
    a.B = MT0 + o * Np + o;
    a.C = nullptr;
    a.CT = S0; a.sCT = sS; a.ldct = (int)h2; a.tout = 1;
    a.M = (int)h2; a.N = (int)h1; a.kb = 0; a.ke = (int)h1; a.khi_mode = 2;
    p.timer.begin(st, ST_CHOL_TRSM);
    launch_gemm(st, a, true, false, pick_bn(h2, h1, nb, false));
    p.timer.end(st);
    p.timer.work(ST_CHOL_TRSM, nb * (double)h2 * h1 * (h1 + NB), 0.0);
  }
  hipStream_t st_t = st;  // the stream T runs on
  if (lane) {
    hipEvent_t s_done = lane_event(*lane);
    hipEventRecord(s_done, st);
    hipStreamWaitEvent(lane->side, s_done, 0);
    st_t = lane->side;
  }
  {  // A22 -= S S'   with A(m, k) = S'[k][m], B(n, k) = S'[k][n]
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = S0; a.B = S0; a.sA = a.sB = sS; a.lda = a.ldb = (int)h2;
    a.C = K0 + (o + h1) * Np + (o + h1);
    a.M = (int)h2; a.N = (int)h2; a.kb = 0; a.ke = (int)h1;
    a.alpha = -1.0; a.beta = 1.0; a.tri = 1; a.tag = 2;
    p.timer.begin(st, ST_CHOL_TRAIL);
    launch_gemm(st, a, false, false, pick_bn(h2, h2, nb, true));
    p.timer.end(st);
    p.timer.work(ST_CHOL_TRAIL, nb * (double)h1 * (double)h2 * (h2 + NB), 0.0);
  }
  {  // T = S * M11 over the dead A21 block;  A(m, k) = S'[k][m],  B(n, k) = M11[k][n]
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = S0; a.sA = sS; a.lda = (int)h2;
    a.B = M0 + o * Np + o;
    a.C = K0 + (o + h1) * Np + o;
    a.M = (int)h2; a.N = (int)h1; a.kb = 0; a.ke = (int)h1; a.klo_mode = 2;
    p.timer.begin(st_t, ST_TRTRI_A);
    launch_gemm(st_t, a, false, false, pick_bn(h2, h1, nb, false));
    p.timer.end(st_t);
    p.timer.work(ST_TRTRI_A, nb * (double)h2 * h1 * (h1 + NB), 0.0);
  }
  if (lane) {
    hipEvent_t t_done = lane_event(*lane);
    hipEventRecord(t_done, lane->side);
    lane->pending = t_done;  // joined by the next S that overwrites the scratch, at the latest before M21 below
  }
  fi_node<R>(p, st, slot0, nb, o + h1, h2);
  lane_join(lane);
  {  // M21 = -M22 * T  (also written transposed into MT);  A(m, k) = M22[m][k] = MT[k][m]
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = MT0 + (o + h1) * Np + (o + h1);
    a.B = K0 + (o + h1) * Np + o;  // B(n, k) = T[k][n]
    a.C = M0 + (o + h1) * Np + o;
    a.CT = MT0 + o * Np + (o + h1); a.sCT = mat; a.ldct = (int)Np; a.tout = 2;
    a.M = (int)h2; a.N = (int)h1; a.kb = 0; a.ke = (int)h2; a.khi_mode = 1; a.alpha = -1.0;
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
  static const bool fuse128 = [] { const char* e = getenv("WV_LEAF128"); return e ? atoi(e) != 0 : true; }();
  if (n == 2 * NB && fuse128) {  // fused node: M of the whole 128 block is available, one forward substitution
    launch_potrf128(p, st, slot0, nb, (int)o);
    p.timer.begin(st, ST_SOLVE);
    hipLaunchKernelGGL(trmv_sub_kernel<R>, dim3((unsigned)(2 * NB / 16), nb), dim3(256), 0, st, mb.M, p.dr, p.dz, Np, o,
                       (int64_t)(2 * NB), slot0);
    p.timer.end(st);
    return;
  }
  const int64_t h1 = (n / NB / 2) * NB, h2 = n - h1;
  R* K0 = mb.K + (int64_t)slot0 * mat;
  R* MT0 = mb.MT + (int64_t)slot0 * mat;
  R* S0 = mb.P + (int64_t)slot0 * p.pbuf_elems();
  const int64_t sS = p.pbuf_elems();
  fi_node<R>(p, st, slot0, nb, o, h1);
  p.timer.begin(st, ST_SOLVE);
  hipLaunchKernelGGL(trmv_sub_kernel<R>, dim3((unsigned)(h1 / 16), nb), dim3(256), 0, st, mb.M, p.dr, p.dz, Np, o, h1,
                     slot0);
  p.timer.end(st);
  {  // S = A21 * M11'   written transposed (same product as in fi_node)
    GemmArgsT<R> a = base_args<R>(p, slot0, nb);
    a.A = K0 + (o + h1) * Np + o;
    a.B = MT0 + o * Np + o;
    a.C = nullptr;
    a.CT = S0; a.sCT = sS; a.ldct = (int)h2; a.tout = 1;
    a.M = (int)h2; a.N = (int)h1; a.kb = 0; a.ke = (int)h1; a.khi_mode = 2;
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
    p.timer.begin(st, ST_CHOL_TRAIL);
    launch_gemm(st, a, false, false, pick_bn(h2, h2, nb, true));
    p.timer.end(st);
    p.timer.work(ST_CHOL_TRAIL, nb * (double)h1 * (double)h2 * (h2 + NB), 0.0);
  }
  fv_node<R>(p, st, slot0, nb, o + h1, h2);
}

void factor_value_only(Problem& p, hipStream_t st, int slot0, int nb) {
  if (SideLane* l = side_lane(p, st, nb)) l->next = 0;
  if (p.f32) fv_node<float>(p, st, slot0, nb, 0, p.Npad);
  else fv_node<double>(p, st, slot0, nb, 0, p.Npad);
}

void factor_inverse(Problem& p, hipStream_t st, int slot0, int nb) {
  if (SideLane* l = side_lane(p, st, nb)) l->next = 0;  // (events of the previous evaluation are all consumed)
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
  p.timer.begin(st, ST_LAUUM);
  // (always 128-wide tiles: one launch per evaluation, and it keeps this product on its own kernel symbol)
  launch_gemm(st, a, false, false, Np > 64 ? 128 : 64);
  p.timer.end(st);
  p.timer.work(ST_LAUUM, nb * (double)Np * Np * Np / 3.0, 0.0);
}

void lauum(Problem& p, hipStream_t st, int slot0, int nb) {
  if (p.f32) lauum_t<float>(p, st, slot0, nb);
  else lauum_t<double>(p, st, slot0, nb);
}

}  // namespace wv
