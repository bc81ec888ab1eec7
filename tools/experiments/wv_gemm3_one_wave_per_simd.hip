// EXPERIMENT, NOT BUILT - a measured negative of round 3 ("one wave per SIMD").
// Kept as source so that the next attempt starts from working code: it was correct (bit-identical to the 128 x 128
// kernels on integer operands and through the whole evaluation pipeline) and slower:
//   version 1 (barrier + DMA issue between iterations, MFMAs in place):  full 1024^3 x 256  56.4 TFLOP/s vs 62.1,
//             T-like 44.6 vs 53.2, M'M 2048 51.6 vs 51.9                       (profiles/r03_probe_big_v1_*.txt)
//   version 2 (this file: DMA issue inside the MFMA block): hipcc's register allocation breaks down with the 256
//             AGPRs exactly full (188 v_accvgpr moves and 100 s_nop per iteration): full 1024^3 47.0  (.._v2_on.txt)
// although the K loop alone sustains 74.6 TFLOP/s in this shape (tools/probe_occ1_loop.hip).  With one workgroup per
// CU every bubble is exposed (DMA issue, barrier skew, the epilogue); the two-workgroups-per-CU kernel hides them behind
// the other workgroup.  To route launches here again: add the file to the Makefile, declare gemm_big_eligible /
// launch_gemm_big in wv_device.hpp, and see commit history of wv_gemm.hip (tri code 4/5 split).
// The float64 MFMA GEMM in its one-wave-per-SIMD shape, for the large K-strided x K-strided products of an evaluation
// (K_y^-1 = M'M, the trailing update A22 -= S S', T = S M11, M21 = -M22 T at the upper levels of the recursion).
//
// Same arithmetic as wv_gemm.hip (v_mfma_f64_4x4x4_4b_f64, the same fragment maps, the contraction walked in the same
// chunks of 16 in the same order - results are bit-identical), different machine shape:
//   * block tile 256 x 128, 4 waves (one per SIMD, up to 512 registers per lane: the 256 accumulator registers of a
//     128 x 64 wave tile live in AGPRs) - 24 LDS fragment reads per 128 MFMAs instead of 20 per 64;
//   * the next k-step's fragments are fetched while this one's MFMAs run, the reads spread between the MFMAs with
//     sched_group_barrier - a single wave per SIMD hides its own LDS latency (tools/probe_occ1_loop.hip: the K loop alone
//     sustains 74.6 TFLOP/s in this shape against 68.5 in the two-workgroups-per-CU shape);
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of
//     THREE stages, issued two to three chunks ahead; one s_barrier per chunk, counted s_waitcnt vmcnt.
// LDS image of a K-strided operand, [k][row] without padding (an LDS-DMA instruction writes 64 lanes x 16 B
// linearly): the 16-row groups are swapped pairwise on odd k - applied to the per-lane SOURCE address, undone in the
// fragment read - so that lanes 0-15 (k even) and 16-31 (k odd) of a fragment read use disjoint bank halves.
// Rows past the operand's extent are clamped: they only feed C elements the epilogue masks.
#include <algorithm>
#include <cstdlib>

#include "wv_device.hpp"

namespace wv {
namespace big {
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int BM = 256, BN = 128, KC = 16, NST = 3;
constexpr int A_ST = BM * KC, B_ST = BN * KC, STG = A_ST + B_ST;  // doubles per stage (48 KB)
constexpr int LDC = BN + 2;                                        // epilogue staging image [128][LDC]
static_assert(128 * LDC <= NST * STG, "the epilogue stages 128 rows of C in the K loop's LDS");
}  // namespace big

// TOUT as in wv_gemm.hip: 0 = write C, 2 = C and C' (g.CT).  TAG only names the symbol (1 = K_y^-1 = M'M, 2 = trailing
// update).  g.tri: 0 = every tile, 5 = the tiles strictly below the diagonal 256 x 256 blocks of a square lower-only
// product (tile row tm holds tiles tn < 2 tm).
template <int TOUT, int TAG>
__global__ __launch_bounds__(256, 1) void gemm_f64_big_kernel(GemmArgs g) {
  using namespace big;
  __shared__ __attribute__((aligned(16))) double lds[NST * STG];

  const int tiles = g.tiles_live;
  int b, tile;
  if (g.nbatch >= 8) {  // same-matrix tiles on one XCD (see wv_gemm.hip)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    b = xcd + 8 * (q / tiles);
    tile = q % tiles;
    if (b >= g.nbatch) return;
  } else {
    b = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  int tm, tn;
  if (g.tri == 5) {
    tm = 1;  // (ascending tile rows: for M'M the longest contractions first)
    while (tile >= 2 * tm) { tile -= 2 * tm; ++tm; }
    tn = tile;
  } else {
    if (g.khi_mode && !g.tri) tile = tiles - 1 - tile;
    tm = tile / g.tiles_n;
    tn = tile % g.tiles_n;
    if (g.khi_mode == 2 || g.klo_mode == 2) tn = (tn + tm) % g.tiles_n;  // Latin-square skew (wv_gemm.hip)
  }
  const int m0 = tm * BM, n0 = tn * BN;

  int kb = g.kb, ke = g.ke;
  if (g.klo_mode == 1) kb = max(kb, m0 + g.koff);
  if (g.klo_mode == 2) kb = max(kb, n0 + g.koff);
  if (g.khi_mode == 1) ke = min(ke, m0 + BM + g.koff);
  if (g.khi_mode == 2) ke = min(ke, n0 + BN + g.koff);
  const int nchunk = max(0, (ke - kb) / KC);
  const int kdir = g.klo_mode ? -1 : 1;  // trimmed-from-below tiles walk K downwards from the common end
  const int kfirst = g.klo_mode ? ke - KC : kb;

  const double* __restrict__ A = g.A + (int64_t)b * g.sA;
  const double* __restrict__ B = g.B + (int64_t)b * g.sB;
  double* __restrict__ C = g.C + (int64_t)b * g.sC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: tell the compiler, so
  const int wm = wave >> 1, wn = wave & 1;                    // that the per-wave branches below are scalar branches
  const int fr = lane & 15, fk = lane >> 4, fb = fk & 1;
  const int fblk = (lane >> 2) & 3, fj = lane & 3;
  int brot[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) brot[s] = 4 * ((fblk + s) & 3) + fj;

  // ---- LDS-DMA plan: per chunk 32 A pieces (k, 128-row half) + 16 B pieces (k), 1 KB each, 12 per wave.
  // Wave w moves A pieces (k = (w >> 1) + 2 t, half w & 1), t < 8, and B pieces k = w + 4 t, t < 4: the k parity - and
  // with it the swizzle - is fixed per wave, so ONE per-lane source offset serves all of a wave's A pieces, one its B's.
  const int hrA = wave & 1, kA0 = wave >> 1, kB0 = wave;
  const int unitA = lane ^ ((kA0 & 1) << 3), unitB = lane ^ ((kB0 & 1) << 3);
  const int rowA = min(m0 + hrA * 128 + 2 * unitA, g.M - 2);
  const int rowB = min(n0 + 2 * unitB, g.N - 2);
  const double* srcA = A + (int64_t)(kfirst + kA0) * g.lda + rowA;  // piece t: + 2 t lda
  const double* srcB = B + (int64_t)(kfirst + kB0) * g.ldb + rowB;  // piece t: + 4 t ldb
  const int64_t stepA = (int64_t)kdir * KC * g.lda, stepB = (int64_t)kdir * KC * g.ldb;
  const int lda2 = 2 * g.lda, ldb4 = 4 * g.ldb;
  auto issue = [&](int stage) {
    double* base = lds + stage * STG;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcA + (int64_t)t * lda2),
                                       (lds_void*)(base + (kA0 + 2 * t) * BM + hrA * 128), 16, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcB + (int64_t)t * ldb4),
                                       (lds_void*)(base + A_ST + (kB0 + 4 * t) * BN), 16, 0, 0);
    srcA += stepA;
    srcB += stepB;
  };

  // ---- fragment read offsets (doubles from the stage base); [.. & 1]: parity of the 16-row group (swapped on odd k)
  int a_par[2], b_par[4][2];
  {
    const int lp = fk * BM + wm * 128 + fr;
    a_par[0] = lp + 16 * fb;
    a_par[1] = lp + 16 * (1 - fb);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int lp = A_ST + fk * BN + wn * 64 + brot[r];
    b_par[r][0] = lp + 16 * fb;
    b_par[r][1] = lp + 16 * (1 - fb);
  }

  double acc[8][4][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  double af[2][8], bf[2][4][4];
  auto fetch = [&](int buf, const double* __restrict__ st, int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) af[buf][i] = st[a_par[i & 1] + 16 * (i & ~1) + s * (4 * BM)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bf[buf][j][r] = st[b_par[r][j & 1] + 16 * (j & ~1) + s * (4 * BN)];
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[buf][i], bf[buf][j][r], acc[i][j][r], 0, 0, 0);
  };
  // Ring discipline (three stages).  Invariant at the top of iteration c: chunks c and c + 1 have landed and are visible
  // to every wave, and every wave is past its last LDS read of chunk c - 1.  The iteration then is ONE basic block of
  // 512 MFMAs per wave (the 256 accumulator registers never move) into which the scheduler spreads
  //   * the 12 LDS-DMA instructions of chunk c + 2 (into the stage chunk c - 1 just left) and their address arithmetic,
  //   * the 48 ds_read2_b64 of the fragments of the following k-steps (the last k-step prefetches chunk c + 1's first),
  // followed by s_waitcnt vmcnt(0) - the DMA issued ~8000 cycles earlier - and one s_barrier.  Past the last chunk the
  // DMA re-reads the last chunk's rows into a stage nobody reads again (no branch in the block).
  // (No per-wave skipping of chunks in which a triangular operand is all zero for a wave's rows, as wv_gemm.hip does:
  // with one wave per SIMD a skipped chunk frees nothing - the other waves of the workgroup still run it - and the
  // zeros contribute exact zeros, so the bits are the same.)
  if (nchunk > 0) {
    issue(0);
    if (nchunk < 2) { srcA -= stepA; srcB -= stepB; }
    issue(1);
    if (nchunk < 3) { srcA -= stepA; srcB -= stepB; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fetch(0, lds, 0);
  }
  int stage = 0;
  for (int c = 0; c < nchunk; ++c) {
    const int nstage = stage == NST - 1 ? 0 : stage + 1;
    const int pstage = stage == 0 ? NST - 1 : stage - 1;  // = (c + 2) % 3
    const double* __restrict__ cur = lds + stage * STG;
    const double* __restrict__ nxt = lds + nstage * STG;
    issue(pstage);                                         // chunk c + 2 (past the end: the last chunk's rows again)
    const int64_t back = (c + 4 > nchunk) ? 1 : 0;         // uniform: no branch
    srcA -= back * stepA;
    srcB -= back * stepB;
    fetch(1, cur, 1); mma(0);
    fetch(0, cur, 2); mma(1);
    fetch(1, cur, 3); mma(0);
    fetch(0, nxt, 0); mma(1);
    // 12 DMA among the first MFMAs, then one ds_read2_b64 after every 10 MFMAs
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int q = 0; q < 36; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage = nstage;
  }

  // ---- epilogue: C = beta C + alpha acc, 128 rows at a time through LDS so that C moves as full 1 KB rows.
  // D lane 16 i' + 4 blk + jj of accumulator (i, j, r) is row 16 i + 4 blk + i', column 16 j + 4((blk + r) & 3) + jj of
  // the wave tile (rows 128 wm .., columns 64 wn ..). ----
  const double alpha = g.alpha, beta = g.beta;
  double* smem = lds;
  constexpr int PAIRS = BN / 2, NIT = 128 * PAIRS / 256;  // 32 row pieces of 16 B per thread and pass
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the ring's last, unused DMA must have landed before LDS is reused)
  __syncthreads();
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    d2 cv[NIT];
    if (beta != 0.0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = tid + 256 * it, row = q / PAIRS, cp = q % PAIRS;
        const int m = m0 + 128 * h + row, n = n0 + 2 * cp;
        cv[it] = (m < g.M && n < g.N) ? *reinterpret_cast<const d2*>(C + (int64_t)m * g.ldc + n) : d2{0.0, 0.0};
      }
    }
    if (wm == h) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            smem[(i * 16 + 4 * fblk + fk) * LDC + wn * 64 + j * 16 + brot[r]] = alpha * acc[i][j][r];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = tid + 256 * it, row = q / PAIRS, cp = q % PAIRS;
      const int m = m0 + 128 * h + row, n = n0 + 2 * cp;
      if (m < g.M && n < g.N) {
        d2 v = *reinterpret_cast<const d2*>(smem + row * LDC + 2 * cp);
        if (beta != 0.0) {
          v.x += beta * cv[it].x;
          v.y += beta * cv[it].y;
        }
        *reinterpret_cast<d2*>(C + (int64_t)m * g.ldc + n) = v;
      }
    }
    if (TOUT == 2) {  // transposed copy: thread -> (column n, row pair)
      double* __restrict__ CT = g.CT + (int64_t)b * g.sCT;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = tid + 256 * it, n = q >> 6, mp = q & 63;
        const int m = m0 + 128 * h + 2 * mp, nn = n0 + n;
        if (m < g.M && nn < g.N) {
          d2 v;
          v.x = smem[(2 * mp) * LDC + n];
          v.y = smem[(2 * mp + 1) * LDC + n];
          if (beta != 0.0) {  // (not used by the schedule: M21 is a beta = 0 product)
            v.x += beta * C[(int64_t)m * g.ldc + nn];
            v.y += beta * C[(int64_t)(m + 1) * g.ldc + nn];
          }
          *reinterpret_cast<d2*>(CT + (int64_t)nn * g.ldct + m) = v;
        }
      }
    }
    __syncthreads();
  }
}

// true when the 256 x 128 kernel can run this K-strided x K-strided product (the caller then routes it here)
bool gemm_big_eligible(const GemmArgs& a, bool a_kc, bool b_kc) {
  if (a_kc || b_kc || a.spans) return false;
  if (a.tout != 0 && a.tout != 2) return false;
  if (a.M % big::BM != 0 || a.N % big::BN != 0) return false;
  if ((a.ke - a.kb) % big::KC != 0 || a.kb % big::KC != 0 || a.koff % big::KC != 0) return false;
  if (a.lda % 2 != 0 || a.ldb % 2 != 0) return false;  // 16-byte LDS-DMA pieces
  return true;
}

// tri_mode 0: all tiles; 5: tiles strictly below the diagonal 256-blocks of a square lower-only product
void launch_gemm_big(hipStream_t st, const GemmArgs& a_in, int tri_mode) {
  GemmArgs a = a_in;
  a.tiles_m = a.M / big::BM;
  a.tiles_n = a.N / big::BN;
  a.tri = tri_mode;
  int tiles = a.tiles_m * a.tiles_n;
  if (tri_mode == 5) tiles = a.tiles_m * (a.tiles_m - 1);
  a.tiles_live = tiles;
  if (tiles <= 0 || a.nbatch <= 0) return;
  const dim3 g((a.nbatch >= 8) ? 8 * ((a.nbatch + 7) / 8) * tiles : a.nbatch * tiles), blk(256);
  if (a.tout == 2) hipLaunchKernelGGL((gemm_f64_big_kernel<2, 0>), g, blk, 0, st, a);
  else if (a.tag == 1) hipLaunchKernelGGL((gemm_f64_big_kernel<0, 1>), g, blk, 0, st, a);
  else if (a.tag == 2) hipLaunchKernelGGL((gemm_f64_big_kernel<0, 2>), g, blk, 0, st, a);
  else hipLaunchKernelGGL((gemm_f64_big_kernel<0, 0>), g, blk, 0, st, a);
}

}  // namespace wv
