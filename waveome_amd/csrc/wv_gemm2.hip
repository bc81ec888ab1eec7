// Second-generation batched float64 GEMM for the large nodes of the factorisation and K_y^-1:
// 256 x 128 block tile, 8 waves (4 x 2, 64 x 64 per wave, v_mfma_f64_4x4x4_4b_f64 as in
// wv_gemm.hip), K chunk 16, THREE LDS stages filled by direct global -> LDS DMA
// (global_load_lds_dwordx4, no staging registers), one raw s_barrier per chunk and counted
// s_waitcnt vmcnt so two chunks stay in flight across the barrier.
//
// Why: rocprofv3 PMC on the register-staged 128 x 128 kernel showed 72-78 % MFMA-busy on full
// GEMMs but 51-60 % on the triangular / trimmed forms the factorisation needs - fewer tiles per
// operand panel -> more L2 misses per FLOP, and a single chunk of prefetch cannot cover them.
// This kernel moves 25 % fewer operand bytes per FLOP and prefetches two chunks deep.
//
// LDS images (no padding - an LDS-DMA instruction writes 64 lanes x 16 B linearly - the swizzle is
// applied to the per-lane SOURCE address and undone in the fragment read):
//   K-contiguous operand, ROWS x 16:  [s = k/4][row][4 k] with the two 16-B halves of a row swapped
//       when bit 3 of the row is set  -> a fragment read (16 rows x 2 k per 32 lanes) hits 32
//       distinct 8-byte bank pairs.
//   K-strided operand, 16 x ROWS:     [k][row] with 16-row groups pairwise swapped on odd k
//       -> lanes 0-15 (k even) and 16-31 (k odd) use disjoint bank halves.
// Rows past the operand's extent are clamped to the last valid row: they only feed C elements
// that the epilogue masks.
#include <cstdlib>

#include "wv_device.hpp"

namespace wv {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int BM2 = 256, BN2 = 128, KC2 = 16, NST = 3;
constexpr int A_ST = BM2 * KC2, B_ST = BN2 * KC2, STG = A_ST + B_ST;  // doubles per stage

// Per-lane source address (element offset from the operand base, at k0 = 0) and LDS destination
// (doubles from the operand's stage base) of DMA piece q.
template <int ROWS, bool KCONTIG>
__device__ inline void dma_piece(int q, int lane, int row0, int nrows, int ld, int64_t& src_off, int& dst) {
  if (KCONTIG) {
    constexpr int RG = ROWS / 32;
    const int s = q / RG, rg = q % RG;
    const int row = rg * 32 + (lane >> 1);
    const int half = (lane & 1) ^ ((row >> 3) & 1);
    const int grow = min(row0 + row, nrows - 1);
    src_off = (int64_t)grow * ld + 4 * s + 2 * half;
    dst = s * (ROWS * 4) + rg * 128;
  } else {
    constexpr int HR = ROWS / 128;
    const int k = q / HR, hr = q % HR;
    const int unit = lane ^ ((k & 1) << 3);
    const int grow = min(row0 + hr * 128 + 2 * unit, nrows - 2);
    src_off = (int64_t)k * ld + grow;
    dst = k * ROWS + hr * 128;
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm2_f64_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[NST * STG];

  const int tiles = g.tiles_m * g.tiles_n;  // (this kernel still launches the skipped tiles)
  int b, tile;
  if (g.nbatch >= 8) {
    int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    b = xcd + 8 * (q / tiles);
    tile = q % tiles;
    if (b >= g.nbatch) return;
  } else {
    b = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
  const int m0 = tm * BM2, n0 = tn * BN2;
  if (g.tri && n0 + g.tri_off > m0 + BM2 - 1) return;

  int kb = g.kb, ke = g.ke;
  if (g.klo_mode == 1) kb = max(kb, m0 + g.koff);
  if (g.klo_mode == 2) kb = max(kb, n0 + g.koff);
  if (g.khi_mode == 1) ke = min(ke, m0 + BM2 + g.koff);
  if (g.khi_mode == 2) ke = min(ke, n0 + BN2 + g.koff);
  const int nchunk = (ke - kb) / KC2;
  const int kdir = g.klo_mode ? -1 : 1;  // see wv_gemm.hip: trimmed-from-below tiles walk K downwards
  const int kfirst = g.klo_mode ? ke - KC2 : kb;

  const double* __restrict__ A = g.A + (int64_t)b * g.sA;
  const double* __restrict__ B = g.B + (int64_t)b * g.sB;
  double* __restrict__ C = g.C + (int64_t)b * g.sC;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4, fb = fk & 1;
  const int fblk = (lane >> 2) & 3, fj = lane & 3;
  int brot[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) brot[s] = 4 * ((fblk + s) & 3) + fj;

  // ---- DMA plan: 48 pieces per chunk (32 A + 16 B), 6 per wave ----
  const double* src[6];
  int dst[6];
  const int64_t astep = (int64_t)kdir * KC2 * (A_KC ? 1 : g.lda);
  const int64_t bstep = (int64_t)kdir * KC2 * (B_KC ? 1 : g.ldb);
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int p = wave + 8 * t;
    int64_t so;
    int d;
    if (t < 4) {  // pieces 0..31: A
      dma_piece<BM2, A_KC>(p, lane, m0, g.M, g.lda, so, d);
      src[t] = A + so + (int64_t)kfirst * (A_KC ? 1 : g.lda);
      dst[t] = d;
    } else {      // pieces 32..47: B
      dma_piece<BN2, B_KC>(p - 32, lane, n0, g.N, g.ldb, so, d);
      src[t] = B + so + (int64_t)kfirst * (B_KC ? 1 : g.ldb);
      dst[t] = A_ST + d;
    }
  }
  auto issue = [&](int stage) {
    double* base = lds + stage * STG;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      __builtin_amdgcn_global_load_lds((glb_void*)src[t], (lds_void*)(base + dst[t]), 16, 0, 0);
      src[t] += (t < 4) ? astep : bstep;
    }
  };

  // ---- fragment read offsets (doubles from the operand's stage base) ----
  // a_par[p] / b_par[r][p]: p = parity of the 16-row group index (only the K-strided image
  // distinguishes the two: groups are swapped pairwise on odd k).
  int a_par[2], b_par[4][2];
  if (A_KC) {
    a_par[0] = a_par[1] = (wm * 64 + fr) * 4 + (((fk >> 1) ^ ((fr >> 3) & 1)) << 1) + (fk & 1);
  } else {
    const int lp = fk * BM2 + (wm >> 1) * 128 + 2 * ((wm & 1) * 32 + (fr >> 1)) + (fr & 1);
    a_par[0] = lp + 16 * fb;
    a_par[1] = lp + 16 * (1 - fb);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (B_KC) {
      b_par[r][0] = b_par[r][1] =
          (wn * 64 + brot[r]) * 4 + (((fk >> 1) ^ ((brot[r] >> 3) & 1)) << 1) + (fk & 1);
    } else {
      const int lp = fk * BN2 + 2 * (wn * 32 + (brot[r] >> 1)) + (brot[r] & 1);
      b_par[r][0] = lp + 16 * fb;
      b_par[r][1] = lp + 16 * (1 - fb);
    }
  }
  // immediates: fragment i -> A_KC ? 64 i : 16 (i & ~1); k-step s -> A_KC ? s*BM2*4 : s*4*BM2 (same)
  double acc[4][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  if (nchunk > 0) issue(0);
  if (nchunk > 1) issue(1);
  int stage = 0;
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < nchunk) issue(stage >= 1 ? stage - 1 : NST - 1);  // (c + 2) % 3 == (stage + 2) % 3
    const double* __restrict__ sa = lds + stage * STG;
    const double* __restrict__ sb = sa + A_ST;
#pragma unroll
    for (int s = 0; s < KC2 / 4; ++s) {
      double af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = sa[a_par[i & 1] + (A_KC ? 64 * i : 16 * (i & ~1)) + s * (BM2 * 4)];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double bf = sb[b_par[r][j & 1] + (B_KC ? 64 * j : 16 * (j & ~1)) + s * (BN2 * 4)];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf, acc[i][j][r], 0, 0, 0);
        }
      }
    }
    stage = (stage == NST - 1) ? 0 : stage + 1;
  }

  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + 4 * fblk + fk;
    double cv[4][4];
    // read the whole strip first so the loads are in flight together, then write it
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + j * 16 + brot[r];
        cv[j][r] = (beta != 0.0 && m < g.M && n < g.N) ? C[(int64_t)m * g.ldc + n] : 0.0;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + j * 16 + brot[r];
        if (m < g.M && n < g.N) C[(int64_t)m * g.ldc + n] = alpha * acc[i][j][r] + beta * cv[j][r];
      }
  }
}

void launch_gemm2(hipStream_t st, const GemmArgs& a_in, bool a_kc, bool b_kc) {
  GemmArgs a = a_in;
  a.tiles_m = (a.M + BM2 - 1) / BM2;
  a.tiles_n = (a.N + BN2 - 1) / BN2;
  if (a.M <= 0 || a.N <= 0 || a.nbatch <= 0) return;
  int tiles = a.tiles_m * a.tiles_n;
  int grid = (a.nbatch >= 8) ? 8 * ((a.nbatch + 7) / 8) * tiles : a.nbatch * tiles;
  dim3 g(grid), blk(512);
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm2_f64_kernel<true, true>), g, blk, 0, st, a);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm2_f64_kernel<true, false>), g, blk, 0, st, a);
  else if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm2_f64_kernel<false, false>), g, blk, 0, st, a);
  else hipLaunchKernelGGL((gemm2_f64_kernel<false, true>), g, blk, 0, st, a);
}

}  // namespace wv
