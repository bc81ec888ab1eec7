// Batched float32 GEMM family on v_mfma_f32_16x16x4_f32 (gfx950): the float32-storage twin of
// wv_gemm.hip, serving the same products of the factor-and-invert schedule (wv_chol.hip) when a
// problem was created on a float32 context (the reference's `set_precision("float32")`,
// waveome/utilities.py:34-49; BASELINE config C5).  Same GemmArgs semantics: per-tile contraction
// trimming (klo_mode / khi_mode), lower-only tile enumeration (tri), Latin-square skew of the tile
// order, transposed outputs (tout), K walked downwards for tiles trimmed from below.
//
// v_mfma_f32_16x16x4_f32: 8 passes (32 cycles), 2048 FLOP -> 256 FLOP / clk / CU, the dense fp32
// matrix rate (157 TFLOP/s at 2.4 GHz).  Lane maps (lane l, fr = l & 15, fk = l >> 4):
//   A: lane holds A[row fr][k fk]      B: lane holds B[k fk][col fr]
//   D: register r of lane holds D[row 4 fk + r][col fr]
// i.e. the operand fragments are the ones the float64 kernel reads (row l & 15, k = l >> 4), one MFMA
// covers the whole 16 x 16 block, so there is no rotation of B fragments.
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 (4 x 4 MFMA blocks,
// 64 accumulator VGPRs), K chunk 32, global -> registers -> LDS double-buffered, one barrier per chunk.
// LDS images, conflict-free for the ds_read_b32 fragment reads (64 lanes = 16 rows x 4 k):
//   K-contiguous operand  [rows][32 + 4]   row stride 36 floats: 36 r mod 64 = the 16 multiples of 4
//   K-strided   operand   [32][rows + 16]  row stride 144 floats = 16 mod 64
// B is always K-strided (every product of the schedule reads B by rows of K).
#include <cstdio>
#include <cstdlib>

#include "wv_device.hpp"

namespace wv {

typedef float f4 __attribute__((ext_vector_type(4)));

namespace g32 {
constexpr int BM = 128, BN = 128, KC = 32;

template <bool KCONTIG>
struct Tile {
  static constexpr int ROWS = 128;
  static constexpr int LD = KCONTIG ? (KC + 4) : (ROWS + 16);
  static constexpr int SIZE = KCONTIG ? ROWS * LD : KC * LD;
  static constexpr int NLOAD = ROWS * KC / 4 / 256;  // float4 loads per thread per chunk (= 4)
  __device__ static inline int at(int row, int k) { return KCONTIG ? row * LD + k : k * LD + row; }

  // A chunk's NLOAD loads of one thread are "uniform piece base + ONE per-thread byte offset": piece i of a K-contiguous
  // operand is its rows 32 i .. 32 i + 31, of a K-strided one its k rows 8 i .. 8 i + 7.  The offset is computed once per
  // tile; the piece bases are scalar arithmetic per chunk - no per-lane 64-bit addresses, masks or zero fill (with the
  // float64 partial sums of the accumulator flush this kernel sits at its 256-register limit: the per-chunk addresses
  // were what spilled, and a reload of a spill waits on the prefetch in flight).  Rows past the operand's extent (extents
  // are multiples of 64: launch_gemm checks) are redirected to valid rows: they only feed C elements the epilogue masks.
  __device__ static inline uint32_t goffset(int ld, int row0, int nrows, int tid) {
    if (KCONTIG) return ((uint32_t)(tid >> 3) * (uint32_t)ld + 4u * (tid & 7)) * 4u;
    return ((uint32_t)(tid >> 5) * (uint32_t)ld + (uint32_t)min(4 * (tid & 31), nrows - 4 - row0)) * 4u;
  }
  __device__ static inline void gload(f4 (&reg)[NLOAD], const float* __restrict__ base, int ld, int row0, int nrows,
                                      int k0, uint32_t off) {
    // (buffer loads: the descriptor - four scalar registers - holds the chunk's 64-bit base, the piece offset is the
    //  instruction's scalar offset, the lane's part its 32-bit vector offset)
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const float* cbase = base + (KCONTIG ? (int64_t)row0 * ld + k0 : (int64_t)k0 * ld + row0);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cbase), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int piece = KCONTIG ? (row0 + 32 * i < nrows ? 32 * i : 0) : 8 * i;  // (uniform)
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, piece * ld * 4, 0);
      reg[i] = __builtin_bit_cast(f4, v);
    }
  }
  __device__ static inline void sstore(float* __restrict__ s, const f4 (&reg)[NLOAD], int tid) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int q = tid + 256 * i;
      if (KCONTIG) {
        const int row = q >> 3, kp = q & 7;
        *reinterpret_cast<f4*>(s + row * LD + 4 * kp) = reg[i];
      } else {
        const int k = q >> 5, rp = q & 31;
        *reinterpret_cast<f4*>(s + k * LD + 4 * rp) = reg[i];
      }
    }
  }
};
}  // namespace g32

// TOUT: 0 = write C; 1 = write only C' (g.CT, leading dimension g.ldct); 2 = both.  TAG = 1 gives the
// K_y^-1 = M'M launch its own symbol for profilers.
template <bool A_KC, int TOUT, int TAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgsF g) {
  using namespace g32;
  using TA = Tile<A_KC>;
  using TB = Tile<false>;
  constexpr int LDC = BN + 4;  // epilogue staging image [64][LDC] of half the C tile
  constexpr int SMEM = 2 * (TA::SIZE + TB::SIZE) > 64 * LDC ? 2 * (TA::SIZE + TB::SIZE) : 64 * LDC;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float(*sA)[TA::SIZE] = reinterpret_cast<float(*)[TA::SIZE]>(smem);
  float(*sB)[TB::SIZE] = reinterpret_cast<float(*)[TB::SIZE]>(smem + 2 * TA::SIZE);

  // ---- block -> (batch, tile): same-matrix tiles on one XCD when there are >= 8 matrices ----
  const int tiles = g.tiles_live;
  int b, tile;
  if (g.nbatch >= 8) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    b = xcd + 8 * (q / tiles);
    tile = q % tiles;
    if (b >= g.nbatch) return;
  } else {
    b = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  if (g.khi_mode && !g.tri) tile = tiles - 1 - tile;  // longest contraction first
  int tm, tn;
  if (g.tri) {
    tm = 0;
    for (;;) {
      const int cnt = min(g.tiles_n, max(0, (tm * BM + BM - 1 - g.tri_off) / BN + 1));
      if (tile < cnt) break;
      tile -= cnt;
      ++tm;
    }
    tn = tile;
  } else {
    tm = tile / g.tiles_n;
    tn = tile % g.tiles_n;
    if (g.khi_mode == 2 || g.klo_mode == 2) {  // column-dependent contraction lengths: column-major (see wv_gemm.hip)
      tn = tile / g.tiles_m;
      tm = tile % g.tiles_m;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;

  int kb = g.kb, ke = g.ke;
  if (g.klo_mode == 1) kb = max(kb, m0 + g.koff);
  if (g.klo_mode == 2) kb = max(kb, n0 + g.koff);
  if (g.khi_mode == 1) ke = min(ke, m0 + BM + g.koff);
  if (g.khi_mode == 2) ke = min(ke, n0 + BN + g.koff);
  // Block structure (GemmArgsT::spans, as in wv_gemm.hip): only the k that can meet a non-zero of both operands
  const int32_t* span_tab = nullptr;
  if (g.spans) {
    const int t = g.spec_span[g.slot_spec[g.slot0 + b]];
    if (t >= 0) {
      span_tab = g.spans + (int64_t)t * g.span_nblk * 2;
      int lo_r = 1 << 30, hi_r = 0, lo_c = 1 << 30, hi_c = 0;
      const int gr0 = (g.sp_r0 + m0) >> 6, gr1 = (g.sp_r0 + min(m0 + BM, g.M) - 1) >> 6;
      const int gc0 = (g.sp_c0 + n0) >> 6, gc1 = (g.sp_c0 + min(n0 + BN, g.N) - 1) >> 6;
      for (int q = gr0; q <= gr1; ++q) { lo_r = min(lo_r, span_tab[2 * q]); hi_r = max(hi_r, span_tab[2 * q + 1]); }
      for (int q = gc0; q <= gc1; ++q) { lo_c = min(lo_c, span_tab[2 * q]); hi_c = max(hi_c, span_tab[2 * q + 1]); }
      kb = max(kb, max(lo_r, lo_c) - g.sp_k0);
      ke = max(kb, min(ke, min(hi_r, hi_c) - g.sp_k0));
      if (ke == kb && g.beta == 1.0 && TOUT == 0) return;  // C += 0
    }
  }
  // chunks are 32 wide while operand extents are multiples of 64: round the range outwards onto the
  // chunk grid anchored at g.kb (the extra rows are zeros of the triangular operand or inside [g.kb, g.ke))
  kb = g.kb + (kb - g.kb) / KC * KC;
  ke = min(g.ke, g.kb + (ke - g.kb + KC - 1) / KC * KC);

  const float* __restrict__ A = g.A + (int64_t)b * g.sA;
  const float* __restrict__ B = g.B + (int64_t)b * g.sB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  // Long contractions in float32 lose digits to the accumulation itself: 2048 dependent roundings at K = 8192, each of the
  // size of the running sum - that, times the condition number, is what the float32 factor's error E is made of
  // (tools/probe_f32_budget.py: |E|_F^2 = 0.46 at N = 16384, cond 6e4; 5e-4 with this).  So the MFMA accumulators are only
  // ever FLUSH chunks (512 k, 128 roundings) old: after that they are added into float64 sums and cleared - 64 conversions
  // + 64 float64 adds per 2048 MFMAs; the product leaves the kernel rounded to float32 once.  Measured at C5 (8 fits,
  // N = 16384, profiles/r05_f32_sweep.txt): FLUSH 4 / 8 / 16 cost 12 / 5 / 1 ms of 356 and leave the corrected LML 2e-8 /
  // 4e-8 / 8e-8 from float64 (2e-5 without).
  constexpr int FLUSH = 16;
  double dacc[4][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) dacc[i][j][r] = 0.0;
  auto flush = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dacc[i][j][r] += (double)acc[i][j][r];
        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
      }
  };

  // per-wave contraction range at 64-row / 64-column granularity (triangular operands, lower-only products)
  int wkb = kb, wke = ke;
  {
    const int wr = m0 + wm * 64, wc = n0 + wn * 64;
    if (g.klo_mode == 1) wkb = max(wkb, wr + g.koff);
    if (g.klo_mode == 2) wkb = max(wkb, wc + g.koff);
    if (g.khi_mode == 1) wke = min(wke, wr + 64 + g.koff);
    if (g.khi_mode == 2) wke = min(wke, wc + 64 + g.koff);
    if (g.tri && wc + g.tri_off > wr + 63) wke = wkb;
    if (span_tab) {  // the wave's own 64 rows and 64 columns
      const int qr = (g.sp_r0 + min(wr, g.M - 1)) >> 6, qc = (g.sp_c0 + min(wc, g.N - 1)) >> 6;
      wkb = max(wkb, max(span_tab[2 * qr], span_tab[2 * qc]) - g.sp_k0);
      wke = min(wke, min(span_tab[2 * qr + 1], span_tab[2 * qc + 1]) - g.sp_k0);
    }
  }
  f4 ra[TA::NLOAD], rb[TB::NLOAD];
  const int nchunk = (ke - kb + KC - 1) / KC;
  const int kstep = g.klo_mode ? -KC : KC;  // tiles trimmed from below walk K downwards from the common end
  kb = g.klo_mode ? ke - KC : kb;
  const uint32_t offa = TA::goffset(g.lda, m0, g.M, tid), offb = TB::goffset(g.ldb, n0, g.N, tid);
  if (nchunk > 0) {
    TA::gload(ra, A, g.lda, m0, g.M, kb, offa);
    TB::gload(rb, B, g.ldb, n0, g.N, kb, offb);
    TA::sstore(sA[0], ra, tid);
    TB::sstore(sB[0], rb, tid);
  }
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunk) {
      TA::gload(ra, A, g.lda, m0, g.M, kb + (c + 1) * kstep, offa);
      TB::gload(rb, B, g.ldb, n0, g.N, kb + (c + 1) * kstep, offb);
    }
    const float* __restrict__ a_s = sA[cur];
    const float* __restrict__ b_s = sB[cur];
    const int kc0 = kb + c * kstep;
    if (kc0 + KC > wkb && kc0 < wke) {
#pragma unroll
      for (int s = 0; s < KC / 4; ++s) {
        float af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = a_s[TA::at(wm * 64 + i * 16 + fr, s * 4 + fk)];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = b_s[TB::at(wn * 64 + j * 16 + fr, s * 4 + fk)];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (c + 1 < nchunk) {
      TA::sstore(sA[cur ^ 1], ra, tid);
      TB::sstore(sB[cur ^ 1], rb, tid);
    }
    // (flush points sit on the absolute chunk grid: a contraction range trimmed by block spans - exact zeros left out -
    //  keeps the same partial sums, so block and dense products agree bit for bit)
    if ((((kc0 - g.kb) / KC) & (FLUSH - 1)) == (kstep > 0 ? FLUSH - 1 : 0)) flush();
    __syncthreads();
  }
  flush();

  // ---- epilogue: C = beta C + alpha acc, staged through LDS so that C moves as full rows ----
  float* __restrict__ C = g.C + (int64_t)b * g.sC;
  const float beta = (float)g.beta;
  constexpr int QUADS = BN / 4;             // float4 per row
  constexpr int NROWIT = 64 * QUADS / 256;  // = 8
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f4 cv[NROWIT];
    if (TOUT != 1 && beta != 0.f) {
#pragma unroll
      for (int it = 0; it < NROWIT; ++it) {
        const int q = tid + 256 * it, row = q / QUADS, cp = q % QUADS;
        const int m = m0 + 64 * h + row, n = n0 + 4 * cp;
        cv[it] = (m < g.M && n < g.N) ? *reinterpret_cast<const f4*>(C + (int64_t)m * g.ldc + n) : f4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (wm == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            smem[(i * 16 + 4 * fk + r) * LDC + wn * 64 + j * 16 + fr] = (float)(g.alpha * dacc[i][j][r]);
    }
    __syncthreads();
    if (TOUT != 1) {
#pragma unroll
      for (int it = 0; it < NROWIT; ++it) {
        const int q = tid + 256 * it, row = q / QUADS, cp = q % QUADS;
        const int m = m0 + 64 * h + row, n = n0 + 4 * cp;
        if (m < g.M && n < g.N) {
          f4 v = *reinterpret_cast<const f4*>(smem + row * LDC + 4 * cp);
          if (beta != 0.f) v += beta * cv[it];
          *reinterpret_cast<f4*>(C + (int64_t)m * g.ldc + n) = v;
        }
      }
    }
    if (TOUT != 0) {
      // transposed copy: thread -> (column n, 4 consecutive rows), 16 lanes x 16 B per row of C'
      float* __restrict__ CT = g.CT + (int64_t)b * g.sCT;
#pragma unroll
      for (int it = 0; it < NROWIT; ++it) {
        const int q = tid + 256 * it, n = q >> 4, mp = q & 15;
        const int m = m0 + 64 * h + 4 * mp, nn = n0 + n;
        if (m < g.M && nn < g.N) {
          f4 v;
          v.x = smem[(4 * mp) * LDC + n];
          v.y = smem[(4 * mp + 1) * LDC + n];
          v.z = smem[(4 * mp + 2) * LDC + n];
          v.w = smem[(4 * mp + 3) * LDC + n];
          *reinterpret_cast<f4*>(CT + (int64_t)nn * g.ldct + m) = v;
        }
      }
    }
    __syncthreads();
  }
}

void launch_gemm(hipStream_t st, const GemmArgsF& a_in, bool a_kc, bool b_kc, int /*bn*/) {
  using namespace g32;
  GemmArgsF a = a_in;
  if (b_kc) {
    fail_launch("launch_gemm(float32): K-contiguous B is not used by the schedule and not built");
    return;
  }
  if (a.M <= 0 || a.N <= 0 || a.nbatch <= 0) return;
  if ((a.M | a.N) & 63) {  // (the kernel's operand loads redirect whole 32-row pieces past the extent)
    fail_launch("launch_gemm(float32): extents must be multiples of 64");
    return;
  }
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  int tiles = a.tiles_m * a.tiles_n;
  if (a.tri) {
    tiles = 0;
    for (int tm = 0; tm < a.tiles_m; ++tm)
      tiles += std::min(a.tiles_n, std::max(0, (tm * BM + BM - 1 - a.tri_off) / BN + 1));
  }
  a.tiles_live = tiles;
  if (tiles <= 0) return;
  const int grid = (a.nbatch >= 8) ? 8 * ((a.nbatch + 7) / 8) * tiles : a.nbatch * tiles;
  dim3 g(grid), blk(256);
  if (a.tag == 1 && !a_kc && a.tout == 0) {
    hipLaunchKernelGGL((gemm_f32_kernel<false, 0, 1>), g, blk, 0, st, a);
  } else if (a.tout == 0) {
    if (a_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, 0>), g, blk, 0, st, a);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, 0>), g, blk, 0, st, a);
  } else if (a.tout == 1 && a_kc) {
    hipLaunchKernelGGL((gemm_f32_kernel<true, 1>), g, blk, 0, st, a);
  } else if (a.tout == 2 && !a_kc) {
    hipLaunchKernelGGL((gemm_f32_kernel<false, 2>), g, blk, 0, st, a);
  } else {
    fail_launch("launch_gemm(float32): unsupported transposed-output variant");
  }
}

}  // namespace wv
