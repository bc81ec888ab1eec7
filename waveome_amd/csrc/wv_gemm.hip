// Batched float64 GEMM family on v_mfma_f64_4x4x4_4b_f64 (gfx950).
//
// One kernel template serves every O(N^3) step of the evaluation:
//   NT  C -= L21 L21'          Cholesky trailing / panel updates, TRSM via inv(L11)
//   NN  P  = M22 L21, M21 = -P Mkk     triangular inversion
//   TN  K^-1 = M' M            inverse from the triangular inverse
// (reference: the tf.linalg.cholesky / triangular_solve / matmul TensorFlow issues, forward and
//  reverse mode, for gpflow's log marginal likelihood - waveome/model_types_DEPR.py:49-56.)
//
// Why the 4x4x4 four-block MFMA: measured on MI355X (tools/probe_rates.py) back-to-back
// v_mfma_f64_16x16x4_f64 sustains 36 TFLOP/s (128 cycles / instruction) while
// v_mfma_f64_4x4x4_4b_f64 sustains 73 TFLOP/s (16 cycles / 512 FLOP), i.e. the datasheet's
// 78.6 TF FP64 matrix rate is only reachable with the four-block form.
// Lane maps, measured with one-hot operands (tools/probe_mfma4.py); blk = block 0..3:
//   A_blk[i][k] : lane 16 k + 4 blk + i      B_blk[k][j] : lane 16 k + 4 blk + j
//   D_blk[i][j] : lane 16 i + 4 blk + j      D_blk = A_blk B_blk + C_blk   (4x4x4 each)
// A fragment register covers 16 rows x 4 k (block blk = rows 4 blk..4 blk+3), the same
// lane -> (row, k) map as the 16x16x4 form.  A B fragment covers 16 columns x 4 k; loading it
// with the column groups rotated by s (block blk holds columns 4((blk+s)&3)..) for s = 0..3
// pairs every row group with every column group, so 4 A fragments x (4 column sets x 4
// rotations) = 64 MFMAs fill a 64 x 64 wave tile per k-step of 4.
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile 128 x BN (BN = 128 or 64), K chunk 16; launches too small to fill
// the GPU with those use a 64 x 64 block tile, the four waves side by side (TAG bit 4 of the kernel template).
// Operand tiles are staged global -> registers -> LDS, double-buffered, one barrier per chunk.
// LDS images keep the fragment reads (lane l: row l&15 permuted within its 16, k = l>>4)
// bank-conflict free with ds_read_b64:
//   K-contiguous operand  [rows][16+2]   (row stride 18 doubles: 18*r mod 32 distinct, even)
//   K-strided   operand   [16][rows+16]  (row stride = 16 mod 32 doubles)
#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#include "wv_device.hpp"

namespace wv {

// Timing / scheduling experiments (switching parts of the kernel off - results become WRONG -, alternative launch
// orders, one workgroup per CU) exist only in the `make probes` build (-DWV_EXPERIMENTS, tools/libwv_probes.so,
// loaded through WAVEOME_AMD_LIB by tools/probe_gemm_parts.py).  The release library has no such code path: WV_X(bit)
// is the constant false and the GemmArgs fields the switches live in do not exist.
#ifdef WV_EXPERIMENTS
#define WV_X(bit) ((g.xskip & (bit)) != 0)
#else
#define WV_X(bit) false
#endif

typedef double d2 __attribute__((ext_vector_type(2)));

#ifdef WV_EXPERIMENTS  // probes build: shader-clock stamps of workgroup 0 / thread 0 of the small-tile kernel (tools/probe_small_stamps.py)
__device__ unsigned long long wv_gemm_stamps[8];
#define WV_GSTAMP(i) do { if (SMALL && bid == 0 && threadIdx.x == 0) wv_gemm_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WV_GSTAMP(i) do { } while (0)
#endif

constexpr int BM = 128;
constexpr int KC = 16;

template <int ROWS, bool KCONTIG>
struct Tile {
  static constexpr int LD = KCONTIG ? (KC + 2) : (ROWS + 16);
  static constexpr int SIZE = KCONTIG ? ROWS * LD : KC * LD;
  static constexpr int NLOAD = ROWS / 32;  // double2 loads per thread per chunk
  __device__ static inline int at(int row, int k) { return KCONTIG ? row * LD + k : k * LD + row; }

  // K loop loads: per-thread 32-bit byte offsets from the tile's (row0, k0) element, computed once per tile; a
  // chunk's loads are then "uniform chunk base + lane offset" (global_load saddr form) - no 64-bit multiply-adds,
  // no bounds masks and no zero fill per chunk (55 vector instructions per chunk in round 1, each of which delays
  // the wave's own next MFMA).  Rows past the operand's extent are clamped to its last rows: they only feed C
  // elements the epilogue masks.  Offsets stay below 128 * ld * 8 bytes, so 32 bits hold for any ld < 2^22.
  __device__ static inline void goffsets(uint32_t (&off)[NLOAD], int ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int q = tid + 256 * i;
      if (KCONTIG) {
        const int row = min(q >> 3, nrows - 1 - row0), kp = q & 7;
        off[i] = ((uint32_t)row * (uint32_t)ld + 2u * kp) * 8u;
      } else {
        const int k = q / (ROWS / 2), row = min(2 * (q % (ROWS / 2)), nrows - 2 - row0);
        off[i] = ((uint32_t)k * (uint32_t)ld + (uint32_t)row) * 8u;
      }
    }
  }
  // the tile's element (row0, k0): row0 * ld + k0 for K-contiguous operands, k0 * ld + row0 for K-strided ones
  __device__ static inline const char* chunk_base(const double* __restrict__ base, int ld, int row0, int k0) {
    return reinterpret_cast<const char*>(base + (KCONTIG ? (int64_t)row0 * ld + k0 : (int64_t)k0 * ld + row0));
  }
  __device__ static inline void gload(d2 (&reg)[NLOAD], const char* __restrict__ cbase, const uint32_t (&off)[NLOAD]) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) reg[i] = *reinterpret_cast<const d2*>(cbase + off[i]);
  }
  __device__ static inline void sstore(double* __restrict__ s, const d2 (&reg)[NLOAD], int tid) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      int q = tid + 256 * i;
      if (KCONTIG) {
        int row = q >> 3, kp = q & 7;
        *reinterpret_cast<d2*>(s + row * LD + 2 * kp) = reg[i];
      } else {
        int k = q / (ROWS / 2), rp = q % (ROWS / 2);
        *reinterpret_cast<d2*>(s + k * LD + 2 * rp) = reg[i];
      }
    }
  }
};

// One K chunk (4 k-steps) for one wave of a DIAGONAL tile of a square lower-only product (see TRIOPT in the kernel).
//   LOWER: the wave owns a diagonal 64 x 64 block and computes only accumulator blocks i >= j (40 of 64 MFMAs);
//   else : the wave owns 32 columns (column groups 0, 1 relative to bcol) of the full block below (32 MFMAs).
// Compile-time masks only - a wave-uniform runtime mask (scalar branches between the MFMAs) ran 70 % slower.
template <bool LOWER, typename TA, typename TB, int NJ>
__device__ inline void mma_chunk_diag(double (&acc)[4][NJ][4], const double* __restrict__ a_s,
                                      const double* __restrict__ b_s, int arow, int bcol, const int (&brot)[4], int fr,
                                      int fk) {
  for (int s = 0; s < KC / 4; ++s) {
    double af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = a_s[TA::at(arow + i * 16 + fr, s * 4 + fk)];
#pragma unroll
    for (int j = 0; j < (LOWER ? NJ : NJ / 2); ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double bf = b_s[TB::at(bcol + j * 16 + brot[r], s * 4 + fk)];
#pragma unroll
        for (int i = (LOWER ? j : 0); i < 4; ++i)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf, acc[i][j][r], 0, 0, 0);
      }
    }
  }
}

// OCC = waves per SIMD the register allocator may assume (2: <= 256 VGPRs, two workgroups per CU;
// 1: up to 512 VGPRs, one workgroup per CU, k-steps fully unrolled and software-pipelined).
// TOUT: 0 = write C; 1 = write only C' (to g.CT, leading dimension g.ldct); 2 = write both.
// The transposed copy keeps later products on the K-strided (fastest) load path.
// TAG only gives a launch its own symbol so that profilers report it separately: 1 = K_y^-1 = M'M,
// 2 = the Cholesky trailing update A22 -= S S'.  TAG bit 4 (16): the SMALL tile of the few-fits regime - block tile
// 64 x 64, the four waves side by side (64 x 16 each): four times the workgroups of a 128 x 128 tiling, for products
// whose larger tiles would leave most of the 256 CUs idle (one fit in flight: every product of the recursion).
template <bool A_KC, bool B_KC, int BN, int OCC, int TOUT, int TAG = 0>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bid) {
  constexpr bool SMALL = (TAG & 16) != 0;
  static_assert(!SMALL || BN == 64, "the small tile is 64 x 64");
  WV_GSTAMP(0);
  constexpr int BM = SMALL ? 64 : wv::BM;   // block tile rows (shadows the 128 of the regular tiling)
  constexpr int WCOLS = SMALL ? BN / 4 : BN / 2;  // columns of a wave's sub-tile (its rows: 64)
  using TA = Tile<BM, A_KC>;
  using TB = Tile<BN, B_KC>;
#ifdef WV_EXPERIMENTS
  constexpr bool NOLDS = (TAG & 8) != 0;  // timing experiment only: fragments are constants, no LDS reads in the K loop
#else
  constexpr bool NOLDS = false;
#endif
  constexpr int NJ = WCOLS / 16;  // B fragments per wave
  // one LDS block: [2][A tile] [2][B tile] during the K loop, re-used as a [64][BN + 2] staging
  // image of half the C tile in the epilogue
  constexpr int LDC = BN + 2;
  constexpr int SMEM = 2 * (TA::SIZE + TB::SIZE) > 64 * LDC ? 2 * (TA::SIZE + TB::SIZE) : 64 * LDC;
  __shared__ __attribute__((aligned(16))) double smem[SMEM];
  double(*sA)[TA::SIZE] = reinterpret_cast<double(*)[TA::SIZE]>(smem);
  double(*sB)[TB::SIZE] = reinterpret_cast<double(*)[TB::SIZE]>(smem + 2 * TA::SIZE);

  // ---- block -> (batch, tile) with same-matrix tiles kept on one XCD (L2 reuse of panels).
  // g.tiles_live counts only the tiles that do work (lower tiles when g.tri): no workgroup is
  // launched just to exit - an empty workgroup still holds one of the CU's two slots for ~2 us. ----
  const int tiles = g.tiles_live;
  int b, tile;
  if (g.nbatch >= 8) {
    int xcd = bid & 7, q = bid >> 3;
    b = xcd + 8 * (q / tiles);
    tile = q % tiles;
    if (b >= g.nbatch) return;
  } else {
    b = bid / tiles;
    tile = bid % tiles;
  }
  // longest contraction first: ranges trimmed from above (khi_mode) grow with the tile index
  if (g.khi_mode && !g.tri) tile = tiles - 1 - tile;
  // When the contraction length depends on the tile COLUMN it repeats with period tiles_n in launch
  // order; workgroups appear to be dealt to an XCD's shader engines in order, round-robin, so a
  // power-of-two period leaves some engines with only short tiles and the in-order dealer stalls on
  // the full ones (PMC: 1.3 instead of 2 resident waves per SIMD).  The fix is the Latin-square skew below; the
  // older per-matrix rotation of the launch order survives in the probes build only (WV_GEMM_LATIN=0 / 2).
#ifdef WV_EXPERIMENTS
  const bool latin = g.latin != 0;
  if ((g.khi_mode == 2 || g.klo_mode == 2) && !g.tri && g.latin != 1) tile = (tile + 3 * (b >> 3) + (b & 7)) % tiles;
#else
  constexpr bool latin = true;
#endif
  int tm, tn;
  if (g.tri) {
#ifdef WV_EXPERIMENTS
    // (rotating lower-only products or row-dependent lengths the same way measured no gain)
    if (g.rot_tri > 0) tile = (tile + ((5 * (b >> 3) + (b & 7)) & 31)) % tiles;
#endif
    if (g.tri == 2) {  // the diagonal tiles only (square tiles, tri_off == 0)
      tm = tile;
      tn = tile;
#ifdef WV_EXPERIMENTS
    } else if (g.tri == 3 && g.tri_order == 1 && (g.tiles_m & 3) == 0) {
      // L2 experiment (WV_TRI_ORDER=1): the strictly-lower tiles in 4 x 4 super-blocks instead of row by row, so that
      // the 64 tiles an XCD holds at a time touch ~16 operand panels instead of ~21.  Measured: no effect (+-0.5 %,
      // profiles/r03_probe_tri_order_*.txt) - the product is not sensitive to operand re-fetch.
      int sbm = 0, sbn = 0;
      for (;;) {
        const int cnt = sbn < sbm ? 16 : 6;
        if (tile < cnt) break;
        tile -= cnt;
        if (++sbn > sbm) { sbn = 0; ++sbm; }
      }
      if (sbn < sbm) {
        tm = 4 * sbm + (tile >> 2);
        tn = 4 * sbn + (tile & 3);
      } else {
        const int rm[6] = {1, 2, 2, 3, 3, 3}, rn[6] = {0, 0, 1, 0, 1, 2};
        tm = 4 * sbm + rm[tile];
        tn = 4 * sbn + rn[tile];
      }
#endif
    } else {
      tm = 0;
      for (;;) {
        int cnt = min(g.tiles_n, max(0, (tm * BM + BM - 1 - g.tri_off) / BN + 1));
        if (g.tri == 3) cnt = min(cnt, tm);  // ... or all lower tiles but those
        if (tile < cnt) break;
        tile -= cnt;
        ++tm;
      }
      tn = tile;
    }
  } else {
    tm = tile / g.tiles_n;
    tn = tile % g.tiles_n;
    // column-dependent lengths: skew every tile row by its index (Latin square) so that any
    // power-of-two stride through the launch order meets every length equally often
    if (latin && (g.khi_mode == 2 || g.klo_mode == 2)) tn = (tn + tm) % g.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  int kb = g.kb, ke = g.ke;
  if (g.klo_mode == 1) kb = max(kb, m0 + g.koff);
  if (g.klo_mode == 2) kb = max(kb, n0 + g.koff);
  if (g.khi_mode == 1) ke = min(ke, m0 + BM + g.koff);
  if (g.khi_mode == 2) ke = min(ke, n0 + BN + g.koff);
  // Block structure: only the k that can meet a non-zero of both operands (GemmArgsT::spans; tables hold multiples
  // of 16, origins are multiples of 64).  An empty range leaves the epilogue: zeros for beta = 0, nothing for beta = 1.
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

  const double* __restrict__ A = g.A + (int64_t)b * g.sA;
  const double* __restrict__ B = g.B + (int64_t)b * g.sB;
  double* __restrict__ C = g.C + (int64_t)b * g.sC;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = SMALL ? 0 : wave >> 1, wn = SMALL ? wave : wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  // B-fragment column (within its 16) for rotation s: 4*((blk + s) & 3) + j
  const int fblk = (lane >> 2) & 3, fj = lane & 3;
  int brot[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) brot[s] = 4 * ((fblk + s) & 3) + fj;

  double acc[4][NJ][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  // TRIOPT (TAG bit 2): this instantiation is launched on the DIAGONAL tiles of a square lower-only product only
  // (g.tri == 2).  Of such a tile's four 64 x 64 blocks three are needed, two of them only in their lower
  // triangle, yet in the general kernel the tile takes as long as a full one (its time is set by the wave that
  // owns the full block).  Here the two diagonal waves skip the 16 x 16 accumulator blocks above the diagonal
  // (40 of 64 MFMAs per k-step) and the wave whose own block lies above the diagonal takes over the right 32
  // columns of the full block (32 MFMAs each): no partial sums to merge.  (ewm, ewn) = the 64 x 64 block a wave
  // works on, ecol = its first column inside that block, half = it owns 32 columns only.
  constexpr bool TRIOPT = (TAG & 4) != 0 && BN == 128 && TOUT == 0 && !SMALL;
  // On a diagonal tile of M'M / S S' the two operand tiles are the SAME rows of the same matrix (m0 == n0, A == B, both
  // K-strided): ONEOP loads and stages the panel once and reads both fragment kinds from it - half the global loads and
  // LDS writes of the launch whose fixed costs weigh most (57 % MFMA-busy in round 2).  The launcher guarantees A == B.
  constexpr bool ONEOP = TRIOPT && !A_KC && !B_KC;
  int ewm = wm, ewn = wn, ecol = 0;
  bool half = false;
  if (TRIOPT && wm != wn) {
    half = true;
    ecol = wm == 1 ? 0 : 32;
    ewm = 1;
    ewn = 0;
  }
  // Per-wave contraction range at 64-row / (BN/2)-column granularity: a wave whose 64 x BN/2
  // sub-tile needs fewer chunks (triangular operand) or none (above the diagonal of a lower-only
  // product) skips the MFMAs - it still loads and synchronises - and its SIMD time goes to the
  // co-resident workgroup.
  int wkb = kb, wke = ke;
  {
    const int wr = m0 + ewm * 64, wc = n0 + ewn * WCOLS;
    if (g.klo_mode == 1) wkb = max(wkb, wr + g.koff);
    if (g.klo_mode == 2) wkb = max(wkb, wc + g.koff);
    if (g.khi_mode == 1) wke = min(wke, wr + 64 + g.koff);
    if (g.khi_mode == 2) wke = min(wke, wc + WCOLS + g.koff);
    if (g.tri && !TRIOPT && wc + g.tri_off > wr + 63) wke = wkb;  // nothing to do
    if (span_tab) {  // the wave's own 64 rows and (at most 64, granule-aligned) columns
      const int qr = (g.sp_r0 + min(wr, g.M - 1)) >> 6, qc = (g.sp_c0 + min(wc, g.N - 1)) >> 6;
      wkb = max(wkb, max(span_tab[2 * qr], span_tab[2 * qc]) - g.sp_k0);
      wke = min(wke, min(span_tab[2 * qr + 1], span_tab[2 * qc + 1]) - g.sp_k0);
    }
  }
  d2 ra[TA::NLOAD], rb[TB::NLOAD];
  const int nchunk = (ke - kb) / KC;
  // Tiles whose contraction range starts at a tile-dependent offset (klo_mode) walk K downwards
  // from the common end so that all tiles streaming the same panel touch the same rows at the same
  // time (L2 reuse); the others already share their start.
  const int kstep = g.klo_mode ? -KC : KC;
  kb = g.klo_mode ? ke - KC : kb;
  uint32_t offa[TA::NLOAD], offb[TB::NLOAD];
  TA::goffsets(offa, g.lda, m0, g.M, tid);
  if (!ONEOP) TB::goffsets(offb, g.ldb, n0, g.N, tid);
  if (nchunk > 0) {
    TA::gload(ra, TA::chunk_base(A, g.lda, m0, kb), offa);
    if (!ONEOP) TB::gload(rb, TB::chunk_base(B, g.ldb, n0, kb), offb);
    TA::sstore(sA[0], ra, tid);
    if (!ONEOP) TB::sstore(sB[0], rb, tid);
  }
  __syncthreads();
  WV_GSTAMP(1);
  for (int c = 0; c < nchunk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunk && !WV_X(4)) {
      const int kn = kb + (c + 1) * kstep;
      TA::gload(ra, TA::chunk_base(A, g.lda, m0, kn), offa);
      if (!ONEOP) TB::gload(rb, TB::chunk_base(B, g.ldb, n0, kn), offb);
    }
    const double* __restrict__ a_s = sA[cur];
    const double* __restrict__ b_s = ONEOP ? sA[cur] : sB[cur];
    const int kc0 = kb + c * kstep;  // first k of this chunk
    if (kc0 >= wkb && kc0 < wke && !WV_X(8)) {
      if constexpr (TRIOPT) {
        if (half) mma_chunk_diag<false, TA, TB, NJ>(acc, a_s, b_s, 64, ecol, brot, fr, fk);
        else mma_chunk_diag<true, TA, TB, NJ>(acc, a_s, b_s, wm * 64, wn * WCOLS, brot, fr, fk);
      } else {
        // One chunk for the A fragments [I0, I1) (16 rows each) and B column sets [J0, J1) (16 columns each) of the
        // wave: all of them, except where a triangular operand leaves whole fragments zero in the 64 k next to the
        // wave's own rows / columns (10 of 16 fragment-chunks there).  Which side and which end depends on the product
        // and is fixed per instantiation (FAM), so that each kernel carries three partial variants, not twelve:
        //   1  K_y^-1 = M'M (TAG 1)   A(m, k) = M[k][m]    zero for k < m   -> A fragments [0, live)
        //   2  M21 = -M22 T (TOUT 2)  A(m, k) = M22[m][k]  zero for k > m   -> A fragments [4 - live, 4)
        //   3  S = A21 M11' (TOUT 1)  B(n, k) = M11[n][k]  zero for k > n   -> B column sets [4 - live, 4)
        //   4  T = S M11 (plain)      B(n, k) = M11[k][n]  zero for k < n   -> B column sets [0, live)
        auto chunk = [&](auto i0c, auto i1c, auto j0c, auto j1c) {
          constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
          constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
          if constexpr (SMALL && !NOLDS) {
            // small tile: 16 MFMAs per k-step cannot hide the LDS latency of their own 8 fragment reads, and at ~100
            // VGPRs there is room to fetch the next k-step's fragments while this one's MFMAs run
            double af[2][4], bq[2][4];
            auto fetch = [&](int s, int buf) {
#pragma unroll
              for (int i = I0; i < I1; ++i) af[buf][i] = a_s[TA::at(i * 16 + fr, s * 4 + fk)];
#pragma unroll
              for (int r = 0; r < 4; ++r) bq[buf][r] = b_s[TB::at(wn * WCOLS + brot[r], s * 4 + fk)];
            };
            fetch(0, 0);
#pragma unroll
            for (int s = 0; s < KC / 4; ++s) {
              if (s + 1 < KC / 4) fetch(s + 1, (s + 1) & 1);
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = I0; i < I1; ++i)
                  acc[i][0][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[s & 1][i], bq[s & 1][r], acc[i][0][r], 0, 0, 0);
            }
            return;
          }
#pragma unroll OCC == 1 ? 4 : 1
          for (int s = 0; s < KC / 4; ++s) {
            double af[4];
#pragma unroll
            for (int i = I0; i < I1; ++i) af[i] = NOLDS ? 1.0 + i + lane * 1e-3 : a_s[TA::at(wm * 64 + i * 16 + fr, s * 4 + fk)];
#pragma unroll
            for (int j = J0; j < (J1 < NJ ? J1 : NJ); ++j) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const double bf = NOLDS ? 0.5 + j + r + lane * 1e-3 : b_s[TB::at(wn * WCOLS + j * 16 + brot[r], s * 4 + fk)];
#pragma unroll
                for (int i = I0; i < I1; ++i)
                  acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf, acc[i][j][r], 0, 0, 0);
              }
            }
          }
        };
        using std::integral_constant;
        using c0 = integral_constant<int, 0>; using c1 = integral_constant<int, 1>; using c2 = integral_constant<int, 2>;
        using c3 = integral_constant<int, 3>; using c4 = integral_constant<int, 4>;
        // (the small tile's waves sit side by side, 16 columns each: its column sets are whole waves, already trimmed
        //  by wkb / wke; its four A fragments are skipped like the big tile's - only M21 has its own small symbol)
        constexpr int FAM = NOLDS ? 0 : SMALL ? (TOUT == 2 ? 2 : 0) : BN != 128 ? 0
                            : (TAG & 1) ? 1 : TOUT == 2 ? 2 : TOUT == 1 ? 3 : TAG == 0 ? 4 : 0;
        int live = 4;
        if (g.koff == 0) {
          const int wr = m0 + wm * 64, wc = n0 + wn * WCOLS;
          if (FAM == 1 && g.klo_mode == 1) live = min(4, (kc0 - wr) / 16 + 1);
          if (FAM == 2 && g.khi_mode == 1) live = 4 - min(3, max(0, (kc0 - wr) / 16));
          if (FAM == 3 && g.khi_mode == 2) live = 4 - min(3, max(0, (kc0 - wc) / 16));
          if (FAM == 4 && g.klo_mode == 2) live = min(4, (kc0 - wc) / 16 + 1);
        }
        if (FAM == 0 || live >= 4) {
          chunk(c0{}, c4{}, c0{}, c4{});
        } else if constexpr (FAM == 1) {
          if (live == 3) chunk(c0{}, c3{}, c0{}, c4{});
          else if (live == 2) chunk(c0{}, c2{}, c0{}, c4{});
          else chunk(c0{}, c1{}, c0{}, c4{});
        } else if constexpr (FAM == 2) {
          if (live == 3) chunk(c1{}, c4{}, c0{}, c4{});
          else if (live == 2) chunk(c2{}, c4{}, c0{}, c4{});
          else chunk(c3{}, c4{}, c0{}, c4{});
        } else if constexpr (FAM == 3) {
          if (live == 3) chunk(c0{}, c4{}, c1{}, c4{});
          else if (live == 2) chunk(c0{}, c4{}, c2{}, c4{});
          else chunk(c0{}, c4{}, c3{}, c4{});
        } else if constexpr (FAM == 4) {
          if (live == 3) chunk(c0{}, c4{}, c0{}, c3{});
          else if (live == 2) chunk(c0{}, c4{}, c0{}, c2{});
          else chunk(c0{}, c4{}, c0{}, c1{});
        }
      }
    }
    if (c + 1 < nchunk) {
      TA::sstore(sA[cur ^ 1], ra, tid);
      if (!ONEOP) TB::sstore(sB[cur ^ 1], rb, tid);
    }
    if (!WV_X(16)) __syncthreads();
  }

  // ---- epilogue: C = beta*C + alpha*acc.  D lane 16 i' + 4 blk + j of accumulator (i, j, r)
  // is row 16 i + 4 blk + i', column 16 j + 4((blk + r) & 3) + j of the wave tile. ----
  // ---- epilogue: C = beta*C + alpha*acc, staged through LDS so that C is read and written as
  // full rows (64 lanes x 16 B = 1 KB per wave instruction) instead of 32-byte fragments.
  // D lane 16 i' + 4 blk + jj of accumulator (i, j, r) is row 16 i + 4 blk + i', column
  // 16 j + 4((blk + r) & 3) + jj of the wave tile. ----
  WV_GSTAMP(2);
  const double alpha = g.alpha, beta = g.beta;
  if (WV_X(2)) {  // (timing experiment) keep the accumulators alive with one store
    double sink = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sink += acc[i][j][r];
    if (sink == 1.2345e-300) C[0] = sink;
    return;
  }
  constexpr int PAIRS = BN / 2;             // double2 per row
  constexpr int NROWIT = 64 * PAIRS / 256;  // row-pieces per thread per half
#pragma unroll
  for (int h = 0; h < BM / 64; ++h) {
    // issue this half's C reads first: their latency hides behind the LDS staging and the barrier
    d2 cv[NROWIT];
    if (beta != 0.0) {
#pragma unroll
      for (int it = 0; it < NROWIT; ++it) {
        const int q = tid + 256 * it, row = q / PAIRS, cp = q % PAIRS;
        const int m = m0 + 64 * h + row, n = n0 + 2 * cp;
        cv[it] = (m < g.M && n < g.N) ? *reinterpret_cast<const d2*>(C + (int64_t)m * g.ldc + n) : d2{0.0, 0.0};
      }
    }
    if (ewm == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          // (TRIOPT) a wave that owns 32 columns writes its two column groups; blocks a diagonal wave skipped hold
          // zeros: alpha * 0 leaves beta * C, and nobody reads above the diagonal of a lower-only product
          if (TRIOPT && half && j >= NJ / 2) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            smem[(i * 16 + 4 * fblk + fk) * LDC + ewn * WCOLS + ecol + j * 16 + brot[r]] = alpha * acc[i][j][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NROWIT; ++it) {
      const int q = tid + 256 * it, row = q / PAIRS, cp = q % PAIRS;
      const int m = m0 + 64 * h + row, n = n0 + 2 * cp;
      // (TRIOPT: the upper right 64 x 64 block of a diagonal tile is computed by nobody, its staging words are stale)
      if (m < g.M && n < g.N && !(TRIOPT && h == 0 && 2 * cp >= BN / 2)) {
        d2 v = *reinterpret_cast<const d2*>(smem + row * LDC + 2 * cp);
        if (beta != 0.0) {
          v.x += beta * cv[it].x;
          v.y += beta * cv[it].y;
        }
        if (TOUT != 1 && (!WV_X(1) || v.x == 1.2345e-300)) *reinterpret_cast<d2*>(C + (int64_t)m * g.ldc + n) = v;
      }
    }
    if (TOUT != 0) {
      // transposed copy: thread -> (column n, row pair), 32 lanes x 16 B = 512 B of one row of C'
      double* __restrict__ CT = g.CT + (int64_t)b * g.sCT;
#pragma unroll
      for (int it = 0; it < NROWIT; ++it) {
        const int q = tid + 256 * it, n = q >> 5, mp = q & 31;
        const int m = m0 + 64 * h + 2 * mp, nn = n0 + n;
        if (m < g.M && nn < g.N) {
          d2 v;
          v.x = smem[(2 * mp) * LDC + n];
          v.y = smem[(2 * mp + 1) * LDC + n];
          *reinterpret_cast<d2*>(CT + (int64_t)nn * g.ldct + m) = v;
        }
      }
    }
    __syncthreads();
  }
  WV_GSTAMP(3);
}

template <bool A_KC, bool B_KC, int BN, int OCC, int TOUT, int TAG = 0>
__global__ __launch_bounds__(256, OCC) void gemm_f64_kernel(GemmArgs g) {
  gemm_body<A_KC, B_KC, BN, OCC, TOUT, TAG>(g, blockIdx.x);
}

// Two independent products of the same operand layout in ONE launch: workgroups [0, n0) work on g0, the rest on g1.
// The recursion's trailing update A22 -= S S' and T = S M11 both only need S; with few fits in flight neither
// fills the GPU, and a launch of both costs no stream events (an event record / wait pair between two streams
// measured ~6.5 us each on the critical path) and one launch less.
template <bool A_KC, bool B_KC, int BN, int OCC, int TOUT, int TAG = 0>
__global__ __launch_bounds__(256, OCC) void gemm_f64_pair_kernel(GemmArgs g0, GemmArgs g1, int n0) {
  const bool first = (int)blockIdx.x < n0;
  const GemmArgs* g = first ? &g0 : &g1;
  gemm_body<A_KC, B_KC, BN, OCC, TOUT, TAG>(*g, first ? (int)blockIdx.x : (int)blockIdx.x - n0);
}

// Tile counts, launch-order switches and the grid size of one product; 0 workgroups = nothing to launch, -1 = refused.
static int prepare_gemm(GemmArgs& a, int BM, int bn) {
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + bn - 1) / bn;
  if (a.M <= 0 || a.N <= 0 || a.nbatch <= 0) return 0;
  // the kernel walks the contraction in chunks of KC = 16 from 16-aligned offsets (Npad is a multiple of 64, so
  // every schedule in wv_chol.hip satisfies this); anything else would silently drop a remainder
  if ((a.ke - a.kb) % KC != 0 || a.kb % KC != 0 || a.koff % KC != 0) {
    fail_launch("launch_gemm: contraction range / offset not a multiple of 16");
    return -1;
  }
  int tiles = a.tiles_m * a.tiles_n;
  if (a.tri) {
    tiles = 0;
    for (int tm = 0; tm < a.tiles_m; ++tm)
      tiles += std::min(a.tiles_n, std::max(0, (tm * BM + BM - 1 - a.tri_off) / bn + 1));
  }
  a.tiles_live = tiles;
#ifdef WV_EXPERIMENTS
  {
    static int lat = -1;
    if (lat < 0) { const char* e = getenv("WV_GEMM_LATIN"); lat = e ? atoi(e) : 1; }
    a.latin = lat;  // 0: per-matrix rotation, 1: Latin-square skew only (what the release build does), 2: both
  }
  {
    // timing experiments only (results become wrong): 1 = no global C stores, 2 = no epilogue at all,
    // 4 = no global operand loads inside the K loop, 8 = no MFMAs, 16 = no barrier in the K loop,
    // 32 = full K-strided products on the no-LDS-reads instantiation
    static int dg = -1;
    if (dg < 0) { const char* e = getenv("WV_GEMM_DBG"); dg = e ? atoi(e) : 0; }
    a.xskip = dg;
  }
  {
    static int rt = -1;
    if (rt < 0) { const char* e = getenv("WV_ROT_TRI"); rt = e ? atoi(e) : 0; }
    a.rot_tri = rt;
  }
  {
    static int to = -1;
    if (to < 0) { const char* e = getenv("WV_TRI_ORDER"); to = e ? atoi(e) : 0; }
    a.tri_order = to;
  }
#endif
  if (tiles <= 0) return 0;
  return (a.nbatch >= 8) ? 8 * ((a.nbatch + 7) / 8) * tiles : a.nbatch * tiles;
}

// Both products K-strided x K-strided with plain output (the layout of A22 -= S S' and T = S M11); tile code 32 or 64.
void launch_gemm_pair(hipStream_t st, const GemmArgs& a0_in, const GemmArgs& a1_in, int tile) {
  if (tile != 32 && tile != 64) {
    fail_launch("launch_gemm_pair: tile code must be 32 or 64");
    return;
  }
  GemmArgs a0 = a0_in, a1 = a1_in;
  if (a0.tout != 0 || a1.tout != 0) {
    fail_launch("launch_gemm_pair: plain outputs only");
    return;
  }
  const int bm = tile == 32 ? 64 : wv::BM;
  const int n0 = prepare_gemm(a0, bm, 64), n1 = prepare_gemm(a1, bm, 64);
  if (n0 < 0 || n1 < 0) return;
  if (n0 == 0 || n1 == 0) {  // (degenerate: one product is empty)
    if (n0 > 0) launch_gemm(st, a0_in, false, false, tile);
    if (n1 > 0) launch_gemm(st, a1_in, false, false, tile);
    return;
  }
  const dim3 g(n0 + n1), blk(256);
  if (tile == 32) hipLaunchKernelGGL((gemm_f64_pair_kernel<false, false, 64, 2, 0, 16>), g, blk, 0, st, a0, a1, n0);
  else hipLaunchKernelGGL((gemm_f64_pair_kernel<false, false, 64, 2, 0, 0>), g, blk, 0, st, a0, a1, n0);
}

// `tile`: 128 = block tile 128 x 128, 64 = 128 x 64, 32 = the small 64 x 64 tile (few fits in flight)
void launch_gemm(hipStream_t st, const GemmArgs& a_in, bool a_kc, bool b_kc, int tile) {
  if (tile != 32 && tile != 64 && tile != 128) {
    fail_launch("launch_gemm: tile code must be 32, 64 or 128");
    return;
  }
  const bool small = tile == 32;
  const int bn = small ? 64 : tile;
  const int BM = small ? 64 : wv::BM;  // (shadows the regular 128 in the tile counts below)
  GemmArgs a = a_in;
  const int grid = prepare_gemm(a, BM, bn);
  if (grid <= 0) return;
  const int tiles = a.tiles_live;
  dim3 g(grid), blk(256);
#ifdef WV_EXPERIMENTS
  static int occ = -1;
  if (occ < 0) {
    const char* e = getenv("WV_GEMM_OCC");
    occ = (e && atoi(e) == 1) ? 1 : 2;
  }
  if ((a.xskip & 32) && !a_kc && !b_kc && bn == 128 && a.tout == 0) {  // (timing experiment)
    hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 2, 0, 8>), g, blk, 0, st, a);
    return;
  }
#endif
  if (small) {  // the three operand layouts of the recursion (S, trailing update / T / M'M, M21) and the NT form
    if (a.tout == 1 && a_kc && !b_kc) hipLaunchKernelGGL((gemm_f64_kernel<true, false, 64, 2, 1, 16>), g, blk, 0, st, a);
    else if (a.tout == 2 && !a_kc && !b_kc) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 64, 2, 2, 16>), g, blk, 0, st, a);
    else if (a.tout == 0 && !a_kc && !b_kc) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 64, 2, 0, 16>), g, blk, 0, st, a);
    else if (a.tout == 0 && a_kc && b_kc) hipLaunchKernelGGL((gemm_f64_kernel<true, true, 64, 2, 0, 16>), g, blk, 0, st, a);
    else fail_launch("launch_gemm: operand layout not built for the small tile");
    return;
  }
  if ((a.tag == 1 || a.tag == 2) && !a_kc && !b_kc && bn == 128 && a.tout == 0) {
    // K_y^-1 = M'M (tag 1) and the Cholesky trailing update A22 -= S S' (tag 2): square lower-only products.
    // Their diagonal tiles go to a second launch whose waves share the three needed 64 x 64 blocks evenly
    // (TRIOPT in the kernel): trailing update -6.5 %, M'M -2 % at N = 2048.  WV_TRI_SPLIT=0: one launch.
    static const bool split = [] { const char* e = getenv("WV_TRI_SPLIT"); return e ? atoi(e) != 0 : true; }();
    // (only when the diagonal launch - one workgroup per tile row and matrix - can fill the GPU by itself: with few
    // fits in flight it is a serial tail of a few long tiles, 427 us of 1.19 ms at N = 4096 with one fit)
    static const int split_min = [] { const char* e = getenv("WV_TRI_SPLIT_MIN"); return e ? atoi(e) : 256; }();
    // (a.A == a.B: both products multiply a matrix by its own transpose - the diagonal launch stages ONE operand panel)
    const bool can_split = split && a.tri == 1 && a.tri_off == 0 && a.M == a.N && a.tiles_m > 1 && a.M % BM == 0 &&
                           a.nbatch * a.tiles_m >= split_min && a.A == a.B && a.lda == a.ldb && a.sA == a.sB;
    GemmArgs o = a;
    if (can_split) {
      o.tri = 3;
      o.tiles_live = tiles - a.tiles_m;
      g = dim3((a.nbatch >= 8) ? 8 * ((a.nbatch + 7) / 8) * o.tiles_live : a.nbatch * o.tiles_live);
    }
    if (a.tag == 1) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 2, 0, 1>), g, blk, 0, st, o);
    else hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 2, 0, 2>), g, blk, 0, st, o);
    if (can_split) {
      GemmArgs d = a;
      d.tri = 2;
      d.tiles_live = a.tiles_m;
      const dim3 gd((a.nbatch >= 8) ? 8 * ((a.nbatch + 7) / 8) * d.tiles_live : a.nbatch * d.tiles_live);
      if (a.tag == 1) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 2, 0, 5>), gd, blk, 0, st, d);
      else hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 2, 0, 6>), gd, blk, 0, st, d);
    }
    return;
  }
  if (a.tout != 0) {
    // transposed outputs exist for the two products that feed K-strided operands downstream
    if (a.tout == 1 && a_kc && !b_kc) {
      if (bn == 128) hipLaunchKernelGGL((gemm_f64_kernel<true, false, 128, 2, 1>), g, blk, 0, st, a);
      else hipLaunchKernelGGL((gemm_f64_kernel<true, false, 64, 2, 1>), g, blk, 0, st, a);
    } else if (a.tout == 2 && !a_kc && !b_kc) {
      if (bn == 128) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 128, 2, 2>), g, blk, 0, st, a);
      else hipLaunchKernelGGL((gemm_f64_kernel<false, false, 64, 2, 2>), g, blk, 0, st, a);
    } else {
      fail_launch("launch_gemm: unsupported transposed-output variant");
    }
    return;
  }
#ifdef WV_EXPERIMENTS
#define WV_LAUNCH(AK, BK, BNN)                                                  \
  do {                                                                          \
    if (occ == 1) hipLaunchKernelGGL((gemm_f64_kernel<AK, BK, BNN, 1, 0>), g, blk, 0, st, a); \
    else hipLaunchKernelGGL((gemm_f64_kernel<AK, BK, BNN, 2, 0>), g, blk, 0, st, a);          \
  } while (0)
#else
#define WV_LAUNCH(AK, BK, BNN) hipLaunchKernelGGL((gemm_f64_kernel<AK, BK, BNN, 2, 0>), g, blk, 0, st, a)
#endif
  if (bn == 128) {
    if (a_kc && b_kc) WV_LAUNCH(true, true, 128);
    else if (a_kc && !b_kc) WV_LAUNCH(true, false, 128);
    else if (!a_kc && !b_kc) WV_LAUNCH(false, false, 128);
    else WV_LAUNCH(false, true, 128);
  } else {
    if (a_kc && b_kc) WV_LAUNCH(true, true, 64);
    else if (a_kc && !b_kc) WV_LAUNCH(true, false, 64);
    else if (!a_kc && !b_kc) WV_LAUNCH(false, false, 64);
    else WV_LAUNCH(false, true, 64);
  }
#undef WV_LAUNCH
}

#ifdef WV_EXPERIMENTS
int read_gemm_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wv_gemm_stamps), sizeof(wv_gemm_stamps));
}
#endif

}  // namespace wv
