// Shared declarations of the device path (gfx950 only).
//
// HBM layout per problem (all float64, row-major):
//   X      [Npad][D]         design matrix, padded rows are zero
//   Y      [P][Npad]         one contiguous row per output column, padding zero
// per batch slot s (max_batch slots):
//   Kbuf   [Npad][Npad]      K_y (lower) -> L (lower, in place) -> K_y^-1 (lower)
//   Mbuf   [Npad][Npad]      M = L^-1 (lower; strictly upper part stays zero)
//   MTbuf  [Npad][Npad]      M' (upper): K-strided image of M for the products that read M by rows
//   Pbuf   [Npad/2+64][Npad/2+64] + [Npad][64]  L21 scratch of the recursive factorisation
//   theta  [n_params]        constrained parameter table of the slot's fit
//   feat   [2][max_feat][Npad]  row- and column-side kernel features (x_d, rint x_d, sinpi / cospi(x_d / period))
//   r, z, alpha [Npad]       residual y - c, L^-1 r, K_y^-1 r
// Npad = N rounded up to a multiple of 64; padded rows/cols of K_y are the identity so
// log|K_y|, r' K_y^-1 r and the leading N x N block of K_y^-1 are unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "wv_spec.hpp"

struct wv_problem;  // (the C-ABI handle: wv_api.hip)

namespace wv {

constexpr int NB = 64;          // diagonal block / panel width
constexpr int GT = 64;          // Gram / trace tile edge
constexpr int MAX_FEAT = 64;    // feature slots per side staged in LDS by the Gram kernels
constexpr int MAX_TERM_F = 4;   // factors per term
constexpr int FEW_FITS = 8;     // at most this many fits in a launch: Gram / trace passes run ONE tile per workgroup

// ---- device-side spec --------------------------------------------------------------
struct DevFactor {
  int32_t type;
  int32_t p[3];       // parameter indices
  int32_t degree;
  int32_t ndims;
  int32_t slot0;      // first feature slot (see FeatOp); consecutive per active dim
  int32_t trainable;  // bit k set: parameter k of this factor is trainable
  int32_t sbase;      // first gradient-sum slot of this factor in the trace pass (-1: nothing trainable)
  int32_t nsum;       // gradient sums the factor's type defines (1: variance only, 2: + lengthscale / offset, 3: + period)
};
enum { FEAT_RAW = 0, FEAT_RINT = 1, FEAT_SINPI = 2, FEAT_COSPI = 3 };
struct FeatOp {
  int32_t kind;
  int32_t dim;
  int32_t param;   // period parameter for SINPI/COSPI
  int32_t sparam;  // >= 0: the ROW-side value is multiplied by sqrt(32/ln 2) / theta[sparam] (a Periodic factor's
                   // lengthscale folded into its sin / cos features, see wv_gram.hip); -1: none
};
struct DevSpec {
  int32_t n_terms, n_factors, n_feat, n_params;
  int32_t noise_param, mean_param, n_dims, n_red;  // n_red: gradient-sum slots of the trace pass (noise included)
  int32_t term_start[65];
  DevFactor factors[128];
  FeatOp feat[MAX_FEAT];
  int16_t fslots[512];  // per factor: feature slot per active dim (Periodic: sin, cos, raw)
};

// ---- GEMM launch description --------------------------------------------------------
// C[m][n] = beta*C[m][n] + alpha * sum_{k in [kb,ke)} A(m,k) * B(n,k),  per batch slot.
// A(m,k) = a_kcontig ? A[m*lda + k] : A[k*lda + m];  B(n,k) likewise.
template <typename T>
struct GemmArgsT {
  const T* A;
  const T* B;
  T* C;
  int64_t sA, sB, sC;  // batch strides (elements)
  int32_t lda, ldb, ldc;
  int32_t M, N;        // extent of C
  int32_t kb, ke;      // default contraction range
  // triangular operands trim the contraction range per tile (m0 / n0 = tile origin in C):
  //   klo_mode 1: kb = max(kb, m0 + koff)        2: kb = max(kb, n0 + koff)
  //   khi_mode 1: ke = min(ke, m0 + BM + koff)   2: ke = min(ke, n0 + BN + koff)
  int32_t klo_mode, khi_mode, koff;
  int32_t tri;         // 1: skip tiles whose first column (n0 + tri_off) > last row (m0 + BM - 1)
  int32_t tri_off;
  double alpha, beta;
  int32_t nbatch;
  int32_t tiles_m, tiles_n;
  T* CT;               // transposed output (tout != 0): CT[n][m], batch stride sCT, leading dimension ldct
  int64_t sCT;
  int32_t ldct, tout;
  int32_t tag;         // 1: launch the separately named instantiation (K_y^-1 = M'M)
  int32_t tiles_live;  // tiles that are launched (lower tiles only when tri)
  int32_t notrim;      // 1: no partial-chunk streams next to a triangular operand's diagonal (full chunks over the zeros)
#ifdef WV_EXPERIMENTS   // `make probes` only (wv_gemm.hip): the release kernels have no switch that can change a result
  int32_t latin;       // 0: per-matrix rotation of the launch order, 1: Latin-square skew (release behaviour), 2: both
  int32_t rot_tri;     // rotate the launch order of lower-only products per matrix
  int32_t xskip;       // WV_GEMM_DBG: parts of the kernel switched off for timing (results become wrong)
  int32_t tri_order;   // WV_TRI_ORDER=1: strictly-lower tiles of M'M / the trailing update in 4 x 4 super-blocks (L2 experiment)
  int32_t stagger, stagger_mode;  // WV_GEMM_STAGGER=<cycles>[,mode]: some workgroups of a launch's first wave wait before their tile
#endif
  // Block structure (wv_spec.hpp, block_column): when `spans` is set, a matrix whose structure (slot_spec -> spec_span)
  // has a span table only walks the k that can meet a non-zero in both operands: with (sp_r0, sp_c0, sp_k0) the
  // K_y-index of the product's row 0, column 0 and k = 0, a tile's range is cut to span(rows) ^ span(columns).
  const int32_t* spans;      // [tables][span_nblk][2]
  const int32_t* spec_span;  // [specs]  table index or -1
  const int32_t* slot_spec;  // [slots]
  int32_t span_nblk, slot0, sp_r0, sp_c0, sp_k0;
};
using GemmArgs = GemmArgsT<double>;
using GemmArgsF = GemmArgsT<float>;  // float32 storage path (wv_gemm32.hip)

enum Stage {
  ST_GRAM = 0, ST_POTRF_DIAG, ST_CHOL_TRSM, ST_CHOL_PANEL, ST_CHOL_TRAIL, ST_TRTRI_A, ST_TRTRI_B,
  ST_LAUUM, ST_SOLVE, ST_TRACE, ST_REDUCE,
  ST_LAUUM_FEW,  // K_y^-1 = M'M on the small / 64-wide tiles of the few-fits regime (other kernel symbols than ST_LAUUM's)
  ST_COUNT
};
extern const char* const kStageNames[ST_COUNT];

struct StageTimer {
  bool enabled = false;
  uint32_t mask = 0xffffffffu;  // bit s: stage s is timed
  bool cur = false;
  struct Rec { hipEvent_t a, b; int stage; };
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;
  double ms[ST_COUNT] = {0};
  int64_t launches[ST_COUNT] = {0};
  double flops[ST_COUNT] = {0};
  double bytes[ST_COUNT] = {0};
  void begin(hipStream_t s, int stage);
  void end(hipStream_t s);
  void collect();
  void reset();
  void work(int stage, double fl, double by) { if (enabled && (mask >> stage & 1u)) { flops[stage] += fl; bytes[stage] += by; } }
  ~StageTimer();
};

struct Ctx {
  int device = 0;
  int dtype = 0;  // 0: float64 matrices, 1: float32 matrices (vectors, reductions and host I/O stay float64)
  hipDeviceProp_t prop;
};

// One group = a set of batch slots that advance together on one stream.
struct Group {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  int slot0 = 0, nslots = 0;
  // pinned host staging
  double* h_theta = nullptr;   // [nslots][n_params]
  int32_t* h_col = nullptr;    // [nslots]
  double* h_out = nullptr;     // [nslots][out_stride]
  int32_t* h_status = nullptr; // [nslots]
  int32_t* h_spec = nullptr;   // [nslots] structure index of each slot's task (multi-structure problems)
};

struct Problem {
  Ctx* ctx = nullptr;
  Spec spec;      // structure 0 (the only one of a single-structure problem)
  DevSpec hspec;
  // Multi-structure problems (wv_problem_create_multi): several kernel structures share X, Y and the
  // workspace; every slot carries the index of its task's structure.  All device tables use the common
  // stride np_stride = max n_params (each DevSpec's n_params field holds that stride; the unused tail of
  // a slot's parameter row / gradient record is zero).
  std::vector<Spec> specs;
  // block structure: one span table per column of X that some structure is block diagonal in (built on demand from
  // the host copy of X), and per structure the index of its table (-1: dense)
  std::vector<double> hX;             // [N][D]
  std::vector<int> span_table_of_col; // [D] table index or -1
  int n_span_tables = 0;
  // Independent diagonal blocks (wv_chol.hip, fi_node): decoupled[g] != 0 = in EVERY structure of the problem no row
  // below 64 g can share a label with a row at or above it, i.e. K_y(rows >= 64 g, columns < 64 g) is structurally zero
  // for every slot: the recursion then treats the two sides of a split at 64 g as independent problems (no S, no update,
  // no T, no M21 - those blocks of M stay the zeros they were created as) and factors runs of independent 128-nodes in one
  // launch.  Recomputed whenever a structure is registered; all zero once a dense structure exists.
  std::vector<std::vector<int32_t>> hspan_tabs;   // host copies of the span tables
  std::vector<int32_t> spec_table;                // [specs] table index or -1
  std::vector<uint8_t> decoupled;                 // [Npad / 64 + 1]
  bool ever_dense = false;                        // a dense structure was registered at some point: never decouple again
  bool decoupled_at(int64_t row) const { return !decoupled.empty() && row % 64 == 0 && decoupled[row / 64] != 0; }
  int32_t* dspans = nullptr;          // [D][Npad / 64][2]
  int32_t* dspec_span = nullptr;      // [spec_cap]
  // missing outputs (wv_problem_caps::nan_is_missing): per output column the rows that are observed (1) - masked
  // rows are treated like padding rows (identity row of K_y, zero residual, no gradient weight) - and their count
  uint8_t* dmask = nullptr;           // [P][Npad], null when nothing is missing
  int32_t* dnobs = nullptr;           // [P]
  int np_stride = 0, max_feat = 0;
  int32_t* dslotspec = nullptr;  // [max_batch]
  int64_t N = 0, Npad = 0;
  int D = 0, P = 0, max_batch = 0;
  int out_stride = 0;  // doubles per slot in the result record
  // device memory
  double *dX = nullptr, *dY = nullptr;
  DevSpec* dspec = nullptr;
  double *Kbuf = nullptr, *Mbuf = nullptr, *MTbuf = nullptr, *Pbuf = nullptr;
  // float32 problems (ctx dtype 1) keep the four N x N work matrices in these instead
  int f32 = 0;
  float *Kbuf32 = nullptr, *Mbuf32 = nullptr, *MTbuf32 = nullptr, *Pbuf32 = nullptr;
  double *dtheta = nullptr, *dr = nullptr, *dz = nullptr, *dalpha = nullptr;
  double *dfeat = nullptr;        // [max_batch][2 sides][max_feat][Npad] per-row kernel features of each slot's fit
  double *dfconst = nullptr;      // [max_batch][max_fact][4] per-factor constants derived from theta (see wv_gram.hip)
  int max_fact = 0, max_red = 1;  // largest n_factors / n_red over the problem's structures
  bool any_multidim = false;      // some factor of some structure acts on more than one column
  bool any_product = false;       // some structure has a product term (or may get one: capacity for later structures)
  int spec_cap = 0;               // entries of the device structure table (>= specs.size(): wv_problem_add_spec)
  double *dsolve_part = nullptr;  // [max_batch][Npad/64][Npad] partial alphas of the fused solve (Npad <= 2048); [max_batch][8][Npad] above
  // float32 contexts only (the float64-grade value of a float32 evaluation, wv_chol.hip launch_f32_refine):
  // [max_batch][F32R_VECS][Npad]: probe v, Mt'v, K_y Mt'v, K_y alpha, E v, E zt, Mt'E v, Mt'E zt, K_y Mt'E v, K_y Mt'E zt,
  // E^2 v, E^2 zt, diag(Mt'Mt) in float64
  double *df32r = nullptr;
  double *dlogdet = nullptr;   // [max_batch][nblk] partial log-determinants
  double *dpart = nullptr;     // [max_batch][ntiles][n_params] trace partials
  // few fits in flight (<= FEW_FITS): per-quad gradient accumulators of every tile, [groups][quad_fits][ntiles][quad_stride],
  // summed by trace_group_kernel in the order of the many-fits regime (null: that regime is used for few fits too)
  double *dquad = nullptr;
  int64_t quad_stride = 0;
  int quad_fits = 0;
  double *dout = nullptr;      // [max_batch][out_stride]
  int32_t *dcol = nullptr, *dstatus = nullptr;
  int64_t workspace_bytes = 0;
  std::vector<Group> groups;
  StageTimer timer;
  bool open_queue = false;  // a wv_queue owns the groups' staging buffers: every other device entry point is refused
  // Post-fit calls (wv_predict, wv_whitened_means, wv_whitened_state) are float64, complete-case computations.  A float32
  // problem, or one with missing outputs, serves them on a float64 single-column twin of itself, made on first use from
  // its own device copies (wv_api.hip, post_fit_twin): column twin_col of Y, and - with missing outputs - only the rows
  // observed for that column (twin_rows: their indices in this problem; empty = all rows).
  Ctx ctx64;
  ::wv_problem* twin = nullptr;
  int twin_col = -1;
  int twin_slots = 0;                 // batch slots the twin was created with (grown when wv_whitened_means asks for more)
  std::vector<int64_t> twin_rows;
  std::vector<double> twin_hostX;     // host copy of X (missing-output problems: the twin is re-made per column)
  int64_t pbuf_elems() const { int64_t h = Npad / 2 + NB; return std::max<int64_t>(h * h, Npad * NB); }
  int nblk() const { return (int)(Npad / NB); }
  int ntiles() const { int t = (int)(Npad / GT); return t * (t + 1) / 2; }
};

// result record layout per slot: [0]=lml, [1]=quad(z'z), [2]=logdet, [3..3+n_params)=dLML/dtheta
inline int out_stride_for(int n_params) { return 3 + n_params; }
constexpr int F32R_V = 0, F32R_W = 1, F32R_YV = 2, F32R_YA = 3, F32R_EV = 4, F32R_EA = 5, F32R_W2V = 6, F32R_W2A = 7,
              F32R_Y2V = 8, F32R_Y2A = 9, F32R_E2V = 10, F32R_E2A = 11, F32R_PD = 12, F32R_VECS = 13;

// ---- environment switches --------------------------------------------------------------
// The release library reads exactly the switches of kTuningSwitches (wv_api.hip; documented one by one in
// waveome_amd/csrc/README.md, each covered by a bit-identity test in tests/test_gpu_parity.py): tuning_env() returns
// nullptr for any other name.  experiment_env() is for everything that only exists to measure an alternative: it reads the
// environment in the diagnostic build (-DWV_EXPERIMENTS) and is the constant nullptr in the release library, which
// therefore does not even contain those names.
const char* tuning_env(const char* name);
#ifdef WV_EXPERIMENTS
inline const char* experiment_env(const char* name) { return std::getenv(name); }
#else
#define experiment_env(name) (static_cast<const char*>(nullptr))
#endif

// ---- kernel launchers (each enqueues on `st`, no host sync) ---------------------------
void launch_gram(Problem& p, hipStream_t st, int slot0, int nb, int write_upper);
void launch_gram_masked(Problem& p, hipStream_t st, int slot0, int nb, uint64_t term_mask, double diag);
void launch_gram_rect(Problem& p, hipStream_t st, const double* dXnew, int64_t Mrow, int64_t Mpad,
                      const double* dtheta_row, double* out);
void launch_trmv(Problem& p, hipStream_t st, const double* Tmat, const double* in, double* out, int slot0,
                 int nb, bool trans);
void launch_potrf_diag(Problem& p, hipStream_t st, int slot0, int nb, int j0);
void launch_gemm(hipStream_t st, const GemmArgs& a, bool a_kcontig, bool b_kcontig, int bn);
void launch_gemm_pair(hipStream_t st, const GemmArgs& a0, const GemmArgs& a1, int tile);  // two products, one launch
void launch_gemm(hipStream_t st, const GemmArgsF& a, bool a_kcontig, bool b_kcontig, int bn);
void launch_resid(Problem& p, hipStream_t st, int slot0, int nb);
void launch_solve(Problem& p, hipStream_t st, int slot0, int nb, bool need_alpha);
void launch_trace(Problem& p, hipStream_t st, int slot0, int nb);
void launch_gram_matvec2(Problem& p, hipStream_t st, int slot0, int nb, const double* w1, const double* w2, double* y1,
                         double* y2, int64_t vec_stride, int64_t alpha_stride);
void launch_f32_refine(Problem& p, hipStream_t st, int slot0, int nb);
int f32_order();
void launch_reduce(Problem& p, hipStream_t st, int slot0, int nb, bool with_grad);
int trace_nwg(const Problem& p, int nb);
int trace_part_stride(const Problem& p);

// schedules
void cholesky(Problem& p, hipStream_t st, int slot0, int nb);
void trtri(Problem& p, hipStream_t st, int slot0, int nb);
void lauum(Problem& p, hipStream_t st, int slot0, int nb);
void factor_inverse(Problem& p, hipStream_t st, int slot0, int nb);  // recursive potrf + trtri
void factor_value_only(Problem& p, hipStream_t st, int slot0, int nb);  // log-det and z = L^-1 r only

// the N x N work matrices of a problem in their storage type
template <typename T> struct MatBufs { T *K, *M, *MT, *P; };
template <typename T> inline MatBufs<T> mat_bufs(Problem& p);
template <> inline MatBufs<double> mat_bufs<double>(Problem& p) { return {p.Kbuf, p.Mbuf, p.MTbuf, p.Pbuf}; }
template <> inline MatBufs<float> mat_bufs<float>(Problem& p) { return {p.Kbuf32, p.Mbuf32, p.MTbuf32, p.Pbuf32}; }

// 4 consecutive matrix elements <-> 4 doubles (16-byte aligned in float32, two 16-byte halves in float64)
__device__ inline void load4(const double* __restrict__ src, double (&v)[4]) {
  const double2 a = *reinterpret_cast<const double2*>(src), b = *reinterpret_cast<const double2*>(src + 2);
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ inline void load4(const float* __restrict__ src, double (&v)[4]) {
  const float4 a = *reinterpret_cast<const float4*>(src);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ inline void store4(double* __restrict__ dst, const double (&v)[4]) {
  *reinterpret_cast<double2*>(dst) = make_double2(v[0], v[1]);
  *reinterpret_cast<double2*>(dst + 2) = make_double2(v[2], v[3]);
}
__device__ inline void store4(float* __restrict__ dst, const double (&v)[4]) {
  *reinterpret_cast<float4*>(dst) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

std::string build_devspec(const Spec& s, DevSpec& out);

#define WV_HIP_TRY(expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      wv::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
      return WV_E_HIP;                                                                     \
    }                                                                                      \
  } while (0)

}  // namespace wv
