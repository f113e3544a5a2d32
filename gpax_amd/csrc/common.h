// common.h — context, device buffers, launch/profiling helpers shared by the libgpx sources.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gpx.h"

namespace gpx {

constexpr int GPX_TILE_COUNTERS = 1024;
constexpr int TILE = 128;      // MFMA GEMM block tile and diagonal-block size
constexpr int OUTER_TILES = 4; // default outer blocking of the right-looking sweeps (4*128 = 512); ctx->outer_tiles
// A single-sample Cholesky of up to this many tile rows runs as ONE outer block (plain right-looking, K = 128 updates of
// everything to the right, one stream): up to N = 5120 the look-ahead's cross-stream waits and the chain work it puts
// BEFORE the next diagonal block cost more than its larger K returns (round 4, one box: potrf N = 1024 0.439 -> 0.377 ms,
// 2048 0.950 -> 0.812, 4096 2.207 -> 1.995, 5120 2.96 -> 2.88; 6144 3.90 -> 4.00: the blocked schedule from there on).
// Batched sweeps keep the blocked schedule: B times the update, GEMM-bound.  FINISH_...: the END of a larger
// factorisation is finished the same way once no more rows than this are left (linalg.hip; N = 8192 potrf -1 %).
constexpr int ONE_BLOCK_TILES = 40;
constexpr int FINISH_ONE_BLOCK_TILES = 32;
constexpr double AUG_BIG = 1e300;
constexpr double SQRT5 = 2.23606797749978969641;
constexpr double MATERN_EPS = 1e-12; // gpax/kernels/kernels.py:20-21
// k-step 32 of the latency shapes for SINGLE-SAMPLE sweeps over matrices of up to this many tile rows (gpx_ctx::small_bk)
constexpr int SMALL_BK_ROWS = 40;
// tile rows below which the single-sample far update of the Cholesky waits for U1 of the same block (linalg.hip)
constexpr int FAR_AFTER_U1 = 40;
enum { GPX_POTF2_SLIM = 0, GPX_POTF2_TILE = 2 }; // gpx_ctx::potf2_mode

// k(r2) of the three kernels (gpax/kernels/kernels.py:44-117) — shared by the Gram build (gram.hip) and the fused small-N
// fit step (fit_small.hip), so that both produce the same matrix bit for bit
template <int KIND>
__device__ __forceinline__ double kernel_value(double r2, double scale) {
  if (KIND == GPX_KERNEL_R2) { // gpx_gram only: the squared scaled distance itself (kernels.py:28-41)
    return r2;
  } else if (KIND == GPX_KERNEL_RBF) {
    return scale * exp(-0.5 * r2);
  } else if (KIND == GPX_KERNEL_PERIODIC) { // r2 carries sum_k (sin(pi (x_k - z_k) / p) / l_k)^2
    return scale * exp(-2.0 * r2);
  } else {
    const double r = sqrt(r2 + MATERN_EPS);
    const double s5r = SQRT5 * r;
    return scale * (1.0 + s5r + (5.0 / 3.0) * r2) * exp(-s5r);
  }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Leading dimension for an internal matrix with `cols` logical columns: multiple of 16
// doubles (128-B rows) and never a multiple of 512 doubles, so that a 128-row tile load does
// not put every row on the same HBM channel set.
inline int64_t pick_ld(int64_t cols) {
  int64_t ld = round_up64(cols, 16);
  if (ld % 512 == 0) ld += 16;
  return ld;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      p = nullptr;
      cap = 0;
      if (e != hipSuccess) return e;
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  double* d() const { return static_cast<double*>(p); }
  int* i() const { return static_cast<int*>(p); }
};

// Pinned host staging buffer (grow-only): bulk sweep inputs / outputs cross PCIe from page-locked memory, so
// that hipMemcpyAsync is a plain DMA enqueue and never falls into the runtime's pageable-copy path.
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipHostFree(p);
      p = nullptr;
      cap = 0;
      if (e != hipSuccess) return e;
    }
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  double* d() const { return static_cast<double*>(p); }
};

struct KernelParams {
  int kind;
  int d;
  double inv_ell[GPX_MAX_DIM];
  double scale;
  double pi_over_p; // pi / period (periodic kernel only)
};
inline int n_ell(const KernelParams& kp) { return kp.d + (kp.kind == GPX_KERNEL_PERIODIC ? 1 : 0); }

// Hyper-parameters of one posterior sample as the batched launches read them from a device
// table (entry b of a launch's batch): the kernel parameters plus the three diagonal terms.
struct ThetaDev {
  KernelParams kp;
  double diag_train; // noise + jitter            (K's diagonal, gp.py:160 / kernels.py:63-65)
  double diag_pred;  // noise_p + jitter          (k_pp's diagonal, gp.py:267)
  double kdiag_pred; // k(x, x) + noise_p + jitter (diag of k_pp, for the variance-only path)
};

// Per-task inputs of a batched launch (vExactGP, gpax/models/vgp.py:70-78: T independent GPs, each
// with its own X / X_new / y): batch entry b reads task (b % mod); mod = 0: one shared input.
struct TaskStride {
  int mod = 0;
  int64_t x_bs = 0, z_bs = 0;
};

// Batch layout of the device-resident pipeline: B independent samples per launch (the vmap of
// gpax/models/gp.py:393-395 as a grid dimension).  Element b lives at base + b * stride.
struct BatchPlan {
  int B = 1;
  bool sweep = false;           // predictive sweep (sizes split-K for the nominal batch at this N)
  const ThetaDev* th = nullptr; // device table (B entries) or nullptr: by-value ctx->theta
  const double* yres = nullptr; // y residuals, element stride y_bs (0 = shared by the batch)
  int64_t y_bs = 0;
  int y_mod = 0;                // > 0: entry b reads yres + (b % y_mod) * y_bs (per-task residuals)
  int64_t k_bs = 0, linv_bs = 0;                     // K / Linv
  int64_t mean_bs = 0;                               // mean, var
  int64_t cov_bs = 0, covlinv_bs = 0, splitk_bs = 0; // Cov, CovLinv, SplitK
  int64_t eps_bs = 0;                                // eps, draws
  int* info_train = nullptr;                         // B ints each
  int* info_cov = nullptr;
  double* scal = nullptr; // per sample [quad, sumlog, grad(ell.., scale, noise)], stride scal_bs
  int64_t scal_bs = 0;
  const double* pred_diag = nullptr; // per-sample, per-test-point variance added to diag(cov) (M each)
  int64_t pd_bs = 0;
};

struct ProfAcc {
  int64_t launches = 0;
  double work = 0.0;
  double bytes = 0.0; // algorithmic bytes of the MFMA classes: the C entries a launch reads and writes (16 B each)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double ms = 0.0;
};

} // namespace gpx

struct gpx_ctx {
  int device = -1;
  hipStream_t stream = nullptr;  // main stream (API copies, trailing updates)
  hipStream_t pstream = nullptr; // high-priority panel stream (lookahead)
  hipStream_t xstream = nullptr; // side stream, lowest priority: work with time to spare beside a chain (the L^-T trees of
                                 // the sparse path while the next factorisation's chain or the tall solve runs; sparse.hip)
  hipStream_t s = nullptr;       // stream the launch helpers currently target
  // diagonal-block kernel (potf2.hip): GPX_POTF2 = slim (default: placed at once beside two resident trailing-update
  // workgroups) | tile (round 2: the tests' reference)
  int potf2_mode = gpx::GPX_POTF2_SLIM;
  // k-step of the latency shapes: GPX_SMALL_BK = 16 | 32 forces it; default 0 = per driver call (small_bk_now): 32 for
  // SINGLE-SAMPLE sweeps over matrices of up to SMALL_BK_ROWS tile rows — nothing saturates the chip there and the
  // fatter k-step halves the barriers of every chain launch (potrf -2 ... -5 % at N = 512 ... 4096) — 16 otherwise (34 /
  // 42 KB of LDS no longer fit beside two resident trailing-update workgroups: +2 % at N = 16384; batched small-N
  // sweeps lose 3 - 6 % of their occupancy-bound throughput).  Never changes a bit.
  int small_bk = 0;
  int small_bk_now = 16;
  // latency-shape GEMM: 5 = round 5 (gemm_tile.h lat_tile: LDS-direct staging, ring of 64-B k-slices, one barrier per
  // slice) | 1 = round 1 (register staging, padded LDS, two barriers per k-step of 16 / 32) | 0 = by driver (default):
  // round 5 wherever the chain has the chip to itself (one-outer-block factorisations, the L^-T trees, the sparse path's
  // M x M chains and products: C2 fit step -1.6 %, C5 sparse step -1.5 %, tree levels at K >= 512 1.2 - 1.4 x), round 1
  // inside the BLOCKED two-stream sweeps, whose chain launches share every SIMD with two MFMA-saturating waves of the
  // trailing update — there a barrier costs what the slowest of four contended waves costs, and 16 of them per K = 128
  // lose to 8 (C3: gemm_other 18.4 -> 20.3 ms per predict, potrf 29.4 -> 29.7; profiles/r05/lat_gemm_ab.md).  Same bits.
  // GPX_LAT_GEMM=r5|r1 / gpx_debug_set_lat_gemm force one everywhere.
  int lat_gemm = 0;
  int lat_now = 5; // the kernel the launches of the current driver call take (set by the drivers from lat_gemm)
  bool lat_lin = true; // lower-tile launches of the round-5 latency shape enumerate only the live tiles (GPX_LAT_LIN=0: square grid)
  hipEvent_t evD = nullptr;
  // one-outer-block chains (a single sample with the chip to itself): the panel TRSM rides in the potf2 launch (potf2.hip
  // potf2_trsm_kernel; GPX_POTF2_TRSM=0 / gpx_debug_set_potf2 "nofuse": the three-launch step)
  bool potf2_trsm = true;
  gpx::DevBuf chain_flag; // the flags (one per stream) workgroup 0 of potf2_trsm_kernel publishes L^-1 through
  bool chain_flag_zeroed = false, chain_attr_set = false;
  unsigned chain_epoch[3] = {0, 0, 0}; // per stream of the context (main, panel, side)
  bool serialise_trailing = false; // measurement mode: every Cholesky trailing update runs alone on the chip (linalg.hip)
  // > 0 while a driver whose own panel chain holds no potf2 (the right-looking TRSM sweeps, the K^-1 = W W^T product)
  // is queueing launches: its big-tile GEMMs run persistently (GPX_PERSIST_SCOPE=0 disables)
  int persist_scope = 0;
  bool persist_scope_ok = true;
  gpx::DevBuf tile_counters;
  unsigned tile_counter_seq = 0;
  std::vector<hipEvent_t> tile_counter_ev;      // per ring slot: recorded behind the last kernel that used it ...
  std::vector<hipStream_t> tile_counter_stream; // ... and the stream it ran on
  int tail_tiles = 72; // GPX_TAIL_TILES: an outer block is in the chain-bound tail when fewer tile rows than this remain (0: no tail)
  std::vector<hipEvent_t> evP, evU; // per-outer-block panel / next-panel-update events
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  hipDeviceProp_t prop;
  int lazy_group = 2; // GPX_LAZY_GROUP: outer blocks whose far (bulk) trailing update is applied in one launch while the
                      // factorisation is GEMM-bound (linalg.hip): K = 1024 at the default outer block of 512 columns
  int outer_tiles = gpx::OUTER_TILES; // GPX_OUTER_TILES (experiments): K of the trailing update = 128 * outer_tiles
  bool outer_tiles_set = false;       // ... given explicitly: no ONE_BLOCK_TILES rule
  unsigned func_attr_mask = 0; // kernels whose dynamic-LDS attribute this context has set on ITS device (bit per variant)

  // ---- training state -------------------------------------------------------------------
  int N = 0, d = 0;
  int T = 1;         // tasks: X holds T training sets of N points (vExactGP); 1 everywhere else
  int Np = 0;        // padded order of the augmented matrix: round_up(N + 1, 128)
  int64_t ldk = 0;   // leading dimension of K/W buffers
  gpx::DevBuf X;     // T x N x d
  gpx::DevBuf K;     // Np x ldk : Gram -> L (lower) -> K^-1 (lower)
  gpx::DevBuf W;     // Np x ldk : L^-T (upper), allocated on first gradient
  gpx::DevBuf Wscr;  // Np x ldk : scratch of the L^-T tree (linalg.hip: T and the transposed C blocks of one level)
  // GPX_SGP_SOLVE = inverse (1, default: factor Kuu, L^-T tree, then W = Kfu Luu^-T as ONE GEMM against Luu^-1) | ride (2,
  // round 5: W as a blocked solve that follows the Cholesky chain of Kuu group by group, the trees on the side stream —
  // measured equal: the chain runs 2.7 x slower beside the tall GEMMs, profiles/r05/sparse_ride.md) | sweep (0, round 2:
  // right-looking sweeps)  (sparse.hip)
  int sgp_inverse = 1;
  int linvt_tree = 1; // GPX_LINVT=tree|sweep: L^-T by the block-recursive inverse (default) or the right-looking sweep
  gpx::DevBuf Linv;  // (Np/128) x 128 x 128 inverses of the diagonal blocks of L
  gpx::DevBuf yres;  // N
  gpx::DevBuf diagv; // N: per-point variance added to K's diagonal (gpx_set_diag), when has_diag
  bool has_diag = false;
  gpx::DevBuf scal;  // small device scalars (lml pieces, gradient, info)
  gpx::DevBuf part;  // reduction partials
  gpx::DevBuf alpha; // N
  gpx::KernelParams theta{};
  double noise = 0, jitter = 0;
  // N <= 128: the whole fit step (Gram, factorisation, lml terms, alpha, K^-1, gradient contraction) as ONE launch
  // (fit_small.hip; GPX_FIT_SMALL=0: the general launch sequence).  small_grad_ready: the gradient of the factorisation in
  // K already sits in the plan's scal / ctx->alpha — dev_grad has nothing left to launch
  bool fit_small = true;
  bool small_grad_ready = false;
  bool small_no_kinv = false; // the last gradient came from the one-launch step: K^-1 was never stored (gpx_lml_grad_diag)
  unsigned fit_small_attr = 0; // fit_small_kernel variants whose dynamic-LDS limit this context has raised on its device
  bool factored = false;
  bool have_kinv = false; // K holds K^-1 and alpha is resident (after the gradient pass)
  bool fused_vt = false; // rows Np.. of K hold k_pX L^-T from a fused factorisation

  // ---- posterior state ------------------------------------------------------------------
  int M = 0, Mp = 0;   // test points riding along in the factorisation (Mp: padded rows)
  int cM = 0, cMp = 0; // covariance block: cov / chol / draws are formed for cM test points at a time
                       // (= M unless a sweep slices X_new, predict_in_batches)
  int64_t ldv = 0, ldc = 0;
  gpx::DevBuf Xnew;   // T x M x d
  gpx::DevBuf Vt;     // Mp x ldv : k_pX -> k_pX L^-T
  gpx::DevBuf Cov;    // Mp x ldc : posterior covariance -> its Cholesky factor (lower)
  gpx::DevBuf CovLinv;// (Mp/128) x 128 x 128
  gpx::DevBuf SplitK; // split-K partial slabs
  gpx::DevBuf mean;   // Mp
  gpx::DevBuf var;    // Mp
  gpx::DevBuf eps;    // n_pad x ldc
  gpx::DevBuf draws;  // n_pad x ldc
  double noise_p = 0;
  bool have_post = false;
  bool cov_factored = false;

  // ---- batched sweep state (gpx_predict_sweep / gpx_sweep_resident) -----------------------
  gpx::DevBuf st_eps, st_yres, st_means, st_samples, st_infos, st_vars, st_pred; // sweep I/O staging (grow-only)
  gpx::PinBuf pin_in, pin_out, pin_x;                                            // page-locked host side of it
  gpx::PinBuf pin_gen; // gpx_fit_batch, general sequence: [theta table | residuals | results] staged page-locked (round 6)
  gpx::PinBuf pin_fit; // small-N fit batches: [theta table | residuals | results] read and written by the kernel itself
  gpx::DevBuf thtab;   // S x ThetaDev
  gpx::DevBuf binfo;   // 2 x B ints (train / cov pivots of the batch in flight)
  gpx::DevBuf bscal;   // B x 32 doubles: lml pieces + gradient of every entry of a fit batch
  gpx::DevBuf byres;   // B x N residuals of a fit batch
  std::vector<double> h_bscal;
  std::vector<int> h_binfo;
  std::vector<gpx::ThetaDev> h_thtab;           // host image of thtab (outlives the async upload)
  int64_t sweep_batches = 0, sweep_samples = 0; // statistics: batches launched, samples processed
  int last_batch = 0;                           // samples per launch chosen by the last sweep

  void* sgp = nullptr; // sparse-GP state (sparse.hip)
  uint64_t train_gen = 0; // bumped by every upload of X

  // ---- generic scratch for unit-test entry points ---------------------------------------
  gpx::DevBuf tA, tB, tC;

  // ---- profiling ------------------------------------------------------------------------
  bool prof_on = false;
  gpx::ProfAcc prof[GPX_PROF_NCLASS];
};

namespace gpx {

inline int fail(gpx_ctx* ctx, const char* what, hipError_t e, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  if (ctx) ctx->err = buf;
  return -2;
}
inline int bad_arg(gpx_ctx* ctx, const char* msg) {
  if (ctx) ctx->err = std::string("bad argument: ") + msg;
  return -1;
}

#define GPX_HIP(ctx, expr)                                                   \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return gpx::fail((ctx), #expr, _e, __FILE__, __LINE__); \
  } while (0)

#define GPX_TRY(expr)      \
  do {                     \
    int _rc = (expr);      \
    if (_rc < 0) return _rc; \
  } while (0)

// roctx ranges around the pipeline stages (tracing) — `rocprofv3 --marker-trace` shows them on the host
// timeline.  The roctx library (librocprofiler-sdk-roctx.so.1, else libroctx64.so.4) is bound with dlopen the first
// time a range is opened and only when GPX_ROCTX=1, so an untraced run pays one branch per stage.
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
  explicit RoctxRange(const char* name) { roctx_push(name); }
  ~RoctxRange() { roctx_pop(); }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;
};

// Bracket a kernel launch with events when profiling is on.
struct ProfScope {
  gpx_ctx* ctx;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t st = nullptr;
  ProfScope(gpx_ctx* c, int cls_, double work, double bytes = 0.0) : ctx(c), cls(cls_) {
    if (!ctx->prof_on) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      a = b = nullptr;
      return;
    }
    ctx->prof[cls].launches += 1;
    ctx->prof[cls].work += work;
    ctx->prof[cls].bytes += bytes;
    st = ctx->s;
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    ctx->prof[cls].pending.emplace_back(a, b);
  }
};

// ---- kernels implemented across the .hip files ------------------------------------------

// gram.hip
int launch_gram(gpx_ctx* ctx, const KernelParams& kp, const double* dX, int n, const double* dZ,
                int m, double diag_add, int add_diag, int lower_only, double* dOut, int64_t ld);
// th != nullptr: batched launch, entry b uses th[b].kp and (diag_sel 1: diag_train, 2: diag_pred,
// 0: none) instead of the by-value kp / diag_add, and writes dOut + b * out_bs.
int launch_gram_padded(gpx_ctx* ctx, const KernelParams& kp, const double* dX, int n, int n_pad,
                       const double* dZ, int m, int m_pad, double diag_add, int add_diag,
                       int lower_only, double* dOut, int64_t ld, int batch = 1, int64_t out_bs = 0,
                       const ThetaDev* th = nullptr, int diag_sel = 0, TaskStride ts = TaskStride(),
                       const double* diag_vec = nullptr);
int launch_augment(gpx_ctx* ctx, double* dK, int64_t ld, int N, int Np, const double* dy,
                   int batch = 1, int64_t k_bs = 0, int64_t y_bs = 0, int y_mod = 0, int* dInfo = nullptr);
int launch_pad_identity(gpx_ctx* ctx, double* dA, int64_t ld, int n, int np);

// gemm_f64.hip
struct GemmArgs {
  const double* A;
  int64_t lda;
  const double* B;
  int64_t ldb;
  double* C;
  int64_t ldc;
  int K;       // reduction length, multiple of 16
  double alpha, beta;
  int lower;   // skip tiles with (tj_off + bx) > (ti_off + by)
  int ti_off, tj_off;
  int ktri;    // k range starts at (ti_off + by) * 128 (upper-triangular operands)
  int kupper;  // k range ends at (tj_off + bx + 1) * 128 (B lower triangular, e.g. chol factor)
  int kcol;    // k range starts at (tj_off + bx) * 128 (B upper triangular, e.g. L^-T as the right factor)
  int kchunk;  // split-K chunk (multiple of 16), 0 = no split
  int64_t c_split_stride;
  int nsplit;  // grid.z = nsplit * batch (filled in by launch_gemm_nt)
  int batch;   // independent problems per launch (0 or 1 = one); element b at base + b * *_bs
  int64_t a_bs, b_bs, c_bs;
  int batch2;  // > 1: two-level batch — entry = outer * batch2 + inner, at base + outer * *_bs + inner * *_bs2
  int64_t a_bs2, b_bs2, c_bs2;
  int latency_shape; // 1: take the 64 x 64 shapes whatever the tile count (many short, unequal k ranges: split-K of a triangular product)
  int big_shape;     // 1: take the 128 x 128 throughput shape whatever the tile count (tall operands: 256 tiles of K >= 256 fill the chip once)
};
int launch_gemm_nt(gpx_ctx* ctx, const GemmArgs& g, int tiles_m, int tiles_n, int splits,
                   int prof_cls, double work);
int mfma_peak(gpx_ctx* ctx, double* tflops);

// potf2.hip
struct GemmArgs;
int launch_potf2_trsm(gpx_ctx* ctx, double* dA, int64_t lda, double* dLinv, int* dInfo, int info_base, const GemmArgs& g,
                      int below);
int launch_potf2_inv(gpx_ctx* ctx, double* dA, int64_t lda, double* dLinv, int* dInfo,
                     int info_base, int batch = 1, int64_t a_bs = 0, int64_t linv_bs = 0);

// fit_small.hip: lml + gradient of `batch` hyper-parameter vectors at N <= 128 in one launch
int launch_fit_small(gpx_ctx* ctx, const KernelParams& kp, double diag_train, const ThetaDev* th, TaskStride ts,
                     const double* dX, int N, const double* dy, int64_t y_bs, int y_mod, double* dK, int64_t ldk,
                     int64_t k_bs, double* dLinv, int64_t linv_bs, double* dalpha, int64_t alpha_bs, double* dscal,
                     int64_t scal_bs, int* dinfo, int want_grad, int batch);

// linalg.hip
int potrf_lower(gpx_ctx* ctx, double* dA, int64_t lda, int np, int extra_tiles, double* dLinv,
                int* dInfo, int batch = 1, int64_t a_bs = 0, int64_t linv_bs = 0);
// diagonal blocks kb0 .. kb1-1 of a plain right-looking factorisation of nblk tile rows (potf2 + inverse, panel TRSM, K = 128
// update of everything to the right), queued on ctx->s: a caller that pipelines other work behind the chain (sparse.hip)
int potrf_steps(gpx_ctx* ctx, double* dA, int64_t lda, int nblk, int kb0, int kb1, double* dLinv, int* dInfo);
int trsm_right_lt(gpx_ctx* ctx, double* dB, int64_t ldb, int rows_p, const double* dL,
                  int64_t ldl, const double* dLinv, int nblk, int upper_rows, int batch = 1,
                  int64_t b_bs = 0, int64_t l_bs = 0, int64_t linv_bs = 0);
int linv_t_tree(gpx_ctx* ctx, double* dW, int64_t ldw, const double* dL, int64_t ldl, const double* dLinv, int nt,
                double* dS, int64_t lds, int batch = 1, int64_t w_bs = 0, int64_t l_bs = 0, int64_t linv_bs = 0,
                int64_t s_bs = 0);
int launch_set_identity(gpx_ctx* ctx, double* dA, int64_t ld, int np, int batch = 1, int64_t a_bs = 0);
int launch_lml_terms(gpx_ctx* ctx, const double* dL, int64_t ld, int N, double* dOut2, int batch = 1,
                     int64_t l_bs = 0, int64_t out_bs = 0);
int launch_rowdot(gpx_ctx* ctx, const double* dV, int64_t ldv, int rows, int cols,
                  const double* dw, double kdiag, double* dmean, double* dvar,
                  int col_start_by_row, int batch = 1, int64_t v_bs = 0, int64_t w_bs = 0,
                  int64_t out_bs = 0, const ThetaDev* th = nullptr, const double* pred_diag = nullptr,
                  int64_t pd_bs = 0);
int launch_grad_diag(gpx_ctx* ctx, const double* dKinv, int64_t ld, int N, const double* dalpha, double* dout);
int launch_cov_finalize(gpx_ctx* ctx, const KernelParams& kp, const double* dXnew, int M, int Mp,
                        const double* dPart, int splits, int64_t split_stride, int64_t ldp,
                        double diag_add, double* dCov, int64_t ldc, int batch = 1,
                        int64_t part_bs = 0, int64_t cov_bs = 0, const ThetaDev* th = nullptr,
                        TaskStride ts = TaskStride(), const double* pred_diag = nullptr, int64_t pd_bs = 0);
int launch_grad_contract(gpx_ctx* ctx, const KernelParams& kp, const double* dX, int N,
                         const double* dKinv, int64_t ld, const double* dalpha, double* dpart,
                         int* nblocks_out, int batch = 1, int64_t k_bs = 0, int64_t alpha_bs = 0,
                         const ThetaDev* th = nullptr, TaskStride ts = TaskStride());
int launch_grad_reduce(gpx_ctx* ctx, const double* dpart, int nblocks, int nvals, double* dout,
                       int batch = 1, int64_t out_bs = 0);
void sgp_release(gpx_ctx* ctx);
// api.hip: gpx_predict_sweep with device-resident inputs / outputs (node-level sweep, multi.hip)
int sweep_device_io(gpx_ctx* ctx, int kind, int S, const double* ells, const double* scales, const double* noises,
                    const double* d_X, int N, int d, const double* d_yres, int yres_rows, const double* d_Xnew, int M,
                    int noiseless, double jitter, const double* d_eps, int n, double* d_means, double* d_samples,
                    int* d_infos, double* d_vars, int m_slice);
int ctx_cov_block(const gpx_ctx* ctx);
int launch_add_mean(gpx_ctx* ctx, double* ddraws, int64_t ld, int n, int M, const double* dmean,
                    int batch = 1, int64_t d_bs = 0, int64_t mean_bs = 0);

} // namespace gpx
