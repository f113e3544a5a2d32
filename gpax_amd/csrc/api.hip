// api.hip — the C-ABI of libgpx (declared in include/gpx.h): host orchestration of the
// device-resident exact-GP pipeline.  Each entry point cites the gpax interface it replaces in
// gpx.h; this file only sequences kernels on the context's stream and moves data H<->D.
#include <dlfcn.h>

#include "common.h"

using namespace gpx;

namespace gpx {
namespace {
int (*g_roctx_push)(const char*) = nullptr;
int (*g_roctx_pop)() = nullptr;
int g_roctx_state = 0; // 0 = not looked at, 1 = bound, -1 = off / unavailable
void roctx_bind() {
  g_roctx_state = -1;
  const char* e = getenv("GPX_ROCTX");
  if (!(e && e[0] == '1')) return;
  for (const char* n : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"}) {
    if (void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) {
      g_roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      g_roctx_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (g_roctx_push && g_roctx_pop) {
        g_roctx_state = 1;
        return;
      }
    }
  }
}
} // namespace
void roctx_push(const char* name) {
  if (g_roctx_state == 0) roctx_bind();
  if (g_roctx_state == 1) (void)g_roctx_push(name);
}
void roctx_pop() {
  if (g_roctx_state == 1) (void)g_roctx_pop();
}
} // namespace gpx

namespace {

constexpr int SC_QUAD = 0, SC_SUMLOG = 1, SC_GRAD = 2; // doubles in ctx->scal
constexpr int SC_INT_OFF = 512;                         // ints start at byte 2048
constexpr int SI_TRAIN = 0, SI_COV = 1;
constexpr double LOG_2PI = 1.83787706640934548356;

inline int* sc_int(gpx_ctx* ctx) { return ctx->scal.i() + SC_INT_OFF; }

int ensure(gpx_ctx* ctx, DevBuf& b, size_t bytes) {
  hipError_t e = b.ensure(bytes);
  if (e != hipSuccess) return fail(ctx, "hipMalloc", e, __FILE__, __LINE__);
  return 0;
}

int set_theta(gpx_ctx* ctx, int kind, int d, const double* ell, double scale) {
  if (kind != GPX_KERNEL_RBF && kind != GPX_KERNEL_MATERN52 && kind != GPX_KERNEL_PERIODIC)
    return bad_arg(ctx, "kernel kind");
  if (d < 1 || d > GPX_MAX_DIM) return bad_arg(ctx, "input dimension must be 1..16");
  ctx->theta.kind = kind;
  ctx->theta.d = d;
  for (int c = 0; c < GPX_MAX_DIM; ++c) ctx->theta.inv_ell[c] = (c < d) ? 1.0 / ell[c] : 0.0;
  ctx->theta.scale = scale;
  ctx->theta.pi_over_p = (kind == GPX_KERNEL_PERIODIC) ? 3.14159265358979323846 / ell[d] : 0.0;
  return 0;
}

double kdiag_value(const KernelParams& kp) {
  if (kp.kind != GPX_KERNEL_MATERN52) return kp.scale;
  const double r = std::sqrt(MATERN_EPS);
  const double s5r = SQRT5 * r;
  return kp.scale * (1.0 + s5r) * std::exp(-s5r);
}

// ---- batch layout ------------------------------------------------------------------------------
// Split-K slabs of the posterior-covariance SYRK (lower tiles only): enough slabs to give the
// launch ~512 workgroups, each slab a multiple of 128 k-columns.
struct CovSplit {
  int splits, kchunk;
};
// Samples per launch the sweep aims for at this N (before the memory / S caps), r = 16384 / Np:
// 1 at N = 16384 (one sample's trailing update fills the chip), ~2 r^2 up to N ~ 4700 (7 at 8192: with 3 contexts
// in flight B = 15 measured 2 % slower), ~4 r^2 below (60 at 4096, 226 at 2048, 256 = cap from ~1900 down;
// N = 2048: 3 360 -> 3 520 posteriors/s, N = 4096 on one context: 916 -> 956 against 2 r^2).
int nominal_batch(const gpx_ctx* ctx) {
  const double r = 16384.0 / (double)ctx->Np;
  int B = (int)((r >= 3.5 ? 4.0 : 2.0) * r * r);
  return B < 1 ? 1 : (B > 256 ? 256 : B);
}

// `sweep`: the slab count is sized for the sweep's nominal batch at this N (a function of N only, so that a
// sweep's results do not depend on the batch size actually used); otherwise for a single sample.
CovSplit cov_split(const gpx_ctx* ctx, bool sweep = false) {
  const int nt = (ctx->N + TILE - 1) / TILE;
  const int mt = ctx->cMp / TILE;
  const int ktot = nt * TILE;
  const int lower_tiles = mt * (mt + 1) / 2 * (sweep ? nominal_batch(ctx) : 1);
  int splits = (512 + lower_tiles - 1) / lower_tiles;
  const int max_splits = ktot / 256 > 0 ? ktot / 256 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int kchunk = round_up((ktot + splits - 1) / splits, TILE);
  splits = (ktot + kchunk - 1) / kchunk;
  return CovSplit{splits, kchunk};
}

// Per-task input strides: training inputs for both Gram operands / k_pX's second operand, X_new for
// k_pX's first operand and k_pp.
TaskStride ts_train(const gpx_ctx* ctx) {
  TaskStride t;
  if (ctx->T > 1) {
    t.mod = ctx->T;
    t.x_bs = t.z_bs = (int64_t)ctx->N * ctx->d;
  }
  return t;
}
TaskStride ts_new(const gpx_ctx* ctx) {
  TaskStride t;
  if (ctx->T > 1) {
    t.mod = ctx->T;
    t.x_bs = (int64_t)ctx->M * ctx->d;
    t.z_bs = (int64_t)ctx->N * ctx->d;
  }
  return t;
}

// Plan over the context's own buffers: B samples per launch, sample b at base + b * stride.
// B = 1 with th = nullptr is the eager single-theta path of gpx_factor / gpx_posterior.
BatchPlan make_plan(gpx_ctx* ctx, int B, int n_pad, bool fused, bool sweep = false) {
  BatchPlan p;
  p.B = B;
  p.sweep = sweep;
  p.yres = ctx->yres.d();
  p.k_bs = (int64_t)(ctx->Np + (fused ? ctx->Mp : 0)) * ctx->ldk;
  p.linv_bs = (int64_t)(ctx->Np / TILE) * TILE * TILE;
  p.mean_bs = ctx->Mp;
  p.cov_bs = (int64_t)ctx->cMp * ctx->ldc;
  p.covlinv_bs = (int64_t)(ctx->cMp / TILE) * TILE * TILE;
  p.splitk_bs = (ctx->cMp > 0) ? (int64_t)cov_split(ctx, sweep).splits * p.cov_bs : 0;
  p.eps_bs = (int64_t)n_pad * ctx->ldc;
  p.info_train = sc_int(ctx) + SI_TRAIN;
  p.info_cov = sc_int(ctx) + SI_COV;
  p.scal = ctx->scal.d();
  p.scal_bs = 0;
  return p;
}

// Gram (lower tiles) + augmentation + blocked Cholesky + lml reductions; all async.
// fused: the k_pX rows of the resident X_new ride below the square matrix (rows Np ..), so the
// factorisation also leaves Vt = k_pX L^-T there (see potrf_lower).
int dev_factor(gpx_ctx* ctx, bool fused, const BatchPlan& bp, bool want_lml) {
  const int N = ctx->N, Np = ctx->Np, B = bp.B;
  const int extra = fused ? ctx->Mp / TILE : 0;
  GPX_TRY(ensure(ctx, ctx->K, (size_t)((B - 1) * bp.k_bs + (int64_t)(Np + (fused ? ctx->Mp : 0)) * ctx->ldk) *
                                  sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->Linv, (size_t)B * bp.linv_bs * sizeof(double)));
  double* K = ctx->K.d();
  ctx->small_grad_ready = false;
  if (ctx->fit_small && !fused && N <= TILE && !ctx->has_diag) {
    // one launch: the factor in K, its inverse in Linv, [quad, sumlog] AND the gradient (a few microseconds more than the
    // lml alone; gpx_lml_grad / the batch's dev_grad then has nothing left to do), alpha (fit_small.hip)
    RoctxRange r("gpx:fit_small (gram, potf2, lml, gradient: one launch)");
    GPX_TRY(ensure(ctx, ctx->alpha, (size_t)B * ctx->Np * sizeof(double)));
    GPX_TRY(launch_fit_small(ctx, ctx->theta, ctx->noise + ctx->jitter, bp.th, ts_train(ctx), ctx->X.d(), N, bp.yres,
                             bp.y_bs, bp.y_mod, K, ctx->ldk, bp.k_bs, ctx->Linv.d(), bp.linv_bs, ctx->alpha.d(), ctx->Np,
                             bp.scal, bp.scal_bs, bp.info_train, 1, B));
    ctx->small_grad_ready = true;
    ctx->factored = (B == 1);
    ctx->have_kinv = false;
    ctx->fused_vt = false;
    ctx->have_post = false;
    return 0;
  }
  {
    RoctxRange r("gpx:gram");
    GPX_TRY(launch_gram_padded(ctx, ctx->theta, ctx->X.d(), N, N, ctx->X.d(), N, Np,
                               ctx->noise + ctx->jitter, 1, 1, K, ctx->ldk, B, bp.k_bs, bp.th, 1, ts_train(ctx),
                               ctx->has_diag ? ctx->diagv.d() : nullptr));
    GPX_TRY(launch_augment(ctx, K, ctx->ldk, N, Np, bp.yres, B, bp.k_bs, bp.y_bs, bp.y_mod, bp.info_train));
  }
  RoctxRange r_potrf(fused ? "gpx:potrf+trsm(k_pX ride-along)" : "gpx:potrf");
  if (fused) {
    // k_pX = kernel(X_new, X_train, params, jitter=0.0): no diagonal term (gp.py:268)
    GPX_TRY(launch_gram_padded(ctx, ctx->theta, ctx->Xnew.d(), ctx->M, ctx->Mp, ctx->X.d(), N, Np, 0.0,
                               0, 0, K + (int64_t)Np * ctx->ldk, ctx->ldk, B, bp.k_bs, bp.th, 0, ts_new(ctx)));
  }
  // (the pivot report bp.info_train was cleared by the augmentation kernel)
  // N a multiple of 128 (every BASELINE size): the augmentation row opens a tile of its own (rows N .. N + 127 = [y | 1e300],
  // identity padding).  Nothing in it needs factoring — what the path reads is w = y L^-T in row N — so that tile rides
  // along below the square part like the k_pX rows instead of costing a diagonal-block step of the serial chain
  // (potf2 + TRSM + update: 5 -> 4 steps at N = 512).  Same operations on row N either way: bit-identical.
  const int Nf = ((N + TILE - 1) / TILE) * TILE; // order of the part that is factored
  GPX_TRY(potrf_lower(ctx, K, ctx->ldk, Nf, extra + (Np - Nf) / TILE, ctx->Linv.d(), bp.info_train, B, bp.k_bs, bp.linv_bs));
  if (want_lml) GPX_TRY(launch_lml_terms(ctx, K, ctx->ldk, N, bp.scal + SC_QUAD, B, bp.k_bs, bp.scal_bs));
  ctx->factored = (B == 1);
  ctx->have_kinv = false;
  ctx->fused_vt = fused;
  ctx->have_post = false;
  return 0;
}
int dev_factor(gpx_ctx* ctx, bool fused) { return dev_factor(ctx, fused, make_plan(ctx, 1, 0, fused), true); }

// L^-T (upper) into W, K^-1 = L^-T L^-1 (lower) over K, alpha = L^-T w, gradient contraction.
// Per sample b the results land in bp.scal + b * bp.scal_bs: [quad, sumlog, grad...].
int dev_grad(gpx_ctx* ctx, const BatchPlan& bp) {
  const int N = ctx->N, B = bp.B;
  if (ctx->small_grad_ready) { // fit_small.hip left the gradient and alpha behind the factorisation
    ctx->small_grad_ready = false;
    ctx->factored = false; // (as after the general path, whose K^-1 overwrites the factor)
    ctx->have_kinv = false;
    ctx->small_no_kinv = (B == 1); // the one-launch step never stores K^-1: gpx_lml_grad_diag re-runs the general sequence
    return 0;
  }
  ctx->small_no_kinv = false;
  const int nt = (N + TILE - 1) / TILE; // tiles that carry real rows (excludes a pure aug tile)
  const int n128 = nt * TILE;
  const int64_t w_bs = (int64_t)ctx->Np * ctx->ldk, alpha_bs = ctx->Np;
  GPX_TRY(ensure(ctx, ctx->W, (size_t)B * w_bs * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->alpha, (size_t)B * alpha_bs * sizeof(double)));
  double* W = ctx->W.d();
  double* K = ctx->K.d();
  RoctxRange r_grad("gpx:lml_grad (L^-T trsm, K^-1 syrk, contraction)");
  if (ctx->linvt_tree) {
    GPX_TRY(ensure(ctx, ctx->Wscr, (size_t)B * w_bs * sizeof(double)));
    GPX_TRY(linv_t_tree(ctx, W, ctx->ldk, K, ctx->ldk, ctx->Linv.d(), nt, ctx->Wscr.d(), ctx->ldk, B, w_bs, bp.k_bs,
                        bp.linv_bs, w_bs));
  } else {
    GPX_TRY(launch_set_identity(ctx, W, ctx->ldk, n128, B, w_bs));
    GPX_TRY(trsm_right_lt(ctx, W, ctx->ldk, nt, K, ctx->ldk, ctx->Linv.d(), nt, 1, B, w_bs, bp.k_bs, bp.linv_bs));
  }
  // alpha_i = sum_{k>=i} W[i][k] w[k], w = row N of the augmented factor (read before K is
  // overwritten by K^-1)
  GPX_TRY(launch_rowdot(ctx, W, ctx->ldk, N, N, K + (int64_t)N * ctx->ldk, 0.0, ctx->alpha.d(),
                        nullptr, 1, B, w_bs, bp.k_bs, alpha_bs, nullptr));
  {
    GemmArgs g{};
    g.A = W;
    g.lda = ctx->ldk;
    g.B = W;
    g.ldb = ctx->ldk;
    g.C = K;
    g.ldc = ctx->ldk;
    g.K = n128;
    g.alpha = 1.0;
    g.beta = 0.0;
    g.lower = 1;
    g.ktri = 1;
    g.batch = B;
    g.a_bs = w_bs;
    g.b_bs = w_bs;
    g.c_bs = bp.k_bs;
    const double n = (double)n128;
    // tiles of very different length (k range [row, N)) and nothing else in flight: persistent, dynamically scheduled
    if (ctx->persist_scope_ok) ctx->persist_scope += 1;
    const int rc_kinv = launch_gemm_nt(ctx, g, nt, nt, 0, GPX_PROF_GEMM_OTHER, n * n * n / 3.0);
    if (ctx->persist_scope_ok) ctx->persist_scope -= 1;
    GPX_TRY(rc_kinv);
  }
  const int nt64 = (N + 63) / 64;
  GPX_TRY(ensure(ctx, ctx->part, (size_t)B * nt64 * (nt64 + 1) / 2 * (GPX_MAX_DIM + 3) * sizeof(double)));
  int nblocks = 0;
  GPX_TRY(launch_grad_contract(ctx, ctx->theta, ctx->X.d(), N, K, ctx->ldk, ctx->alpha.d(),
                               ctx->part.d(), &nblocks, B, bp.k_bs, alpha_bs, bp.th, ts_train(ctx)));
  GPX_TRY(launch_grad_reduce(ctx, ctx->part.d(), nblocks, n_ell(ctx->theta) + 2, bp.scal + SC_GRAD, B, bp.scal_bs));
  ctx->factored = false; // K now holds K^-1
  ctx->small_grad_ready = false;
  ctx->have_kinv = (B == 1);
  return 0;
}
int dev_grad(gpx_ctx* ctx) { return dev_grad(ctx, make_plan(ctx, 1, 0, false)); }

// Xnew: (T, M, d) — T = ctx->T task-specific test sets (T = 1: the usual (M, d)).
// on_device: Xnew already lives on this context's GPU (node-level sweep: it arrived by RCCL broadcast).
int set_xnew(gpx_ctx* ctx, const double* Xnew, int M, bool on_device = false) {
  if (M < 1) return bad_arg(ctx, "M must be >= 1");
  ctx->M = M;
  ctx->Mp = round_up(M, TILE);
  ctx->cM = M;
  ctx->cMp = ctx->Mp;
  ctx->ldv = pick_ld(ctx->Np);
  ctx->ldc = pick_ld(ctx->cMp);
  GPX_TRY(ensure(ctx, ctx->Xnew, (size_t)ctx->T * M * ctx->d * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->mean, (size_t)ctx->Mp * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->var, (size_t)ctx->Mp * sizeof(double)));
  const size_t xb = (size_t)ctx->T * M * ctx->d * sizeof(double);
  if (on_device) {
    GPX_HIP(ctx, hipMemcpyAsync(ctx->Xnew.d(), Xnew, xb, hipMemcpyDeviceToDevice, ctx->stream));
  } else if (xb > (64u << 10)) { // large X_new (predict_in_batches grids): through page-locked staging, see DESIGN.md 10
    GPX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the previous use of the staging buffer has been consumed
    GPX_HIP(ctx, ctx->pin_x.ensure(xb));
    std::memcpy(ctx->pin_x.p, Xnew, xb);
    GPX_HIP(ctx, hipMemcpyAsync(ctx->Xnew.d(), ctx->pin_x.p, xb, hipMemcpyHostToDevice, ctx->stream));
  } else {
    GPX_HIP(ctx, hipMemcpyAsync(ctx->Xnew.d(), Xnew, xb, hipMemcpyHostToDevice, ctx->stream));
  }
  return 0;
}

// k_pX -> Vt = k_pX L^-T, mean = Vt w, var (all M test points; skipped when !want_mean), and (optionally)
// cov = k_pp - Vt Vt^T for the covariance block of mc test points starting at m0 (default: all of them).
int dev_posterior(gpx_ctx* ctx, bool want_cov, const BatchPlan& bp, bool want_mean = true, int m0 = 0, int mc = -1) {
  const int N = ctx->N, M = ctx->M, Mp = ctx->Mp, B = bp.B;
  const int nt = (N + TILE - 1) / TILE;
  const int mt = Mp / TILE;
  if (mc < 0) mc = ctx->cM;
  const int cMp = ctx->cMp, cmt = cMp / TILE;
  const double* K = ctx->K.d();
  KernelParams kp = ctx->theta;
  RoctxRange r_post(want_cov ? "gpx:posterior (trsm, mean, cov syrk)" : "gpx:posterior (trsm, mean, var)");
  double* Vt;
  int64_t ldv, v_bs;
  if (ctx->fused_vt) { // already solved during the factorisation
    Vt = ctx->K.d() + (int64_t)ctx->Np * ctx->ldk;
    ldv = ctx->ldk;
    v_bs = bp.k_bs;
  } else {
    if (B != 1) return bad_arg(ctx, "batched posterior needs the fused factorisation");
    GPX_TRY(ensure(ctx, ctx->Vt, (size_t)ctx->Mp * ctx->ldv * sizeof(double)));
    Vt = ctx->Vt.d();
    ldv = ctx->ldv;
    v_bs = 0;
    // k_pX = kernel(X_new, X_train, params, jitter=0.0): no diagonal term (gp.py:268)
    GPX_TRY(launch_gram_padded(ctx, kp, ctx->Xnew.d(), M, Mp, ctx->X.d(), N, nt * TILE, 0.0, 0, 0, Vt, ldv, 1, 0,
                               bp.th, 0));
    GPX_TRY(trsm_right_lt(ctx, Vt, ldv, mt, K, ctx->ldk, ctx->Linv.d(), nt, 0));
  }
  if (want_mean) {
    GPX_TRY(ensure(ctx, ctx->mean, (size_t)B * bp.mean_bs * sizeof(double)));
    GPX_TRY(ensure(ctx, ctx->var, (size_t)B * bp.mean_bs * sizeof(double)));
    const double kd = kdiag_value(kp) + ctx->noise_p + ctx->jitter;
    GPX_TRY(launch_rowdot(ctx, Vt, ldv, M, N, K + (int64_t)N * ctx->ldk, kd, ctx->mean.d(), ctx->var.d(), 0, B, v_bs,
                          bp.k_bs, bp.mean_bs, bp.th, bp.pred_diag, bp.pd_bs));
  }
  ctx->cov_factored = false;
  if (want_cov) {
    const CovSplit cs = cov_split(ctx, bp.sweep);
    const int ktot = nt * TILE;
    const int64_t ldp = ctx->ldc;
    const int64_t stride = (int64_t)cMp * ldp;
    const double* Vs = Vt + (int64_t)m0 * ldv; // rows of this covariance block
    GPX_TRY(ensure(ctx, ctx->Cov, (size_t)B * bp.cov_bs * sizeof(double)));
    GPX_TRY(ensure(ctx, ctx->SplitK, (size_t)B * cs.splits * stride * sizeof(double)));
    GemmArgs g{};
    g.A = Vs;
    g.lda = ldv;
    g.B = Vs;
    g.ldb = ldv;
    g.C = ctx->SplitK.d();
    g.ldc = ldp;
    g.K = ktot;
    g.alpha = 1.0;
    g.beta = 0.0;
    g.lower = 1;
    g.kchunk = cs.kchunk;
    g.c_split_stride = stride;
    g.batch = B;
    g.a_bs = v_bs;
    g.b_bs = v_bs;
    g.c_bs = bp.splitk_bs;
    const double m = (double)cMp;
    GPX_TRY(launch_gemm_nt(ctx, g, cmt, cmt, cs.splits, GPX_PROF_GEMM_OTHER, m * (m + 1.0) * ktot));
    GPX_TRY(launch_cov_finalize(ctx, kp, ctx->Xnew.d() + (int64_t)m0 * ctx->d, mc, cMp, ctx->SplitK.d(), cs.splits,
                                stride, ldp, ctx->noise_p + ctx->jitter, ctx->Cov.d(), ctx->ldc, B, bp.splitk_bs,
                                bp.cov_bs, bp.th, ts_new(ctx), bp.pred_diag ? bp.pred_diag + m0 : nullptr, bp.pd_bs));
  }
  (void)mt;
  ctx->have_post = want_cov && B == 1;
  return 0;
}
int dev_posterior(gpx_ctx* ctx, bool want_cov) {
  return dev_posterior(ctx, want_cov, make_plan(ctx, 1, 0, ctx->fused_vt));
}

// chol(cov) (once per posterior) and draws = mean + eps Lc^T; eps already on device, padded.
// m0 / mc: the covariance block the resident Cov belongs to (its mean starts at mean + m0).
int dev_draw(gpx_ctx* ctx, int n_pad, int n, const BatchPlan& bp, int m0 = 0, int mc = -1) {
  const int Mp = ctx->cMp, mt = Mp / TILE, B = bp.B;
  if (mc < 0) mc = ctx->cM;
  RoctxRange r_draw("gpx:mvn_draw (chol(cov), L eps)");
  if (!ctx->cov_factored) {
    GPX_TRY(ensure(ctx, ctx->CovLinv, (size_t)B * bp.covlinv_bs * sizeof(double)));
    // first block resets the pivot report; later blocks keep the first failure (potf2 only writes a zero slot)
    if (m0 == 0) GPX_HIP(ctx, hipMemsetAsync(bp.info_cov, 0, (size_t)B * sizeof(int), ctx->stream));
    GPX_TRY(potrf_lower(ctx, ctx->Cov.d(), ctx->ldc, Mp, 0, ctx->CovLinv.d(), bp.info_cov, B, bp.cov_bs,
                        bp.covlinv_bs));
    ctx->cov_factored = (B == 1);
  }
  GemmArgs g{};
  g.A = ctx->eps.d();
  g.lda = ctx->ldc;
  g.B = ctx->Cov.d();
  g.ldb = ctx->ldc;
  g.C = ctx->draws.d();
  g.ldc = ctx->ldc;
  g.K = Mp;
  g.alpha = 1.0;
  g.beta = 0.0;
  g.kupper = 1;
  g.batch = B;
  g.a_bs = bp.eps_bs;
  g.b_bs = bp.cov_bs;
  g.c_bs = bp.eps_bs;
  GPX_TRY(launch_gemm_nt(ctx, g, n_pad / TILE, mt, 0, GPX_PROF_GEMM_OTHER,
                         2.0 * n_pad * (double)Mp * Mp / 2.0));
  GPX_TRY(launch_add_mean(ctx, ctx->draws.d(), ctx->ldc, n, mc, ctx->mean.d() + m0, B, bp.eps_bs, bp.mean_bs));
  return 0;
}
int dev_draw(gpx_ctx* ctx, int n_pad, int n) { return dev_draw(ctx, n_pad, n, make_plan(ctx, 1, n_pad, ctx->fused_vt)); }

int upload_eps(gpx_ctx* ctx, const double* eps, int n, int* n_pad_out) {
  const int n_pad = round_up(n, TILE);
  *n_pad_out = n_pad;
  GPX_TRY(ensure(ctx, ctx->eps, (size_t)n_pad * ctx->ldc * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->draws, (size_t)n_pad * ctx->ldc * sizeof(double)));
  GPX_HIP(ctx, hipMemsetAsync(ctx->eps.d(), 0, (size_t)n_pad * ctx->ldc * sizeof(double), ctx->stream));
  if (eps) {
    GPX_HIP(ctx, hipMemcpy2DAsync(ctx->eps.d(), ctx->ldc * sizeof(double), eps,
                                  (size_t)ctx->M * sizeof(double), (size_t)ctx->M * sizeof(double), n,
                                  hipMemcpyHostToDevice, ctx->stream));
  }
  return 0;
}

// ---- batched predictive sweep ---------------------------------------------------------------
// The vmap of ExactGP.predict (gpax/models/gp.py:393-395) as a grid dimension: B theta samples
// advance through the same launch sequence together (grid.z / one potf2 workgroup each), so a
// launch carries B times the work and the serial chain of small kernels is paid once per batch.
// Each sample's arithmetic is the single-sample arithmetic (same kernels, same tile shapes, same
// accumulation order): results do not depend on B.

// eps (S, n, Mtot) contiguous -> per-sample padded slabs dst[b][n_pad][ldc] of the covariance block
// [m0, m0 + mc) of the test points
__global__ __launch_bounds__(256) void eps_gather_kernel(double* __restrict__ dst, int64_t ldc,
                                                         int64_t eps_bs, const double* __restrict__ src,
                                                         int n, int Mtot, int m0, int mc) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y, b = blockIdx.z;
  if (a < mc) dst[(int64_t)b * eps_bs + (int64_t)r * ldc + a] = src[((int64_t)b * n + r) * Mtot + m0 + a];
}

// per-batch means / variances / pivot reports -> the sweep's contiguous outputs (b, M), (b, 2)
__global__ __launch_bounds__(256) void sweep_store_kernel(const double* __restrict__ mean, int64_t mean_bs, int M,
                                                          const int* __restrict__ info_train,
                                                          const int* __restrict__ info_cov, int has_cov,
                                                          double* __restrict__ means, int* __restrict__ infos,
                                                          const double* __restrict__ var,
                                                          double* __restrict__ vars) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.z;
  if (a < M) {
    means[(int64_t)b * M + a] = mean[(int64_t)b * mean_bs + a];
    if (vars != nullptr) vars[(int64_t)b * M + a] = var[(int64_t)b * mean_bs + a];
  }
  if (a == 0 && infos != nullptr) {
    infos[2 * b] = info_train[b];
    infos[2 * b + 1] = has_cov ? info_cov[b] : 0;
  }
}

// draws of one covariance block -> samples (b, n, Mtot)[.., m0 : m0 + mc]
__global__ __launch_bounds__(256) void draws_store_kernel(const double* __restrict__ draws, int64_t ldc,
                                                          int64_t eps_bs, int n, int Mtot, int m0, int mc,
                                                          double* __restrict__ samples) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y, b = blockIdx.z;
  if (a < mc) samples[((int64_t)b * n + r) * Mtot + m0 + a] = draws[(int64_t)b * eps_bs + (int64_t)r * ldc + a];
}

struct SweepIO {
  int kind = 0, S = 0, n = 0, noiseless = 0;
  double jitter = 0.0;
  const double *ells = nullptr, *scales = nullptr, *noises = nullptr; // host tables
  const double* dYres = nullptr; // device (S, N) or nullptr: ctx->yres shared by all samples
  int y_mod = 0;                 // > 0: dYres is (y_mod, N), sample s reads row s % y_mod (per-task residuals)
  const double* dEps = nullptr;  // device (S, n, M) or nullptr: whatever is resident in ctx->eps
  double* dMeans = nullptr;      // device outputs (nullptr: results stay in the batch buffers)
  double* dSamples = nullptr;
  int* dInfos = nullptr;
  double* dVars = nullptr; // device (S, M) posterior variances (diag of cov) or nullptr
  const double* dPredDiag = nullptr; // device (S, M): per-sample variances added to diag(cov_s) / var_s
  int m_slice = 0; // > 0: covariances / draws per block of m_slice test points (predict_in_batches semantics)
};

// Samples per launch: enough that the small-N pipeline fills the chip, bounded by memory.
int pick_batch(gpx_ctx* ctx, int S, int n_pad, bool want_cov) {
  int forced = 0;
  if (const char* e = getenv("GPX_SWEEP_BATCH")) forced = atoi(e);
  // auto: nominal_batch() — measured in tools/small_n_sweep.py / multi_ctx.py / c4_sweep.py
  int B = forced > 0 ? forced : nominal_batch(ctx);
  // nominal_batch() stops at 256 (the split-K slab counts of a sweep are sized for it: a function of N only).  Where that
  // cap binds (N < ~1900) a launch of 256 samples is still partly latency: up to 1024 samples per launch below N = 1152
  // (+8 ... 13 % posteriors/s at N = 128 ... 512; the memory budget below still applies).  A sample's arithmetic does not
  // depend on the batch it rides in.
  if (forced <= 0 && B == 256 && ctx->Np <= 1152) B = 1024;
  if (B > 1024) B = 1024;
  const BatchPlan p = make_plan(ctx, 1, n_pad, true, true);
  double per = (double)p.k_bs + p.linv_bs + 2.0 * p.mean_bs;
  if (want_cov) per += (double)p.cov_bs + p.splitk_bs + p.covlinv_bs + 2.0 * p.eps_bs;
  per *= sizeof(double);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    double budget = (double)free_b / 3.0;
    const double cap = 24.0 * 1024 * 1024 * 1024;
    if (budget > cap) budget = cap;
    // buffers this context already holds are reused, not added
    budget += (double)ctx->K.cap;
    const int fit = (int)(budget / per);
    if (B > fit) B = fit;
  }
  if (B > S) B = S;
  if (B < 1) B = 1;
  if (ctx->T > 1) { // batches start on a task boundary: entry b of a batch is task b % T
    B -= B % ctx->T;
    if (B < ctx->T) B = ctx->T;
  }
  return B;
}

int fill_theta_table(gpx_ctx* ctx, const SweepIO& io) {
  const int d = ctx->d;
  const int stride = d + (io.kind == GPX_KERNEL_PERIODIC ? 1 : 0);
  ctx->h_thtab.resize((size_t)io.S);
  const KernelParams saved = ctx->theta;
  for (int s = 0; s < io.S; ++s) {
    int rc = set_theta(ctx, io.kind, d, io.ells + (int64_t)s * stride, io.scales[s]);
    if (rc < 0) {
      ctx->theta = saved;
      return rc;
    }
    ThetaDev& t = ctx->h_thtab[(size_t)s];
    t.kp = ctx->theta;
    const double noise_p = io.noiseless ? 0.0 : io.noises[s];
    t.diag_train = io.noises[s] + io.jitter;
    t.diag_pred = noise_p + io.jitter;
    t.kdiag_pred = kdiag_value(t.kp) + noise_p + io.jitter;
  }
  // ctx->theta keeps the structural fields (kind, d) the launchers dispatch on
  GPX_TRY(ensure(ctx, ctx->thtab, (size_t)io.S * sizeof(ThetaDev)));
  GPX_HIP(ctx, hipMemcpyAsync(ctx->thtab.p, ctx->h_thtab.data(), (size_t)io.S * sizeof(ThetaDev),
                              hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// Enqueue the whole sweep on the context's streams (asynchronous; the caller synchronises).
int sweep_core(gpx_ctx* ctx, const SweepIO& io) {
  const int N = ctx->N, M = ctx->M, n = io.n, S = io.S;
  const int n_pad = round_up(n > 0 ? n : 1, TILE);
  ctx->jitter = io.jitter;
  // covariance blocks: all M test points at once, or slices of m_slice points that share ONE factorisation
  // per sample (ExactGP.predict_in_batches, gp.py:325-349, re-factorises K for every slice)
  const int Ms = (io.m_slice > 0 && io.m_slice < M) ? io.m_slice : M;
  const int J = (M + Ms - 1) / Ms;
  ctx->cM = Ms;
  ctx->cMp = round_up(Ms, TILE);
  ctx->ldc = pick_ld(ctx->cMp);
  // ride-along rows must cover the (padded) last block: rows beyond M are zero
  ctx->Mp = round_up(M, TILE);
  if ((J - 1) * Ms + ctx->cMp > ctx->Mp) ctx->Mp = round_up((J - 1) * Ms + ctx->cMp, TILE);
  GPX_TRY(fill_theta_table(ctx, io));
  const int B = pick_batch(ctx, S, n_pad, n > 0);
  BatchPlan bp = make_plan(ctx, B, n_pad, true, true);
  GPX_TRY(ensure(ctx, ctx->binfo, (size_t)2 * B * sizeof(int)));
  bp.info_train = ctx->binfo.i();
  bp.info_cov = ctx->binfo.i() + B;
  if (n > 0) {
    const size_t eb = (size_t)B * bp.eps_bs * sizeof(double);
    const bool grew = ctx->eps.cap < eb;
    GPX_TRY(ensure(ctx, ctx->eps, eb));
    GPX_TRY(ensure(ctx, ctx->draws, eb));
    // padding rows / columns of eps must be zero; resident eps (no dEps) is kept unless reallocated
    if (io.dEps != nullptr || grew) GPX_HIP(ctx, hipMemsetAsync(ctx->eps.p, 0, eb, ctx->stream));
  }
  const ThetaDev* table = static_cast<const ThetaDev*>(ctx->thtab.p);
  for (int s0 = 0; s0 < S; s0 += B) {
    const int b = (S - s0 < B) ? S - s0 : B;
    bp.B = b;
    bp.th = table + s0;
    if (io.dYres != nullptr && io.y_mod > 0) {
      bp.yres = io.dYres; // s0 is a multiple of y_mod (= T)
      bp.y_bs = N;
      bp.y_mod = io.y_mod;
    } else if (io.dYres != nullptr) {
      bp.yres = io.dYres + (int64_t)s0 * N;
      bp.y_bs = N;
    } else {
      bp.yres = ctx->yres.d();
      bp.y_bs = 0;
    }
    bp.pred_diag = io.dPredDiag ? io.dPredDiag + (int64_t)s0 * M : nullptr;
    bp.pd_bs = M;
    GPX_TRY(dev_factor(ctx, true, bp, false));
    GPX_TRY(dev_posterior(ctx, false, bp)); // means / variances of all M test points
    for (int j = 0; j < J && n > 0; ++j) {
      const int m0 = j * Ms, mc = (M - m0 < Ms) ? M - m0 : Ms;
      if (io.dEps != nullptr) {
        if (mc < Ms) // ragged last block: clear the columns the previous block filled
          GPX_HIP(ctx, hipMemsetAsync(ctx->eps.p, 0, (size_t)b * bp.eps_bs * sizeof(double), ctx->stream));
        dim3 grid((mc + 255) / 256, n, b);
        eps_gather_kernel<<<grid, 256, 0, ctx->stream>>>(ctx->eps.d(), ctx->ldc, bp.eps_bs,
                                                          io.dEps + (int64_t)s0 * n * M, n, M, m0, mc);
        GPX_HIP(ctx, hipGetLastError());
      }
      GPX_TRY(dev_posterior(ctx, true, bp, false, m0, mc));
      GPX_TRY(dev_draw(ctx, n_pad, n, bp, m0, mc));
      if (io.dSamples != nullptr) {
        dim3 grid((mc + 255) / 256, n, b);
        draws_store_kernel<<<grid, 256, 0, ctx->stream>>>(ctx->draws.d(), ctx->ldc, bp.eps_bs, n, M, m0, mc,
                                                           io.dSamples + (int64_t)s0 * n * M);
        GPX_HIP(ctx, hipGetLastError());
      }
    }
    if (io.dMeans != nullptr) {
      dim3 grid((M + 255) / 256, 1, b);
      sweep_store_kernel<<<grid, 256, 0, ctx->stream>>>(
          ctx->mean.d(), bp.mean_bs, M, bp.info_train, bp.info_cov, n > 0 ? 1 : 0, io.dMeans + (int64_t)s0 * M,
          io.dInfos ? io.dInfos + 2 * s0 : nullptr, ctx->var.d(), io.dVars ? io.dVars + (int64_t)s0 * M : nullptr);
      GPX_HIP(ctx, hipGetLastError());
    }
    ctx->sweep_batches += 1;
    ctx->sweep_samples += b;
  }
  ctx->last_batch = B;
  // the context no longer holds a single factored theta
  ctx->factored = false;
  ctx->small_grad_ready = false;
  ctx->have_post = false;
  ctx->cov_factored = false;
  return 0;
}

void drain_profile(gpx_ctx* ctx) {
  for (int c = 0; c < GPX_PROF_NCLASS; ++c) {
    for (auto& pr : ctx->prof[c].pending) {
      float ms = 0.f;
      if (hipEventSynchronize(pr.second) == hipSuccess &&
          hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
        ctx->prof[c].ms += ms;
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
    ctx->prof[c].pending.clear();
  }
}

} // namespace

extern "C" {

int gpx_init(int device, gpx_ctx** out) {
  if (!out) return -1;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  gpx_ctx* ctx = new gpx_ctx();
  *out = ctx; // returned even on failure so the caller can read gpx_last_error
  if (e != hipSuccess || count <= 0) {
    ctx->err = std::string("no HIP device available: ") + hipGetErrorString(e);
    return -2;
  }
  if (device < 0 || device >= count) {
    ctx->err = "device ordinal out of range";
    return -1;
  }
  GPX_HIP(ctx, hipSetDevice(device));
  ctx->device = device;
  GPX_HIP(ctx, hipGetDeviceProperties(&ctx->prop, device));
  if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    ctx->err = std::string("libgpx is built for gfx950 only; device is ") + ctx->prop.gcnArchName;
    return -3;
  }
  {
    int lo = 0, hi = 0; // numerically lower = higher priority
    GPX_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
    // The library's switches (12 in all, this list + GPX_ROCTX, GPX_SWEEP_BATCH, GPX_NODE_TRANSPORT, GPX_RANK_FILE_TIMEOUT):
    // each selects the ONE alternate of a kernel / schedule that the bit-identity tests compare the default against.
    // Everything else rounds 1 - 3 measured and rejected (CU reservations, early diagonal, split U1 / far updates,
    // XCD-aware tile order, cooperative panel kernel, column / blocked potf2) left the library in round 4:
    // profiles/r04/pruned.md names the commit each one last lived in.
    if (const char* e = getenv("GPX_PERSIST_SCOPE")) ctx->persist_scope_ok = (e[0] != '0');
    if (const char* e = getenv("GPX_LAZY_GROUP")) {
      const int lg = atoi(e);
      if (lg >= 1 && lg <= 16) ctx->lazy_group = lg;
    }
    if (const char* e = getenv("GPX_OUTER_TILES")) {
      const int ot = atoi(e);
      if (ot >= 1 && ot <= 128) {
        ctx->outer_tiles = ot;
        ctx->outer_tiles_set = true;
      }
    }
    if (const char* e = getenv("GPX_TAIL_TILES")) ctx->tail_tiles = atoi(e);
    GPX_HIP(ctx, hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, lo));
    GPX_HIP(ctx, hipStreamCreateWithPriority(&ctx->pstream, hipStreamNonBlocking, hi));
    GPX_HIP(ctx, hipStreamCreateWithPriority(&ctx->xstream, hipStreamNonBlocking, lo));
    GPX_HIP(ctx, hipEventCreateWithFlags(&ctx->evD, hipEventDisableTiming));
    if (const char* e = getenv("GPX_POTF2")) {
      if (gpx_debug_set_potf2(ctx, e) != 0) return -1;
    }
    if (const char* e = getenv("GPX_POTF2_TRSM")) ctx->potf2_trsm = (e[0] != '0');
    {
      // the one-launch fit step wants up to ~100 KB of dynamic LDS per workgroup, the fused potf2 + TRSM launch 128 KB
      // (gfx950: 160 KB per CU): a device that offers less keeps the general launch sequences
      int optin = 0;
      if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) optin = 0;
      const size_t lds_max = std::max((size_t)(optin > 0 ? optin : 0), (size_t)ctx->prop.maxSharedMemoryPerMultiProcessor);
      if (lds_max < (size_t)100 * 1024) ctx->fit_small = false;
      if (lds_max < (size_t)128 * 1024) ctx->potf2_trsm = false;
    }
    if (const char* e = getenv("GPX_FIT_SMALL")) ctx->fit_small = (e[0] != '0');
    if (const char* e = getenv("GPX_LAT_LIN")) ctx->lat_lin = (e[0] != '0');
    if (const char* e = getenv("GPX_LAT_GEMM")) ctx->lat_gemm = (std::strcmp(e, "r1") == 0) ? 1 : ((std::strcmp(e, "r5") == 0) ? 5 : 0);
    if (const char* e = getenv("GPX_SMALL_BK")) ctx->small_bk = (atoi(e) == 32) ? 32 : (atoi(e) == 16 ? 16 : 0);
    if (const char* e = getenv("GPX_LINVT")) ctx->linvt_tree = (std::strcmp(e, "sweep") == 0) ? 0 : 1;
    if (const char* e = getenv("GPX_SGP_SOLVE"))
      ctx->sgp_inverse = (std::strcmp(e, "sweep") == 0) ? 0 : ((std::strcmp(e, "ride") == 0) ? 2 : 1);
    ctx->s = ctx->stream;
  }
  GPX_HIP(ctx, hipEventCreate(&ctx->ev0));
  GPX_HIP(ctx, hipEventCreate(&ctx->ev1));
  GPX_TRY(ensure(ctx, ctx->scal, 8192));
  GPX_HIP(ctx, hipMemsetAsync(ctx->scal.p, 0, 8192, ctx->stream));
  return 0;
}

int gpx_device_count(void) {
  int count = 0;
  return (hipGetDeviceCount(&count) == hipSuccess) ? count : 0;
}

int gpx_device_pci(int device, int* domain, int* bus, int* dev) {
  int d = 0, b = 0, v = 0;
  if (hipDeviceGetAttribute(&d, hipDeviceAttributePciDomainID, device) != hipSuccess ||
      hipDeviceGetAttribute(&b, hipDeviceAttributePciBusId, device) != hipSuccess ||
      hipDeviceGetAttribute(&v, hipDeviceAttributePciDeviceId, device) != hipSuccess)
    return -2;
  if (domain) *domain = d;
  if (bus) *bus = b;
  if (dev) *dev = v;
  return 0;
}

void gpx_destroy(gpx_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) {
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    drain_profile(ctx);
    DevBuf* bufs[] = {&ctx->X,    &ctx->K,   &ctx->W,       &ctx->Linv,   &ctx->yres, &ctx->scal,
                      &ctx->part, &ctx->alpha, &ctx->Xnew,  &ctx->Vt,     &ctx->Cov,  &ctx->CovLinv,
                      &ctx->SplitK, &ctx->mean, &ctx->var,  &ctx->eps,    &ctx->draws, &ctx->tA,
                      &ctx->tB,   &ctx->tC,  &ctx->thtab,   &ctx->binfo, &ctx->bscal, &ctx->byres, &ctx->diagv, &ctx->st_eps, &ctx->st_yres,
                      &ctx->st_means, &ctx->st_samples, &ctx->st_infos, &ctx->st_vars, &ctx->st_pred, &ctx->tile_counters, &ctx->chain_flag};
    ctx->pin_gen.release();
    for (DevBuf* b : bufs) b->release();
    ctx->pin_in.release();
    ctx->pin_out.release();
    ctx->pin_x.release();
    ctx->pin_fit.release();
    sgp_release(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->evP) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->evU) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->tile_counter_ev)
      if (e) (void)hipEventDestroy(e);
    if (ctx->evD) (void)hipEventDestroy(ctx->evD);
    if (ctx->pstream) (void)hipStreamDestroy(ctx->pstream);
    if (ctx->xstream) (void)hipStreamDestroy(ctx->xstream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  }
  delete ctx;
}

const char* gpx_last_error(const gpx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gpx_device_info(gpx_ctx* ctx, char* name, int name_len, int* num_cu, int64_t* hbm_bytes,
                    int* clock_khz) {
  if (!ctx || ctx->device < 0) return -1;
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  }
  if (num_cu) *num_cu = ctx->prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
  if (clock_khz) *clock_khz = ctx->prop.clockRate;
  return 0;
}

int gpx_synchronize(gpx_ctx* ctx) {
  if (!ctx || ctx->device < 0) return -1;
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int gpx_gram(gpx_ctx* ctx, int kind, const double* X, int n, const double* Z, int m, int d,
             const double* ell, double scale, double diag_add, int add_diag, double* out) {
  if (!ctx || ctx->device < 0) return -1;
  if (n < 0 || m < 0) return bad_arg(ctx, "negative size");
  if (n == 0 || m == 0) return 0;
  if (!X || !Z || !ell || !out) return bad_arg(ctx, "null pointer");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  KernelParams kp{};
  {
    KernelParams saved = ctx->theta;
    // GPX_KERNEL_R2: the lengthscales as for the RBF kernel; only this entry point knows the kind
    GPX_TRY(set_theta(ctx, kind == GPX_KERNEL_R2 ? GPX_KERNEL_RBF : kind, d, ell, scale));
    kp = ctx->theta;
    if (kind == GPX_KERNEL_R2) kp.kind = GPX_KERNEL_R2;
    ctx->theta = saved;
  }
  const int64_t ld = pick_ld(m);
  GPX_TRY(ensure(ctx, ctx->tA, (size_t)n * d * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->tB, (size_t)m * d * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->tC, (size_t)n * ld * sizeof(double)));
  GPX_HIP(ctx, hipMemcpyAsync(ctx->tA.d(), X, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  GPX_HIP(ctx, hipMemcpyAsync(ctx->tB.d(), Z, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  GPX_TRY(launch_gram(ctx, kp, ctx->tA.d(), n, ctx->tB.d(), m, diag_add, add_diag, 0, ctx->tC.d(), ld));
  GPX_HIP(ctx, hipMemcpy2DAsync(out, (size_t)m * sizeof(double), ctx->tC.d(), ld * sizeof(double),
                                (size_t)m * sizeof(double), n, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

static int set_train_impl(gpx_ctx* ctx, const double* X, int T, int N, int d, bool on_device) {
  if (!ctx || ctx->device < 0) return -1;
  if (!X) return bad_arg(ctx, "null X");
  if (N < 1) return bad_arg(ctx, "N must be >= 1");
  if (T < 1) return bad_arg(ctx, "T must be >= 1");
  if (d < 1 || d > GPX_MAX_DIM) return bad_arg(ctx, "input dimension must be 1..16");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  ctx->N = N;
  ctx->train_gen += 1; // cached results that depend on X (sparse.hip forward pass) belong to the previous upload
  ctx->d = d;
  ctx->T = T;
  ctx->M = 0; // X_new of a previous training set does not carry over
  ctx->has_diag = false;
  ctx->Np = round_up(N + 1, TILE);
  ctx->ldk = pick_ld(ctx->Np);
  GPX_TRY(ensure(ctx, ctx->X, (size_t)T * N * d * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->K, (size_t)ctx->Np * ctx->ldk * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->Linv, (size_t)(ctx->Np / TILE) * TILE * TILE * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->yres, (size_t)N * sizeof(double)));
  GPX_HIP(ctx, hipMemcpyAsync(ctx->X.d(), X, (size_t)T * N * d * sizeof(double),
                              on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->factored = false;
  ctx->small_grad_ready = false;
  ctx->have_post = false;
  return 0;
}

int gpx_set_train_tasks(gpx_ctx* ctx, const double* X, int T, int N, int d) {
  return set_train_impl(ctx, X, T, N, d, false);
}

int gpx_set_train(gpx_ctx* ctx, const double* X, int N, int d) { return gpx_set_train_tasks(ctx, X, 1, N, d); }

int gpx_set_diag(gpx_ctx* ctx, const double* v, int n) {
  if (!ctx || ctx->device < 0) return -1;
  if (v == nullptr || n == 0) {
    ctx->has_diag = false;
    ctx->factored = false;
    ctx->small_grad_ready = false;
    return 0;
  }
  if (ctx->N < 1 || n != ctx->N) return bad_arg(ctx, "gpx_set_diag: need one value per training point");
  if (ctx->T != 1) return bad_arg(ctx, "gpx_set_diag: not available with per-task training sets");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  GPX_TRY(ensure(ctx, ctx->diagv, (size_t)n * sizeof(double)));
  GPX_HIP(ctx, hipMemcpyAsync(ctx->diagv.d(), v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->has_diag = true;
  ctx->factored = false;
  ctx->small_grad_ready = false;
  return 0;
}

int gpx_factor(gpx_ctx* ctx, int kind, const double* ell, double scale, double noise,
               double jitter, const double* yres, double* lml, int* info) {
  if (!ctx || ctx->device < 0) return -1;
  if (ctx->N < 1) return bad_arg(ctx, "gpx_set_train must be called first");
  if (ctx->T != 1) return bad_arg(ctx, "per-task training sets: use gpx_fit_batch / gpx_predict_sweep");
  if (!ell || !yres) return bad_arg(ctx, "null pointer");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  GPX_TRY(set_theta(ctx, kind, ctx->d, ell, scale));
  ctx->noise = noise;
  ctx->jitter = jitter;
  GPX_HIP(ctx, hipMemcpyAsync(ctx->yres.d(), yres, (size_t)ctx->N * sizeof(double),
                              hipMemcpyHostToDevice, ctx->stream));
  GPX_TRY(dev_factor(ctx, false));
  double h[2];
  int hinfo = 0;
  GPX_HIP(ctx, hipMemcpyAsync(h, ctx->scal.d() + SC_QUAD, 2 * sizeof(double), hipMemcpyDeviceToHost,
                              ctx->stream));
  GPX_HIP(ctx, hipMemcpyAsync(&hinfo, sc_int(ctx) + SI_TRAIN, sizeof(int), hipMemcpyDeviceToHost,
                              ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // a failure at the augmentation pivot itself (order N + 1) is not a failure of K
  if (hinfo > ctx->N) hinfo = 0;
  if (info) *info = hinfo;
  if (lml) {
    *lml = (hinfo != 0) ? NAN : (-0.5 * h[0] - h[1] - 0.5 * ctx->N * LOG_2PI);
  }
  return 0;
}

int gpx_lml_grad(gpx_ctx* ctx, double* grad_ell, double* grad_scale, double* grad_noise,
                 double* alpha) {
  if (!ctx || ctx->device < 0) return -1;
  if (!ctx->factored) return bad_arg(ctx, "gpx_lml_grad must follow gpx_factor");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  GPX_TRY(dev_grad(ctx));
  double h[GPX_MAX_DIM + 3];
  const int ne = n_ell(ctx->theta);
  GPX_HIP(ctx, hipMemcpyAsync(h, ctx->scal.d() + SC_GRAD, (ne + 2) * sizeof(double),
                              hipMemcpyDeviceToHost, ctx->stream));
  if (alpha) {
    GPX_HIP(ctx, hipMemcpyAsync(alpha, ctx->alpha.d(), (size_t)ctx->N * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
  }
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (grad_ell)
    for (int c = 0; c < ne; ++c) grad_ell[c] = h[c];
  if (grad_scale) *grad_scale = h[ne];
  if (grad_noise) *grad_noise = h[ne + 1];
  return 0;
}

int gpx_lml_grad_diag(gpx_ctx* ctx, double* grad_diag) {
  if (!ctx || ctx->device < 0) return -1;
  if (!grad_diag) return bad_arg(ctx, "null pointer");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->have_kinv && ctx->small_no_kinv) {
    // the gradient came from the one-launch fit step (N <= 128, no per-point diagonal: fit_small.hip), which contracts K^-1
    // out of its accumulators and never stores it: once more through the general sequence at the same theta / residuals
    const bool fs = ctx->fit_small;
    ctx->fit_small = false;
    int rc = dev_factor(ctx, false);
    if (rc == 0) rc = dev_grad(ctx);
    ctx->fit_small = fs;
    if (rc != 0) return rc;
  }
  if (!ctx->have_kinv) return bad_arg(ctx, "gpx_lml_grad_diag must follow gpx_lml_grad (K^-1 and alpha resident)");
  const int N = ctx->N;
  GPX_TRY(ensure(ctx, ctx->byres, (size_t)N * sizeof(double)));
  GPX_TRY(launch_grad_diag(ctx, ctx->K.d(), ctx->ldk, N, ctx->alpha.d(), ctx->byres.d()));
  GPX_HIP(ctx, hipMemcpyAsync(grad_diag, ctx->byres.d(), (size_t)N * sizeof(double), hipMemcpyDeviceToHost,
                              ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int gpx_fit_batch(gpx_ctx* ctx, int kind, int B, const double* ells, const double* scales,
                  const double* noises, double jitter, const double* yres, int yres_rows,
                  double* lml, int* info, double* grad, double* alpha) {
  if (!ctx || ctx->device < 0) return -1;
  if (ctx->N < 1) return bad_arg(ctx, "gpx_set_train must be called first");
  if (B < 0) return bad_arg(ctx, "negative batch");
  if (B == 0) return 0;
  if (!ells || !scales || !noises || !yres || !lml) return bad_arg(ctx, "null pointer");
  if (yres_rows != 1 && yres_rows != B && !(ctx->T > 1 && yres_rows == ctx->T))
    return bad_arg(ctx, "yres_rows must be 1, B or the task count");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  const int N = ctx->N;
  constexpr int SB = 32; // doubles per sample in bscal
  if (ctx->fit_small && N <= TILE && !ctx->has_diag && grad != nullptr) {
    // Small N: ONE kernel launch (fit_small.hip) that reads its hyper-parameters and residuals from, and writes its
    // results to, page-locked host memory — no copy calls at all around it: at N = 25 the five hipMemcpyAsync of the
    // general sequence cost more than the arithmetic (profiles/r05/fit_small.json).  X stays resident on the device.
    const int d = ctx->d, stride = d + (kind == GPX_KERNEL_PERIODIC ? 1 : 0);
    const size_t th_b = round_up64((int64_t)B * sizeof(ThetaDev), 64), y_b = round_up64((int64_t)yres_rows * N * 8, 64),
                 sc_b = (size_t)B * SB * 8, in_b = round_up64((int64_t)B * sizeof(int), 64), al_b = (size_t)B * TILE * 8;
    if (ctx->pin_fit.cap < th_b + y_b + sc_b + in_b + al_b) {
      GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      GPX_HIP(ctx, ctx->pin_fit.ensure(2 * (th_b + y_b + sc_b + in_b + al_b)));
    }
    char* base = static_cast<char*>(ctx->pin_fit.p);
    ThetaDev* th = reinterpret_cast<ThetaDev*>(base);
    double* hy = reinterpret_cast<double*>(base + th_b);
    double* hsc = reinterpret_cast<double*>(base + th_b + y_b);
    int* hin = reinterpret_cast<int*>(base + th_b + y_b + sc_b);
    double* hal = reinterpret_cast<double*>(base + th_b + y_b + sc_b + in_b);
    const KernelParams saved = ctx->theta;
    for (int b = 0; b < B; ++b) {
      const int rc = set_theta(ctx, kind, d, ells + (int64_t)b * stride, scales[b]);
      if (rc < 0) {
        ctx->theta = saved;
        return rc;
      }
      th[b].kp = ctx->theta;
      th[b].diag_train = noises[b] + jitter;
      th[b].diag_pred = th[b].diag_train;
      th[b].kdiag_pred = 0.0;
    }
    ctx->jitter = jitter;
    std::memcpy(hy, yres, (size_t)yres_rows * N * 8);
    GPX_TRY(ensure(ctx, ctx->K, (size_t)B * ctx->Np * ctx->ldk * sizeof(double)));
    GPX_TRY(ensure(ctx, ctx->Linv, (size_t)B * TILE * TILE * sizeof(double)));
    ctx->s = ctx->stream;
    GPX_TRY(launch_fit_small(ctx, ctx->theta, 0.0, th, ts_train(ctx), ctx->X.d(), N, hy, yres_rows == 1 ? 0 : N,
                             (yres_rows == 1 || yres_rows == B) ? 0 : yres_rows, ctx->K.d(), ctx->ldk,
                             (int64_t)ctx->Np * ctx->ldk, ctx->Linv.d(), (int64_t)TILE * TILE, hal, TILE, hsc, SB, hin, 1, B));
    GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->factored = false;
    ctx->small_grad_ready = false;
    ctx->have_post = false;
    ctx->have_kinv = false;
    const int ne = n_ell(ctx->theta);
    for (int b = 0; b < B; ++b) {
      int hinfo = hin[b];
      if (hinfo > N) hinfo = 0; // a failure at the augmentation pivot itself is not a failure of K
      if (info) info[b] = hinfo;
      const double* h = hsc + (size_t)b * SB;
      lml[b] = (hinfo != 0) ? NAN : (-0.5 * h[SC_QUAD] - h[SC_SUMLOG] - 0.5 * N * LOG_2PI);
      for (int c = 0; c < ne + 2; ++c) grad[(int64_t)b * (ne + 2) + c] = (hinfo != 0) ? NAN : h[SC_GRAD + c];
      if (alpha) std::memcpy(alpha + (int64_t)b * N, hal + (size_t)b * TILE, (size_t)N * 8);
    }
    return 0;
  }
  // The general launch sequence.  Everything that crosses PCIe around it goes through ONE page-locked region of the context
  // — [theta table | residuals | lml terms and gradient | pivot reports | alpha] — with asynchronous copies and a single
  // synchronisation: a copy between the device and PAGEABLE memory (the caller's arrays, a std::vector) is staged by the
  // runtime and waited for call by call, ~10 - 20 us each, which at N = 512 (one leapfrog of C1's NUTS: 0.26 ms on the device)
  // was a quarter of the host-side overhead of this call.
  ctx->jitter = jitter;
  const int d_ = ctx->d, stride_ = d_ + (kind == GPX_KERNEL_PERIODIC ? 1 : 0);
  const bool want_alpha = grad != nullptr && alpha != nullptr;
  const size_t th_b = round_up64((int64_t)B * sizeof(ThetaDev), 64), y_b = round_up64((int64_t)yres_rows * N * 8, 64),
               sc_b = round_up64((int64_t)B * SB * 8, 64), in_b = round_up64((int64_t)B * sizeof(int), 64),
               al_b = want_alpha ? (size_t)B * N * 8 : 0;
  if (ctx->pin_gen.cap < th_b + y_b + sc_b + in_b + al_b) {
    GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GPX_HIP(ctx, ctx->pin_gen.ensure(2 * (th_b + y_b + sc_b + in_b + al_b)));
  }
  char* pbase = static_cast<char*>(ctx->pin_gen.p);
  ThetaDev* hth = reinterpret_cast<ThetaDev*>(pbase);
  double* hy = reinterpret_cast<double*>(pbase + th_b);
  double* hsc = reinterpret_cast<double*>(pbase + th_b + y_b);
  int* hin = reinterpret_cast<int*>(pbase + th_b + y_b + sc_b);
  double* hal = reinterpret_cast<double*>(pbase + th_b + y_b + sc_b + in_b);
  {
    const KernelParams saved = ctx->theta;
    for (int b = 0; b < B; ++b) { // (fill_theta_table's entries, built in place)
      const int rc = set_theta(ctx, kind, d_, ells + (int64_t)b * stride_, scales[b]);
      if (rc < 0) {
        ctx->theta = saved;
        return rc;
      }
      hth[b].kp = ctx->theta;
      hth[b].diag_train = noises[b] + jitter;
      hth[b].diag_pred = noises[b] + jitter;
      hth[b].kdiag_pred = kdiag_value(hth[b].kp) + noises[b] + jitter;
    }
  }
  GPX_TRY(ensure(ctx, ctx->thtab, (size_t)B * sizeof(ThetaDev)));
  GPX_HIP(ctx, hipMemcpyAsync(ctx->thtab.p, hth, (size_t)B * sizeof(ThetaDev), hipMemcpyHostToDevice, ctx->stream));
  BatchPlan bp = make_plan(ctx, B, 0, false);
  bp.th = static_cast<const ThetaDev*>(ctx->thtab.p);
  GPX_TRY(ensure(ctx, ctx->binfo, (size_t)2 * B * sizeof(int)));
  GPX_TRY(ensure(ctx, ctx->bscal, (size_t)B * SB * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->byres, (size_t)yres_rows * N * sizeof(double)));
  bp.info_train = ctx->binfo.i();
  bp.info_cov = ctx->binfo.i() + B;
  bp.scal = ctx->bscal.d();
  bp.scal_bs = SB;
  std::memcpy(hy, yres, (size_t)yres_rows * N * sizeof(double));
  if (yres_rows != 1) {
    GPX_HIP(ctx, hipMemcpyAsync(ctx->byres.d(), hy, (size_t)yres_rows * N * sizeof(double), hipMemcpyHostToDevice,
                                ctx->stream));
    bp.yres = ctx->byres.d();
    bp.y_bs = N;
    bp.y_mod = (yres_rows == B) ? 0 : yres_rows;
  } else {
    GPX_HIP(ctx, hipMemcpyAsync(ctx->yres.d(), hy, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    bp.yres = ctx->yres.d();
    bp.y_bs = 0;
  }
  GPX_TRY(dev_factor(ctx, false, bp, true));
  const int ne = n_ell(ctx->theta);
  if (grad) GPX_TRY(dev_grad(ctx, bp));
  GPX_HIP(ctx, hipMemcpyAsync(hsc, ctx->bscal.d(), (size_t)B * SB * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipMemcpyAsync(hin, bp.info_train, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (want_alpha)
    GPX_HIP(ctx, hipMemcpy2DAsync(hal, (size_t)N * sizeof(double), ctx->alpha.d(), (size_t)ctx->Np * sizeof(double),
                                  (size_t)N * sizeof(double), B, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (want_alpha) std::memcpy(alpha, hal, (size_t)B * N * sizeof(double));
  ctx->factored = false;
  ctx->small_grad_ready = false;
  ctx->have_post = false;
  ctx->h_binfo.assign(hin, hin + B);
  ctx->h_bscal.assign(hsc, hsc + (size_t)B * SB);
  for (int b = 0; b < B; ++b) {
    int hinfo = ctx->h_binfo[(size_t)b];
    if (hinfo > N) hinfo = 0; // a failure at the augmentation pivot itself is not a failure of K
    if (info) info[b] = hinfo;
    const double* h = ctx->h_bscal.data() + (size_t)b * SB;
    lml[b] = (hinfo != 0) ? NAN : (-0.5 * h[SC_QUAD] - h[SC_SUMLOG] - 0.5 * N * LOG_2PI);
    if (grad)
      for (int c = 0; c < ne + 2; ++c) grad[(int64_t)b * (ne + 2) + c] = (hinfo != 0) ? NAN : h[SC_GRAD + c];
  }
  return 0;
}

int gpx_posterior(gpx_ctx* ctx, const double* Xnew, int M, double noise_p, double jitter,
                  double* mean, double* cov, double* var) {
  if (!ctx || ctx->device < 0) return -1;
  if (!ctx->factored) return bad_arg(ctx, "gpx_posterior must follow gpx_factor");
  if (!Xnew) return bad_arg(ctx, "null Xnew");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  GPX_TRY(set_xnew(ctx, Xnew, M));
  ctx->noise_p = noise_p;
  const double saved_jitter = ctx->jitter;
  ctx->jitter = jitter;
  ctx->fused_vt = false; // this X_new was not part of the factorisation: solve here
  int rc = dev_posterior(ctx, cov != nullptr);
  ctx->jitter = saved_jitter;
  GPX_TRY(rc);
  if (mean)
    GPX_HIP(ctx, hipMemcpyAsync(mean, ctx->mean.d(), (size_t)M * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
  if (var)
    GPX_HIP(ctx, hipMemcpyAsync(var, ctx->var.d(), (size_t)M * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
  if (cov)
    GPX_HIP(ctx, hipMemcpy2DAsync(cov, (size_t)M * sizeof(double), ctx->Cov.d(),
                                  ctx->ldc * sizeof(double), (size_t)M * sizeof(double), M,
                                  hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int gpx_mvn_draw(gpx_ctx* ctx, const double* eps, int n, double* out, int* info) {
  if (!ctx || ctx->device < 0) return -1;
  if (!ctx->have_post) return bad_arg(ctx, "gpx_mvn_draw must follow gpx_posterior with cov");
  if (n < 1 || !eps || !out) return bad_arg(ctx, "bad draw arguments");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  int n_pad = 0;
  GPX_TRY(upload_eps(ctx, eps, n, &n_pad));
  GPX_TRY(dev_draw(ctx, n_pad, n));
  int hinfo = 0;
  GPX_HIP(ctx, hipMemcpy2DAsync(out, (size_t)ctx->M * sizeof(double), ctx->draws.d(),
                                ctx->ldc * sizeof(double), (size_t)ctx->M * sizeof(double), n,
                                hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipMemcpyAsync(&hinfo, sc_int(ctx) + SI_COV, sizeof(int), hipMemcpyDeviceToHost,
                              ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo > ctx->M) hinfo = 0;
  if (info) *info = hinfo;
  if (hinfo != 0)
    for (int64_t t = 0; t < (int64_t)n * ctx->M; ++t) out[t] = NAN;
  return 0;
}

int gpx_predict_sweep(gpx_ctx* ctx, int kind, int S, const double* ells, const double* scales,
                      const double* noises, const double* yres, int yres_rows,
                      const double* Xnew, int M, int noiseless, double jitter,
                      const double* eps, int n, double* means, double* samples, int* infos, double* vars,
                      const double* pred_diag, int m_slice) {
  if (!ctx || ctx->device < 0) return -1;
  if (ctx->N < 1) return bad_arg(ctx, "gpx_set_train must be called first");
  if (S < 0 || n < 0) return bad_arg(ctx, "negative count");
  if (S == 0) return 0;
  if (!ells || !scales || !noises || !yres || !Xnew || !means) return bad_arg(ctx, "null pointer");
  if (n > 0 && (!eps || !samples)) return bad_arg(ctx, "eps/samples required when n > 0");
  if (yres_rows != 1 && yres_rows != S && !(ctx->T > 1 && yres_rows == ctx->T))
    return bad_arg(ctx, "yres_rows must be 1, S or the task count");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  const int N = ctx->N;
  GPX_TRY(set_xnew(ctx, Xnew, M));
  // device staging for all inputs/outputs of the sweep: nothing crosses PCIe inside the loop.  The staging
  // buffers live in the context and only grow: a hipMalloc / hipFree pair per call costs ~20 ms once the
  // context holds multi-GB batch buffers (measured at N = 512, B = 256), more than the sweep itself.
  DevBuf &dEps = ctx->st_eps, &dYres = ctx->st_yres, &dMeans = ctx->st_means, &dSamples = ctx->st_samples,
         &dInfos = ctx->st_infos, &dVars = ctx->st_vars, &dPred = ctx->st_pred;
  int rc = 0;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(ctx->stream); // nothing may still read the caller's host buffers
  };
#define SWEEP_TRY(expr)  \
  do {                   \
    rc = (expr);         \
    if (rc < 0) {        \
      cleanup();         \
      return rc;         \
    }                    \
  } while (0)
#define SWEEP_HIP(expr)                                            \
  do {                                                             \
    hipError_t _e = (expr);                                        \
    if (_e != hipSuccess) {                                        \
      cleanup();                                                   \
      return fail(ctx, #expr, _e, __FILE__, __LINE__);             \
    }                                                              \
  } while (0)
  const bool strided = yres_rows != 1;
  SWEEP_TRY(ensure(ctx, dMeans, (size_t)S * M * sizeof(double)));
  SWEEP_TRY(ensure(ctx, dInfos, (size_t)2 * S * sizeof(int)));
  SWEEP_HIP(hipMemsetAsync(dInfos.p, 0, (size_t)2 * S * sizeof(int), ctx->stream));
  if (vars) SWEEP_TRY(ensure(ctx, dVars, (size_t)S * M * sizeof(double)));
  if (pred_diag) {
    SWEEP_TRY(ensure(ctx, dPred, (size_t)S * M * sizeof(double)));
    SWEEP_HIP(hipMemcpyAsync(dPred.d(), pred_diag, (size_t)S * M * sizeof(double), hipMemcpyHostToDevice,
                             ctx->stream));
  }
  if (strided) {
    SWEEP_TRY(ensure(ctx, dYres, (size_t)yres_rows * N * sizeof(double)));
    SWEEP_HIP(hipMemcpyAsync(dYres.d(), yres, (size_t)yres_rows * N * sizeof(double), hipMemcpyHostToDevice,
                             ctx->stream));
  } else {
    SWEEP_HIP(hipMemcpyAsync(ctx->yres.d(), yres, (size_t)N * sizeof(double), hipMemcpyHostToDevice,
                             ctx->stream));
  }
  const size_t draws_b = (size_t)S * n * M * sizeof(double), means_b = (size_t)S * M * sizeof(double);
  if (n > 0) {
    SWEEP_TRY(ensure(ctx, dEps, draws_b));
    SWEEP_TRY(ensure(ctx, dSamples, draws_b));
    SWEEP_HIP(ctx->pin_in.ensure(draws_b));
    std::memcpy(ctx->pin_in.p, eps, draws_b);
    SWEEP_HIP(hipMemcpyAsync(dEps.d(), ctx->pin_in.p, draws_b, hipMemcpyHostToDevice, ctx->stream));
  }
  SweepIO io;
  io.kind = kind;
  io.S = S;
  io.n = n;
  io.noiseless = noiseless;
  io.jitter = jitter;
  io.ells = ells;
  io.scales = scales;
  io.noises = noises;
  io.dYres = strided ? dYres.d() : nullptr;
  io.y_mod = (strided && yres_rows != S) ? yres_rows : 0;
  io.dEps = n > 0 ? dEps.d() : nullptr;
  io.dMeans = dMeans.d();
  io.dSamples = n > 0 ? dSamples.d() : nullptr;
  io.dInfos = dInfos.i();
  io.dVars = vars ? dVars.d() : nullptr;
  io.dPredDiag = pred_diag ? dPred.d() : nullptr;
  io.m_slice = m_slice;
  SWEEP_TRY(sweep_core(ctx, io));
  // results come back through the page-locked buffer: [means | samples | vars | infos]
  const size_t infos_b = (size_t)2 * S * sizeof(int);
  SWEEP_HIP(ctx->pin_out.ensure(2 * means_b + draws_b + infos_b));
  char* po = static_cast<char*>(ctx->pin_out.p);
  SWEEP_HIP(hipMemcpyAsync(po, dMeans.d(), means_b, hipMemcpyDeviceToHost, ctx->stream));
  if (n > 0) SWEEP_HIP(hipMemcpyAsync(po + means_b, dSamples.d(), draws_b, hipMemcpyDeviceToHost, ctx->stream));
  if (vars) SWEEP_HIP(hipMemcpyAsync(po + means_b + draws_b, dVars.d(), means_b, hipMemcpyDeviceToHost, ctx->stream));
  SWEEP_HIP(hipMemcpyAsync(po + 2 * means_b + draws_b, dInfos.p, infos_b, hipMemcpyDeviceToHost, ctx->stream));
  SWEEP_HIP(hipStreamSynchronize(ctx->stream));
  std::memcpy(means, po, means_b);
  if (n > 0) std::memcpy(samples, po + means_b, draws_b);
  if (vars) std::memcpy(vars, po + means_b + draws_b, means_b);
  const int* hinfos = reinterpret_cast<const int*>(po + 2 * means_b + draws_b);
  cleanup();
#undef SWEEP_TRY
#undef SWEEP_HIP
  for (int s = 0; s < S; ++s) {
    int it = hinfos[2 * s], ic = hinfos[2 * s + 1];
    if (it > N) it = 0;
    if (ic > ctx->cM) ic = 0;
    const int code = it != 0 ? it : (ic != 0 ? -ic : 0);
    if (infos) infos[s] = code;
    if (it != 0)
      for (int a = 0; a < M; ++a) {
        means[(int64_t)s * M + a] = NAN;
        if (vars) vars[(int64_t)s * M + a] = NAN;
      }
    if (code != 0 && n > 0)
      for (int64_t t = 0; t < (int64_t)n * M; ++t) samples[(int64_t)s * n * M + t] = NAN;
  }
  return 0;
}

int gpx_sweep_stats(gpx_ctx* ctx, int64_t* batches, int64_t* samples, int* last_batch) {
  if (!ctx) return -1;
  if (batches) *batches = ctx->sweep_batches;
  if (samples) *samples = ctx->sweep_samples;
  if (last_batch) *last_batch = ctx->last_batch;
  return 0;
}

int gpx_profile_enable(gpx_ctx* ctx, int on) {
  if (!ctx) return -1;
  ctx->prof_on = on != 0;
  return 0;
}

int gpx_profile_reset(gpx_ctx* ctx) {
  if (!ctx || ctx->device < 0) return -1;
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  drain_profile(ctx);
  for (int c = 0; c < GPX_PROF_NCLASS; ++c) {
    ctx->prof[c].launches = 0;
    ctx->prof[c].work = 0.0;
    ctx->prof[c].bytes = 0.0;
    ctx->prof[c].ms = 0.0;
  }
  return 0;
}

int gpx_profile_read(gpx_ctx* ctx, int cls, int64_t* launches, double* total_ms,
                     double* total_work) {
  if (!ctx || ctx->device < 0) return -1;
  if (cls < 0 || cls >= GPX_PROF_NCLASS) return bad_arg(ctx, "profile class");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  drain_profile(ctx);
  if (launches) *launches = ctx->prof[cls].launches;
  if (total_ms) *total_ms = ctx->prof[cls].ms;
  if (total_work) *total_work = ctx->prof[cls].work;
  return 0;
}

int gpx_profile_read_bytes(gpx_ctx* ctx, int cls, double* total_bytes) {
  if (!ctx || ctx->device < 0) return -1;
  if (cls < 0 || cls >= GPX_PROF_NCLASS) return bad_arg(ctx, "profile class");
  if (total_bytes) *total_bytes = ctx->prof[cls].bytes;
  return 0;
}

int gpx_debug_set_potf2(gpx_ctx* ctx, const char* mode) {
  if (!ctx || !mode) return -1;
  const std::string v(mode);
  if (v == "slim") ctx->potf2_mode = gpx::GPX_POTF2_SLIM;
  else if (v == "tile") ctx->potf2_mode = gpx::GPX_POTF2_TILE;
  else if (v == "fuse") ctx->potf2_trsm = true;    // the panel TRSM rides in the potf2 launch (one-outer-block chains)
  else if (v == "nofuse") ctx->potf2_trsm = false; // ... or is a launch of its own
  else return bad_arg(ctx, "potf2 kernel: slim | tile | fuse | nofuse");
  return 0;
}

int gpx_debug_set_serialise_trailing(gpx_ctx* ctx, int on) {
  if (!ctx) return -1;
  ctx->serialise_trailing = on != 0;
  return 0;
}

int gpx_debug_set_lat_gemm(gpx_ctx* ctx, const char* mode) {
  if (!ctx || !mode) return -1;
  const std::string v(mode);
  if (v == "r5") ctx->lat_gemm = 5;
  else if (v == "r1") ctx->lat_gemm = 1;
  else if (v == "auto") ctx->lat_gemm = 0;
  else return bad_arg(ctx, "latency-shape GEMM: auto | r5 | r1");
  return 0;
}

int gpx_debug_gemm_time(gpx_ctx* ctx, int tiles_m, int tiles_n, int K, int mode, int lower, int shape, int reps,
                        double* ms_per_launch) {
  if (!ctx || ctx->device < 0) return -1;
  if (tiles_m < 1 || tiles_n < 1 || K < 16 || K % 16 != 0 || reps < 1 || mode < 0 || mode > 2 || !ms_per_launch)
    return bad_arg(ctx, "gemm timing arguments");
  if (mode == 2 && (tiles_n != 1 || K != TILE)) return bad_arg(ctx, "in-place form: tiles_n = 1, K = 128");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t lda = pick_ld(K), ldc = pick_ld((int64_t)tiles_n * TILE);
  const size_t ab = (size_t)tiles_m * TILE * lda * sizeof(double), bb = (size_t)tiles_n * TILE * lda * sizeof(double),
               cb = (size_t)tiles_m * TILE * ldc * sizeof(double);
  GPX_TRY(ensure(ctx, ctx->tA, ab));
  GPX_TRY(ensure(ctx, ctx->tB, bb));
  GPX_TRY(ensure(ctx, ctx->tC, cb));
  // every byte 0x3f: the double 0x3f3f3f3f3f3f3f3f = 4.8e-4 — finite, non-zero, and small enough that repeated updates stay finite
  GPX_HIP(ctx, hipMemsetAsync(ctx->tA.p, 0x3f, ab, ctx->stream));
  GPX_HIP(ctx, hipMemsetAsync(ctx->tB.p, 0x3f, bb, ctx->stream));
  GPX_HIP(ctx, hipMemsetAsync(ctx->tC.p, 0x3f, cb, ctx->stream));
  GemmArgs g{};
  g.A = ctx->tA.d();
  g.lda = lda;
  g.B = ctx->tB.d();
  g.ldb = lda;
  g.C = mode == 2 ? ctx->tA.d() : ctx->tC.d();
  g.ldc = mode == 2 ? lda : ldc;
  g.K = K;
  g.alpha = mode == 1 ? -1.0 : (mode == 2 ? 1e-3 : 1.0); // (in place: the result feeds the next repetition — keep it small)
  g.beta = mode == 1 ? 1.0 : 0.0;
  g.lower = lower;
  g.latency_shape = shape == 1;
  g.big_shape = shape == 2;
  ctx->s = ctx->stream;
  ctx->small_bk_now = ctx->small_bk != 0 ? ctx->small_bk : 16;
  const double work = 2.0 * tiles_m * TILE * (double)tiles_n * TILE * K * (lower ? 0.5 : 1.0);
  for (int r = 0; r < 3; ++r) GPX_TRY(launch_gemm_nt(ctx, g, tiles_m, tiles_n, 0, GPX_PROF_GEMM_OTHER, work)); // warm-up
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  GPX_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int r = 0; r < reps; ++r) GPX_TRY(launch_gemm_nt(ctx, g, tiles_m, tiles_n, 0, GPX_PROF_GEMM_OTHER, work));
  GPX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  GPX_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  GPX_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_per_launch = (double)ms / reps;
  return 0;
}

int gpx_time_stage(gpx_ctx* ctx, int stage, int reps, double* elapsed_ms) {
  if (!ctx || ctx->device < 0) return -1;
  if (ctx->N < 1) return bad_arg(ctx, "gpx_set_train / gpx_factor must be called first");
  if (reps < 1) return bad_arg(ctx, "reps must be >= 1");
  if (stage >= GPX_STAGE_POSTERIOR && ctx->M < 1) return bad_arg(ctx, "gpx_posterior must be called first");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  int n_pad = 0;
  if (stage == GPX_STAGE_PREDICT) {
    // one draw per pass from a fixed device-resident eps (zeros are fine for timing: the GEMM
    // does the same work) — keep whatever eps the last gpx_mvn_draw uploaded if present
    n_pad = TILE;
    GPX_TRY(ensure(ctx, ctx->eps, (size_t)n_pad * ctx->ldc * sizeof(double)));
    GPX_TRY(ensure(ctx, ctx->draws, (size_t)n_pad * ctx->ldc * sizeof(double)));
  }
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  GPX_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int r = 0; r < reps; ++r) {
    switch (stage) {
      case GPX_STAGE_GRAM:
        GPX_TRY(launch_gram_padded(ctx, ctx->theta, ctx->X.d(), ctx->N, ctx->N, ctx->X.d(), ctx->N,
                                   ctx->Np, ctx->noise + ctx->jitter, 1, 1, ctx->K.d(), ctx->ldk, 1, 0, nullptr, 0,
                                   TaskStride(), ctx->has_diag ? ctx->diagv.d() : nullptr));
        ctx->factored = false;
        ctx->small_grad_ready = false;
        break;
      case GPX_STAGE_POTRF:
        GPX_TRY(dev_factor(ctx, false));
        break;
      case GPX_STAGE_FITSTEP:
        GPX_TRY(dev_factor(ctx, false));
        GPX_TRY(dev_grad(ctx));
        break;
      case GPX_STAGE_POSTERIOR:
        GPX_TRY(dev_factor(ctx, true));
        GPX_TRY(dev_posterior(ctx, true));
        break;
      case GPX_STAGE_PREDICT:
        GPX_TRY(dev_factor(ctx, true));
        GPX_TRY(dev_posterior(ctx, true));
        GPX_TRY(dev_draw(ctx, n_pad, 1));
        break;
      default:
        return bad_arg(ctx, "unknown stage");
    }
  }
  GPX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  GPX_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  GPX_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  if (elapsed_ms) *elapsed_ms = ms;
  return 0;
}

int gpx_sweep_resident(gpx_ctx* ctx, int kind, int S, const double* ells, const double* scales,
                       const double* noises, int noiseless, double jitter, int n_draws,
                       double* elapsed_ms) {
  if (!ctx || ctx->device < 0) return -1;
  if (ctx->N < 1 || ctx->M < 1) return bad_arg(ctx, "gpx_factor and gpx_posterior must be called first");
  if (S < 1 || !ells || !scales || !noises || n_draws < 0) return bad_arg(ctx, "sweep arguments");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  SweepIO io;
  io.kind = kind;
  io.S = S;
  io.n = n_draws;
  io.noiseless = noiseless;
  io.jitter = jitter;
  io.ells = ells;
  io.scales = scales;
  io.noises = noises;
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  GPX_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  GPX_TRY(sweep_core(ctx, io));
  GPX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  GPX_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  GPX_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  if (elapsed_ms) *elapsed_ms = ms;
  return 0;
}

int gpx_mfma_f64_peak(gpx_ctx* ctx, double* tflops) {
  if (!ctx || ctx->device < 0 || !tflops) return -1;
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  return mfma_peak(ctx, tflops); // tflops[0..2]: TFLOP/s, cycles per MFMA, effective MHz
}

int gpx_gemm_nt(gpx_ctx* ctx, int M, int N, int K, double alpha, const double* A,
                const double* B, double beta, double* C) {
  if (!ctx || ctx->device < 0) return -1;
  if (M < 1 || N < 1 || K < 1 || !A || !B || !C) return bad_arg(ctx, "gemm arguments");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  const int Mp = round_up(M, TILE), Np = round_up(N, TILE), Kp = round_up(K, 32);
  const int64_t lda = pick_ld(Kp), ldc = pick_ld(Np);
  GPX_TRY(ensure(ctx, ctx->tA, (size_t)Mp * lda * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->tB, (size_t)Np * lda * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->tC, (size_t)Mp * ldc * sizeof(double)));
  GPX_HIP(ctx, hipMemsetAsync(ctx->tA.p, 0, (size_t)Mp * lda * sizeof(double), ctx->stream));
  GPX_HIP(ctx, hipMemsetAsync(ctx->tB.p, 0, (size_t)Np * lda * sizeof(double), ctx->stream));
  GPX_HIP(ctx, hipMemsetAsync(ctx->tC.p, 0, (size_t)Mp * ldc * sizeof(double), ctx->stream));
  GPX_HIP(ctx, hipMemcpy2DAsync(ctx->tA.d(), lda * sizeof(double), A, (size_t)K * sizeof(double),
                                (size_t)K * sizeof(double), M, hipMemcpyHostToDevice, ctx->stream));
  GPX_HIP(ctx, hipMemcpy2DAsync(ctx->tB.d(), lda * sizeof(double), B, (size_t)K * sizeof(double),
                                (size_t)K * sizeof(double), N, hipMemcpyHostToDevice, ctx->stream));
  if (beta != 0.0)
    GPX_HIP(ctx, hipMemcpy2DAsync(ctx->tC.d(), ldc * sizeof(double), C, (size_t)N * sizeof(double),
                                  (size_t)N * sizeof(double), M, hipMemcpyHostToDevice, ctx->stream));
  GemmArgs g{};
  g.A = ctx->tA.d();
  g.lda = lda;
  g.B = ctx->tB.d();
  g.ldb = lda;
  g.C = ctx->tC.d();
  g.ldc = ldc;
  g.K = Kp;
  g.alpha = alpha;
  g.beta = beta;
  GPX_TRY(launch_gemm_nt(ctx, g, Mp / TILE, Np / TILE, 0, GPX_PROF_GEMM_OTHER,
                         2.0 * M * (double)N * K));
  GPX_HIP(ctx, hipMemcpy2DAsync(C, (size_t)N * sizeof(double), ctx->tC.d(), ldc * sizeof(double),
                                (size_t)N * sizeof(double), M, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int gpx_potrf(gpx_ctx* ctx, int n, const double* A, double* L, int* info) {
  if (!ctx || ctx->device < 0) return -1;
  if (n < 1 || !A || !L) return bad_arg(ctx, "potrf arguments");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  const int np = round_up(n, TILE);
  const int64_t ld = pick_ld(np);
  GPX_TRY(ensure(ctx, ctx->tA, (size_t)np * ld * sizeof(double)));
  GPX_TRY(ensure(ctx, ctx->tB, (size_t)(np / TILE) * TILE * TILE * sizeof(double)));
  GPX_HIP(ctx, hipMemsetAsync(ctx->tA.p, 0, (size_t)np * ld * sizeof(double), ctx->stream));
  GPX_HIP(ctx, hipMemcpy2DAsync(ctx->tA.d(), ld * sizeof(double), A, (size_t)n * sizeof(double),
                                (size_t)n * sizeof(double), n, hipMemcpyHostToDevice, ctx->stream));
  GPX_TRY(launch_pad_identity(ctx, ctx->tA.d(), ld, n, np));
  GPX_HIP(ctx, hipMemsetAsync(sc_int(ctx) + SI_TRAIN, 0, sizeof(int), ctx->stream));
  GPX_TRY(potrf_lower(ctx, ctx->tA.d(), ld, np, 0, ctx->tB.d(), sc_int(ctx) + SI_TRAIN));
  int hinfo = 0;
  GPX_HIP(ctx, hipMemcpy2DAsync(L, (size_t)n * sizeof(double), ctx->tA.d(), ld * sizeof(double),
                                (size_t)n * sizeof(double), n, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipMemcpyAsync(&hinfo, sc_int(ctx) + SI_TRAIN, sizeof(int), hipMemcpyDeviceToHost,
                              ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) L[(int64_t)i * n + j] = 0.0;
  if (hinfo > n) hinfo = 0;
  if (info) *info = hinfo;
  return 0;
}

} // extern "C"

// ---- device-resident I/O variant of gpx_predict_sweep (node-level sweep, multi.hip) -----------------------------
// Every d_* pointer lives on ctx's device (inputs arrived by RCCL broadcast, outputs are gathered by RCCL); the
// theta tables are host arrays.  Enqueues on ctx->stream and returns; the caller synchronises.  d_infos receives
// 2 ints per sample (train pivot, cov pivot) exactly as gpx_predict_sweep decodes them.
namespace gpx {
int sweep_device_io(gpx_ctx* ctx, int kind, int S, const double* ells, const double* scales, const double* noises,
                    const double* d_X, int N, int d, const double* d_yres, int yres_rows, const double* d_Xnew, int M,
                    int noiseless, double jitter, const double* d_eps, int n, double* d_means, double* d_samples,
                    int* d_infos, double* d_vars, int m_slice) {
  if (!ctx || ctx->device < 0) return -1;
  if (S < 1 || n < 0 || !ells || !scales || !noises || !d_yres || !d_Xnew || !d_means || !d_infos)
    return bad_arg(ctx, "sweep_device_io arguments");
  if (yres_rows != 1 && yres_rows != S) return bad_arg(ctx, "yres_rows must be 1 or S");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  if (d_X != nullptr) GPX_TRY(set_train_impl(ctx, d_X, 1, N, d, true));
  if (ctx->N != N || ctx->d != d) return bad_arg(ctx, "training set mismatch");
  GPX_TRY(set_xnew(ctx, d_Xnew, M, true));
  if (yres_rows == 1)
    GPX_HIP(ctx, hipMemcpyAsync(ctx->yres.d(), d_yres, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  GPX_HIP(ctx, hipMemsetAsync(d_infos, 0, (size_t)2 * S * sizeof(int), ctx->stream));
  SweepIO io;
  io.kind = kind;
  io.S = S;
  io.n = n;
  io.noiseless = noiseless;
  io.jitter = jitter;
  io.ells = ells;
  io.scales = scales;
  io.noises = noises;
  io.dYres = (yres_rows == 1) ? nullptr : d_yres;
  io.dEps = n > 0 ? d_eps : nullptr;
  io.dMeans = d_means;
  io.dSamples = n > 0 ? d_samples : nullptr;
  io.dInfos = d_infos;
  io.dVars = d_vars;
  io.m_slice = m_slice;
  return sweep_core(ctx, io);
}
int ctx_cov_block(const gpx_ctx* ctx) { return ctx->cM; }
} // namespace gpx

