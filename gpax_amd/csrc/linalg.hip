// linalg.hip — blocked drivers (Cholesky, right-looking triangular sweeps) and the small
// bandwidth-bound kernels of the exact-GP path.  All matrices are row-major, padded to
// multiples of 128; the dense contractions are launches of gemm_nt_kernel (gemm_f64.hip).
//
// Reference seam: the Cholesky / solve_triangular / matmul calls JAX performs underneath
// gpax/models/gp.py:160-164 (MultivariateNormal log_prob), gp.py:271-273 (posterior) and
// gp.py:292 (MVN draw).
#include "common.h"

namespace gpx {

static inline GemmArgs gemm_args(const double* A, int64_t lda, const double* B, int64_t ldb,
                                 double* C, int64_t ldc, int K, double alpha, double beta) {
  GemmArgs g{};
  g.A = A;
  g.lda = lda;
  g.B = B;
  g.ldb = ldb;
  g.C = C;
  g.ldc = ldc;
  g.K = K;
  g.alpha = alpha;
  g.beta = beta;
  return g;
}

// ---- blocked right-looking Cholesky (lower, in place), two-level blocking ------------------
// Outer blocks of OUTER_TILES*128 columns keep the big trailing SYRK at K = 512 (C-tile HBM
// traffic / flop is 1/4 of a K = 128 update); inside an outer block the panel is advanced 128
// columns at a time: potf2+inverse (1 workgroup) -> panel TRSM as GEMM with the inverse ->
// update of the remaining columns of the outer block only.
//
// extra_tiles > 0: the matrix has extra_tiles*128 more ROWS below the square part (the k_pX
// rows of the posterior).  They take part in every panel TRSM and trailing update, which is
// exactly the right-looking solve of  X L^T = k_pX : after the sweep those rows hold
// k_pX L^-T, at the trailing GEMM's efficiency and with no separate serial chain.
//
// Look-ahead on two streams.  For outer block k (columns [ob, oe)):
//   P(k)  panel stream: potf2 / TRSM / inner updates of the block's own columns;
//   U1(k) panel stream: trailing update restricted to the NEXT outer block's columns;
//   U2(k) main stream : trailing update of everything to the right of that.
// P(k+1) only needs U1(k), so it runs concurrently with U2(k) — the latency-bound panel work
// hides behind the big SYRK, and U2(k) follows U2(k-1) with no gap.  Every C tile still receives
// its updates in a fixed order (U2(k-1) before U1(k) by event), so results are bit-reproducible.
struct BatchStrides {
  int batch;
  int64_t a_bs, linv_bs;
};
static inline void set_batch(GemmArgs& g, int batch, int64_t a_bs, int64_t b_bs, int64_t c_bs) {
  g.batch = batch;
  g.a_bs = a_bs;
  g.b_bs = b_bs;
  g.c_bs = c_bs;
}

// One outer block (diagonal blocks ob .. oe-1).  Per diagonal block kb: potf2 (+ inverse) -> panel TRSM (GEMM with
// the inverse) -> update of the outer block's remaining columns.
// (Round 2 overlapped the next potf2 with that update — "early diagonal", with and without fusing the update into the
// potf2 kernel — and round 3 ran a whole block's chain as one cooperative kernel; both bit-identical, both slower:
// profiles/r02/chain_experiments.md, profiles/r03/panel_kernel.md, tools/exp/panel.hip.  Round 4 ran the inner updates
// left-looking — column kb brought up to date in ONE update of K = 128 (kb - ob) right before its potf2, a column of
// the block read and written once instead of (kb - ob) times; bit-identical — potrf 29.1 -> 29.3 ms at C3, the fit
// step at N = 512 0.565 -> 0.609 ms: the longer update sits on the chain right before the potf2.  Not kept.  Round 5 let
// the workgroup of the NEXT diagonal block inside update(kb) go on into that block's potf2 — two launches per step and no
// cross-stream wait; bit-identical, potrf 0.19 -> 0.23 ms at N = 512, 1.99 -> 2.12 ms at N = 4096: one workgroup updates
// its block at ONE CU's MFMA rate and then factors beside four MFMA-saturating workgroups.  profiles/r05/step_fused.md.)
static int panel_block(gpx_ctx* ctx, double* dA, int64_t lda, int nblk, int extra, int ob, int oe,
                       double* dLinv, int* dInfo, const BatchStrides& bs, int kb_end = -1) {
  if (kb_end < 0) kb_end = oe; // (kb_end < oe: the first steps of the block only — potrf_steps)
  for (int kb = ob; kb < kb_end; ++kb) {
    double* Akk = dA + (int64_t)kb * TILE * lda + (int64_t)kb * TILE;
    double* Li = dLinv + (int64_t)kb * TILE * TILE;
    const int below = nblk - kb - 1 + extra;
    double* Apan = dA + (int64_t)(kb + 1) * TILE * lda + (int64_t)kb * TILE;
    // a single sample with the chip to itself (round-5 latency shapes in force): the panel TRSM rides in the potf2 launch
    // (potf2.hip potf2_trsm_kernel: the same strip body, the same bits) — two launches per step instead of three
    const bool fused = ctx->potf2_trsm && below > 0 && bs.batch == 1 && ctx->potf2_mode == GPX_POTF2_SLIM &&
                       (ctx->lat_gemm != 0 ? ctx->lat_gemm : ctx->lat_now) == 5;
    if (fused) {
      GemmArgs g = gemm_args(Apan, lda, Li, TILE, Apan, lda, TILE, 1.0, 0.0);
      GPX_TRY(launch_potf2_trsm(ctx, Akk, lda, Li, dInfo, kb * TILE, g, below));
    } else {
      GPX_TRY(launch_potf2_inv(ctx, Akk, lda, Li, dInfo, kb * TILE, bs.batch, bs.a_bs, bs.linv_bs));
    }
    if (below <= 0) continue;
    if (!fused) { // panel TRSM, in place: A[kb+1.., kb] <- A[kb+1.., kb] * Linv^T
      GemmArgs g = gemm_args(Apan, lda, Li, TILE, Apan, lda, TILE, 1.0, 0.0);
      set_batch(g, bs.batch, bs.a_bs, bs.linv_bs, bs.a_bs);
      GPX_TRY(launch_gemm_nt(ctx, g, below, 1, 0, GPX_PROF_GEMM_OTHER,
                             2.0 * below * TILE * (double)TILE * TILE));
    }
    const int inner_cols = oe - kb - 1;
    if (inner_cols > 0) { // update the rest of the outer block's columns (lower tiles)
      double* Cin = dA + (int64_t)(kb + 1) * TILE * lda + (int64_t)(kb + 1) * TILE;
      GemmArgs g = gemm_args(Apan, lda, Apan, lda, Cin, lda, TILE, -1.0, 1.0);
      set_batch(g, bs.batch, bs.a_bs, bs.a_bs, bs.a_bs);
      g.lower = 1;
      g.ti_off = kb + 1;
      g.tj_off = kb + 1;
      GPX_TRY(launch_gemm_nt(ctx, g, below, inner_cols, 0, GPX_PROF_GEMM_OTHER,
                             2.0 * below * inner_cols * (double)TILE * TILE * TILE));
    }
  }
  return 0;
}

// C[rows r0.., cols c0..c1) -= Pan[rows, ob..oe) * Pan[cols, ob..oe)^T, lower tiles only.
static int trailing_update(gpx_ctx* ctx, double* dA, int64_t lda, int nblk, int extra, int ob,
                           int oe, int r0, int c0, int c1, int prof_cls, const BatchStrides& bs) {
  const int rows = nblk + extra - r0, cols = c1 - c0;
  if (rows <= 0 || cols <= 0) return 0;
  const int K = (oe - ob) * TILE;
  const double* PanR = dA + (int64_t)r0 * TILE * lda + (int64_t)ob * TILE;
  const double* PanC = dA + (int64_t)c0 * TILE * lda + (int64_t)ob * TILE;
  double* Ctr = dA + (int64_t)r0 * TILE * lda + (int64_t)c0 * TILE;
  GemmArgs g = gemm_args(PanR, lda, PanC, lda, Ctr, lda, K, -1.0, 1.0);
  set_batch(g, bs.batch, bs.a_bs, bs.a_bs, bs.a_bs);
  g.lower = 1;
  g.ti_off = r0;
  g.tj_off = c0;
  // algorithmic flops: 2K per updated entry with column <= row
  double entries = 0.0;
  for (int i = 0; i < rows; ++i) {
    const int R = r0 + i; // absolute tile row: full tiles in columns c0 .. min(c1, R) - 1, half a tile on the diagonal
    const int full = (R < c1 ? R : c1) - c0;
    if (full > 0) entries += (double)full * TILE * TILE;
    if (R >= c0 && R < c1) entries += 0.5 * TILE * (TILE + 1.0);
  }
  // measurement mode (gpx_debug_set_serialise_trailing): the bulk update runs ALONE on the chip — it starts when the
  // other stream of the look-ahead has drained and that stream resumes when it is done — so that the HIP events around it
  // (ProfScope) give the kernel's rate on its real operands and shapes without the panel chain sharing the SIMDs
  const bool alone = ctx->serialise_trailing && prof_cls == GPX_PROF_GEMM_TRAILING;
  hipStream_t other = (ctx->s == ctx->stream) ? ctx->pstream : ctx->stream;
  if (alone) {
    GPX_HIP(ctx, hipEventRecord(ctx->evD, other));
    GPX_HIP(ctx, hipStreamWaitEvent(ctx->s, ctx->evD, 0));
  }
  const int rc = launch_gemm_nt(ctx, g, rows, cols, 0, prof_cls, 2.0 * K * entries);
  if (alone && rc >= 0) {
    GPX_HIP(ctx, hipEventRecord(ctx->evD, ctx->s));
    GPX_HIP(ctx, hipStreamWaitEvent(other, ctx->evD, 0));
  }
  return rc;
}

static int ensure_events(gpx_ctx* ctx, int nouter) {
  while ((int)ctx->evP.size() < nouter + 1) {
    hipEvent_t e1, e2;
    GPX_HIP(ctx, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    GPX_HIP(ctx, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    ctx->evP.push_back(e1);
    ctx->evU.push_back(e2);
  }
  return 0;
}

// Lazy far updates (ctx->lazy_group = G outer blocks).  With accumulators that start from -C (gemm_f64.hip) an
// update with K = 512 followed by another with K = 512 is, bit for bit, ONE fma chain of length 1024 — a C tile
// stored and reloaded between the two is the same double.  So how the k range of a tile's updates is cut into
// launches is free, and the bulk of the trailing matrix can take the panels of G outer blocks in one launch
// (K = 512 G: half / a quarter of the C read-modify-write traffic and of the per-tile prologue / epilogue, the
// difference between 56 and 63 / 69 TFLOP/s for this kernel on random data), as long as every tile still receives
// its panels in ascending order.  Outer blocks are grouped G at a time; ge(k) = last block of k's group.
//   P(k)        panel stream: factor outer block k (potf2 / TRSM / inner updates)
//   U1(k)       panel stream: K = 512 update from block k alone of the columns that cannot wait for the group's far
//               update: outer columns k+1 .. ge(k)+1 (P(k+1) needs column k+1 next)
//   far(g)      main stream, at k = ge(k), K = all columns of the group: first the outer columns ge+2 .. ge+G+1
//               ("near-far": the next group's U1 launches write them, and only wait for THIS launch), then the bulk
//               ge+G+2 .. end.
// Column c thus receives whole groups while group_end <= c - 2, then single blocks — ascending throughout; the
// event after the near-far launch orders the next group's singles behind it.  G = 1 is the round-1 schedule
// (U1 = next block's columns, U2 = the rest), except that U1 now waits for the near-far slice only.
int potrf_lower(gpx_ctx* ctx, double* dA, int64_t lda, int np, int extra_tiles, double* dLinv,
                int* dInfo, int batch, int64_t a_bs, int64_t linv_bs) {
  const BatchStrides bs{batch > 1 ? batch : 1, a_bs, linv_bs};
  const int nblk = np / TILE;
  ctx->small_bk_now = ctx->small_bk != 0 ? ctx->small_bk : ((bs.batch == 1 && nblk + extra_tiles <= SMALL_BK_ROWS) ? 32 : 16);
  const bool one_block = !ctx->outer_tiles_set && bs.batch == 1 && nblk <= ONE_BLOCK_TILES && nblk + extra_tiles <= ONE_BLOCK_TILES + 8;
  const int OT = one_block ? nblk : ctx->outer_tiles;
  const int G = ctx->lazy_group > 0 ? ctx->lazy_group : 1;
  // ADAPTIVE schedule (the timeline of one C3 factorisation, profiles/r02/timeline_c3.md): while the remaining matrix is
  // large the trailing updates follow each other without gaps and the panel chain hides behind them (GEMM-bound
  // "head"): bulk updates take `lazy_group` outer blocks at a time (larger K).  Once an outer block's chain takes
  // longer than the trailing update it overlaps with, the chain sets the pace (chain-bound "tail": a block that leaves
  // fewer than `tail_tiles` tile rows behind it): one outer block per update.  Every combination gives the same bits.
  const int tail_tiles = ctx->tail_tiles;
  const int nouter = (nblk + OT - 1) / OT;
  // latency-shape kernel of this call's chain launches (common.h lat_gemm): round 5 with the chip to itself, round 1 beside
  // the trailing updates of the blocked schedule; whatever follows this call starts from round 5 again
  ctx->lat_now = (nouter == 1) ? 5 : 1;
  struct LatReset {
    gpx_ctx* c;
    ~LatReset() { c->lat_now = 5; }
  } lat_reset{ctx};
  if (nouter == 1) {
    // one outer block (N <= 512, and single-sample factorisations up to ONE_BLOCK_TILES tile rows): nothing to look ahead over — the chain runs on the main stream, without the two
    // cross-stream event waits of the look-ahead (~10 - 18 us each at this size)
    ctx->s = ctx->stream;
    return panel_block(ctx, dA, lda, nblk, extra_tiles, 0, nblk, dLinv, dInfo, bs);
  }
  GPX_TRY(ensure_events(ctx, nouter));
  hipStream_t smain = ctx->stream, span = ctx->pstream;
  auto ob_of = [&](int k) { return k < nouter ? k * OT : nblk; }; // first tile column of outer block k (clamped)
  auto in_tail = [&](int k) { return k < nouter && tail_tiles > 0 && (nblk - (k * OT + OT) + extra_tiles) < tail_tiles; };
  // the panel stream starts after everything already queued on the main stream (Gram etc.)
  GPX_HIP(ctx, hipEventRecord(ctx->evU[nouter], smain));
  GPX_HIP(ctx, hipStreamWaitEvent(span, ctx->evU[nouter], 0));
  int rc = 0;
  // groups: consecutive head blocks are grouped G at a time, tail blocks stay single
  std::vector<int> gfirst((size_t)nouter), glast((size_t)nouter);
  for (int k = 0; k < nouter;) {
    int len = 1;
    if (!in_tail(k))
      while (len < G && k + len < nouter && !in_tail(k + len)) ++len;
    for (int j = 0; j < len; ++j) {
      gfirst[(size_t)(k + j)] = k;
      glast[(size_t)(k + j)] = k + len - 1;
    }
    k += len;
  }
  for (int k = 0; k < nouter && rc >= 0; ++k) {
    const int ob = ob_of(k), oe = ob_of(k + 1);
    const int gs = gfirst[(size_t)k], ge = glast[(size_t)k];
    // The END of a large single-sample factorisation is a mid-size one: once no more than FINISH_ONE_BLOCK_TILES tile rows are
    // left and block k starts a group (its columns have every update from the blocks before it through the U1
    // launches, the columns behind it through the far updates issued so far), the rest runs as one outer block on the
    // panel stream, behind whatever the main stream still has in flight — as a factorisation of that size would.
    if (k > 0 && k == gs && !ctx->outer_tiles_set && bs.batch == 1 && nblk - ob <= FINISH_ONE_BLOCK_TILES &&
        nblk - ob + extra_tiles <= FINISH_ONE_BLOCK_TILES + 8) {
      GPX_HIP(ctx, hipEventRecord(ctx->evD, smain));
      GPX_HIP(ctx, hipStreamWaitEvent(span, ctx->evD, 0));
      ctx->s = span;
      if (ctx->small_bk == 0) ctx->small_bk_now = 32;
      rc = panel_block(ctx, dA, lda, nblk, extra_tiles, ob, nblk, dLinv, dInfo, bs);
      break;
    }
    // P(k) on the panel stream (it follows U1(k-1) there, in stream order)
    ctx->s = span;
    rc = panel_block(ctx, dA, lda, nblk, extra_tiles, ob, oe, dLinv, dInfo, bs);
    if (rc < 0) break;
    GPX_HIP(ctx, hipEventRecord(ctx->evP[k], span));
    if (oe >= nblk) {
      GPX_HIP(ctx, hipStreamWaitEvent(smain, ctx->evP[k], 0));
      break;
    }
    // U1(k): block k alone (K = its columns) onto outer columns k+1 .. ge+1, on the PANEL stream.  The first block of
    // a group writes columns the previous group's far update also wrote: wait for the launch that did (fixed order).
    if (k == gs && k > 0) GPX_HIP(ctx, hipStreamWaitEvent(span, ctx->evU[k - 1], 0));
    const int u1_end = ob_of(ge + 2);
    rc = trailing_update(ctx, dA, lda, nblk, extra_tiles, ob, oe, oe, oe, u1_end, GPX_PROF_GEMM_OTHER, bs);
    if (rc < 0) break;
    if (k == ge) { // far update of the whole group on the main stream
      GPX_HIP(ctx, hipStreamWaitEvent(smain, ctx->evP[k], 0));
      // Deep in the chain-bound tail (fewer than FAR_AFTER_U1 tile rows left: the far update is shorter than the chain it
      // hides behind) it starts only when U1(k) is done, so that U1 — on the chain — has the chip to itself instead of
      // sharing it with the update that has time to spare (potrf -0.5 ... -2.1 % at N = 4096 ... 16384).  Single-sample
      // launches only: a batched update is B times longer and sets the pace itself.
      if (bs.batch == 1 && (nblk - oe + extra_tiles) < FAR_AFTER_U1) {
        GPX_HIP(ctx, hipEventRecord(ctx->evD, span));
        GPX_HIP(ctx, hipStreamWaitEvent(smain, ctx->evD, 0));
      }
      const int gob = ob_of(gs);
      // The next group's U1 launches write outer columns ge+2 .. ge_next+1.  If that group has several blocks those
      // columns get a ("near-far") launch of their own, so that its first U1 need not wait for the bulk; a single
      // next block is not worth a launch of < 512 tiles (61 launches at 32.4 ms against 31 at 28.7 ms per
      // factorisation): one launch then, the round-1 schedule.
      const int gn = (k + 1 < nouter) ? glast[(size_t)(k + 1)] : k + 1;
      const int n0 = ob_of(ge + 2), n1 = (gn - ge >= 2) ? ob_of(gn + 2) : nblk;
      ctx->s = smain;
      rc = trailing_update(ctx, dA, lda, nblk, extra_tiles, gob, oe, n0, n0, n1, GPX_PROF_GEMM_TRAILING, bs);
      if (rc < 0) break;
      GPX_HIP(ctx, hipEventRecord(ctx->evU[k], smain)); // what the next group's first U1 waits for
      rc = trailing_update(ctx, dA, lda, nblk, extra_tiles, gob, oe, n1, n1, nblk, GPX_PROF_GEMM_TRAILING, bs);
    }
  }
  // everything queued on the panel stream happens-before whatever follows on the main stream
  GPX_HIP(ctx, hipEventRecord(ctx->evP[nouter], span));
  GPX_HIP(ctx, hipStreamWaitEvent(smain, ctx->evP[nouter], 0));
  ctx->s = smain;
  return rc;
}

int potrf_steps(gpx_ctx* ctx, double* dA, int64_t lda, int nblk, int kb0, int kb1, double* dLinv, int* dInfo) {
  const BatchStrides bs{1, 0, 0};
  ctx->small_bk_now = ctx->small_bk != 0 ? ctx->small_bk : (nblk <= SMALL_BK_ROWS ? 32 : 16);
  return panel_block(ctx, dA, lda, nblk, 0, kb0, nblk, dLinv, dInfo, bs, kb1 < nblk ? kb1 : nblk);
}

// ---- right-looking solve of  X * L^T = B  in place (B: rows_t*128 x nblk*128) --------------
// Column block i:  X_i = B_i * Linv_i^T, then B_{>i} -= X_i * L[>i, i]^T.  Two-level blocking
// as in potrf_lower.  upper_rows != 0: B is upper triangular (the L^-T build of the gradient),
// so column block i only involves row tiles 0..i.
// Same two-stream look-ahead as potrf_lower: the latency-bound column steps of outer block k + 1 and the
// update of just its columns (U1) run on the panel stream while the bulk update U2(k) of everything further
// right runs on the main stream.
int trsm_right_lt(gpx_ctx* ctx, double* dB, int64_t ldb, int rows_t, const double* dL,
                  int64_t ldl, const double* dLinv, int nblk, int upper_rows, int batch,
                  int64_t b_bs, int64_t l_bs, int64_t linv_bs) {
  if (batch < 1) batch = 1;
  ctx->small_bk_now = ctx->small_bk != 0 ? ctx->small_bk : ((batch == 1 && (rows_t > nblk ? rows_t : nblk) <= SMALL_BK_ROWS) ? 32 : 16);
  const int OT = ctx->outer_tiles;
  const int nouter = (nblk + OT - 1) / OT;
  GPX_TRY(ensure_events(ctx, nouter));
  hipStream_t smain = ctx->stream, span = ctx->pstream;
  ctx->lat_now = (nouter == 1) ? 5 : 1; // (as in potrf_lower)
  struct LatReset {
    gpx_ctx* c;
    ~LatReset() { c->lat_now = 5; }
  } lat_reset{ctx};
  GPX_HIP(ctx, hipEventRecord(ctx->evU[nouter], smain));
  GPX_HIP(ctx, hipStreamWaitEvent(span, ctx->evU[nouter], 0));
  // This sweep's panel chain has no diagonal-block factorisation (the blocks of L come inverted): every chain launch
  // takes a shape that fits next to two resident big-tile workgroups, so the bulk updates can run as persistent,
  // dynamically scheduled launches (gemm_f64.hip) — measured on the fit step at C3: 86.1 -> 81.3 ms.
  struct Scope {
    gpx_ctx* c;
    explicit Scope(gpx_ctx* c_) : c(c_) { if (c->persist_scope_ok) c->persist_scope += 1; }
    ~Scope() { if (c->persist_scope_ok) c->persist_scope -= 1; }
  } scope(ctx);
  // rest update of outer block [ob, oe): B[:, c0:c1) -= B[:, ob:oe) L[c0:c1, ob:oe)^T
  auto rest_update = [&](int ob, int oe, int c0, int c1) -> int {
    if (c1 <= c0) return 0;
    const int rt = upper_rows ? oe : rows_t;
    const int K = (oe - ob) * TILE;
    const double* Lsub = dL + (int64_t)c0 * TILE * ldl + (int64_t)ob * TILE;
    GemmArgs g = gemm_args(dB + (int64_t)ob * TILE, ldb, Lsub, ldl, dB + (int64_t)c0 * TILE, ldb, K, -1.0, 1.0);
    set_batch(g, batch, b_bs, l_bs, b_bs);
    return launch_gemm_nt(ctx, g, rt, c1 - c0, 0, GPX_PROF_GEMM_OTHER, 2.0 * rt * (c1 - c0) * (double)TILE * TILE * K);
  };
  int rc = 0;
  for (int k = 0; k < nouter && rc >= 0; ++k) {
    const int ob = k * OT;
    const int oe = (ob + OT < nblk) ? ob + OT : nblk;
    const int oe2 = (oe + OT < nblk) ? oe + OT : nblk;
    ctx->s = span;
    for (int i = ob; i < oe && rc >= 0; ++i) {
      const int rt = upper_rows ? (i + 1) : rows_t;
      double* Bi = dB + (int64_t)i * TILE;
      {
        GemmArgs g = gemm_args(Bi, ldb, dLinv + (int64_t)i * TILE * TILE, TILE, Bi, ldb, TILE, 1.0,
                               0.0);
        set_batch(g, batch, b_bs, linv_bs, b_bs);
        rc = launch_gemm_nt(ctx, g, rt, 1, 0, GPX_PROF_GEMM_OTHER, 2.0 * rt * TILE * (double)TILE * TILE);
        if (rc < 0) break;
      }
      const int inner_cols = oe - i - 1;
      if (inner_cols > 0) {
        const double* Lsub = dL + (int64_t)(i + 1) * TILE * ldl + (int64_t)i * TILE;
        GemmArgs g = gemm_args(Bi, ldb, Lsub, ldl, dB + (int64_t)(i + 1) * TILE, ldb, TILE, -1.0,
                               1.0);
        set_batch(g, batch, b_bs, l_bs, b_bs);
        rc = launch_gemm_nt(ctx, g, rt, inner_cols, 0, GPX_PROF_GEMM_OTHER,
                            2.0 * rt * inner_cols * (double)TILE * TILE * TILE);
      }
    }
    if (rc < 0) break;
    GPX_HIP(ctx, hipEventRecord(ctx->evP[k], span));
    GPX_HIP(ctx, hipStreamWaitEvent(smain, ctx->evP[k], 0));
    if (oe < nblk) {
      // U1(k): the next outer block's columns, after U2(k-1) which also wrote them (fixed accumulation order)
      if (k > 0) GPX_HIP(ctx, hipStreamWaitEvent(span, ctx->evU[k - 1], 0));
      rc = rest_update(ob, oe, oe, oe2);
      if (rc < 0) break;
      ctx->s = smain;
      // (starting a short far update only after U1, as potrf_lower does in its tail, was measured here too: the fit
      // step gets 0.4 - 0.9 % slower — profiles/r02/chain_experiments.md)
      rc = rest_update(ob, oe, oe2, nblk);
      GPX_HIP(ctx, hipEventRecord(ctx->evU[k], smain));
    }
  }
  ctx->s = smain;
  return rc;
}

// ---- W = L^-T (upper) by the block-recursive inverse, bottom-up --------------------------------
// L = [[A, 0], [B, C]]  =>  L^-T = [[A^-T, Y], [0, C^-T]],  Y = -A^-T B^T C^-T.
// Level s (s = 1, 2, 4, ... 128-tiles): every aligned pair of finished diagonal blocks (A: s tiles, C: up to s tiles)
// gets its Y from two GEMMs whose k ranges are trimmed to the triangles:
//   T = W_A B^T            (A operand upper triangular: k starts at the row tile          — ktri)
//   Y = -T V_C^T, V_C = W_C^T   (B operand lower triangular: k ends at the column tile    — kupper)
// The NT kernels read both operands by rows, so W_C is transposed once per level (a bandwidth-bound pass over
// s x s per pair).  All pairs of a level — and all samples of a batched fit step — go in ONE launch each (two-level
// batch, gemm_tile.h), a ragged last pair (C shorter than A) in its own.  There is no serial column chain: the
// right-looking sweep (trsm_right_lt with upper_rows) needs 2 launches per 128 columns that each wait for the one
// before; this needs 3 per level, log2(n / 128) levels, and 98 % of the flops sit in the three top levels with
// K >= 1024.  Same N^3 / 3 flops.  The diagonal 128-blocks come from potf2's inverses (level 0: W_kk = Linv_k^T).
// Only tiles on or above the diagonal of W are written; nothing downstream reads the others (rowdot starts at the
// row's own tile, the K^-1 product runs with ktri).  dS: scratch of the same shape as W (T and V_C of one level).
__global__ __launch_bounds__(256) void transpose_blocks_kernel(const double* __restrict__ in, int64_t ldi,
                                                               double* __restrict__ out, int64_t ldo, int n,
                                                               int inner, int64_t in_bs, int64_t in_bs2,
                                                               int64_t out_bs, int64_t out_bs2, int upper_only) {
  __shared__ double t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x; // 32 x 32 block (bi, bj) of the input
  // upper_only: the input is upper triangular in 128-tiles — tiles strictly below the diagonal hold nothing
  if (upper_only && (bi >> 2) > (bj >> 2)) return;
  const int e = blockIdx.z, e1 = e / inner, e2 = e - e1 * inner;
  in += (int64_t)e1 * in_bs + (int64_t)e2 * in_bs2;
  out += (int64_t)e1 * out_bs + (int64_t)e2 * out_bs2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    t[r][tx] = (i < n && j < n) ? in[(int64_t)i * ldi + j] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r, j = bi * 32 + tx;
    if (i < n && j < n) out[(int64_t)i * ldo + j] = t[tx][r];
  }
}

int linv_t_tree(gpx_ctx* ctx, double* dW, int64_t ldw, const double* dL, int64_t ldl, const double* dLinv, int nt,
                double* dS, int64_t lds_, int batch, int64_t w_bs, int64_t l_bs, int64_t linv_bs, int64_t s_bs) {
  if (batch < 1) batch = 1;
  ctx->small_bk_now = ctx->small_bk != 0 ? ctx->small_bk : 16;
  auto transpose = [&](const double* in, int64_t ldi, double* out, int64_t ldo, int n, int inner, int64_t in_bs,
                       int64_t in_bs2, int64_t out_bs, int64_t out_bs2, int upper_only) -> int {
    dim3 grid((n + 31) / 32, (n + 31) / 32, batch * inner);
    transpose_blocks_kernel<<<grid, 256, 0, ctx->s>>>(in, ldi, out, ldo, n, inner, in_bs, in_bs2, out_bs, out_bs2,
                                                      upper_only);
    GPX_HIP(ctx, hipGetLastError());
    return 0;
  };
  // level 0: the diagonal 128-blocks
  GPX_TRY(transpose(dLinv, TILE, dW, ldw, TILE, nt, linv_bs, (int64_t)TILE * TILE, w_bs, (int64_t)TILE * (ldw + 1), 0));
  struct Scope { // tiles of very different length (trimmed k ranges), nothing else in flight: persistent, dynamically scheduled
    gpx_ctx* c;
    explicit Scope(gpx_ctx* c_) : c(c_) { if (c->persist_scope_ok) c->persist_scope += 1; }
    ~Scope() { if (c->persist_scope_ok) c->persist_scope -= 1; }
  } scope(ctx);
  // pairs [first, first + count) of level s, C blocks c tiles wide
  auto level = [&](int s, int first, int count, int c) -> int {
    if (count <= 0 || c <= 0) return 0;
    const int64_t a0 = (int64_t)first * 2 * s * TILE;   // first row / column of the first pair's A block
    const int64_t c0 = a0 + (int64_t)s * TILE;          //                                        C block
    const int64_t pw = (int64_t)2 * s * TILE * (ldw + 1), pl = (int64_t)2 * s * TILE * (ldl + 1),
                  ps = (int64_t)2 * s * TILE * (lds_ + 1);
    const double t3 = (double)TILE * TILE * TILE;
    double* Vc = dS + c0 * lds_ + c0;
    double* Tm = dS + a0 * lds_ + c0;
    GPX_TRY(transpose(dW + c0 * ldw + c0, ldw, Vc, lds_, c * TILE, count, w_bs, pw, s_bs, ps, 1));
    {
      GemmArgs g = gemm_args(dW + a0 * ldw + a0, ldw, dL + c0 * ldl + a0, ldl, Tm, lds_, s * TILE, 1.0, 0.0);
      g.ktri = 1;
      set_batch(g, batch * count, w_bs, l_bs, s_bs);
      g.batch2 = count;
      g.a_bs2 = pw;
      g.b_bs2 = pl;
      g.c_bs2 = ps;
      GPX_TRY(launch_gemm_nt(ctx, g, s, c, 0, GPX_PROF_GEMM_OTHER, (double)s * s * c * t3));
    }
    {
      GemmArgs g = gemm_args(Tm, lds_, Vc, lds_, dW + a0 * ldw + c0, ldw, c * TILE, -1.0, 0.0);
      g.kupper = 1;
      set_batch(g, batch * count, s_bs, s_bs, w_bs);
      g.batch2 = count;
      g.a_bs2 = ps;
      g.b_bs2 = ps;
      g.c_bs2 = pw;
      GPX_TRY(launch_gemm_nt(ctx, g, s, c, 0, GPX_PROF_GEMM_OTHER, (double)s * c * c * t3));
    }
    return 0;
  };
  for (int s = 1; s < nt; s *= 2) {
    const int full = nt / (2 * s), rest = nt - full * 2 * s;
    GPX_TRY(level(s, 0, full, s));
    if (rest > s) GPX_TRY(level(s, full, 1, rest - s));
  }
  return 0;
}

// ---- small kernels ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void set_identity_kernel(double* __restrict__ A, int64_t ld,
                                                           int np, int64_t a_bs) {
  const int i = blockIdx.y;
  double* row = A + (int64_t)blockIdx.z * a_bs + (int64_t)i * ld;
  for (int j = (blockIdx.x * 256 + threadIdx.x) * 2; j < np; j += gridDim.x * 512) {
    *reinterpret_cast<double2*>(row + j) = make_double2(j == i ? 1.0 : 0.0, j + 1 == i ? 1.0 : 0.0);
  }
}

int launch_set_identity(gpx_ctx* ctx, double* dA, int64_t ld, int np, int batch, int64_t a_bs) {
  dim3 grid(min(32, (np / 2 + 255) / 256), np, batch > 1 ? batch : 1);
  set_identity_kernel<<<grid, 256, 0, ctx->s>>>(dA, ld, np, a_bs);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Block-wide deterministic sum (fixed tree): result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* red /* >= 16 doubles */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
  }
  return s;
}

// out[0] = sum_{k<N} L[N][k]^2 (= |L^-1 y|^2: row N of the augmented factor),
// out[1] = sum_{i<N} log L[i][i].
__global__ __launch_bounds__(1024) void lml_terms_kernel(const double* __restrict__ L, int64_t ld,
                                                         int N, double* __restrict__ out,
                                                         int64_t l_bs, int64_t out_bs) {
  __shared__ double red[16];
  L += (int64_t)blockIdx.x * l_bs; // one workgroup per batch entry
  out += (int64_t)blockIdx.x * out_bs;
  double q = 0.0, s = 0.0;
  const double* w = L + (int64_t)N * ld;
  for (int k = threadIdx.x; k < N; k += 1024) {
    const double wk = w[k];
    q = fma(wk, wk, q);
    s += log(L[(int64_t)k * ld + k]);
  }
  const double qs = block_sum(q, red);
  const double ss = block_sum(s, red);
  if (threadIdx.x == 0) {
    out[0] = qs;
    out[1] = ss;
  }
}

int launch_lml_terms(gpx_ctx* ctx, const double* dL, int64_t ld, int N, double* dOut2, int batch,
                     int64_t l_bs, int64_t out_bs) {
  lml_terms_kernel<<<batch > 1 ? batch : 1, 1024, 0, ctx->s>>>(dL, ld, N, dOut2, l_bs, out_bs);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// mean[r] = sum_k V[r][k] w[k];  var[r] = kdiag - sum_k V[r][k]^2.  WPR = 1: one wave per row, four rows per workgroup.
// WPR = 4 (rows of >= 4096 columns): the four waves of a workgroup share one row, k interleaved in steps of 256, partial
// sums added in wave order — a 2048 x 16316 operand (u = W^T y of the sparse bound, the predictive mean / variance at
// C3) is otherwise 2048 waves on a 256-CU chip: 2 TB/s.  Which form a launch takes depends on `cols` only, so single
// and batched launches of one shape keep giving the same bits.
// col_start_by_row: V is upper triangular, start at k = r (alpha = L^-T w).
template <int WPR>
__global__ __launch_bounds__(256) void rowdot_kernel(const double* __restrict__ V, int64_t ldv,
                                                     int rows, int cols,
                                                     const double* __restrict__ w, double kdiag,
                                                     double* __restrict__ mean,
                                                     double* __restrict__ var,
                                                     int col_start_by_row, int64_t v_bs,
                                                     int64_t w_bs, int64_t out_bs,
                                                     const ThetaDev* __restrict__ th,
                                                     const double* __restrict__ pred_diag,
                                                     int64_t pd_bs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = (WPR == 1) ? blockIdx.x * 4 + wave : blockIdx.x;
  if (r >= rows) return;
  const int bb = blockIdx.y; // batch entry
  V += (int64_t)bb * v_bs;
  w += (int64_t)bb * w_bs;
  if (th != nullptr) kdiag = th[bb].kdiag_pred;
  const double* v = V + (int64_t)r * ldv;
  double m = 0.0, q = 0.0;
  int k0 = col_start_by_row ? (r & ~63) : 0;
  for (int k = k0 + (WPR == 1 ? 0 : wave * 64) + lane; k < cols; k += 64 * WPR) {
    const double x = (col_start_by_row && k < r) ? 0.0 : v[k];
    m = fma(x, w[k], m);
    q = fma(x, x, q);
  }
  m = wave_sum(m);
  q = wave_sum(q);
  if constexpr (WPR > 1) {
    __shared__ double red[2][WPR];
    if (lane == 0) {
      red[0][wave] = m;
      red[1][wave] = q;
    }
    __syncthreads();
    if (wave != 0) return;
    m = red[0][0];
    q = red[1][0];
#pragma unroll
    for (int i = 1; i < WPR; ++i) {
      m += red[0][i];
      q += red[1][i];
    }
  }
  if (lane == 0) {
    if (mean) mean[(int64_t)bb * out_bs + r] = m;
    if (var) var[(int64_t)bb * out_bs + r] = kdiag - q + (pred_diag ? pred_diag[(int64_t)bb * pd_bs + r] : 0.0);
  }
}

int launch_rowdot(gpx_ctx* ctx, const double* dV, int64_t ldv, int rows, int cols,
                  const double* dw, double kdiag, double* dmean, double* dvar,
                  int col_start_by_row, int batch, int64_t v_bs, int64_t w_bs, int64_t out_bs,
                  const ThetaDev* th, const double* pred_diag, int64_t pd_bs) {
  if (rows <= 0) return 0;
  if (cols >= 4096) {
    dim3 grid(rows, batch > 1 ? batch : 1);
    rowdot_kernel<4><<<grid, 256, 0, ctx->s>>>(dV, ldv, rows, cols, dw, kdiag, dmean, dvar, col_start_by_row, v_bs, w_bs,
                                               out_bs, th, pred_diag, pd_bs);
  } else {
    dim3 grid((rows + 3) / 4, batch > 1 ? batch : 1);
    rowdot_kernel<1><<<grid, 256, 0, ctx->s>>>(dV, ldv, rows, cols, dw, kdiag, dmean, dvar, col_start_by_row, v_bs, w_bs,
                                               out_bs, th, pred_diag, pd_bs);
  }
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

__device__ __forceinline__ double kval_rt(int kind, double r2, double scale) {
  if (kind == GPX_KERNEL_RBF) return scale * exp(-0.5 * r2);
  if (kind == GPX_KERNEL_PERIODIC) return scale * exp(-2.0 * r2);
  const double r = sqrt(r2 + MATERN_EPS);
  const double s5r = SQRT5 * r;
  return scale * (1.0 + s5r + (5.0 / 3.0) * r2) * exp(-s5r);
}

// cov[a][b] = k_pp(a, b) - sum_z P_z[max(a,b)][min(a,b)]   (a, b < M); identity padding.
// k_pp = kernel(X_new, X_new, theta, noise_p, jitter)  (gpax/models/gp.py:267) evaluated on
// the fly; the split-K slabs only hold lower tiles.
__global__ __launch_bounds__(256) void cov_finalize_kernel(KernelParams kpv,
                                                           const double* __restrict__ Xn, int M,
                                                           int Mp, const double* __restrict__ P,
                                                           int splits, int64_t split_stride,
                                                           int64_t ldp, double diag_add,
                                                           double* __restrict__ Cov, int64_t ldc,
                                                           int64_t part_bs, int64_t cov_bs,
                                                           const ThetaDev* __restrict__ th,
                                                           TaskStride ts,
                                                           const double* __restrict__ pred_diag,
                                                           int64_t pd_bs) {
  if (ts.mod > 0) Xn += (blockIdx.z % ts.mod) * ts.x_bs; // per-task X_new
  const int b = blockIdx.x * 64 + (threadIdx.x & 63);
  const int a0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
  if (b >= Mp) return;
  const ThetaDev* t = (th != nullptr) ? th + blockIdx.z : nullptr; // batch entry blockIdx.z
  const int d = kpv.d, kind = kpv.kind;                            // structural, same for the batch
  const double k_scale = t ? t->kp.scale : kpv.scale;
  const double pi_over_p = t ? t->kp.pi_over_p : kpv.pi_over_p;
  auto inv_ell = [&](int c) -> double { return t ? t->kp.inv_ell[c] : kpv.inv_ell[c]; };
  if (t != nullptr) diag_add = t->diag_pred;
  P += (int64_t)blockIdx.z * part_bs;
  Cov += (int64_t)blockIdx.z * cov_bs;
  for (int t = 0; t < 4; ++t) {
    const int a = a0 + t;
    if (a >= Mp) return;
    double v;
    if (a < M && b < M) {
      double r2 = 0.0;
      for (int c = 0; c < d; ++c) {
        double u = Xn[(int64_t)a * d + c] - Xn[(int64_t)b * d + c];
        u = (kind == GPX_KERNEL_PERIODIC) ? sin(u * pi_over_p) * inv_ell(c) : u * inv_ell(c);
        r2 = fma(u, u, r2);
      }
      v = kval_rt(kind, r2, k_scale);
      if (a == b) v += diag_add + (pred_diag ? pred_diag[(int64_t)blockIdx.z * pd_bs + a] : 0.0);
      const int hi = a > b ? a : b, lo = a > b ? b : a;
      double acc = 0.0;
      for (int z = 0; z < splits; ++z) acc += P[(int64_t)z * split_stride + (int64_t)hi * ldp + lo];
      v -= acc;
    } else {
      v = (a == b) ? 1.0 : 0.0;
    }
    Cov[(int64_t)a * ldc + b] = v;
  }
}

int launch_cov_finalize(gpx_ctx* ctx, const KernelParams& kp, const double* dXnew, int M, int Mp,
                        const double* dPart, int splits, int64_t split_stride, int64_t ldp,
                        double diag_add, double* dCov, int64_t ldc, int batch, int64_t part_bs,
                        int64_t cov_bs, const ThetaDev* th, TaskStride ts, const double* pred_diag,
                        int64_t pd_bs) {
  dim3 grid((Mp + 63) / 64, (Mp + 15) / 16, batch > 1 ? batch : 1);
  cov_finalize_kernel<<<grid, 256, 0, ctx->s>>>(kp, dXnew, M, Mp, dPart, splits, split_stride,
                                                ldp, diag_add, dCov, ldc, part_bs, cov_bs, th, ts, pred_diag, pd_bs);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// ---- gradient contraction --------------------------------------------------------------------
// d lml / d theta = 1/2 sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta, accumulated over the
// lower triangle (off-diagonal weight 1, diagonal weight 1/2) with dK evaluated on the fly:
//   d/d scale : k_ij / scale          d/d noise : delta_ij
//   d/d ell_m : dk/dr2 * (-2 u_m^2 / ell_m),  u_m = (x_im - x_jm) / ell_m
//   RBF: dk/dr2 = -k/2;  Matern52: dk/dr2 = -(5/6) s e^{-sqrt5 r} (1 + sqrt5 r2 / r).
// Output per block: [g_ell(0..d), g_scale, g_noise]; reduced in fixed order afterwards.
constexpr int GC_TILE = 64;
constexpr int GC_MAXV = GPX_MAX_DIM + 3;

// KIND / D are compile-time (D = 0: generic d <= 16) so the per-dimension accumulators stay in
// registers; batch entry blockIdx.y reads its hyper-parameters from the device table `th`.
template <int KIND, int D>
__global__ __launch_bounds__(256) void grad_contract_kernel(KernelParams kpv,
                                                            const double* __restrict__ X, int N,
                                                            const double* __restrict__ Kinv,
                                                            int64_t ld,
                                                            const double* __restrict__ alpha,
                                                            double* __restrict__ part,
                                                            int64_t k_bs, int64_t alpha_bs,
                                                            int64_t part_bs,
                                                            const ThetaDev* __restrict__ th,
                                                            TaskStride ts) {
  constexpr int DM = (D > 0) ? D : GPX_MAX_DIM;
  if (ts.mod > 0) X += (blockIdx.y % ts.mod) * ts.x_bs; // per-task training inputs
  const ThetaDev* t = (th != nullptr) ? th + blockIdx.y : nullptr;
  const double k_scale = t ? t->kp.scale : kpv.scale;
  const double pi_over_p = t ? t->kp.pi_over_p : kpv.pi_over_p;
  const int d = (D > 0) ? D : kpv.d;
  double inv_ell[DM];
#pragma unroll
  for (int c = 0; c < DM; ++c) inv_ell[c] = (c < d) ? (t ? t->kp.inv_ell[c] : kpv.inv_ell[c]) : 0.0;
  Kinv += (int64_t)blockIdx.y * k_bs;
  alpha += (int64_t)blockIdx.y * alpha_bs;
  part += (int64_t)blockIdx.y * part_bs;
  // linear block id -> lower-triangle tile (ti >= tj)
  const int bid = blockIdx.x;
  int ti = (int)((sqrt(8.0 * bid + 1.0) - 1.0) * 0.5);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > bid) --ti;
  const int tj = bid - (int)((int64_t)ti * (ti + 1) / 2);
  constexpr bool PER = (KIND == GPX_KERNEL_PERIODIC);
  const int ne = d + (PER ? 1 : 0); // [ell(0..d), (period), scale, noise]
  __shared__ double red[16];
  double acc_ell[DM], acc_p = 0.0, acc_s = 0.0, acc_n = 0.0;
#pragma unroll
  for (int c = 0; c < DM; ++c) acc_ell[c] = 0.0;
  const int j = tj * GC_TILE + (threadIdx.x & 63);
  const int ibase = ti * GC_TILE + (threadIdx.x >> 6);
  if (j < N) {
    const double aj = alpha[j];
    double xj[DM];
#pragma unroll
    for (int c = 0; c < DM; ++c) xj[c] = (c < d) ? X[(int64_t)j * d + c] : 0.0;
    for (int tt = 0; tt < GC_TILE / 4; ++tt) {
      const int i = ibase + 4 * tt;
      if (i >= N || j > i) continue;
      const double G = alpha[i] * aj - Kinv[(int64_t)i * ld + j];
      const double wgt = (i == j) ? 0.5 : 1.0;
      const double wg = wgt * G;
      if (PER) {
        // k = s exp(-2 sum (S_m / l_m)^2), S_m = sin(pi delta_m / p)
        double q = 0.0, dp = 0.0, s2[DM];
#pragma unroll
        for (int c = 0; c < DM; ++c) {
          s2[c] = 0.0;
          if (c < d) {
            const double delta = X[(int64_t)i * d + c] - xj[c];
            const double sn = sin(delta * pi_over_p), cs = cos(delta * pi_over_p);
            s2[c] = sn * sn * inv_ell[c] * inv_ell[c];
            q += s2[c];
            dp += sn * cs * delta * inv_ell[c] * inv_ell[c];
          }
        }
        const double kv = k_scale * exp(-2.0 * q);
#pragma unroll
        for (int c = 0; c < DM; ++c) acc_ell[c] += wg * kv * 4.0 * s2[c] * inv_ell[c];          // d/d l_m
        acc_p += wg * kv * 4.0 * dp * pi_over_p * pi_over_p / 3.14159265358979323846;          // d/d p
        acc_s += wg * kv / k_scale;
      } else {
        double r2 = 0.0;
        double u2[DM];
#pragma unroll
        for (int c = 0; c < DM; ++c) {
          u2[c] = 0.0;
          if (c < d) {
            const double u = (X[(int64_t)i * d + c] - xj[c]) * inv_ell[c];
            u2[c] = u * u;
            r2 += u2[c];
          }
        }
        double kv, dk;
        if (KIND == GPX_KERNEL_RBF) {
          kv = k_scale * exp(-0.5 * r2);
          dk = -0.5 * kv;
        } else {
          const double r = sqrt(r2 + MATERN_EPS);
          const double e = exp(-SQRT5 * r);
          kv = k_scale * (1.0 + SQRT5 * r + (5.0 / 3.0) * r2) * e;
          dk = -(5.0 / 6.0) * k_scale * e * (1.0 + SQRT5 * r2 / r);
        }
#pragma unroll
        for (int c = 0; c < DM; ++c) acc_ell[c] += wg * dk * (-2.0 * u2[c] * inv_ell[c]);
        acc_s += wg * kv / k_scale;
      }
      if (i == j) acc_n += wg;
    }
  }
  double* out = part + (int64_t)bid * GC_MAXV;
#pragma unroll
  for (int c = 0; c < DM; ++c) {
    if (c < d) { // uniform condition
      const double sum = block_sum(acc_ell[c], red);
      if (threadIdx.x == 0) out[c] = sum;
    }
  }
  if (PER) {
    const double sum = block_sum(acc_p, red);
    if (threadIdx.x == 0) out[d] = sum;
  }
  {
    const double ss = block_sum(acc_s, red);
    const double sn = block_sum(acc_n, red);
    if (threadIdx.x == 0) {
      out[ne] = ss;
      out[ne + 1] = sn;
    }
  }
}

__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* __restrict__ part,
                                                          int nblocks, int nvals,
                                                          double* __restrict__ out, int64_t part_bs,
                                                          int64_t out_bs) {
  __shared__ double red[16];
  part += (int64_t)blockIdx.x * part_bs; // one workgroup per batch entry
  out += (int64_t)blockIdx.x * out_bs;
  for (int c = 0; c < nvals; ++c) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(int64_t)b * GC_MAXV + c];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) out[c] = t;
  }
}

int launch_grad_contract(gpx_ctx* ctx, const KernelParams& kp, const double* dX, int N,
                         const double* dKinv, int64_t ld, const double* dalpha, double* dpart,
                         int* nblocks_out, int batch, int64_t k_bs, int64_t alpha_bs,
                         const ThetaDev* th, TaskStride ts) {
  const int nt = (N + GC_TILE - 1) / GC_TILE;
  const int nblocks = nt * (nt + 1) / 2;
  *nblocks_out = nblocks;
  dim3 grid(nblocks, batch > 1 ? batch : 1);
  const int64_t part_bs = (int64_t)nblocks * GC_MAXV;
#define GPX_GC_LAUNCH(KIND, DD)                                                                         \
  grad_contract_kernel<KIND, DD><<<grid, 256, 0, ctx->s>>>(kp, dX, N, dKinv, ld, dalpha, dpart, k_bs, \
                                                           alpha_bs, part_bs, th, ts)
#define GPX_GC_KIND(KIND)            \
  switch (kp.d) {                    \
    case 1: GPX_GC_LAUNCH(KIND, 1); break; \
    case 2: GPX_GC_LAUNCH(KIND, 2); break; \
    case 3: GPX_GC_LAUNCH(KIND, 3); break; \
    case 4: GPX_GC_LAUNCH(KIND, 4); break; \
    default: GPX_GC_LAUNCH(KIND, 0);  \
  }
  if (kp.kind == GPX_KERNEL_RBF) {
    GPX_GC_KIND(GPX_KERNEL_RBF)
  } else if (kp.kind == GPX_KERNEL_PERIODIC) {
    GPX_GC_KIND(GPX_KERNEL_PERIODIC)
  } else {
    GPX_GC_KIND(GPX_KERNEL_MATERN52)
  }
#undef GPX_GC_KIND
#undef GPX_GC_LAUNCH
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_grad_reduce(gpx_ctx* ctx, const double* dpart, int nblocks, int nvals, double* dout,
                       int batch, int64_t out_bs) {
  grad_reduce_kernel<<<batch > 1 ? batch : 1, 256, 0, ctx->s>>>(dpart, nblocks, nvals, dout,
                                                                 (int64_t)nblocks * GC_MAXV, out_bs);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// d lml / d v_i for a per-point diagonal v (K = k + diag(v)): 1/2 (alpha_i^2 - Kinv_ii)
__global__ __launch_bounds__(256) void grad_diag_kernel(const double* __restrict__ Kinv, int64_t ld, int N,
                                                        const double* __restrict__ alpha,
                                                        double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) out[i] = 0.5 * (alpha[i] * alpha[i] - Kinv[(int64_t)i * ld + i]);
}

int launch_grad_diag(gpx_ctx* ctx, const double* dKinv, int64_t ld, int N, const double* dalpha, double* dout) {
  grad_diag_kernel<<<(N + 255) / 256, 256, 0, ctx->s>>>(dKinv, ld, N, dalpha, dout);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// draws[s][a] += mean[a]
__global__ __launch_bounds__(256) void add_mean_kernel(double* __restrict__ D, int64_t ld, int n,
                                                       int M, const double* __restrict__ mean,
                                                       int64_t d_bs, int64_t mean_bs) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  D += (int64_t)blockIdx.z * d_bs;
  mean += (int64_t)blockIdx.z * mean_bs;
  if (a < M && s < n) D[(int64_t)s * ld + a] += mean[a];
}

int launch_add_mean(gpx_ctx* ctx, double* ddraws, int64_t ld, int n, int M, const double* dmean,
                    int batch, int64_t d_bs, int64_t mean_bs) {
  if (n <= 0) return 0;
  dim3 grid((M + 255) / 256, n, batch > 1 ? batch : 1);
  add_mean_kernel<<<grid, 256, 0, ctx->s>>>(ddraws, ld, n, M, dmean, d_bs, mean_bs);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

} // namespace gpx
