/*
 * gpx.h — C-ABI of libgpx: the MI355X (gfx950) exact-GP hot path behind gpax's
 * kernel / ExactGP / viGP / viSparseGP Python surface.
 *
 * The reference (ziatdinovmax/gpax, pure Python on JAX/NumPyro) has no FFI; its seam for this
 * path is the kernel-callable protocol and the model methods cited per entry point below
 * (paths relative to the reference checkout).  Every entry point names the reference
 * interface it replaces.  INTEGRATION.md shows the ctypes binding a gpax maintainer would add.
 *
 * Conventions
 *   - All pointers are caller-owned HOST buffers, C-order (row-major) fp64, unless the
 *     parameter name starts with `d_` (device pointer).  The library copies H<->D.
 *   - One gpx_ctx per GPU; calls on one ctx must be serialised by the caller; distinct
 *     ctxs are independent.  Functions block until results are in the host buffers.
 *   - Return value: 0 = ok; < 0 = bad argument / HIP failure (text via gpx_last_error).
 *     Numerical failure (non-positive Cholesky pivot) is NOT an error return: it is
 *     reported LAPACK-style through an `info` out-parameter (1-based order of the first
 *     bad pivot) and outputs are NaN-filled, mirroring JAX's silent-NaN Cholesky that
 *     gpax's `filter_nans` relies on (gpax/models/gp.py:396-398).  Never aborts.
 *   - fp64 end to end.  Parity tolerance vs the reference algorithm: 1e-6 relative
 *     (BASELINE.json north_star); tests assert 1e-8 or tighter.
 */
#ifndef GPX_H
#define GPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpx_ctx gpx_ctx;

/* kernel family selector — gpax/kernels/kernels.py:227-241 (get_kernel registry) */
#define GPX_KERNEL_RBF 0      /* gpax/kernels/kernels.py:44-65 */
#define GPX_KERNEL_MATERN52 1 /* gpax/kernels/kernels.py:68-91 */
#define GPX_KERNEL_PERIODIC 2 /* gpax/kernels/kernels.py:94-117; `ell` then carries d + 1 values:
                                 the d lengthscales followed by the period (and gradients likewise) */
#define GPX_KERNEL_R2 3       /* gpx_gram ONLY: the squared scaled distance itself, sum_k ((x_k - z_k) / ell_k)^2
                               * (square_scaled_distance, gpax/kernels/kernels.py:28-41); scale is ignored */

#define GPX_MAX_DIM 16 /* max input dimension d handled by the fused kernels */

/* profiling classes for gpx_profile_read */
#define GPX_PROF_GEMM_TRAILING 0 /* potrf outer trailing update (syrk, K = outer block) */
#define GPX_PROF_GEMM_OTHER 1    /* every other MFMA GEMM launch */
#define GPX_PROF_POTF2 2         /* diagonal-block factor+inverse */
#define GPX_PROF_GRAM 3          /* Gram builds */
#define GPX_PROF_NCLASS 4

/* ---- lifecycle ------------------------------------------------------------------------ */

/* Replaces JAX device selection (`device=` kwarg -> jax.device_put, gpax/models/gp.py:201-203).
 * Fails (<0) when no gfx950-capable HIP device `device` exists. */
int gpx_init(int device, gpx_ctx** out);
int gpx_device_count(void); /* HIP devices visible to this process (0 when there is none / no driver) */
/* PCI address of visible device `device` — the physical GPU behind an ordinal (launchers narrow the visible set per
 * process; bench.py counts distinct GPUs with it).  Any output pointer may be NULL. */
int gpx_device_pci(int device, int* domain, int* bus, int* dev);
void gpx_destroy(gpx_ctx* ctx);
const char* gpx_last_error(const gpx_ctx* ctx);
int gpx_device_info(gpx_ctx* ctx, char* name, int name_len, int* num_cu, int64_t* hbm_bytes,
                    int* clock_khz);
int gpx_synchronize(gpx_ctx* ctx);

/* ---- Gram matrix: gpax/kernels/kernels.py:28-91 ------------------------------------------
 * out[i*m + j] = scale * k(||(X_i - Z_j)/ell||)  (+ diag_add on i == j when add_diag != 0).
 * The reference adds (noise + jitter) * eye iff X.shape == Z.shape (kernels.py:63,89); the
 * host wrapper decides add_diag by that same rule and passes diag_add = noise + jitter.
 * ell has d entries (the wrapper broadcasts scalar / (1,) / (1,1) lengthscales). */
int gpx_gram(gpx_ctx* ctx, int kind, const double* X, int n, const double* Z, int m, int d,
             const double* ell, double scale, double diag_add, int add_diag, double* out);

/* ---- training state: ExactGP._set_data / fit, gpax/models/gp.py:166-220,410-414 -----------
 * Uploads X_train (N,d) and sizes the device workspaces (Gram/factor buffer etc.). */
int gpx_set_train(gpx_ctx* ctx, const double* X, int N, int d);

/* Per-point variances v (N) added to the diagonal of the training covariance on top of
 * (noise + jitter): K + diag(v) — the measured noise of MeasuredNoiseGP.model
 * (gpax/models/mngp.py:92-98).  Stays in force for every factorisation of this training set until
 * cleared with v = NULL (or the next gpx_set_train). */
int gpx_set_diag(gpx_ctx* ctx, const double* v, int n);

/* T independent training sets X (T,N,d) of vector-valued GPs (vExactGP._set_data / model,
 * gpax/models/vgp.py:62-96,199-208: `jax.vmap(self.kernel)` over the task axis).  Afterwards the
 * batched entry points (gpx_fit_batch, gpx_predict_sweep) treat entry b as task b % T — its own
 * X[t], X_new[t] and (yres_rows = T) y residuals; the single-theta entry points refuse T > 1. */
int gpx_set_train_tasks(gpx_ctx* ctx, const double* X, int T, int N, int d);

/* ---- log marginal likelihood: ExactGP.model, gpax/models/gp.py:137-164 --------------------
 * (NumPyro MultivariateNormal(loc, covariance_matrix=k).log_prob(y): Cholesky, triangular
 * solve, sum log diag.)  Builds K = kernel(X,X,theta,noise,jitter) on device, factors it
 * (blocked right-looking fp64 MFMA Cholesky) and returns
 *   lml = -1/2 yres^T K^-1 yres - sum log L_ii - N/2 log(2 pi),   yres = y - mean_fn(X).
 * The factor stays resident for gpx_lml_grad / gpx_posterior.  *info > 0 => lml = NaN. */
int gpx_factor(gpx_ctx* ctx, int kind, const double* ell, double scale, double noise,
               double jitter, const double* yres, double* lml, int* info);

/* Analytic gradient of the lml above w.r.t. (ell[0..d), scale, noise) — replaces JAX
 * reverse-mode autodiff through gp.py:137-164 used by NUTS (gp.py:207-218) and SVI
 * (vigp.py:108-120).  Also returns alpha = K^-1 yres (d lml / d yres = -alpha) for the
 * mean-function chain rule.  Must follow gpx_factor; CONSUMES the factor (K^-1 overwrites it). */
int gpx_lml_grad(gpx_ctx* ctx, double* grad_ell, double* grad_scale, double* grad_noise,
                 double* alpha);

/* d lml / d v for the per-point diagonal of gpx_set_diag (K = k + (noise + jitter) I + diag(v)):
 * grad_diag[i] = 1/2 (alpha_i^2 - (K^-1)_ii).  Call after gpx_lml_grad (K^-1 and alpha resident; when that gradient came
 * from the one-launch fit step of N <= 128 — no gpx_set_diag in force — which never stores K^-1, this call re-runs the
 * general sequence at the same theta first).
 * VarNoiseGP.model (gpax/models/hskgp.py:124-153) differentiates through v = exp(log_var). */
int gpx_lml_grad_diag(gpx_ctx* ctx, double* grad_diag);

/* ---- batched fit step: B hyper-parameter vectors through gpx_factor + gpx_lml_grad at once ----
 * The chains of MCMC(num_chains > 1, chain_method='parallel'|'vectorized') (gpax/models/gp.py:173-174,
 * 207-218) ask for one log-likelihood gradient each per leapfrog; this entry evaluates them as ONE
 * launch sequence with the chain as a grid dimension.  For entry b:
 *   theta_b = (ells[b*ne .. ), scales[b], noises[b]), ne = d (+1: period);
 *   yres (yres_rows, N): yres_rows = 1 (shared), B (one per entry) or T (per task, row b % T);
 *   after gpx_set_train_tasks entry b is task b % T: X[b % T] (vExactGP.model, vgp.py:62-96)
 *   lml[b], info[b] as gpx_factor; grad[b*(ne+2) ..] = [d/d ell.., d/d scale, d/d noise];
 *   alpha[b*N ..] = K_b^-1 yres_b.  grad == NULL: values only.  Entries with info != 0 are NaN.
 * The per-entry arithmetic is that of gpx_factor / gpx_lml_grad (bit-identical results). */
int gpx_fit_batch(gpx_ctx* ctx, int kind, int B, const double* ells, const double* scales,
                  const double* noises, double jitter, const double* yres, int yres_rows,
                  double* lml, int* info, double* grad, double* alpha);

/* ---- one NUTS transition on the host side of the library (csrc/nuts.hip) -----------------------------------------
 * numpyro.infer.NUTS as ExactGP.fit drives it (gpax/models/gp.py:207-218), for the default model: every site of the
 * unconstrained vector u = log(theta) LogNormal-distributed (gp.py:222-247), no mean function.  Leapfrogs, recursive
 * doubling, multinomial / biased progressive sampling and the generalised U-turn test around gpx_fit_batch(B = 1) — the
 * loop of gpax_amd/infer/nuts.py statement for statement, without a Python frame per leapfrog (what dominates at the N = 25
 * of examples/gpax_simpleGP.ipynb).  Adaptation stays with the caller.
 *   dim = ne + 2, ne = d (+ 1: period);  idx_ell[c]: position in u of the device's lengthscale entry c, idx_scale /
 *   idx_noise likewise (together a permutation of 0 .. dim - 1);
 *   prior_loc / prior_scale / prior_const (dim each): LogNormal(loc, scale) per element, const = log(scale) + log(2 pi) / 2;
 *   yres (N): residuals (host);  u, U, g: in — the current point, its potential and gradient; out — the new ones;
 *   p0 (dim): the momentum (drawn by the caller);  eps, inv_mass (dim), max_tree_depth: as nuts_transition;
 *   pcg_state4: {state hi, state lo, inc hi, inc lo} of a numpy.random.PCG64 — the uniforms are drawn from it in the
 *   order the Python loop draws them; state hi / lo are advanced on return;
 *   accept (mean acceptance probability), n_leapfrog, diverging: the transition's diagnostics.
 * Returns 0, or the error of a device call inside a leapfrog (gpx_last_error). */
int gpx_nuts_transition(gpx_ctx* ctx, int kind, int dim, int ne, const int* idx_ell, int idx_scale, int idx_noise,
                        const double* prior_loc, const double* prior_scale, const double* prior_const, double jitter,
                        const double* yres, double* u, double* U, double* g, const double* p0, double eps,
                        const double* inv_mass, int max_tree_depth, uint64_t* pcg_state4, double* accept, int* n_leapfrog,
                        int* diverging);
/* The potential gpx_nuts_transition integrates, alone: U(u) = -(log p(y | theta) + log prior(theta) + log |d theta / d u|)
 * at theta = exp(u) and its gradient (arguments as above) — tests drive the Python loop of gpax_amd/infer/nuts.py with it
 * to show that the two loops are the same algorithm (identical chains when the scalar arithmetic is shared). */
int gpx_nuts_potential(gpx_ctx* ctx, int kind, int dim, int ne, const int* idx_ell, int idx_scale, int idx_noise,
                       const double* prior_loc, const double* prior_scale, const double* prior_const, double jitter,
                       const double* yres, const double* u, double* U, double* g);
/* Host only (no GPU, no context): n doubles of numpy.random.PCG64 from the state {state hi, state lo, inc hi, inc lo}
 * (advanced on return) — Generator.uniform()'s stream, as gpx_nuts_transition draws it (tests/test_samplers.py). */
int gpx_debug_pcg64_doubles(uint64_t* state4, int n, double* out);

/* ---- posterior: ExactGP.get_mvn_posterior, gpax/models/gp.py:253-277 ----------------------
 * Must follow gpx_factor (same theta).  mean (M), cov (M*M, may be NULL), var (M, may be
 * NULL; = diag(cov), what viGP.predict returns, gpax/models/vigp.py:184-185).
 *   k_pp = kernel(Xnew,Xnew,theta,noise_p,jitter); k_pX = kernel(Xnew,X,theta) (no diag)
 *   mean = k_pX K^-1 yres;  cov = k_pp - k_pX K^-1 k_pX^T    (POTRF+TRSM route; the
 * reference's explicit inverse, gp.py:271, is not replicated).
 * The M x M posterior covariance stays resident for gpx_mvn_draw. */
int gpx_posterior(gpx_ctx* ctx, const double* Xnew, int M, double noise_p, double jitter,
                  double* mean, double* cov, double* var);

/* ---- MVN draw: ExactGP._predict, gpax/models/gp.py:279-293 --------------------------------
 * out[s*M + a] = mean[a] + sum_b Lc[a][b] eps[s*M + b], Lc = chol(cov) of the last
 * gpx_posterior.  eps are caller-supplied standard normals (JAX threefry streams are not
 * reproducible here; parity is defined given eps).  *info > 0 => out is NaN-filled. */
int gpx_mvn_draw(gpx_ctx* ctx, const double* eps, int n, double* out, int* info);

/* ---- predictive sweep: ExactGP.predict, gpax/models/gp.py:351-399 -------------------------
 * The jax.vmap over S posterior samples, run as a device-resident loop over batches of B
 * samples: the batch is a grid dimension of every launch (B is picked from N and free HBM,
 * B = 1 at N = 16384, 60 at 4096, 256 below N ~ 1900; GPX_SWEEP_BATCH forces it), so at most B*N*N is
 * materialised and small-N sweeps are not launch-bound.  Results do not depend on B.
 * For sample s:
 *   theta_s = (ells[s*d .. s*d+d), scales[s], noises[s]);
 *   yres (yres_rows, N): yres_rows = 1 (no sampled mean-function parameters), S (one row per sample)
 *   or T (per task, row s % T, after gpx_set_train_tasks — then Xnew is (T, M, d) and sample s is
 *   task s % T: samples ordered [s][t], vExactGP.predict, gp.py:351-399 through vgp.py:151-176);
 *   mean_s, cov_s as gpx_posterior with noise_p = noiseless ? 0 : noises[s];
 *   samples[s] = mean_s + chol(cov_s) eps[s]   (n draws).
 * means (S*M), samples (S*n*M), infos (S; bit 0.. = train-factor info, negative = draw chol
 * failed).  Rows whose info != 0 are NaN-filled.  eps may be NULL when n == 0.
 * vars (S*M) or NULL: diag(cov_s), for callers that draw from the marginals only
 * (MeasuredNoiseGP._predict, gpax/models/mngp.py:170-181).
 * m_slice > 0: ExactGP.predict_in_batches (gpax/models/gp.py:325-349) — the test points are treated in blocks
 * of m_slice: cov / chol / draws per block (draws of different blocks are independent, as when predict is
 * called per slice), but K(theta_s) is factored ONCE per sample, all blocks' k_pX rows riding along; eps and
 * samples keep the (S, n, M) layout.  0: one block of M points.
 * pred_diag (S*M) or NULL: per-sample variances added to the diagonal of cov_s before the draw
 * (the predicted noise variance of VarNoiseGP.get_mvn_posterior, gpax/models/hskgp.py:188-204). */
int gpx_predict_sweep(gpx_ctx* ctx, int kind, int S, const double* ells, const double* scales,
                      const double* noises, const double* yres, int yres_rows,
                      const double* Xnew, int M, int noiseless, double jitter,
                      const double* eps, int n, double* means, double* samples, int* infos, double* vars,
                      const double* pred_diag, int m_slice);

/* ---- variational sparse GP: viSparseGP, gpax/models/sparse_gp.py ----------------------------
 * gpx_sgp_bound: the per-SVI-step objective of viSparseGP.model (sparse_gp.py:62-114): VFE bound =
 * LowRankMultivariateNormal(loc, cov_factor=W, cov_diag=noise).log_prob(y) - trace_term/2 with
 * W = (Luu^-1 Kuf)^T, for inducing points Xu (Mi x d) on the X uploaded by gpx_set_train, and (when
 * want_grad) its analytic gradient w.r.t. (ell[d], scale, noise) and Xu (Mi*d) — replaces JAX
 * autodiff through sparse_gp.py:92-114.  dyres (N, may be NULL) = d bound / d yres for the
 * mean-function chain rule.  *info: >0 chol(Kuu) failed, <0 chol(capacitance) failed => NaN. */
int gpx_sgp_bound(gpx_ctx* ctx, int kind, const double* ell, double scale, double noise, double jitter,
                  const double* Xu, int Mi, const double* yres, int want_grad, double* bound,
                  double* grad_ell, double* grad_scale, double* grad_noise, double* grad_Xu, double* dyres,
                  int* info);

/* viSparseGP.get_mvn_posterior (sparse_gp.py:173-223): Woodbury posterior at Xnew (Ms x d).
 * mean (Ms), cov (Ms*Ms or NULL), var (Ms or NULL).  noise_p = noise * (1 - noiseless).
 * Both sparse entry points keep the forward pass of the previous call (Kuu, Kuf, both factorisations) while the next
 * call brings bit-identical kind / ell / scale / noise / jitter / Xu / yres on the same gpx_set_train upload — the
 * reference's predict_in_batches (sparse_gp.py:225-262) calls the posterior once per slice of X_new with nothing else
 * changed.  The values are those of a call that recomputes everything. */
int gpx_sgp_posterior(gpx_ctx* ctx, int kind, const double* ell, double scale, double noise, double jitter,
                      const double* Xu, int Mi, const double* yres, const double* Xnew, int Ms, double noise_p,
                      double* mean, double* cov, double* var, int* info);

/* ---- node-level predictive sweep: the vmap axis of ExactGP.predict (gpax/models/gp.py:392-395) sharded over the
 * GPUs of ONE node from ONE process.  A gpx_node owns `inflight` contexts on each of `ngpu`
 * devices (devices == NULL: ordinals 0 .. ngpu-1) and one RCCL communicator per device (ncclCommInitAll; RCCL is
 * bound at run time with dlopen, libgpx does not link it).  gpx_predict_sweep_multi = gpx_predict_sweep (T = 1,
 * no pred_diag) with the S samples dealt to the GPUs as they go:
 *   root GPU <- one H2D of [X | X_new | y_res | eps];  ncclBroadcast of that payload over xGMI;
 *   every context of every GPU takes chunks (whole launch batches) from one shared cursor, one host thread each;
 *   ncclSend / ncclRecv (one group) of every GPU's means | draws | vars | pivots to the root;  one D2H.
 * Results equal gpx_predict_sweep's sample by sample (same kernels, same per-sample arithmetic).
 * Environment: GPX_NODE_TRANSPORT=memcpy replaces the two RCCL steps by hipMemcpyPeerAsync (testing on a box
 * with one GPU listed twice, which RCCL refuses); never selected implicitly. */
typedef struct gpx_node gpx_node;
int gpx_node_init(int ngpu, const int* devices, int inflight, gpx_node** out);
void gpx_node_destroy(gpx_node* node);
const char* gpx_node_last_error(const gpx_node* node);
/* transport_rccl: 1 = RCCL, 0 = memcpy test transport; rccl_version as ncclGetVersion reports it (0 without RCCL) */
int gpx_node_info(const gpx_node* node, int* ngpu, int* inflight, int* transport_rccl, int* rccl_version);
/* Samples every GPU of the node worked off in the last gpx_predict_sweep_multi (counts[0 .. min(cap, ngpu))); returns ngpu
 * (0 before the first sweep).  The samples are dealt to the GPUs as they go (one cursor shared by all contexts of all
 * GPUs), so a slower GPU ends up with fewer.  Test hook: GPX_NODE_SLOW="<gpu index>:<microseconds>" at gpx_node_init makes
 * the contexts of that GPU sleep after every chunk. */
int gpx_node_last_shares(const gpx_node* node, int* counts, int cap);
int gpx_predict_sweep_multi(gpx_node* node, int kind, const double* X, int N, int d, int S, const double* ells,
                            const double* scales, const double* noises, const double* yres, int yres_rows,
                            const double* Xnew, int M, int noiseless, double jitter, const double* eps, int n,
                            double* means, double* samples, int* infos, double* vars, int m_slice);

/* ---- the same sharded sweep with ONE PROCESS PER GPU (the launch model of torch.distributed.run / mpirun: RANK,
 * LOCAL_RANK, WORLD_SIZE), without PyTorch.  Every process owns one GPU (`inflight` contexts on it) and one rank of an
 * RCCL communicator (ncclCommInitRank; RCCL bound with dlopen).  Rank 0 creates the 128-byte unique id
 * (gpx_rank_unique_id) and the launcher's rendezvous hands it to the other ranks (gpax_amd/launch.py).
 * gpx_rank_predict_sweep is COLLECTIVE: every rank calls it with the same scalar arguments; the arrays are read
 * (X, ells, scales, noises, yres, Xnew, eps) and written (means, samples, infos, vars) on rank 0 only, other ranks
 * may pass NULL.  Data path: rank 0 H2D of [X | X_new | y_res | eps | theta table] -> ncclBroadcast -> every rank
 * sweeps its contiguous block of the S samples -> ncclSend / ncclRecv of the result blocks to rank 0 -> D2H.
 * Results equal gpx_predict_sweep's sample by sample.  Replaces the vmap of gpax/models/gp.py:392-395 when the
 * caller is launched one process per GPU (chain_method / device placement of the reference: gp.py:173-174,201-203).
 * file_dir != NULL / "": the "file" transport (a directory all ranks share) instead of RCCL — several ranks on one
 * GPU in the tests (RCCL refuses duplicate devices), and bench.py's fallback when RCCL cannot initialise; unique_id
 * is ignored then.
 * gpx_rank_barrier: waits for this rank's own contexts, then for every rank.  gpx_rank_allreduce_max: v[0..n) <-
 * elementwise max over ranks (n <= 64; max-over-ranks timing).  gpx_rank_bcast: count doubles from rank 0's host
 * buffer to every rank's.  All three are collective. */
#define GPX_UNIQUE_ID_BYTES 128
typedef struct gpx_rank gpx_rank;
int gpx_rank_unique_id(char* id /* GPX_UNIQUE_ID_BYTES */, char* errbuf, int errlen);
int gpx_rank_init(int device, int rank, int nranks, const char* unique_id, const char* file_dir, int inflight,
                  gpx_rank** out);
void gpx_rank_destroy(gpx_rank* rk);
const char* gpx_rank_last_error(const gpx_rank* rk);
int gpx_rank_info(const gpx_rank* rk, int* rank, int* nranks, int* inflight, int* transport_rccl, int* rccl_version);
/* PCI address of the GPU this rank drives: what tells ranks that SHARE a device (a launcher whose LOCAL_RANK exceeds the
 * visible devices, a test box) from ranks on distinct GPUs — bench.py reports the number of distinct devices as n_gpus. */
int gpx_rank_device_pci(const gpx_rank* rk, int* domain, int* bus, int* dev);
/* RCCL collective / point-to-point calls this rank has issued so far (all-reduce, broadcast, send, receive).  With
 * GPX_RANK_FORCE_COLLECTIVES=1 a communicator of ONE rank issues them too (diagnostic: the 1-GPU rehearsal of the path). */
int64_t gpx_rank_collective_calls(const gpx_rank* rk);
int gpx_rank_barrier(gpx_rank* rk);
/* Collective.  Every rank times a probe on its own GPU (the trailing-update form of the fp64 MFMA GEMM on resident scratch
 * operands) and the ranks exchange the rates (one all-reduce); from then on gpx_rank_predict_sweep sizes the ranks'
 * contiguous blocks in proportion to them (gpx_shard_ranges_weighted; rates relative to their mean, clamped to
 * [0.85, 1.15]).  speeds (nranks doubles, or NULL) receives them.  Without this call the blocks are equal (gpx_shard_range).
 * Test hook: GPX_RANK_SPEED_SCALE=<factor> multiplies this rank's measured rate. */
int gpx_rank_calibrate(gpx_rank* rk, double* speeds);
int gpx_rank_allreduce_max(gpx_rank* rk, double* v, int n);
int gpx_rank_bcast(gpx_rank* rk, double* buf, int64_t count);
int gpx_rank_predict_sweep(gpx_rank* rk, int kind, const double* X, int N, int d, int S, const double* ells,
                           const double* scales, const double* noises, const double* yres, int yres_rows,
                           const double* Xnew, int M, int noiseless, double jitter, const double* eps, int n,
                           double* means, double* samples, int* infos, double* vars, int want_vars, int m_slice);
/* Host only (no device call): the contiguous block [lo, hi) of part `part` out of `parts` over S samples that both
 * sharded sweeps use (sizes differ by at most one). */
int gpx_shard_range(int S, int part, int parts, int* lo, int* hi);
/* Host only: all `parts` blocks at once, sizes in proportion to weights[part] > 0 (largest-remainder rounding, ties to the
 * lower part); weights == NULL or not all positive: the blocks of gpx_shard_range.  lo / hi: `parts` ints each. */
int gpx_shard_ranges_weighted(int S, const double* weights, int parts, int* lo, int* hi);

/* Sweep statistics since gpx_init: batches launched, samples processed, and the batch size B
 * chosen by the most recent sweep. */
int gpx_sweep_stats(gpx_ctx* ctx, int64_t* batches, int64_t* samples, int* last_batch);

/* ---- measurement hooks (bench.py / profiles) ----------------------------------------------
 * When enabled, HIP events bracket every launch of the profiled kernel classes on the
 * library's own stream; gpx_profile_read returns launches, summed duration and summed
 * ALGORITHMIC flops (MFMA classes) or bytes (Gram) since the last reset. */
int gpx_profile_enable(gpx_ctx* ctx, int on);
int gpx_profile_reset(gpx_ctx* ctx);
int gpx_profile_read(gpx_ctx* ctx, int cls, int64_t* launches, double* total_ms,
                     double* total_work);
/* Algorithmic BYTES of the MFMA classes since the last reset: 16 B (one read, one write) per C entry a launch
 * updates (8 B when beta == 0) — the per-launch figure bench.py's roofline block quotes beside the PMC traffic. */
int gpx_profile_read_bytes(gpx_ctx* ctx, int cls, double* total_bytes);
/* Diagnostic: select the diagonal-block kernel of the blocked Cholesky for the following calls on this context —
 * "slim" (default, csrc/potf2_slim.h) or "tile" (round 2, csrc/potf2_tile.h: the reference of the bit-identity
 * tests).  Both produce the same bits; bench.py uses it for the in-process A/B of the potf2 class.
 * The environment variable GPX_POTF2 sets the same thing at gpx_init. */
int gpx_debug_set_potf2(gpx_ctx* ctx, const char* mode);
/* Diagnostic: the latency-shape GEMM (64 x 64 tiles and 32 x 128 strips of the panel chains) for the following calls on
 * this context — "r5" (csrc/gemm_tile.h lat_tile: LDS-direct staging, ring of k-slices), "r1" (round 1: register
 * staging, the reference of the bit-identity tests) or "auto" (default: r5 wherever a chain has the chip to itself, r1
 * inside the blocked two-stream sweeps; csrc/common.h).  Same bits from all.  GPX_LAT_GEMM at gpx_init. */
int gpx_debug_set_lat_gemm(gpx_ctx* ctx, const char* mode);
/* Measurement mode (bench.py's roofline block): on != 0 — every trailing update of the blocked Cholesky (the SYRK under
 * gpax/models/gp.py:160-164, profile class GPX_PROF_GEMM_TRAILING) of the following calls on this context runs ALONE on the
 * chip: it starts when the panel stream of the look-ahead has drained, and that stream resumes when it is done.  The HIP
 * events gpx_profile_* puts around each launch then time the kernel on its real operands and shapes without the panel
 * chain sharing the SIMDs ("serialised" launch time).  Results are the same bits; only the overlap is given up. */
int gpx_debug_set_serialise_trailing(gpx_ctx* ctx, int on);
/* Diagnostic: device time of ONE GEMM launch of the library on resident scratch operands (constant, non-zero), average
 * over `reps` back-to-back launches between two HIP events: C (tiles_m x tiles_n 128-tiles) from A (tiles_m*128 x K) and
 * B (tiles_n*128 x K).  mode 0: C = A B^T; 1: C -= A B^T (the Cholesky / TRSM update form: accumulators start from -C);
 * 2: the in-place panel TRSM A <- A B^T (tiles_n = 1, K = 128).  lower: lower tiles only.  shape 0: the shape
 * launch_gemm_nt picks for that tile count; 1: latency shapes; 2: the 128 x 128 throughput shape.  What tools/
 * lat_gemm_bench.py and bench.py quote for the chain GEMMs (the kernels under gpax/models/gp.py:160-164's Cholesky). */
int gpx_debug_gemm_time(gpx_ctx* ctx, int tiles_m, int tiles_n, int K, int mode, int lower, int shape, int reps,
                        double* ms_per_launch);

/* Diagnostic, host only (no GPU, no context): the tile list a plain GEMM launch of the library runs over since round 5 —
 * entry i of grid.x -> (by, bx) written to by_bx[2 i], by_bx[2 i + 1], at most `cap` entries; returns the length of the
 * list (< 0: bad arguments).  lower != 0: the live tiles of a lower-triangular launch, row by row (bx <= by + delta,
 * delta = first tile row - first tile column of the launch, in tiles of the launch's shape).  lower == 0: the full
 * tiles_m x tiles_n grid, order 0 row-major, 1 column-major from the LAST column (k range ends at the column tile), 2
 * column-major from the first column (k range starts there).  What tests/test_tile_list.py checks against a brute-force
 * enumeration; the trailing update of the Cholesky under gpax/models/gp.py:160-164 is such a launch. */
int gpx_debug_tile_list(int lower, int delta, int tiles_m, int tiles_n, int order, int cap, int* by_bx);

/* Device-only timed repetitions (inputs resident in HBM; used by bench.py so that `value`
 * excludes PCIe).  Each call runs `reps` passes of the named stage at the theta/Xnew last set
 * by gpx_factor/gpx_posterior and returns the elapsed milliseconds between HIP events. */
#define GPX_STAGE_GRAM 0      /* Gram build of K (N x N, lower tiles) */
#define GPX_STAGE_POTRF 1     /* Gram + blocked Cholesky (+ lml)                    */
#define GPX_STAGE_FITSTEP 2   /* Gram + potrf + lml + potri + gradient contraction  */
#define GPX_STAGE_POSTERIOR 3 /* Gram + potrf + k_pX + trsm + syrk + mean           */
#define GPX_STAGE_PREDICT 4   /* POSTERIOR + chol(cov) + n draws                    */
int gpx_time_stage(gpx_ctx* ctx, int stage, int reps, double* elapsed_ms);

/* The predictive sweep with everything resident in HBM (bench.py's timed region): runs the
 * per-sample pipeline of gpx_predict_sweep for S thetas against the X / yres / Xnew already on
 * the device (set by gpx_set_train, gpx_factor and gpx_posterior), n_draws MVN draws per sample
 * from device-resident eps, outputs left on the device.  No PCIe traffic; returns the elapsed
 * milliseconds between HIP events on the library's stream. */
int gpx_sweep_resident(gpx_ctx* ctx, int kind, int S, const double* ells, const double* scales,
                       const double* noises, int noiseless, double jitter, int n_draws,
                       double* elapsed_ms);

/* Raw MFMA fp64 ceiling microbenchmark (v_mfma_f64_16x16x4_f64 issue rate).  tflops must hold 3
 * doubles: [0] sustained TFLOP/s, [1] shader cycles per MFMA per wave, [2] effective shader MHz. */
int gpx_mfma_f64_peak(gpx_ctx* ctx, double* tflops);

/* Plain GEMM entry used by the unit tests of the MFMA tile kernel:
 * C[M x N] = alpha * A[M x K] * B[N x K]^T + beta * C   (row-major, any sizes). */
int gpx_gemm_nt(gpx_ctx* ctx, int M, int N, int K, double alpha, const double* A,
                const double* B, double beta, double* C);

/* Plain Cholesky entry used by the unit tests: in: A (n x n, symmetric, lower read);
 * out: L (n x n lower, zeros above).  *info as above. */
int gpx_potrf(gpx_ctx* ctx, int n, const double* A, double* L, int* info);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
