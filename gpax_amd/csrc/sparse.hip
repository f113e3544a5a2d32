// sparse.hip — variational sparse GP (VFE / Titsias) on the same HIP primitives.
//
// Replaces gpax/models/sparse_gp.py:62-114 (viSparseGP.model: LowRankMultivariateNormal
// log-density + trace-term factor, with learnable inducing points Xu) and
// sparse_gp.py:173-223 (get_mvn_posterior), plus the reverse-mode gradient JAX provides for the
// SVI loop (sparse_gp.py:151-166).
//
// Forward (Cholesky route, as the reference):
//   Kuu = k(Xu,Xu) + jitter I = Luu Luu^T          W = Kfu Luu^-T   (N x M, right-looking TRSM)
//   A = I + W^T W / s2 = LA LA^T  (M x M)           c = LA^-1 W^T y / s2
//   F = -N/2 log 2pi - N/2 log s2 - sum log diag LA - y^T y / (2 s2) + c^T c / 2
//       - max(0, N kd - |W|_F^2) / (2 s2)
//   (matrix-determinant lemma / Woodbury form of LowRankMVN(W, s2 I).log_prob(y) - trace/2).
// Posterior: V1 = Ksu Luu^-T, V2 = V1 LA^-T, mean = V2 c, cov = Kss - V1 V1^T + V2 V2^T.
// Gradient: matrix adjoints G_uu = dF/dKuu, G_uf = dF/dKuf from the explicit inverses
//   Ki = Kuu^-1 = Tu Tu^T (Tu = Luu^-T),  S1 = (Kuu + Kuf Kfu / s2)^-1 = Tu A^-1 Tu^T,
//   m = S1 Kuf y:   G_uu = -S1/2 - m m^T/(2 s2^2) + Ki - Tu A Tu^T / 2,
//                   G_uf^T = Kfu (Ki - S1)/s2 - (Kfu m) m^T / s2^3 + y m^T / s2^2,
// contracted with dK/dtheta and dK/dXu evaluated on the fly (nothing N x M x d is stored).
#include "common.h"

namespace gpx {

// ---- small kernels ---------------------------------------------------------------------------

// out (cols x rows, ldo) = in^T (rows x cols, ldi); 32x32 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int64_t ldi, int rows,
                                                        int cols, double* __restrict__ out, int64_t ldo) {
  __shared__ double t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = by + r, j = bx + tx;
    t[r][tx] = (i < rows && j < cols) ? in[(int64_t)i * ldi + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = bx + r, i = by + tx;
    if (j < cols && i < rows) out[(int64_t)j * ldo + i] = t[tx][r];
  }
}

static int launch_transpose(gpx_ctx* ctx, const double* in, int64_t ldi, int rows, int cols, double* out,
                            int64_t ldo) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  transpose_kernel<<<grid, 256, 0, ctx->s>>>(in, ldi, rows, cols, out, ldo);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// Full symmetric n x n:  out = diag_val * I + scale * sum_z P_z[max(i,j)][min(i,j)]   (i, j < n_valid),
// identity elsewhere in the n_pad extent (a multiple of 32).  One workgroup per 32 x 32 block of the LOWER triangle: the
// slabs are read along their rows, the sum goes out as block (bi, bj) and — turned through LDS — as block (bj, bi), both
// along rows.  (Rounds 2 / 3 read P[max][min] per output element: the upper half of the output walked DOWN the
// columns of every slab — 211 us at M = 2048 with 11 slabs, 1.7 TB/s.)
__global__ __launch_bounds__(256) void sym_finalize_kernel(const double* __restrict__ P, int splits,
                                                           int64_t split_stride, int64_t ldp, double scale,
                                                           double diag_val, int n_valid, int n_pad,
                                                           double* __restrict__ out, int64_t ldo) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = bj * 32 + tx;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r;
    double v;
    if (i < n_valid && j < n_valid) {
      // (in a diagonal block the entries above the diagonal are taken from their mirror image below)
      const double* p = (bi == bj && tx > r) ? P + (int64_t)j * ldp + i : P + (int64_t)i * ldp + j;
      double acc = 0.0;
      for (int z = 0; z < splits; ++z) acc += p[(int64_t)z * split_stride];
      v = scale * acc + (i == j ? diag_val : 0.0);
    } else {
      v = (i == j) ? 1.0 : 0.0;
    }
    out[(int64_t)i * ldo + j] = v;
    t[r][tx] = v;
  }
  if (bi == bj) return;
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) out[(int64_t)(bj * 32 + r) * ldo + bi * 32 + tx] = t[tx][r];
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ double bsum(double v, double* red) {
  v = wsum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) s += red[w];
  return s;
}

// transpose_kernel that also leaves the sum of squares of its 32 x 32 block of `in` (entries with i < vr, j < vc) in
// part[blockIdx.y * gridDim.x + blockIdx.x]: |W|_F^2 of the sparse bound without another pass over the N x M matrix.
__global__ __launch_bounds__(256) void transpose_sumsq_kernel(const double* __restrict__ in, int64_t ldi, int rows,
                                                              int cols, double* __restrict__ out, int64_t ldo, int vr,
                                                              int vc, double* __restrict__ part) {
  __shared__ double t[32][33];
  __shared__ double red[16];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double sq = 0.0;
  for (int r = ty; r < 32; r += 8) {
    const int i = by + r, j = bx + tx;
    const double x = (i < rows && j < cols) ? in[(int64_t)i * ldi + j] : 0.0;
    t[r][tx] = x;
    if (i < vr && j < vc) sq = fma(x, x, sq);
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = bx + r, i = by + tx;
    if (j < cols && i < rows) out[(int64_t)j * ldo + i] = t[tx][r];
  }
  const double tot = bsum(sq, red);
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// part[block] = sum of squares of rows [block*R, ...) of a rows x cols matrix (fixed order).
__global__ __launch_bounds__(256) void sumsq_rows_kernel(const double* __restrict__ Amat, int64_t ld, int rows,
                                                         int cols, double* __restrict__ part) {
  __shared__ double red[16];
  double s = 0.0;
  const int r0 = blockIdx.x * 8;
  for (int r = r0; r < r0 + 8 && r < rows; ++r) {
    const double* a = Amat + (int64_t)r * ld;
    for (int k = threadIdx.x; k < cols; k += 256) s = fma(a[k], a[k], s);
  }
  const double t = bsum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// Single-block scalar reductions used by the bound:
//  out[0] = sum part[0..np)   (|W|_F^2)      out[1] = sum_{i<M} log LA[i][i]
//  out[2] = sum_{i<M} c_i^2                   out[3] = sum_n y_n^2
//  out[4] = sum_n y_n t_n (t may be null)     out[5] = sum_n t_n^2
__global__ __launch_bounds__(1024) void sgp_scalars_kernel(const double* __restrict__ part, int np,
                                                           const double* __restrict__ LA, int64_t lda, int M,
                                                           const double* __restrict__ c,
                                                           const double* __restrict__ y, int N,
                                                           const double* __restrict__ t,
                                                           double* __restrict__ out) {
  __shared__ double red[16];
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
  for (int k = threadIdx.x; k < np; k += 1024) a0 += part[k];
  for (int k = threadIdx.x; k < M; k += 1024) {
    a1 += log(LA[(int64_t)k * lda + k]);
    a2 = fma(c[k], c[k], a2);
  }
  for (int k = threadIdx.x; k < N; k += 1024) {
    a3 = fma(y[k], y[k], a3);
    if (t) {
      a4 = fma(y[k], t[k], a4);
      a5 = fma(t[k], t[k], a5);
    }
  }
  double r;
  r = bsum(a0, red); if (threadIdx.x == 0) out[0] = r;
  r = bsum(a1, red); if (threadIdx.x == 0) out[1] = r;
  r = bsum(a2, red); if (threadIdx.x == 0) out[2] = r;
  r = bsum(a3, red); if (threadIdx.x == 0) out[3] = r;
  r = bsum(a4, red); if (threadIdx.x == 0) out[4] = r;
  r = bsum(a5, red); if (threadIdx.x == 0) out[5] = r;
}

// v[i] = a * x[i] + b * y[i]   (y may be null)
__global__ __launch_bounds__(256) void axpby_kernel(double* v, double a, const double* x,
                                                    double b, const double* y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = a * x[i] + (y ? b * y[i] : 0.0);
}

static int launch_axpby(gpx_ctx* ctx, double* v, double a, const double* x, double b, const double* y, int n) {
  axpby_kernel<<<(n + 255) / 256, 256, 0, ctx->s>>>(v, a, x, b, y, n);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// C = a * A + b * B (+ diagonal handled by caller), full n x n
__global__ __launch_bounds__(256) void mat_axpby_kernel(double* __restrict__ C, int64_t ldc, double a,
                                                        const double* __restrict__ A, int64_t lda, double b,
                                                        const double* __restrict__ B, int64_t ldb, int n) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j < n) C[(int64_t)i * ldc + j] = a * A[(int64_t)i * lda + j] + b * B[(int64_t)i * ldb + j];
}

// mirror the lower triangle into the upper (n x n)
__global__ __launch_bounds__(256) void symmetrize_kernel(double* __restrict__ A, int64_t ld, int n) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j < n && j > i) A[(int64_t)i * ld + j] = A[(int64_t)j * ld + i];
}

__device__ __forceinline__ void kval_dk(int kind, double r2, double scale, double& kv, double& dk) {
  if (kind == GPX_KERNEL_RBF) {
    kv = scale * exp(-0.5 * r2);
    dk = -0.5 * kv;
  } else {
    const double r = sqrt(r2 + MATERN_EPS);
    const double e = exp(-SQRT5 * r);
    kv = scale * (1.0 + SQRT5 * r + (5.0 / 3.0) * r2) * e;
    dk = -(5.0 / 6.0) * scale * e * (1.0 + SQRT5 * r2 / r);
  }
}

constexpr int SC_NV = 2 * GPX_MAX_DIM + 4;

// Contraction of an adjoint  g[i][j] = G[i][j] + rcoef[i] * mvec[j]  (rows i index points P
// (rows x d): Xu itself for G_uu, X_train for G_uf^T; columns j index the inducing points) with the
// kernel derivatives evaluated on the fly.
// Per block (64 rows x 64 cols) partials: [0..d) d/d ell, [d] d/d scale; the Xu-gradient is
// accumulated per column in gxu_part[blockRow][col][m] (reduced over block rows afterwards in
// fixed order => deterministic).
// KIND / D > 0: kernel and input dimension as compile-time constants (the arrays below live in registers and the kernel
// branch folds: C5's 16 316 x 2039 contraction 446 -> see profiles/r04/c5_sparse.json); D = 0: any d <= GPX_MAX_DIM
// at run time.  Same operations in the same order either way.
template <int KIND, int D>
__global__ __launch_bounds__(256) void sgp_contract_kernel(KernelParams kp, const double* __restrict__ Pts,
                                                           int rows, const double* __restrict__ Xu, int M,
                                                           const double* __restrict__ G, int64_t ldg,
                                                           const double* __restrict__ rcoef,
                                                           const double* __restrict__ mvec,
                                                           double* __restrict__ part,
                                                           double* __restrict__ gxu_part) {
  constexpr int DM = D > 0 ? D : GPX_MAX_DIM;
  __shared__ double red[16];
  __shared__ double colacc[4][64][DM];
  const int d = D > 0 ? D : kp.d;
  const int kind = KIND >= 0 ? KIND : kp.kind;
  const int bj = blockIdx.x, bi = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = bj * 64 + lane; // column = inducing point index
  double acc[DM], acc_scale = 0.0; // (the scale partial in a scalar of its own: acc[d] with a run-time d put the array in scratch)
  double gx[DM];
#pragma unroll
  for (int c = 0; c < DM; ++c) acc[c] = 0.0;
#pragma unroll
  for (int c = 0; c < DM; ++c) gx[c] = 0.0;
  if (j < M) {
    double xu[DM];
#pragma unroll
    for (int c = 0; c < DM; ++c) xu[c] = c < d ? Xu[(int64_t)j * d + c] : 0.0;
    const double mj = mvec[j];
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
      const int i = bi * 64 + wave * 16 + t;
      if (i >= rows) break;
      const double g = fma(rcoef[i], mj, G[(int64_t)i * ldg + j]);
      double r2 = 0.0, diff[DM];
#pragma unroll
      for (int c = 0; c < DM; ++c) {
        if (c >= d) break;
        diff[c] = (xu[c] - Pts[(int64_t)i * d + c]) * kp.inv_ell[c]; // (xu - p)/ell
        r2 = fma(diff[c], diff[c], r2);
      }
      double kv, dk;
      kval_dk(kind, r2, kp.scale, kv, dk);
#pragma unroll
      for (int c = 0; c < DM; ++c) {
        if (c >= d) break;
        acc[c] += g * dk * (-2.0 * diff[c] * diff[c] * kp.inv_ell[c]);
        gx[c] += g * dk * (2.0 * diff[c] * kp.inv_ell[c]); // d k / d xu_c
      }
      acc_scale += g * kv / kp.scale;
    }
  }
#pragma unroll
  for (int c = 0; c < DM; ++c)
    if (c < d) colacc[wave][lane][c] = gx[c];
  __syncthreads();
  if (wave == 0 && j < M) {
#pragma unroll
    for (int c = 0; c < DM; ++c) {
      if (c >= d) break;
      const double s = colacc[0][lane][c] + colacc[1][lane][c] + colacc[2][lane][c] + colacc[3][lane][c];
      gxu_part[((int64_t)bi * M + j) * GPX_MAX_DIM + c] = s;
    }
  }
  const int64_t bid = (int64_t)bi * gridDim.x + bj;
#pragma unroll
  for (int c = 0; c < DM + 2; ++c) {
    if (c >= d + 2) break;
    const double s = bsum(c < d ? acc[c < DM ? c : 0] : (c == d ? acc_scale : 0.0), red);
    if (threadIdx.x == 0) part[bid * SC_NV + c] = s;
  }
}

typedef void (*sgp_contract_fn)(KernelParams, const double*, int, const double*, int, const double*, int64_t, const double*,
                                const double*, double*, double*);
static sgp_contract_fn pick_contract(const KernelParams& kp) {
  const bool rbf = kp.kind == GPX_KERNEL_RBF;
  switch (kp.d) {
    case 1: return rbf ? sgp_contract_kernel<GPX_KERNEL_RBF, 1> : sgp_contract_kernel<GPX_KERNEL_MATERN52, 1>;
    case 2: return rbf ? sgp_contract_kernel<GPX_KERNEL_RBF, 2> : sgp_contract_kernel<GPX_KERNEL_MATERN52, 2>;
    case 3: return rbf ? sgp_contract_kernel<GPX_KERNEL_RBF, 3> : sgp_contract_kernel<GPX_KERNEL_MATERN52, 3>;
    default: return sgp_contract_kernel<-1, 0>;
  }
}

// out[c] = sum_b part[b][c]; gxu[j][c] (+)= factor * sum_bi gxu_part[bi][j][c]
__global__ __launch_bounds__(256) void sgp_reduce_kernel(const double* __restrict__ part, int nblocks, int nv,
                                                         double* __restrict__ out,
                                                         const double* __restrict__ gxu_part, int nbi, int M,
                                                         int d, double factor, int accumulate,
                                                         double* __restrict__ gxu) {
  __shared__ double red[16];
  if (blockIdx.x == 0) {
    for (int c = 0; c < nv; ++c) {
      double s = 0.0;
      for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(int64_t)b * SC_NV + c];
      const double t = bsum(s, red);
      if (threadIdx.x == 0) out[c] = t;
    }
  }
  // one WAVE per output (j, c): its lanes take the block rows b = lane, lane + 64, ... and a fixed-order butterfly adds
  // them up (deterministic; a single thread walking the nbi = N / 64 partials, 130 KB apart, is a chain of dependent
  // loads: C5 97 us for 4078 outputs)
  const int lane = threadIdx.x & 63;
  for (int idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < M * d; idx += gridDim.x * 4) {
    const int j = idx / d, c = idx - j * d;
    double s = 0.0;
    for (int b = lane; b < nbi; b += 64) s += gxu_part[((int64_t)b * M + j) * GPX_MAX_DIM + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
      s *= factor;
      gxu[idx] = accumulate ? gxu[idx] + s : s;
    }
  }
}

// ---- helpers -----------------------------------------------------------------------------------

static inline GemmArgs gargs(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                             int K, double alpha, double beta) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.K = K; g.alpha = alpha; g.beta = beta;
  return g;
}

static int ens(gpx_ctx* ctx, DevBuf& b, size_t bytes) {
  hipError_t e = b.ensure(bytes);
  if (e != hipSuccess) return fail(ctx, "hipMalloc", e, __FILE__, __LINE__);
  return 0;
}

// C (lower, then mirrored to full) = alpha * V V^T over K columns via split-K slabs; diag_val on the diagonal
static int syrk_full(gpx_ctx* ctx, const double* V, int64_t ldv, int nt, int K, double scale, double diag_val,
                     int n_valid, double* out, int64_t ldo) {
  const int np = nt * TILE;
  const int lower_tiles = nt * (nt + 1) / 2;
  // Split count from a cost model instead of "just fill the 512 workgroup slots" (round 2: ceil(512 / tiles) slabs —
  // at C5, 136 tiles x 4 slabs = 544 workgroups, i.e. two rounds with the second one nearly empty: 2.08 ms for
  // 6.8e10 flop).  rounds x (k-tiles per slab) in units of one workgroup k-step, plus the finalisation pass that reads
  // every slab (np^2 x 8 B per slab at ~3 TB/s against 0.218 us per k of a 128 x 128 tile on one of 512 slots).
  // The choice depends on the shape only, so results stay run-to-run identical.
  const int max_splits = K / 256 > 0 ? (K / 256 < 32 ? K / 256 : 32) : 1;
  int splits = 1;
  double best = 1e300;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const int kc = round_up((K + sp - 1) / sp, TILE);
    const int eff = (K + kc - 1) / kc; // slabs actually launched
    const double rounds = std::ceil((double)lower_tiles * eff / 512.0);
    const double cost = rounds * kc * 0.218 + eff * ((double)np * np * 8.0 / 3.0e6);
    if (cost < best - 1e-9) {
      best = cost;
      splits = eff;
    }
  }
  const int kchunk = round_up((K + splits - 1) / splits, TILE);
  splits = (K + kchunk - 1) / kchunk;
  const int64_t ldp = pick_ld(np), stride = (int64_t)np * ldp;
  GPX_TRY(ens(ctx, ctx->SplitK, (size_t)splits * stride * sizeof(double)));
  GemmArgs g = gargs(V, ldv, V, ldv, ctx->SplitK.d(), ldp, K, 1.0, 0.0);
  g.lower = 1;
  g.kchunk = kchunk;
  g.c_split_stride = stride;
  // persistent, dynamically scheduled: in a plain launch workgroup id -> XCD is a fixed rotation, i.e. XCD x gets the tile
  // columns x and x + 8 — with 16 tile columns of a lower triangle that is 24 tiles per slab on one XCD and 10 on another
  if (ctx->persist_scope_ok) ctx->persist_scope += 1;
  const int rc_syrk = launch_gemm_nt(ctx, g, nt, nt, splits, GPX_PROF_GEMM_OTHER, (double)np * (np + 1.0) * K);
  if (ctx->persist_scope_ok) ctx->persist_scope -= 1;
  GPX_TRY(rc_syrk);
  dim3 grid(np / 32, np / 32);
  sym_finalize_kernel<<<grid, 256, 0, ctx->s>>>(ctx->SplitK.d(), splits, stride, ldp, scale, diag_val, n_valid, np,
                                                out, ldo);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// out (full, symmetric, np x np) = the product described by g, whose k ranges are cut by the triangle of its operands
// (ktri: k starts at the row tile) and of which only the lower triangle is computed.  As ONE launch the 64 x 64 tiles
// are all resident at once and the launch lasts as long as its longest tile — k = 0 .. np on one workgroup, ~170 us at
// np = 2048 whether the product has M^3 / 3 or 2 M^3 flop.  Cut into `splits` k slabs (one grid.z each, empty ranges store
// zeros) no workgroup has more than np / splits of k, and the slab sum + mirror is the pass that `symmetrize` was.
static int tri_product_sym(gpx_ctx* ctx, GemmArgs g, int nt, int splits, double* out, int64_t ldo, double work) {
  const int np = nt * TILE;
  const int64_t ldp = pick_ld(np), stride = (int64_t)np * ldp;
  const int kchunk = round_up((g.K + splits - 1) / splits, TILE);
  splits = (g.K + kchunk - 1) / kchunk;
  GPX_TRY(ens(ctx, ctx->SplitK, (size_t)splits * stride * sizeof(double)));
  g.C = ctx->SplitK.d();
  g.ldc = ldp;
  g.lower = 1;
  g.kchunk = kchunk;
  g.c_split_stride = stride;
  g.latency_shape = 1;
  GPX_TRY(launch_gemm_nt(ctx, g, nt, nt, splits, GPX_PROF_GEMM_OTHER, work));
  dim3 grid(np / 32, np / 32);
  sym_finalize_kernel<<<grid, 256, 0, ctx->s>>>(ctx->SplitK.d(), splits, stride, ldp, 1.0, 0.0, np, np, out, ldo);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// T = L^-T (upper, n x n; zero below the diagonal — it enters full GEMMs) from the factor L and its diagonal-block inverses,
// by the block-recursive tree of linalg.hip (scr: scratch of T's shape); V != nullptr: V = T^T = L^-1 (lower) as well.
// Round 3: every triangular solve of the sparse path with many right-hand sides (W = Kfu Luu^-T, V1 = Ksu Luu^-T,
// V2 = V1 LA^-T) is ONE GEMM against L^-1 with the k range cut at the column tile (kupper) instead of a right-looking
// sweep of 2 dependent launches per 128 columns.
static int build_linv_t(gpx_ctx* ctx, const double* L, int64_t ldl, const double* Linv, int nt, double* T,
                        int64_t ldt, double* scr, double* V = nullptr) {
  GPX_HIP(ctx, hipMemsetAsync(T, 0, (size_t)nt * TILE * ldt * sizeof(double), ctx->s));
  GPX_TRY(linv_t_tree(ctx, T, ldt, L, ldl, Linv, nt, scr, ldt));
  if (V != nullptr) GPX_TRY(launch_transpose(ctx, T, ldt, nt * TILE, nt * TILE, V, ldt));
  return 0;
}

// X = B V^T with V = L^-1 lower triangular (rows x nt*128):  X[i][j] = sum_{k <= j} B[i][k] V[j][k]
static int solve_by_inverse(gpx_ctx* ctx, const double* B, int64_t ldb, int rows_t, const double* V, int64_t ldv, int nt,
                            double* X, int64_t ldx) {
  GemmArgs g = gargs(B, ldb, V, ldv, X, ldx, nt * TILE, 1.0, 0.0);
  g.kupper = 1;
  if (ctx->persist_scope_ok) ctx->persist_scope += 1; // long tiles first (TileMap.col_desc)
  const int rc = launch_gemm_nt(ctx, g, rows_t, nt, 0, GPX_PROF_GEMM_OTHER,
                                (double)rows_t * TILE * (double)nt * TILE * (nt + 1.0) * TILE);
  if (ctx->persist_scope_ok) ctx->persist_scope -= 1;
  return rc;
}

} // namespace gpx

using namespace gpx;

// C = a * A + (b0 + b1 u) * B + (d0 + d1 u) * I   (n x n, full; B may be null), u = [thresh - flag[0] > 0]: the
// bound's clipped-trace switch (jnp.clip(trace_term, a_min=0)) decided ON THE DEVICE from the reduced |W|_F^2 in flag[0]
// — the gradient's launches follow the forward pass without a host round trip in between.
__global__ __launch_bounds__(256) void mat_combine_kernel(double* __restrict__ C, int64_t ldc, double a,
                                                          const double* __restrict__ A, int64_t lda, double b0, double b1,
                                                          const double* __restrict__ B, int64_t ldb, double d0, double d1,
                                                          const double* __restrict__ flag, double thresh, int n) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j < n) {
    const double u = (thresh - flag[0] > 0.0) ? 1.0 : 0.0;
    double v = a * A[(int64_t)i * lda + j];
    if (B) v += (b0 + b1 * u) * B[(int64_t)i * ldb + j];
    if (i == j) v += d0 + d1 * u;
    C[(int64_t)i * ldc + j] = v;
  }
}

struct SgpState {
  int M = 0, Mp = 0, Ntp = 0;
  int64_t ldu = 0, ldw = 0, ldt = 0;
  gpx::DevBuf Xu, Kuu, LinvU, Wn, Wt, A, Acopy, LinvA, u, c, cpad, scal, part;
  gpx::DevBuf B0, B1, B2, B3, B4, vvec, tvec, mvec, rcoef_u, rcoef_f, gxu_part, cpart, gXu, T1a, T1;
  gpx::DevBuf Xs, V1, V2, mean, var, var2, Cov;
  gpx::DevBuf Vu, VA, Tscr, Kfu; // Luu^-1, LA^-1 (lower), scratch of the L^-T tree, Kfu before the solve
  gpx::DevBuf TscrU;             // scratch of the Luu^-T tree when it runs on the side stream beside the LA chain / tree
  std::vector<hipEvent_t> evG;   // ride-along forward pass: per column group [chain done, inverse block done] + joins
  gpx::KernelParams kp{};
  double noise = 0, jitter = 0;
  // The forward pass of the last call, kept while the next call brings the very same inputs (kernel, theta, jitter, Xu,
  // yres, X): predict_in_batches calls gpx_sgp_posterior once per slice of X_new with everything else unchanged
  // (sparse_gp.py:173-223 recomputes Kuu, Kuf and both factorisations for every slice).
  bool fwd_valid = false, reuse = false;
  hipEvent_t evFork = nullptr, evJoin = nullptr; // the Kfu Gram build on the panel stream beside the Cholesky of Kuu
  int npartW = 0; // partial sums of |W|_F^2 left in `part` by the forward pass
  uint64_t h_train_gen = 0;
  int h_kind = -1;
  double h_par[GPX_MAX_DIM + 4] = {0};
  double kfu_diag = 0; // added to the diagonal of Kfu (same-shape rule of the reference's kernels; sgp_setup)
  std::vector<double> h_Xu, h_y;
};

static SgpState* sgp_state(gpx_ctx* ctx) {
  if (!ctx->sgp) ctx->sgp = new SgpState();
  return static_cast<SgpState*>(ctx->sgp);
}

namespace gpx {
void sgp_release(gpx_ctx* ctx) {
  if (!ctx->sgp) return;
  SgpState* s = static_cast<SgpState*>(ctx->sgp);
  DevBuf* bufs[] = {&s->Xu, &s->Kuu, &s->LinvU, &s->Wn, &s->Wt, &s->A, &s->Acopy, &s->LinvA, &s->u, &s->c, &s->cpad,
                    &s->scal, &s->part, &s->B0, &s->B1, &s->B2, &s->B3, &s->B4, &s->vvec, &s->tvec, &s->mvec,
                    &s->rcoef_u, &s->rcoef_f, &s->gxu_part, &s->cpart, &s->gXu, &s->T1a, &s->T1, &s->Xs, &s->V1,
                    &s->V2, &s->mean, &s->var, &s->var2, &s->Cov, &s->Vu, &s->VA, &s->Tscr, &s->Kfu, &s->TscrU};
  for (DevBuf* b : bufs) b->release();
  for (hipEvent_t e : s->evG) (void)hipEventDestroy(e);
  if (s->evFork) (void)hipEventDestroy(s->evFork);
  if (s->evJoin) (void)hipEventDestroy(s->evJoin);
  delete s;
  ctx->sgp = nullptr;
}
} // namespace gpx

static double kd_value(const KernelParams& kp) {
  if (kp.kind == GPX_KERNEL_RBF) return kp.scale;
  const double r = std::sqrt(MATERN_EPS);
  return kp.scale * (1.0 + SQRT5 * r) * std::exp(-SQRT5 * r);
}

// kfu_diag: what the reference's call of the kernel for Kuf adds to its diagonal.  The kernels add (noise + jitter) I iff
// the two inputs have the SAME SHAPE (kernels.py:63-65), and viSparseGP.model calls `kernel(Xu, X, params)` with the default
// jitter 1e-6 (sparse_gp.py:96): with as many inducing points as training points Kuf carries 1e-6 on its diagonal in the
// bound — and nothing in get_mvn_posterior, which passes jitter=0 (sparse_gp.py:196).  Part of the reuse key.
static int sgp_setup(gpx_ctx* ctx, SgpState* s, int kind, const double* ell, double scale, double noise,
                     double jitter, const double* Xu, int Mi, const double* yres, double kfu_diag) {
  if (ctx->N < 1) return bad_arg(ctx, "gpx_set_train must be called first");
  if (kind != GPX_KERNEL_RBF && kind != GPX_KERNEL_MATERN52) return bad_arg(ctx, "kernel kind");
  if (Mi < 1 || !Xu || !ell || !yres) return bad_arg(ctx, "sparse GP arguments");
  const int d = ctx->d;
  {
    double par[GPX_MAX_DIM + 4] = {0};
    for (int c = 0; c < d; ++c) par[c] = ell[c];
    par[GPX_MAX_DIM] = scale; par[GPX_MAX_DIM + 1] = noise; par[GPX_MAX_DIM + 2] = jitter; par[GPX_MAX_DIM + 3] = kfu_diag;
    const size_t xb = (size_t)Mi * d * 8, yb = (size_t)ctx->N * 8;
    s->reuse = s->fwd_valid && s->h_train_gen == ctx->train_gen && s->h_kind == kind && s->M == Mi &&
               std::memcmp(par, s->h_par, sizeof(par)) == 0 && s->h_Xu.size() * 8 == xb && s->h_y.size() * 8 == yb &&
               std::memcmp(s->h_Xu.data(), Xu, xb) == 0 && std::memcmp(s->h_y.data(), yres, yb) == 0;
    if (s->reuse) { // everything the forward pass produced is still resident; yres again (the exact-GP entry points share it)
      GPX_HIP(ctx, hipMemcpyAsync(ctx->yres.d(), yres, yb, hipMemcpyHostToDevice, ctx->stream));
      ctx->factored = false;
      ctx->small_grad_ready = false;
      ctx->have_post = false;
      return 0;
    }
    s->fwd_valid = false;
    s->h_train_gen = ctx->train_gen;
    s->h_kind = kind;
    std::memcpy(s->h_par, par, sizeof(par));
    s->h_Xu.assign(Xu, Xu + (size_t)Mi * d);
    s->h_y.assign(yres, yres + ctx->N);
  }
  s->M = Mi;
  s->Mp = round_up(Mi, TILE);
  s->Ntp = round_up(ctx->N, TILE);
  s->ldu = pick_ld(s->Mp);
  s->ldw = pick_ld(s->Mp);
  s->ldt = pick_ld(s->Ntp);
  s->kp.kind = kind;
  s->kp.d = d;
  for (int c = 0; c < GPX_MAX_DIM; ++c) s->kp.inv_ell[c] = c < d ? 1.0 / ell[c] : 0.0;
  s->kp.scale = scale;
  s->noise = noise;
  s->jitter = jitter;
  s->kfu_diag = kfu_diag;
  GPX_TRY(ens(ctx, s->Xu, (size_t)Mi * d * 8));
  GPX_HIP(ctx, hipMemcpyAsync(s->Xu.d(), Xu, (size_t)Mi * d * 8, hipMemcpyHostToDevice, ctx->stream));
  GPX_HIP(ctx, hipMemcpyAsync(ctx->yres.d(), yres, (size_t)ctx->N * 8, hipMemcpyHostToDevice, ctx->stream));
  ctx->factored = false;
  ctx->small_grad_ready = false;
  ctx->have_post = false;
  return 0;
}

// Forward pass shared by the bound and the posterior (yres in ctx->yres on the device).
static int sgp_forward_run(gpx_ctx* ctx, SgpState* s);
static int sgp_forward(gpx_ctx* ctx, SgpState* s) {
  if (s->reuse) return 0;
  GPX_TRY(sgp_forward_run(ctx, s));
  s->fwd_valid = true;
  return 0;
}
// ---- round 5: Kuu = Luu Luu^T and W = Kfu Luu^-T in ONE pipelined pass -------------------------------------------------
// Rounds 3 / 4 factored Kuu (a 16-step chain at M = 2048: 0.92 ms with 255 CUs idle), built Luu^-T by the tree (0.27 ms)
// and only then formed W as one N x M x M GEMM against Luu^-1 (1.12 ms) — three stages in sequence
// (profiles/r04/c5_step_timeline.md).  W is the right-looking solve X Luu^T = Kfu (what the reference does:
// solve_triangular, sparse_gp.py:94), and column group g of that solve needs only the FIRST columns of Luu:
//   chain  (panel stream)  plain right-looking steps of Kuu (potrf_steps), an event after every column group [a, b)
//   tree   (side stream)   the inverse of the group's diagonal block Luu[a:b, a:b] — a 2- or 4-tile L^-T tree, transposed
//   main                   W[:, a:b] = R[:, a:b] inv(Luu[a:b, a:b])^T  (ONE GEMM, k range cut at the column tile), then the
//                          far update R[:, b:] -= W[:, a:b] Luu[b:, a:b]^T (K = 256 / 512, the throughput shape)
// R starts as Kfu and is updated in place; every tile of R receives its groups in ascending order on ONE stream (bit-
// reproducible).  Groups of 2, 2, 4, 4, ... tile columns: the tall solve starts after two chain steps and then always has
// a group's worth of work queued while the chain (57 us per step) runs ahead of it.  The FULL Luu^-T tree (the gradient's
// Tu, the posterior's Luu^-1) is no longer on the path to W: it runs on the side stream beside the chain of the second
// factorisation (sgp_forward_run).
static int sgp_events(gpx_ctx* ctx, SgpState* s, size_t n) {
  while (s->evG.size() < n) {
    hipEvent_t e;
    GPX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->evG.push_back(e);
  }
  return 0;
}

// events of the ride-along pass: evG[0 .. 4] = start / A ready / LA chain done / the two joins, then two per column group
enum { SGP_EV_START = 0, SGP_EV_A = 1, SGP_EV_CHAIN = 2, SGP_EV_JOIN1 = 3, SGP_EV_JOIN2 = 4, SGP_EV_GROUPS = 5 };

static std::vector<int> sgp_groups(int mt) { // boundaries of the column groups: 2, 2, 4, 4, ... tile columns
  std::vector<int> gb(1, 0);
  for (int a = 0, k = 0; a < mt; ++k) {
    a += (k < 2) ? 2 : 4;
    gb.push_back(a < mt ? a : mt);
  }
  return gb;
}

static int sgp_factor_ride(gpx_ctx* ctx, SgpState* s, int* dinfo) {
  const int Mp = s->Mp, Ntp = s->Ntp, mt = Mp / TILE, ntl = Ntp / TILE;
  hipStream_t smain = ctx->stream, schain = ctx->pstream, sside = ctx->xstream;
  const std::vector<int> gb = sgp_groups(mt);
  const int ng = (int)gb.size() - 1;
  GPX_TRY(sgp_events(ctx, s, (size_t)SGP_EV_GROUPS + 2 * ng));
  hipEvent_t evStart = s->evG[SGP_EV_START];
  // chain and side stream start behind what the main stream holds (Xu, Kuu, its identity padding)
  GPX_HIP(ctx, hipEventRecord(evStart, smain));
  GPX_HIP(ctx, hipStreamWaitEvent(schain, evStart, 0));
  GPX_HIP(ctx, hipStreamWaitEvent(sside, evStart, 0));
  double* Tu = s->B0.d();
  GPX_HIP(ctx, hipMemsetAsync(Tu, 0, (size_t)Mp * s->ldu * sizeof(double), sside)); // zero below the diagonal: Tu enters full GEMMs
  // R = Kfu on the main stream, beside the first chain steps
  ctx->s = smain;
  GPX_TRY(launch_gram_padded(ctx, s->kp, ctx->X.d(), ctx->N, Ntp, s->Xu.d(), s->M, Mp, s->kfu_diag, s->kfu_diag != 0.0, 0, s->Kfu.d(), s->ldw));
  int rc = 0;
  for (int g = 0; g < ng && rc >= 0; ++g) {
    const int a = gb[(size_t)g], b = gb[(size_t)g + 1], w = b - a;
    hipEvent_t evC = s->evG[(size_t)SGP_EV_GROUPS + 2 * g], evT = s->evG[(size_t)SGP_EV_GROUPS + 2 * g + 1];
    const int64_t dg = (int64_t)a * TILE * (s->ldu + 1); // the group's diagonal block in an M x M matrix
    ctx->s = schain;
    rc = potrf_steps(ctx, s->Kuu.d(), s->ldu, mt, a, b, s->LinvU.d(), dinfo);
    if (rc < 0) break;
    GPX_HIP(ctx, hipEventRecord(evC, schain));
    ctx->s = sside;
    GPX_HIP(ctx, hipStreamWaitEvent(sside, evC, 0));
    rc = linv_t_tree(ctx, Tu + dg, s->ldu, s->Kuu.d() + dg, s->ldu, s->LinvU.d() + (int64_t)a * TILE * TILE, w,
                     s->TscrU.d() + dg, s->ldu);
    if (rc < 0) break;
    rc = launch_transpose(ctx, Tu + dg, s->ldu, w * TILE, w * TILE, s->Vu.d() + dg, s->ldu);
    if (rc < 0) break;
    GPX_HIP(ctx, hipEventRecord(evT, sside));
    ctx->s = smain;
    GPX_HIP(ctx, hipStreamWaitEvent(smain, evT, 0)); // (evT follows evC: the group's columns of Luu are final as well)
    {
      GemmArgs q = gargs(s->Kfu.d() + (int64_t)a * TILE, s->ldw, s->Vu.d() + dg, s->ldu, s->Wn.d() + (int64_t)a * TILE,
                         s->ldw, w * TILE, 1.0, 0.0);
      q.kupper = 1;
      q.big_shape = 1;
      rc = launch_gemm_nt(ctx, q, ntl, w, 0, GPX_PROF_GEMM_OTHER, (double)Ntp * TILE * TILE * w * (w + 1.0));
      if (rc < 0) break;
    }
    if (b < mt) {
      GemmArgs q = gargs(s->Wn.d() + (int64_t)a * TILE, s->ldw, s->Kuu.d() + (int64_t)b * TILE * s->ldu + (int64_t)a * TILE,
                         s->ldu, s->Kfu.d() + (int64_t)b * TILE, s->ldw, w * TILE, -1.0, 1.0);
      q.big_shape = 1;
      rc = launch_gemm_nt(ctx, q, ntl, mt - b, 0, GPX_PROF_GEMM_OTHER, 2.0 * Ntp * (double)(mt - b) * TILE * w * TILE);
    }
  }
  ctx->s = smain;
  return rc;
}

// Whatever path leaves the forward pass — a launch error included — the main stream waits for what the panel and side
// streams still hold: the next call rewrites Xu / Kuu / the tree buffers on the main stream.
struct SgpJoin {
  gpx_ctx* ctx;
  hipEvent_t e1, e2;
  ~SgpJoin() {
    if (hipEventRecord(e1, ctx->pstream) == hipSuccess) (void)hipStreamWaitEvent(ctx->stream, e1, 0);
    if (hipEventRecord(e2, ctx->xstream) == hipSuccess) (void)hipStreamWaitEvent(ctx->stream, e2, 0);
    ctx->s = ctx->stream;
  }
};

static int sgp_forward_ride(gpx_ctx* ctx, SgpState* s) {
  const int N = ctx->N, M = s->M, Mp = s->Mp, Ntp = s->Ntp;
  const int mt = Mp / TILE;
  const double s2 = s->noise;
  const size_t mm = (size_t)Mp * s->ldu * 8;
  int* dinfo = s->scal.i() + 1024;
  GPX_TRY(ens(ctx, s->Kfu, (size_t)Ntp * s->ldw * 8));
  GPX_TRY(ens(ctx, s->B0, mm));
  GPX_TRY(ens(ctx, s->B1, mm));
  GPX_TRY(ens(ctx, s->Vu, mm));
  GPX_TRY(ens(ctx, s->VA, mm));
  GPX_TRY(ens(ctx, s->Tscr, mm));
  GPX_TRY(ens(ctx, s->TscrU, mm));
  GPX_TRY(sgp_events(ctx, s, (size_t)SGP_EV_GROUPS + 2 * (sgp_groups(mt).size() - 1)));
  SgpJoin join{ctx, s->evG[SGP_EV_JOIN1], s->evG[SGP_EV_JOIN2]};
  hipEvent_t evA = s->evG[SGP_EV_A], evChain = s->evG[SGP_EV_CHAIN];
  // Kuu = kernel(Xu, Xu, params, **jitter): noise defaults to 0 (sparse_gp.py:92)
  GPX_TRY(launch_gram_padded(ctx, s->kp, s->Xu.d(), M, Mp, s->Xu.d(), M, Mp, s->jitter, 1, 1, s->Kuu.d(), s->ldu));
  GPX_TRY(launch_pad_identity(ctx, s->Kuu.d(), s->ldu, M, Mp));
  GPX_TRY(sgp_factor_ride(ctx, s, dinfo));
  { // Wt = W^T, and |W|_F^2 in partial sums on the way (the bound's trace term)
    dim3 grid((Mp + 31) / 32, (Ntp + 31) / 32);
    transpose_sumsq_kernel<<<grid, 256, 0, ctx->s>>>(s->Wn.d(), s->ldw, Ntp, Mp, s->Wt.d(), s->ldt, N, M, s->part.d());
    GPX_HIP(ctx, hipGetLastError());
    s->npartW = (int)(grid.x * grid.y);
  }
  // A = I + Wt Wt^T / s2 (kept in Acopy)
  GPX_TRY(syrk_full(ctx, s->Wt.d(), s->ldt, mt, Ntp, 1.0 / s2, 1.0, M, s->A.d(), s->ldu));
  GPX_HIP(ctx, hipMemcpyAsync(s->Acopy.d(), s->A.d(), mm, hipMemcpyDeviceToDevice, ctx->stream));
  GPX_HIP(ctx, hipEventRecord(evA, ctx->stream));
  // ... factored on the PANEL stream, while the side stream — idle since the last group's inverse block — builds the full
  // Luu^-T (the gradient's Tu) and Luu^-1 (the posterior's) beside that chain, and the main stream forms u = Wt y
  GPX_HIP(ctx, hipStreamWaitEvent(ctx->pstream, evA, 0));
  ctx->s = ctx->pstream;
  GPX_TRY(potrf_steps(ctx, s->A.d(), s->ldu, mt, 0, mt, s->LinvA.d(), dinfo + 1));
  GPX_HIP(ctx, hipEventRecord(evChain, ctx->pstream));
  GPX_HIP(ctx, hipStreamWaitEvent(ctx->xstream, evA, 0)); // (the tall solve has read the last group's block of Luu^-1 by then)
  ctx->s = ctx->xstream;
  GPX_TRY(linv_t_tree(ctx, s->B0.d(), s->ldu, s->Kuu.d(), s->ldu, s->LinvU.d(), mt, s->TscrU.d(), s->ldu));
  GPX_TRY(launch_transpose(ctx, s->B0.d(), s->ldu, Mp, Mp, s->Vu.d(), s->ldu));
  ctx->s = ctx->stream;
  // u = Wt y ; then, behind the chain: TA = LA^-T, LA^-1 and c = LA^-1 u / s2 as one matrix-vector product
  GPX_TRY(launch_rowdot(ctx, s->Wt.d(), s->ldt, M, N, ctx->yres.d(), 0.0, s->u.d(), nullptr, 0));
  GPX_HIP(ctx, hipMemsetAsync(s->c.d(), 0, (size_t)Mp * 8, ctx->stream));
  GPX_TRY(launch_axpby(ctx, s->cpad.d(), 1.0 / s2, s->u.d(), 0.0, nullptr, M)); // cpad[0 .. M) = u / s2
  GPX_HIP(ctx, hipStreamWaitEvent(ctx->stream, evChain, 0));
  GPX_TRY(build_linv_t(ctx, s->A.d(), s->ldu, s->LinvA.d(), mt, s->B1.d(), s->ldu, s->Tscr.d(), s->VA.d()));
  GPX_TRY(launch_rowdot(ctx, s->VA.d(), s->ldu, M, M, s->cpad.d(), 0.0, s->c.d(), nullptr, 0));
  return 0;
}

static int sgp_forward_run(gpx_ctx* ctx, SgpState* s) {
  const int N = ctx->N, M = s->M, Mp = s->Mp, Ntp = s->Ntp;
  const int mt = Mp / TILE, ntl = Ntp / TILE;
  const double s2 = s->noise;
  const size_t mm = (size_t)Mp * s->ldu * 8;
  GPX_TRY(ens(ctx, s->Kuu, mm));
  GPX_TRY(ens(ctx, s->LinvU, (size_t)mt * TILE * TILE * 8));
  GPX_TRY(ens(ctx, s->Wn, (size_t)Ntp * s->ldw * 8));
  GPX_TRY(ens(ctx, s->Wt, (size_t)Mp * s->ldt * 8));
  GPX_TRY(ens(ctx, s->A, mm));
  GPX_TRY(ens(ctx, s->Acopy, mm));
  GPX_TRY(ens(ctx, s->LinvA, (size_t)mt * TILE * TILE * 8));
  GPX_TRY(ens(ctx, s->u, (size_t)Mp * 8));
  GPX_TRY(ens(ctx, s->c, (size_t)Mp * 8));
  GPX_TRY(ens(ctx, s->cpad, (size_t)TILE * s->ldu * 8));
  GPX_TRY(ens(ctx, s->scal, 8192));
  GPX_TRY(ens(ctx, s->part, (size_t)((Ntp / 32) * (Mp / 32) + Mp / 8 + 32) * 8));
  int* dinfo = s->scal.i() + 1024;
  GPX_HIP(ctx, hipMemsetAsync(dinfo, 0, 2 * sizeof(int), ctx->stream));
  if (ctx->sgp_inverse == 2) return sgp_forward_ride(ctx, s);
  GPX_TRY(sgp_events(ctx, s, SGP_EV_GROUPS));
  SgpJoin join{ctx, s->evG[SGP_EV_JOIN1], s->evG[SGP_EV_JOIN2]}; // every way out: the main stream waits for the others
  if (ctx->sgp_inverse) {
    // (rounds 4 / 5, GPX_SGP_SOLVE=inverse: the default) Kfu = k(X, Xu) waits for nothing of the Kuu branch: on the SIDE
    // stream, beside the latency-bound Cholesky chain of Kuu (a 16-step chain at M = 2048 that leaves the chip idle),
    // joined again before the solve that reads it.  (Round 4 queued it on the panel stream, which a blocked factorisation
    // of Kuu — more than ONE_BLOCK_TILES tile rows, or GPX_OUTER_TILES set — would also have used for its chain: the
    // build then delayed the chain instead of hiding beside it.  ADVICE r4.)
    if (!s->evFork) {
      GPX_HIP(ctx, hipEventCreateWithFlags(&s->evFork, hipEventDisableTiming));
      GPX_HIP(ctx, hipEventCreateWithFlags(&s->evJoin, hipEventDisableTiming));
    }
    GPX_TRY(ens(ctx, s->Kfu, (size_t)Ntp * s->ldw * 8));
    GPX_HIP(ctx, hipEventRecord(s->evFork, ctx->stream)); // Xu, X, theta are in place
    GPX_HIP(ctx, hipStreamWaitEvent(ctx->xstream, s->evFork, 0));
    ctx->s = ctx->xstream;
    const int rc_kfu = launch_gram_padded(ctx, s->kp, ctx->X.d(), N, Ntp, s->Xu.d(), M, Mp, s->kfu_diag, s->kfu_diag != 0.0, 0, s->Kfu.d(), s->ldw);
    ctx->s = ctx->stream;
    GPX_TRY(rc_kfu);
    GPX_HIP(ctx, hipEventRecord(s->evJoin, ctx->xstream));
  }
  // Kuu = kernel(Xu, Xu, params, **jitter): noise defaults to 0 (sparse_gp.py:92)
  GPX_TRY(launch_gram_padded(ctx, s->kp, s->Xu.d(), M, Mp, s->Xu.d(), M, Mp, s->jitter, 1, 1, s->Kuu.d(), s->ldu));
  GPX_TRY(launch_pad_identity(ctx, s->Kuu.d(), s->ldu, M, Mp));
  GPX_TRY(potrf_lower(ctx, s->Kuu.d(), s->ldu, Mp, 0, s->LinvU.d(), dinfo));
  // Kfu (N x M), then W = Kfu Luu^-T
  if (ctx->sgp_inverse) {
    GPX_TRY(ens(ctx, s->B0, mm));
    GPX_TRY(ens(ctx, s->Vu, mm));
    GPX_TRY(ens(ctx, s->Tscr, mm));
    GPX_TRY(build_linv_t(ctx, s->Kuu.d(), s->ldu, s->LinvU.d(), mt, s->B0.d(), s->ldu, s->Tscr.d(), s->Vu.d())); // Tu, Luu^-1
    GPX_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->evJoin, 0)); // Kfu (built on the panel stream, above)
    GPX_TRY(solve_by_inverse(ctx, s->Kfu.d(), s->ldw, ntl, s->Vu.d(), s->ldu, mt, s->Wn.d(), s->ldw));
  } else {
    GPX_TRY(launch_gram_padded(ctx, s->kp, ctx->X.d(), N, Ntp, s->Xu.d(), M, Mp, s->kfu_diag, s->kfu_diag != 0.0, 0, s->Wn.d(), s->ldw));
    GPX_TRY(trsm_right_lt(ctx, s->Wn.d(), s->ldw, ntl, s->Kuu.d(), s->ldu, s->LinvU.d(), mt, 0));
  }
  { // Wt = W^T, and |W|_F^2 in partial sums on the way (the bound's trace term)
    dim3 grid((Mp + 31) / 32, (Ntp + 31) / 32);
    transpose_sumsq_kernel<<<grid, 256, 0, ctx->s>>>(s->Wn.d(), s->ldw, Ntp, Mp, s->Wt.d(), s->ldt, N, M, s->part.d());
    GPX_HIP(ctx, hipGetLastError());
    s->npartW = (int)(grid.x * grid.y);
  }
  // A = I + Wt Wt^T / s2 (kept in Acopy), factor
  GPX_TRY(syrk_full(ctx, s->Wt.d(), s->ldt, mt, Ntp, 1.0 / s2, 1.0, M, s->A.d(), s->ldu));
  GPX_HIP(ctx, hipMemcpyAsync(s->Acopy.d(), s->A.d(), mm, hipMemcpyDeviceToDevice, ctx->stream));
  GPX_TRY(potrf_lower(ctx, s->A.d(), s->ldu, Mp, 0, s->LinvA.d(), dinfo + 1));
  // u = Wt y ;  c^T = (u^T / s2) LA^-T  (one-tile-row right TRSM)
  GPX_TRY(launch_rowdot(ctx, s->Wt.d(), s->ldt, M, N, ctx->yres.d(), 0.0, s->u.d(), nullptr, 0));
  if (ctx->sgp_inverse) {
    // TA = LA^-T and LA^-1 once, here: the gradient needs TA, the posterior LA^-1, and c = LA^-1 u / s2 is then one
    // matrix-vector product instead of a one-tile-row sweep of 2 dependent launches per 128 columns (0.4 ms at 16 tiles)
    GPX_TRY(ens(ctx, s->B1, mm));
    GPX_TRY(ens(ctx, s->VA, mm));
    GPX_TRY(build_linv_t(ctx, s->A.d(), s->ldu, s->LinvA.d(), mt, s->B1.d(), s->ldu, s->Tscr.d(), s->VA.d()));
    GPX_HIP(ctx, hipMemsetAsync(s->c.d(), 0, (size_t)Mp * 8, ctx->stream));
    GPX_TRY(launch_axpby(ctx, s->cpad.d(), 1.0 / s2, s->u.d(), 0.0, nullptr, M)); // cpad[0 .. M) = u / s2
    GPX_TRY(launch_rowdot(ctx, s->VA.d(), s->ldu, M, M, s->cpad.d(), 0.0, s->c.d(), nullptr, 0));
    return 0;
  }
  GPX_HIP(ctx, hipMemsetAsync(s->cpad.d(), 0, (size_t)TILE * s->ldu * 8, ctx->stream));
  GPX_TRY(launch_axpby(ctx, s->cpad.d(), 1.0 / s2, s->u.d(), 0.0, nullptr, M));
  GPX_TRY(trsm_right_lt(ctx, s->cpad.d(), s->ldu, 1, s->A.d(), s->ldu, s->LinvA.d(), mt, 0));
  GPX_HIP(ctx, hipMemcpyAsync(s->c.d(), s->cpad.d(), (size_t)Mp * 8, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

static int read_info(gpx_ctx* ctx, SgpState* s, int* out) {
  int hinfo[2];
  GPX_HIP(ctx, hipMemcpyAsync(hinfo, s->scal.i() + 1024, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hinfo[0] > s->M) hinfo[0] = 0;
  if (hinfo[1] > s->M) hinfo[1] = 0;
  *out = hinfo[0] != 0 ? hinfo[0] : (hinfo[1] != 0 ? -hinfo[1] : 0);
  return 0;
}

extern "C" {

int gpx_sgp_bound(gpx_ctx* ctx, int kind, const double* ell, double scale, double noise, double jitter,
                  const double* Xu, int Mi, const double* yres, int want_grad, double* bound,
                  double* grad_ell, double* grad_scale, double* grad_noise, double* grad_Xu, double* dyres,
                  int* info) {
  if (!ctx || ctx->device < 0) return -1;
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  SgpState* s = sgp_state(ctx);
  GPX_TRY(sgp_setup(ctx, s, kind, ell, scale, noise, jitter, Xu, Mi, yres, Mi == ctx->N ? 1e-6 : 0.0));
  GPX_TRY(sgp_forward(ctx, s));
  const int N = ctx->N, d = ctx->d, M = s->M, Mp = s->Mp, Ntp = s->Ntp, mt = Mp / TILE;
  const double s2 = noise, kd = kd_value(s->kp);
  double* sc = s->scal.d();
  const int npart = s->npartW; // |W|_F^2 in partial sums since the forward pass (transpose_sumsq_kernel)
  const size_t mm = (size_t)Mp * s->ldu * 8;
  const double* tvec = nullptr;
  int npartA = 0;
  if (want_grad) {
    GPX_TRY(ens(ctx, s->B0, mm)); GPX_TRY(ens(ctx, s->B1, mm)); GPX_TRY(ens(ctx, s->B2, mm));
    GPX_TRY(ens(ctx, s->B3, mm)); GPX_TRY(ens(ctx, s->B4, mm));
    GPX_TRY(ens(ctx, s->vvec, (size_t)Mp * 8)); GPX_TRY(ens(ctx, s->mvec, (size_t)Mp * 8));
    GPX_TRY(ens(ctx, s->tvec, (size_t)Ntp * 8));
    double* Tu = s->B0.d();
    double* TA = s->B1.d();
    GPX_TRY(ens(ctx, s->Tscr, mm));
    if (!ctx->sgp_inverse) // (otherwise Tu = Luu^-T is there since the forward pass)
      GPX_TRY(build_linv_t(ctx, s->Kuu.d(), s->ldu, s->LinvU.d(), mt, Tu, s->ldu, s->Tscr.d()));
    if (!ctx->sgp_inverse)
      GPX_TRY(build_linv_t(ctx, s->A.d(), s->ldu, s->LinvA.d(), mt, TA, s->ldu, s->Tscr.d())); // TA = LA^-T
    // v = TA c ; m = s2 Tu v ; t = s2 W v
    GPX_TRY(launch_rowdot(ctx, TA, s->ldu, M, M, s->c.d(), 0.0, s->vvec.d(), nullptr, 1));
    GPX_TRY(launch_rowdot(ctx, Tu, s->ldu, M, M, s->vvec.d(), 0.0, s->mvec.d(), nullptr, 1));
    GPX_TRY(launch_axpby(ctx, s->mvec.d(), s2, s->mvec.d(), 0.0, nullptr, M));
    GPX_TRY(launch_rowdot(ctx, s->Wn.d(), s->ldw, N, M, s->vvec.d(), 0.0, s->tvec.d(), nullptr, 0));
    GPX_TRY(launch_axpby(ctx, s->tvec.d(), s2, s->tvec.d(), 0.0, nullptr, N));
    tvec = s->tvec.d();
    // tr(A^-1) = |TA|_F^2
    npartA = (M + 7) / 8;
    sumsq_rows_kernel<<<npartA, 256, 0, ctx->s>>>(TA, s->ldu, M, M, s->part.d() + npart);
    GPX_HIP(ctx, hipGetLastError());
  }
  sgp_scalars_kernel<<<1, 1024, 0, ctx->s>>>(s->part.d(), npart, s->A.d(), s->ldu, M, s->c.d(), ctx->yres.d(), N,
                                             tvec, sc);
  if (want_grad) // second use of the same reducer for tr(A^-1): only out[0] of this call is meaningful
    sgp_scalars_kernel<<<1, 1024, 0, ctx->s>>>(s->part.d() + npart, npartA, s->A.d(), s->ldu, 0, s->c.d(),
                                               ctx->yres.d(), 0, nullptr, sc + 8);
  GPX_HIP(ctx, hipGetLastError());
  double h[16];
  int bad = 0;
  double wF2 = 0.0, yy = 0.0, yt = 0.0, tt = 0.0, trAinv = 0.0, trace_raw = 0.0;
  bool unclipped = false;
  // The reduced scalars and the pivot reports reach the host with ONE synchronisation: right here for a bound-only call,
  // behind the gradient's launches otherwise — those need nothing from the host (the clipped-trace switch is evaluated
  // on the device, mat_combine_kernel), so they queue up behind the forward pass without a round trip in between.
  auto fetch_scalars = [&]() -> int {
    GPX_HIP(ctx, hipMemcpyAsync(h, sc, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GPX_TRY(read_info(ctx, s, &bad));
    const double sumlogLA = h[1], cc = h[2];
    wF2 = h[0], yy = h[3], yt = h[4], tt = h[5], trAinv = h[8];
    trace_raw = N * kd - wF2;
    unclipped = trace_raw > 0.0; // jnp.clip(trace_term, a_min=0): zero value AND zero gradient when clipped
    if (info) *info = bad;
    if (bound) {
      *bound = bad ? NAN
                   : (-0.5 * N * 1.83787706640934548356 - 0.5 * N * std::log(s2) - sumlogLA - 0.5 * yy / s2 +
                      0.5 * cc - (unclipped ? 0.5 * trace_raw / s2 : 0.0));
    }
    return 0;
  };
  if (!want_grad) return fetch_scalars();

  // ---- matrix adjoints in whitened form ------------------------------------------------------
  //   G_uu  = Tu H Tu^T - m m^T / (2 s2^2),   H = -Ainv/2 + (1/2 + [u]/2) I - [u] A/2
  //   G_uf^T = W (Tu R)^T + rcoef_f m^T,        R = ([u] I - Ainv)/s2,  rcoef_f = y/s2^2 - t/s2^3
  double* Tu = s->B0.d();
  double* TA = s->B1.d();
  double* Ainv = s->B2.d();
  {
    GemmArgs g = gargs(TA, s->ldu, TA, s->ldu, Ainv, s->ldu, Mp, 1.0, 0.0);
    g.ktri = 1;
    dim3 gs((Mp + 255) / 256, Mp);
    if (mt >= 8) { // A^-1 = TA TA^T, lower triangle in k slabs, then mirrored
      GPX_TRY(tri_product_sym(ctx, g, mt, 4, Ainv, s->ldu, (double)Mp * Mp * Mp / 3.0));
    } else {
      g.lower = 1;
      GPX_TRY(launch_gemm_nt(ctx, g, mt, mt, 0, GPX_PROF_GEMM_OTHER, (double)Mp * Mp * Mp / 3.0));
      symmetrize_kernel<<<gs, 256, 0, ctx->s>>>(Ainv, s->ldu, Mp);
    }
    // H -> B3 ; R -> B4
    // ([u] = 1 unless the trace term is clipped: N kd - |W|_F^2 > 0, with |W|_F^2 = sc[0] reduced above)
    mat_combine_kernel<<<gs, 256, 0, ctx->s>>>(s->B3.d(), s->ldu, -0.5, Ainv, s->ldu, 0.0, -0.5, s->Acopy.d(), s->ldu,
                                               0.5, 0.5, sc, N * kd, Mp);
    mat_combine_kernel<<<gs, 256, 0, ctx->s>>>(s->B4.d(), s->ldu, -1.0 / s2, Ainv, s->ldu, 0.0, 0.0, nullptr, 0, 0.0,
                                               1.0 / s2, sc, N * kd, Mp);
    GPX_HIP(ctx, hipGetLastError());
  }
  { // E1 = Tu H -> B2 (Ainv dead) ; G0 = E1 Tu^T -> B3 (H dead after E1)
    GemmArgs g = gargs(Tu, s->ldu, s->B3.d(), s->ldu, s->B2.d(), s->ldu, Mp, 1.0, 0.0);
    g.ktri = 1; // ... and row i of the left factor at column i
    GPX_TRY(launch_gemm_nt(ctx, g, mt, mt, 0, GPX_PROF_GEMM_OTHER, (double)Mp * Mp * (Mp + TILE)));
    // G0 = Tu H Tu^T is symmetric: its lower triangle as Tu E1^T (left factor Tu: k starts at the row tile, so the
    // launch order hands out the long tiles first), mirrored afterwards — a quarter of the dense 2 M^3
    GemmArgs h2 = gargs(Tu, s->ldu, s->B2.d(), s->ldu, s->B3.d(), s->ldu, Mp, 1.0, 0.0);
    h2.ktri = 1;
    if (mt >= 8) {
      GPX_TRY(tri_product_sym(ctx, h2, mt, 4, s->B3.d(), s->ldu, (double)Mp * Mp * (Mp + TILE) / 2.0));
    } else {
      h2.lower = 1;
      GPX_TRY(launch_gemm_nt(ctx, h2, mt, mt, 0, GPX_PROF_GEMM_OTHER, (double)Mp * Mp * (Mp + TILE) / 2.0));
      dim3 gs3((Mp + 255) / 256, Mp);
      symmetrize_kernel<<<gs3, 256, 0, ctx->s>>>(s->B3.d(), s->ldu, Mp);
      GPX_HIP(ctx, hipGetLastError());
    }
  }
  GPX_TRY(ens(ctx, s->T1, (size_t)Ntp * s->ldw * 8));
  { // T1 = W R Tu^T as W (Tu R)^T: the M x M product first (M^3, k from the triangle of Tu; R is symmetric), then ONE
    // N x M x M GEMM — instead of W R followed by (W R) Tu^T, i.e. 2 N M^2 + M^3 flop for 3 N M^2 (round 3)
    double* Qt = s->B2.d(); // E1 is dead
    GemmArgs g = gargs(Tu, s->ldu, s->B4.d(), s->ldu, Qt, s->ldu, Mp, 1.0, 0.0);
    g.ktri = 1;
    GPX_TRY(launch_gemm_nt(ctx, g, mt, mt, 0, GPX_PROF_GEMM_OTHER, (double)Mp * Mp * (Mp + TILE)));
    GemmArgs h2 = gargs(s->Wn.d(), s->ldw, Qt, s->ldu, s->T1.d(), s->ldw, Mp, 1.0, 0.0);
    GPX_TRY(launch_gemm_nt(ctx, h2, Ntp / TILE, mt, 0, GPX_PROF_GEMM_OTHER, 2.0 * Ntp * (double)Mp * Mp));
  }
  GPX_TRY(ens(ctx, s->rcoef_u, (size_t)Mp * 8));
  GPX_TRY(ens(ctx, s->rcoef_f, (size_t)Ntp * 8));
  GPX_TRY(launch_axpby(ctx, s->rcoef_u.d(), -0.5 / (s2 * s2), s->mvec.d(), 0.0, nullptr, M));
  GPX_TRY(launch_axpby(ctx, s->rcoef_f.d(), 1.0 / (s2 * s2), ctx->yres.d(), -1.0 / (s2 * s2 * s2), s->tvec.d(), N));
  const int nbj = (M + 63) / 64, nbi_u = (M + 63) / 64, nbi_f = (N + 63) / 64;
  const int nbi_max = nbi_u > nbi_f ? nbi_u : nbi_f;
  GPX_TRY(ens(ctx, s->cpart, (size_t)nbj * nbi_max * SC_NV * 8));
  GPX_TRY(ens(ctx, s->gxu_part, (size_t)nbi_max * M * GPX_MAX_DIM * 8));
  GPX_TRY(ens(ctx, s->gXu, (size_t)M * d * 8));
  double* out_uu = sc + 32;
  double* out_uf = sc + 32 + SC_NV;
  {
    dim3 g1(nbj, nbi_u);
    const sgp_contract_fn contract = pick_contract(s->kp);
    contract<<<g1, 256, 0, ctx->s>>>(s->kp, s->Xu.d(), M, s->Xu.d(), M, s->B3.d(), s->ldu, s->rcoef_u.d(),
                                     s->mvec.d(), s->cpart.d(), s->gxu_part.d());
    sgp_reduce_kernel<<<512, 256, 0, ctx->s>>>(s->cpart.d(), nbj * nbi_u, d + 1, out_uu, s->gxu_part.d(), nbi_u, M, d,
                                              2.0, 0, s->gXu.d());
    dim3 g2(nbj, nbi_f);
    contract<<<g2, 256, 0, ctx->s>>>(s->kp, ctx->X.d(), N, s->Xu.d(), M, s->T1.d(), s->ldw,
                                     s->rcoef_f.d(), s->mvec.d(), s->cpart.d(), s->gxu_part.d());
    sgp_reduce_kernel<<<512, 256, 0, ctx->s>>>(s->cpart.d(), nbj * nbi_f, d + 1, out_uf, s->gxu_part.d(), nbi_f, M, d,
                                              1.0, 1, s->gXu.d());
    GPX_HIP(ctx, hipGetLastError());
  }
  double hu[GPX_MAX_DIM + 2], hf[GPX_MAX_DIM + 2];
  GPX_HIP(ctx, hipMemcpyAsync(hu, out_uu, (d + 1) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(ctx, hipMemcpyAsync(hf, out_uf, (d + 1) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (grad_Xu)
    GPX_HIP(ctx, hipMemcpyAsync(grad_Xu, s->gXu.d(), (size_t)M * d * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (dyres) { // dF/dyres = -(y/s2 - t/s2^2)
    GPX_TRY(launch_axpby(ctx, s->rcoef_f.d(), -1.0 / s2, ctx->yres.d(), 1.0 / (s2 * s2), s->tvec.d(), N));
    GPX_HIP(ctx, hipMemcpyAsync(dyres, s->rcoef_f.d(), (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  GPX_TRY(fetch_scalars()); // (synchronises the stream: everything copied above has arrived)
  const double uu = unclipped ? 1.0 : 0.0;
  if (grad_ell)
    for (int c = 0; c < d; ++c) grad_ell[c] = hu[c] + hf[c];
  if (grad_scale) *grad_scale = hu[d] + hf[d] - uu * 0.5 * N / s2 * (kd / scale);
  if (grad_noise) {
    const double trS1Phi = s2 * (M - trAinv);
    *grad_noise = -0.5 * N / s2 + 0.5 * trS1Phi / (s2 * s2) + 0.5 * yy / (s2 * s2) - yt / (s2 * s2 * s2) +
                  0.5 * tt / (s2 * s2 * s2 * s2) + uu * 0.5 * trace_raw / (s2 * s2);
  }
  if (bad) {
    if (grad_ell) for (int c = 0; c < d; ++c) grad_ell[c] = NAN;
    if (grad_scale) *grad_scale = NAN;
    if (grad_noise) *grad_noise = NAN;
  }
  return 0;
}

/* viSparseGP.get_mvn_posterior, gpax/models/sparse_gp.py:173-223 (no mean function on the device;
 * the host adds mean_fn(X_new)).  mean (Ms), cov (Ms*Ms or NULL), var (Ms or NULL). */
int gpx_sgp_posterior(gpx_ctx* ctx, int kind, const double* ell, double scale, double noise, double jitter,
                      const double* Xu, int Mi, const double* yres, const double* Xnew, int Ms, double noise_p,
                      double* mean, double* cov, double* var, int* info) {
  if (!ctx || ctx->device < 0) return -1;
  if (!Xnew || Ms < 1) return bad_arg(ctx, "sparse posterior arguments");
  GPX_HIP(ctx, hipSetDevice(ctx->device));
  SgpState* s = sgp_state(ctx);
  GPX_TRY(sgp_setup(ctx, s, kind, ell, scale, noise, jitter, Xu, Mi, yres, 0.0));
  GPX_TRY(sgp_forward(ctx, s));
  const int d = ctx->d, M = s->M, Mp = s->Mp, mt = Mp / TILE;
  const int Msp = round_up(Ms, TILE), st = Msp / TILE;
  GPX_TRY(ens(ctx, s->Xs, (size_t)Ms * d * 8));
  GPX_TRY(ens(ctx, s->V1, (size_t)Msp * s->ldw * 8));
  GPX_TRY(ens(ctx, s->V2, (size_t)Msp * s->ldw * 8));
  GPX_TRY(ens(ctx, s->mean, (size_t)Msp * 8));
  GPX_TRY(ens(ctx, s->var, (size_t)Msp * 8));
  GPX_TRY(ens(ctx, s->var2, (size_t)Msp * 8));
  GPX_HIP(ctx, hipMemcpyAsync(s->Xs.d(), Xnew, (size_t)Ms * d * 8, hipMemcpyHostToDevice, ctx->stream));
  // V1 = Ksu Luu^-T ; V2 = V1 LA^-T
  if (ctx->sgp_inverse) { // one GEMM each against Luu^-1 and LA^-1 (both from the forward pass)
    GPX_TRY(launch_gram_padded(ctx, s->kp, s->Xs.d(), Ms, Msp, s->Xu.d(), M, Mp, 0.0, 0, 0, s->V2.d(), s->ldw));
    GPX_TRY(solve_by_inverse(ctx, s->V2.d(), s->ldw, st, s->Vu.d(), s->ldu, mt, s->V1.d(), s->ldw));
    GPX_TRY(solve_by_inverse(ctx, s->V1.d(), s->ldw, st, s->VA.d(), s->ldu, mt, s->V2.d(), s->ldw));
  } else {
    GPX_TRY(launch_gram_padded(ctx, s->kp, s->Xs.d(), Ms, Msp, s->Xu.d(), M, Mp, 0.0, 0, 0, s->V1.d(), s->ldw));
    GPX_TRY(trsm_right_lt(ctx, s->V1.d(), s->ldw, st, s->Kuu.d(), s->ldu, s->LinvU.d(), mt, 0));
    GPX_HIP(ctx, hipMemcpyAsync(s->V2.d(), s->V1.d(), (size_t)Msp * s->ldw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    GPX_TRY(trsm_right_lt(ctx, s->V2.d(), s->ldw, st, s->A.d(), s->ldu, s->LinvA.d(), mt, 0));
  }
  const double kdiag = kd_value(s->kp) + noise_p + jitter; // Kss = kernel(X_new, X_new, params, noise_p, **jitter)
  GPX_TRY(launch_rowdot(ctx, s->V1.d(), s->ldw, Ms, M, s->c.d(), kdiag, nullptr, s->var.d(), 0));   // kd - |V1|^2
  GPX_TRY(launch_rowdot(ctx, s->V2.d(), s->ldw, Ms, M, s->c.d(), 0.0, s->mean.d(), s->var2.d(), 0)); // mean, -|V2|^2
  GPX_TRY(launch_axpby(ctx, s->var.d(), 1.0, s->var.d(), -1.0, s->var2.d(), Ms));
  if (cov) {
    // cov = Kss - V1 V1^T + V2 V2^T : two split-K SYRK slab sets, finalised against k_pp on the fly
    const int64_t ldc = pick_ld(Msp);
    GPX_TRY(ens(ctx, s->Cov, (size_t)Msp * ldc * 8));
    const int64_t ldp = ldc, stride = (int64_t)Msp * ldp;
    GPX_TRY(ens(ctx, ctx->SplitK, (size_t)2 * stride * 8));
    GemmArgs g = gargs(s->V1.d(), s->ldw, s->V1.d(), s->ldw, ctx->SplitK.d(), ldp, Mp, 1.0, 0.0);
    g.lower = 1;
    GPX_TRY(launch_gemm_nt(ctx, g, st, st, 0, GPX_PROF_GEMM_OTHER, (double)Msp * Msp * Mp));
    GemmArgs h2 = gargs(s->V2.d(), s->ldw, s->V2.d(), s->ldw, ctx->SplitK.d() + stride, ldp, Mp, -1.0, 0.0);
    h2.lower = 1;
    GPX_TRY(launch_gemm_nt(ctx, h2, st, st, 0, GPX_PROF_GEMM_OTHER, (double)Msp * Msp * Mp));
    GPX_TRY(launch_cov_finalize(ctx, s->kp, s->Xs.d(), Ms, Msp, ctx->SplitK.d(), 2, stride, ldp, noise_p + jitter,
                                s->Cov.d(), ldc));
    GPX_HIP(ctx, hipMemcpy2DAsync(cov, (size_t)Ms * 8, s->Cov.d(), ldc * 8, (size_t)Ms * 8, Ms, hipMemcpyDeviceToHost,
                                  ctx->stream));
  }
  if (mean) GPX_HIP(ctx, hipMemcpyAsync(mean, s->mean.d(), (size_t)Ms * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (var) GPX_HIP(ctx, hipMemcpyAsync(var, s->var.d(), (size_t)Ms * 8, hipMemcpyDeviceToHost, ctx->stream));
  int bad = 0;
  GPX_TRY(read_info(ctx, s, &bad));
  if (info) *info = bad;
  if (bad) {
    if (mean) for (int a = 0; a < Ms; ++a) mean[a] = NAN;
    if (var) for (int a = 0; a < Ms; ++a) var[a] = NAN;
    if (cov) for (int64_t a = 0; a < (int64_t)Ms * Ms; ++a) cov[a] = NAN;
  }
  return 0;
}

} // extern "C"
