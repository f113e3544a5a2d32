// fit_small.hip — the whole fit step (log marginal likelihood AND its gradient) of a small exact GP as ONE kernel launch:
// one 256-thread workgroup per hyper-parameter vector.
//
// Role on the path: gpax/models/gp.py:137-164 (ExactGP.model under NUTS / SVI: MultivariateNormal(0, K).log_prob(y) and its
// reverse-mode gradient), at the sizes gpax actually runs at — every reference notebook fits N = 6 ... 40 points
// (examples/gpax_simpleGP.ipynb: N = 25, 342 it/s x 7 leapfrogs).  There the launch sequence of the general path (Gram,
// augmentation, potf2, lml terms, tree level 0, row dots, K^-1 product, contraction, reduction: ~10 launches for 4e6
// flop) is nothing but launch latency: 0.17 ms per fit step at N = 128, 0.15 at N = 32 (profiles/r05/bench_first_with_configs.json).
//
// N <= 127, i.e. the augmented matrix [[K, .], [y^T, 1e300]] (+ identity padding) is ONE 128 x 128 block (N = 128: K is the
// block and the augmentation row rides in the tile row below it — a 16-row strip multiplied with L^-1 after the factorisation,
// as the general path's panel TRSM does with it):
//   A  Gram + augmentation row + identity padding -> the block in global memory (L2), L^-1 block preset to the identity;
//      the same arithmetic as gram_kernel / augment_kernel, so the block — and with it the factor the posterior reads
//      afterwards — is bit for bit the general path's
//   B  L and L^-1: potf2_small_body below when at most four 16 x 16 tile rows hold data (N <= 63: every tile in LDS, ~7 us at
//      N <= 31), potf2_slim_body (potf2_slim.h) above that — the same arithmetic as the general path's kernel either way
//   C  w = row N of L (= L^-1 y),  quad = |w|^2,  sumlog = sum log L_ii,  alpha = L^-T w
//   D  K^-1 = L^-T L^-1 tile by tile on the MFMA pipe (operands straight from the L^-1 block: both are read along its
//      rows), each 16 x 16 tile contracted with (alpha alpha^T - K^-1) o dK/dtheta out of the accumulators — dK
//      evaluated on the fly exactly as grad_contract_kernel does — then a fixed-order workgroup reduction.
// Batch entry = blockIdx.x (NUTS chains in lockstep, vExactGP tasks, sweeps of theta): every entry is the same code on its
// own slab, so batched and single launches give the same bits.
#include "common.h"
#include "potf2_slim.h"

namespace gpx {

struct FitSmallArgs {
  const double* X;      // (T x) N x d
  int N, d;
  KernelParams kp;      // by-value hyper-parameters (th == nullptr)
  double diag_train;    // noise + jitter
  const ThetaDev* th;   // per-entry table or nullptr
  TaskStride ts;        // per-task training inputs (entry b reads task b % mod)
  const double* y;      // residuals: entry b reads y + (y_mod > 0 ? b % y_mod : b) * y_bs
  int64_t y_bs;
  int y_mod;
  double* A;            // 128 x lda block per entry
  int64_t lda, a_bs;
  double* Linv;         // 128 x 128 per entry
  int64_t linv_bs;
  double* alpha;        // N per entry
  int64_t alpha_bs;
  double* scal;         // [quad, sumlog, grad(ell.., (period), scale, noise)] per entry
  int64_t scal_bs;
  int* info;            // one int per entry
  int want_grad;
};

constexpr int FS_RED = 32; // doubles of reduction scratch

// ---- the factorisation when at most four 16 x 16 tile rows hold data (N <= 63) ---------------------------------------------------
// potf2_slim_body walks all 8 panels of the 128 x 128 block whatever is in it: its workers visit every tile (26 us, of which
// an N = 25 problem needs the first two tile rows).  Here the nt <= 4 active tile rows live in LDS — C(i,j), i >= j, at slot
// i (i + 1) / 2 + j; the inverse's residual tiles R(i,c), i > c, behind them; Dinv; the column scratch of diag16 — and every
// tile goes through the SAME operations in the same order as in potf2_tile_body (the reference of the bit-identity tests):
// diag16 on the diagonal tile, L(i,p) = C(i,p) Dinv^T, X(p,c) = Dinv R(p,c), C(i,j) -= L(i,p) L(j,p)^T, R(i,c) -= L(i,p) X(p,c)
// for p ascending — so L and L^-1 are bit for bit those of the general path.  The tile jobs of a phase are dealt round
// robin to the four waves; three barriers per panel.  nt = 2: ~7 us, nt = 4: ~14 us.
constexpr int FS_NT = 4;
constexpr int FS_CT = FS_NT * (FS_NT + 1) / 2, FS_RT = FS_NT * (FS_NT - 1) / 2; // 10 Cholesky + 6 residual tiles
constexpr size_t FS_SMALL_LDS = (size_t)((FS_CT + FS_RT + 1) * TSZ + 64) * sizeof(double);

__device__ __forceinline__ void potf2_small_body(double* A, int64_t lda, double* Linv, int* info, double* lds, int nt) {
  double* Ct = lds;                      // C(i,j) -> L(i,j)
  double* Rt = lds + FS_CT * TSZ;        // R(i,c) -> X(i,c), slot i (i - 1) / 2 + c
  double* Dinv = lds + (FS_CT + FS_RT) * TSZ;
  double* col = Dinv + TSZ;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int crow = lane >> 4, ccol = lane & 15;
  // tiles in: the lower tiles of the active part; zero residuals; the strictly-upper tiles of the active part of A are
  // cleared as the general path's kernels do (Linv is preset to the identity by the caller)
  for (int idx = w; idx < nt * (nt + 1) / 2; idx += 4) {
    int i, j;
    lower_tile(idx, i, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ct[idx * TSZ + (crow + 4 * r) * TLD + ccol] = A[(int64_t)(i * TS + crow + 4 * r) * lda + j * TS + ccol];
    if (i > j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Rt[(i * (i - 1) / 2 + j) * TSZ + (crow + 4 * r) * TLD + ccol] = 0.0;
        A[(int64_t)(j * TS + crow + 4 * r) * lda + i * TS + ccol] = 0.0;
      }
    }
  }
  __syncthreads();
  int bad = 0;
  for (int p = 0; p < nt; ++p) {
    double* Dg = Ct + (p * (p + 1) / 2 + p) * TSZ;
    if (w == 0) { // the diagonal tile
      diag16(Dg, Dinv, col, lane, bad, p * TS);
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * q + t;
        A[(int64_t)(p * TS + r) * lda + p * TS + i] = Dg[r * TLD + i];
        Linv[(p * TS + r) * PB + p * TS + i] = Dinv[r * TLD + i];
      }
    }
    __syncthreads();
    // panel TRSM L(i,p), i > p, and the inverse row X(p,c), c < p
    for (int m = w; m < nt - 1; m += 4) {
      if (m < nt - 1 - p) {
        const int i = p + 1 + m;
        double* T = Ct + (i * (i + 1) / 2 + p) * TSZ;
        const pd4_t x = mma_nt(pd4_t{0.0, 0.0, 0.0, 0.0}, T, Dinv, lane, 1.0);
        acc_to_lds(x, T, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(int64_t)(i * TS + crow + 4 * r) * lda + p * TS + ccol] = x[r];
      } else {
        const int c = m - (nt - 1 - p);
        double* T = Rt + (p * (p - 1) / 2 + c) * TSZ;
        const pd4_t x = mma_nn(pd4_t{0.0, 0.0, 0.0, 0.0}, Dinv, T, lane, 1.0);
        acc_to_lds(x, T, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Linv[(p * TS + crow + 4 * r) * PB + c * TS + ccol] = x[r];
      }
    }
    __syncthreads();
    // trailing updates of panel p: C(i,j), i >= j > p, then R(i,c), i > p, c <= p — one list, dealt round robin
    {
      const int nc = (nt - 1 - p) * (nt - p) / 2, nr = (nt - 1 - p) * (p + 1);
      for (int m = w; m < nc + nr; m += 4) {
        if (m < nc) {
          int di, dj;
          lower_tile(m, di, dj); // (i, j) = (p + 1 + di, p + 1 + dj)
          const int i = p + 1 + di, j = p + 1 + dj;
          double* T = Ct + (i * (i + 1) / 2 + j) * TSZ;
          pd4_t acc = lds_to_acc(T, lane);
          acc = mma_nt(acc, Ct + (i * (i + 1) / 2 + p) * TSZ, Ct + (j * (j + 1) / 2 + p) * TSZ, lane, -1.0);
          acc_to_lds(acc, T, lane);
        } else {
          const int q = m - nc, i = p + 1 + q / (p + 1), c = q % (p + 1);
          double* T = Rt + (i * (i - 1) / 2 + c) * TSZ;
          pd4_t acc = lds_to_acc(T, lane);
          acc = mma_nn(acc, Ct + (i * (i + 1) / 2 + p) * TSZ, (c == p) ? Dinv : Rt + (p * (p - 1) / 2 + c) * TSZ, lane, -1.0);
          acc_to_lds(acc, T, lane);
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0 && bad != 0 && info != nullptr) {
    if (*info == 0) *info = bad;
  }
}

// Dynamic LDS: [region R | x (128 x DM) | w | alpha | reduction scratch].  R is the factorisation's LDS (potf2_slim.h) and,
// once that is done, the lower 16 x 16 tiles of L^-1 the K^-1 product reads (nt16 (nt16 + 1) / 2 tiles of 16 x 17 doubles:
// 21 KB up to N = 63 — inside what the factorisation needed anyway — 45 KB up to N = 95, 77 KB at N = 127).
__host__ __device__ constexpr size_t fit_small_region(int N) {
  const size_t nt = (size_t)(N + TS - 1) / TS, tiles = nt * (nt + 1) / 2 * TSZ * sizeof(double);
  const size_t fact = (N + 1 <= FS_NT * TS) ? FS_SMALL_LDS : POTF2_SLIM_LDS; // (the small form needs more LDS: all tiles live there)
  return tiles > fact ? tiles : fact;
}
template <int D>
constexpr size_t fit_small_lds(int N) {
  return fit_small_region(N) + (size_t)(PB * (D > 0 ? D : GPX_MAX_DIM) + 2 * PB + FS_RED) * sizeof(double);
}

__device__ __forceinline__ double fs_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// workgroup sum in a fixed order; the result is valid in thread 0
__device__ __forceinline__ double fs_block_sum(double v, double* red) {
  v = fs_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (threadIdx.x == 0) ? ((red[0] + red[1]) + (red[2] + red[3])) : 0.0;
}

template <int KIND, int D>
__global__ __launch_bounds__(256) void fit_small_kernel(FitSmallArgs a) {
  constexpr int DM = (D > 0) ? D : GPX_MAX_DIM;
  constexpr bool PER = (KIND == GPX_KERNEL_PERIODIC);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* sx = lds + fit_small_region(a.N) / sizeof(double); // PB x DM: training inputs (scaled by 1 / ell unless periodic)
  double* sw = sx + PB * DM;                           // w = L^-1 y
  double* sal = sw + PB;                               // alpha
  double* red = sal + PB;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = a.N, d = (D > 0) ? D : a.d;
  const ThetaDev* t = (a.th != nullptr) ? a.th + b : nullptr;
  const double k_scale = t ? t->kp.scale : a.kp.scale;
  const double pi_over_p = t ? t->kp.pi_over_p : a.kp.pi_over_p;
  const double diag_add = t ? t->diag_train : a.diag_train;
  double inv_ell[DM];
#pragma unroll
  for (int c = 0; c < DM; ++c) inv_ell[c] = (c < d) ? (t ? t->kp.inv_ell[c] : a.kp.inv_ell[c]) : 0.0;
  const double* X = a.X + (a.ts.mod > 0 ? (int64_t)(b % a.ts.mod) * a.ts.x_bs : 0);
  const double* y = a.y + (int64_t)(a.y_mod > 0 ? b % a.y_mod : b) * a.y_bs;
  double* A = a.A + (int64_t)b * a.a_bs;
  double* Linv = a.Linv + (int64_t)b * a.linv_bs;
  const int64_t lda = a.lda;
  int* info = a.info + b;

  // ---- A: the augmented Gram block -----------------------------------------------------------------------------------------
  // thread -> two adjacent columns (2 jj, 2 jj + 1) of every fourth row: one 16-byte store per matrix and row, a wave writes
  // 1 KiB of a row.  y may live in page-locked HOST memory (gpx_fit_batch): its load is issued first, beside that of the
  // hyper-parameters, so that the kernel pays ONE trip over the host link, not two in a row.
  const int jj = tid & 63, h = tid >> 6, j0 = 2 * jj;
  const double y0 = (j0 < N) ? y[j0] : 0.0, y1 = (j0 + 1 < N) ? y[j0 + 1] : 0.0;
  for (int idx = tid; idx < PB * d; idx += 256) {
    const int r = idx / d, c = idx - r * d;
    sx[r * DM + c] = (r < N) ? X[(int64_t)r * d + c] * (PER ? 1.0 : inv_ell[c]) : 0.0;
  }
  if (tid == 0) *info = 0;
  __syncthreads();
  {
    double z0[DM], z1[DM];
#pragma unroll
    for (int c = 0; c < DM; ++c) {
      z0[c] = (c < d) ? sx[j0 * DM + c] : 0.0;
      z1[c] = (c < d) ? sx[(j0 + 1) * DM + c] : 0.0;
    }
    for (int i = h; i < PB; i += 4) {
      double v0, v1;
      if (i < N) {
        double r20 = 0.0, r21 = 0.0;
#pragma unroll
        for (int c = 0; c < DM; ++c) {
          if (c < d) { // uniform
            const double x = sx[i * DM + c];
            double u0 = x - z0[c], u1 = x - z1[c];
            if (PER) {
              u0 = sin(u0 * pi_over_p) * inv_ell[c];
              u1 = sin(u1 * pi_over_p) * inv_ell[c];
            }
            r20 = fma(u0, u0, r20);
            r21 = fma(u1, u1, r21);
          }
        }
        v0 = kernel_value<KIND>(r20, k_scale);
        v1 = kernel_value<KIND>(r21, k_scale);
        if (i == j0) v0 += diag_add;
        if (i == j0 + 1) v1 += diag_add;
        if (j0 >= N) v0 = 0.0;
        if (j0 + 1 >= N) v1 = 0.0;
      } else if (i == N) {
        v0 = (j0 < N) ? y0 : (j0 == N ? AUG_BIG : 0.0);
        v1 = (j0 + 1 < N) ? y1 : (j0 + 1 == N ? AUG_BIG : 0.0);
      } else {
        v0 = (i == j0) ? 1.0 : 0.0;
        v1 = (i == j0 + 1) ? 1.0 : 0.0;
      }
      *reinterpret_cast<double2*>(A + (int64_t)i * lda + j0) = make_double2(v0, v1);
      // what the factorisation does not visit of L^-1 stays the identity
      *reinterpret_cast<double2*>(Linv + i * PB + j0) = make_double2(i == j0 ? 1.0 : 0.0, i == j0 + 1 ? 1.0 : 0.0);
    }
  }
  if (N == PB) {
    // N = 128: K fills the block and the augmentation row opens a tile row of its own below it (rows 128 .. 255 = [y | 1e300],
    // identity padding — what augment_kernel writes there).  Nothing in that tile needs factoring: what the path reads is
    // w = y L^-T in row 128, the panel TRSM of the general path (dev_factor: "rides along below the square part")
    for (int i = PB + h; i < 2 * PB; i += 4) {
      *reinterpret_cast<double2*>(A + (int64_t)i * lda + j0) = (i == PB) ? make_double2(y0, y1) : make_double2(0.0, 0.0);
      const int c0 = PB + j0;
      *reinterpret_cast<double2*>(A + (int64_t)i * lda + c0) =
          make_double2(i == c0 ? (i == PB ? AUG_BIG : 1.0) : 0.0, i == c0 + 1 ? 1.0 : 0.0);
    }
  }
  __syncthreads(); // (workgroup-scope release / acquire: the block is visible to every wave of this workgroup)

  // ---- B: L and L^-1 ----------------------------------------------------------------------------------------------------------
  if (N + 1 <= FS_NT * TS) potf2_small_body(A, lda, Linv, info, lds, (N + 1 + TS - 1) / TS);
  else potf2_slim_body(A, lda, Linv, info, 0, lds, N < PB ? N + 1 : PB);
  __syncthreads();
  if (N == PB) {
    // w = y L^-T into row 128: the 16-row strip [y; 0 ...] times L^-1 (B operand: row j of L^-1, straight from the L2 this
    // workgroup just wrote it through), k ascending in v_mfma_f64_16x16x4_f64 steps from 0 — the operations of the panel TRSM
    // strip of the general path on that row (gemm_tile.h lat_tile / potf2.hip), so the same bits.  Wave w: column tiles 2 w, 2 w + 1.
    const int lane_ = tid & 63, wave_ = tid >> 6, fr_ = lane_ & 15, fk_ = lane_ >> 4;
    double* wrow = A + (int64_t)PB * lda;
    double af[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) af[kk] = (fr_ == 0) ? wrow[4 * kk + fk_] : 0.0;
    __syncthreads(); // every wave holds y before any wave stores its part of w over it
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int j = (2 * wave_ + n) * TS + fr_;
      const double* brow = Linv + j * PB + fk_;
      double bf[32];
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) bf[kk] = brow[4 * kk];
      pd4_t acc = pd4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk], acc, 0, 0, 0);
      if (fk_ == 0) wrow[j] = 1.0 * acc[0]; // D: lane l, register r = row (l >> 4) + 4 r, column l & 15 — row 0 is the strip's y row
    }
    __syncthreads();
  }

  // ---- C: w, quad, sumlog, alpha ------------------------------------------------------------------------------------------------
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int nt16 = (N + TS - 1) / TS;
  const int ntiles = nt16 * (nt16 + 1) / 2;
  // the lower tiles of L^-1 into LDS (tile (kt, ti), kt >= ti, at slot kt (kt + 1) / 2 + ti; rows padded to 17 doubles): one
  // coalesced pass with every load in flight at once.  alpha and the K^-1 product below read L^-1 from there — straight
  // from memory every dependent step would wait for its own operands (the first version of this kernel: 73 us at N = 127,
  // most of it in the 64-step column sums of alpha; profiles/r05/fit_small.json)
  for (int idx = wave; idx < ntiles; idx += 4) {
    int kt, ti;
    lower_tile(idx, kt, ti);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      lds[idx * TSZ + (fk + 4 * r) * TLD + fr] = Linv[(kt * TS + fk + 4 * r) * PB + ti * TS + fr];
  }
  double q = 0.0, sl = 0.0;
  if (tid < PB) {
    const double wk = (tid < N) ? A[(int64_t)N * lda + tid] : 0.0;
    sw[tid] = wk;
    q = wk * wk;
    sl = (tid < N) ? log(A[(int64_t)tid * lda + tid]) : 0.0;
  }
  const double qs = fs_block_sum(q, red); // (its barriers also publish sw and the tiles)
  const double ss = fs_block_sum(sl, red);
  double* out = a.scal + (int64_t)b * a.scal_bs;
  if (tid == 0) {
    out[0] = qs;
    out[1] = ss;
  }
  if (!a.want_grad) return;
  {
    // alpha_i = sum_{k >= i} Linv[k][i] w[k] (alpha = L^-T w): the two halves of the workgroup take the even / the odd k of
    // column i, added even + odd
    const int i = tid & (PB - 1), h = tid >> 7;
    const int ti = i >> 4, ci = i & 15;
    double s = 0.0;
    if (i < N) {
      for (int kt = ti; kt < nt16; ++kt) {
        const double* tile = lds + (kt * (kt + 1) / 2 + ti) * TSZ + ci;
#pragma unroll
        for (int r = h; r < TS; r += 2) s = fma(tile[r * TLD], sw[kt * TS + r], s); // (rows k < i of the diagonal tile hold zeros)
      }
    }
    if (h == 1) sal[i] = s;
    __syncthreads();
    if (h == 0) {
      const double al = (i < N) ? s + sal[i] : 0.0;
      sal[i] = al;
      if (i < N) a.alpha[(int64_t)b * a.alpha_bs + i] = al;
    }
    __syncthreads();
  }

  // ---- D: K^-1 tiles on the MFMA pipe, contracted out of the accumulators --------------------------------------------------------
  const int ne = d + (PER ? 1 : 0);
  double acc_ell[DM], acc_p = 0.0, acc_s = 0.0, acc_n = 0.0;
#pragma unroll
  for (int c = 0; c < DM; ++c) acc_ell[c] = 0.0;
  for (int idx = wave; idx < ntiles; idx += 4) {
    int ti, tj;
    lower_tile(idx, ti, tj);
    pd4_t acc = pd4_t{0.0, 0.0, 0.0, 0.0};
    // Kinv[i][j] = sum_{k >= max(i, j)} Linv[k][i] Linv[k][j]: A operand lane -> [i = fr][k = fk] = tile(kt, ti)[k][i],
    // B operand lane -> [k = fk][j = fr] = tile(kt, tj)[k][j] — the same read pattern for both; k ascending.  Rows >= N of
    // Linv contribute nothing (the augmentation row is ~1e-300, the padding the identity)
    for (int kt = ti; kt < nt16; ++kt) {
      const double* ta = lds + (kt * (kt + 1) / 2 + ti) * TSZ + fk * TLD + fr;
      const double* tb = lds + (kt * (kt + 1) / 2 + tj) * TSZ + fk * TLD + fr;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[4 * kk * TLD], tb[4 * kk * TLD], acc, 0, 0, 0);
    }
    const int j = tj * TS + fr;
    if (j < N) {
      const double aj = sal[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * TS + fk + 4 * r;
        if (i >= N || j > i) continue;
        const double G = sal[i] * aj - acc[r];
        const double wg = ((i == j) ? 0.5 : 1.0) * G;
        if (PER) {
          double qq = 0.0, dp = 0.0, s2[DM];
#pragma unroll
          for (int c = 0; c < DM; ++c) {
            s2[c] = 0.0;
            if (c < d) {
              const double delta = sx[i * DM + c] - sx[j * DM + c];
              const double sn = sin(delta * pi_over_p), cs = cos(delta * pi_over_p);
              s2[c] = sn * sn * inv_ell[c] * inv_ell[c];
              qq += s2[c];
              dp += sn * cs * delta * inv_ell[c] * inv_ell[c];
            }
          }
          const double kv = k_scale * exp(-2.0 * qq);
#pragma unroll
          for (int c = 0; c < DM; ++c) acc_ell[c] += wg * kv * 4.0 * s2[c] * inv_ell[c];
          acc_p += wg * kv * 4.0 * dp * pi_over_p * pi_over_p / 3.14159265358979323846;
          acc_s += wg * kv / k_scale;
        } else {
          double r2 = 0.0, u2[DM];
#pragma unroll
          for (int c = 0; c < DM; ++c) {
            u2[c] = 0.0;
            if (c < d) {
              const double u = sx[i * DM + c] - sx[j * DM + c]; // (both already scaled by 1 / ell)
              u2[c] = u * u;
              r2 += u2[c];
            }
          }
          double kv, dk;
          if (KIND == GPX_KERNEL_RBF) {
            kv = k_scale * exp(-0.5 * r2);
            dk = -0.5 * kv;
          } else {
            const double rr = sqrt(r2 + MATERN_EPS);
            const double e = exp(-SQRT5 * rr);
            kv = k_scale * (1.0 + SQRT5 * rr + (5.0 / 3.0) * r2) * e;
            dk = -(5.0 / 6.0) * k_scale * e * (1.0 + SQRT5 * r2 / rr);
          }
#pragma unroll
          for (int c = 0; c < DM; ++c) acc_ell[c] += wg * dk * (-2.0 * u2[c] * inv_ell[c]);
          acc_s += wg * kv / k_scale;
        }
        if (i == j) acc_n += wg;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < DM; ++c) {
    if (c < d) { // uniform
      const double s = fs_block_sum(acc_ell[c], red);
      if (tid == 0) out[2 + c] = s;
    }
  }
  if (PER) {
    const double s = fs_block_sum(acc_p, red);
    if (tid == 0) out[2 + d] = s;
  }
  {
    const double s1 = fs_block_sum(acc_s, red);
    const double s2 = fs_block_sum(acc_n, red);
    if (tid == 0) {
      out[2 + ne] = s1;
      out[2 + ne + 1] = s2;
    }
  }
}

template <int KIND>
static int fit_small_dispatch(gpx_ctx* ctx, const FitSmallArgs& a, int batch) {
#define GPX_FS_LAUNCH(DD)                                                                                               \
  do {                                                                                                                  \
    const size_t lds = fit_small_lds<DD>(a.N);                                                                          \
    constexpr unsigned bit = 1u << (KIND * 5 + (DD));                                                                   \
    if (lds > 48 * 1024 && !(ctx->fit_small_attr & bit)) {                                                              \
      GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fit_small_kernel<KIND, DD>),                       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fit_small_lds<DD>(PB)));    \
      ctx->fit_small_attr |= bit;                                                                                       \
    }                                                                                                                   \
    fit_small_kernel<KIND, DD><<<batch, 256, lds, ctx->s>>>(a);                                                          \
  } while (0)
  switch (a.d) {
    case 1: GPX_FS_LAUNCH(1); break;
    case 2: GPX_FS_LAUNCH(2); break;
    case 3: GPX_FS_LAUNCH(3); break;
    case 4: GPX_FS_LAUNCH(4); break;
    default: GPX_FS_LAUNCH(0);
  }
#undef GPX_FS_LAUNCH
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// The fit step of `batch` hyper-parameter vectors at N <= 128 as one launch.  K / Linv / alpha / scal as the general path
// lays them out (BatchPlan strides); afterwards K holds L (the factor gpx_posterior reads), Linv its inverse.
int launch_fit_small(gpx_ctx* ctx, const KernelParams& kp, double diag_train, const ThetaDev* th, TaskStride ts,
                     const double* dX, int N, const double* dy, int64_t y_bs, int y_mod, double* dK, int64_t ldk,
                     int64_t k_bs, double* dLinv, int64_t linv_bs, double* dalpha, int64_t alpha_bs, double* dscal,
                     int64_t scal_bs, int* dinfo, int want_grad, int batch) {
  if (N < 1 || N > PB) return bad_arg(ctx, "fit_small: N must be 1..128");
  if (batch < 1) batch = 1;
  FitSmallArgs a{};
  a.X = dX;
  a.N = N;
  a.d = kp.d;
  a.kp = kp;
  a.diag_train = diag_train;
  a.th = th;
  a.ts = ts;
  a.y = dy;
  a.y_bs = y_bs;
  a.y_mod = y_mod;
  a.A = dK;
  a.lda = ldk;
  a.a_bs = k_bs;
  a.Linv = dLinv;
  a.linv_bs = linv_bs;
  a.alpha = dalpha;
  a.alpha_bs = alpha_bs;
  a.scal = dscal;
  a.scal_bs = scal_bs;
  a.info = dinfo;
  a.want_grad = want_grad;
  // flops: Gram n^2 kernel evaluations aside, potf2 2 n^3 / 3 + K^-1 n^3 / 3 — counted under the potf2 class
  ProfScope ps(ctx, GPX_PROF_POTF2, batch * (2.0 * PB * (double)PB * PB / 3.0 + (double)N * N * N / 3.0));
  if (kp.kind == GPX_KERNEL_RBF) return fit_small_dispatch<GPX_KERNEL_RBF>(ctx, a, batch);
  if (kp.kind == GPX_KERNEL_PERIODIC) return fit_small_dispatch<GPX_KERNEL_PERIODIC>(ctx, a, batch);
  return fit_small_dispatch<GPX_KERNEL_MATERN52>(ctx, a, batch);
}

} // namespace gpx
