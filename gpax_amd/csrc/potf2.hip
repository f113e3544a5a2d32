// potf2.hip — 128x128 diagonal-block Cholesky fused with the inverse of the factor, one
// workgroup, register-resident (gfx950).
//
// Role on the path: the serial kernel of the right-looking blocked Cholesky that replaces
// jnp.linalg.cholesky inside NumPyro's MultivariateNormal (gpax/models/gp.py:160-164,292).
// The inverse of the diagonal block turns every panel TRSM into an MFMA GEMM.
//
// Layout: 256 threads as a 16x16 grid (ty, tx); thread owns S[a][b] = A[ty + 16 a][tx + 16 b],
// 64 doubles in VGPRs.  The lower triangle holds the Cholesky working matrix; the strict upper
// triangle (unused by Cholesky) holds the forward-substitution residual of L X = I, transposed:
// S[c][i] = R[i][c] for c < i.  At step j the owners of column j broadcast it through LDS (one
// barrier per step, double-buffered) and ONE update rule serves both halves:
//     S[r][i] -= S[r][j] * S[i][j] / d_j        for i > j and (r >= i  or  r <= j)
// (with S[j][j] read as 1 in the r == j row).  Column scaling by 1/sqrt(d_j) is deferred to the
// end, where it is again uniform per column for both L and L^-1.
// Work is skipped at 16-column granularity (wave-uniform), per-thread predicates elsewhere.
// A non-positive pivot makes sqrt() produce NaN (propagates, like JAX) and sets *info.
#include "common.h"

#include <cstdlib>

namespace gpx {

constexpr int PB = 128;
constexpr int PT_LD = PB + 1;
constexpr size_t POTF2_LDS_BYTES = (size_t)(64 * PT_LD + 2 * PB + PB) * sizeof(double);

template <int JB>
__device__ __forceinline__ void potf2_block(double (&S)[8][8], double* colbuf, double* dsv,
                                            int tx, int ty, int& bad) {
#pragma unroll 1
  for (int jj = 0; jj < 16; ++jj) {
    const int j = JB * 16 + jj;
    double* cb = colbuf + (j & 1) * PB;
    if (tx == jj) {
#pragma unroll
      for (int a = 0; a < 8; ++a) cb[ty + 16 * a] = S[a][JB];
    }
    __syncthreads();
    const double dj = cb[j];
    if (tx == 0 && ty == 0) {
      dsv[j] = dj;
      if (!(dj > 0.0) && bad == 0) bad = j + 1;
    }
    const double ip2 = 1.0 / dj;
    double cr[8], cc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) cr[a] = cb[ty + 16 * a];
    if (ty == jj) cr[JB] = 1.0; // row r == j of the X half
#pragma unroll
    for (int b = JB; b < 8; ++b) cc[b] = cb[tx + 16 * b] * ip2;
    const bool col_gt = tx > jj;   // column in slot JB is beyond j
    const bool row_le = ty <= jj;  // row in slot JB is <= j (X half incl. r == j)
    const bool lower = ty >= tx;   // within a diagonal slot: r >= i
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
      for (int b = JB; b < 8; ++b) {
        bool on;
        if (a < JB) {
          on = (b > JB) || col_gt;
        } else if (a == JB) {
          if (b == JB)
            on = col_gt && (row_le || lower);
          else
            on = row_le;
        } else { // a > JB: Cholesky half only, needs r >= i
          if (b > a) continue;
          if (b == a)
            on = lower && ((b > JB) || col_gt);
          else
            on = (b > JB) || col_gt;
        }
        if (on) S[a][b] = fma(-cr[a], cc[b], S[a][b]);
      }
    }
  }
}

__global__ __launch_bounds__(256, 1) void potf2_inv_kernel(double* A, int64_t lda, double* Linv,
                                                           int* info, int info_base, int64_t a_bs,
                                                           int64_t linv_bs) {
  A += (int64_t)blockIdx.x * a_bs; // one workgroup per batch entry
  Linv += (int64_t)blockIdx.x * linv_bs;
  if (info != nullptr) info += blockIdx.x;
  // latency-critical serial kernel of the factorisation: win issue arbitration against the
  // trailing-update waves it shares a CU with under look-ahead
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* T = lds;                   // 64 x PT_LD transpose staging (one half at a time)
  double* colbuf = lds + 64 * PT_LD; // 2 x PB
  double* dsv = colbuf + 2 * PB;     // PB pivots (d_j before sqrt)
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;

  double S[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int r = ty + 16 * a;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = tx + 16 * b;
      S[a][b] = (r >= i) ? A[(int64_t)r * lda + i] : 0.0;
    }
  }
  int bad = 0;
  potf2_block<0>(S, colbuf, dsv, tx, ty, bad);
  potf2_block<1>(S, colbuf, dsv, tx, ty, bad);
  potf2_block<2>(S, colbuf, dsv, tx, ty, bad);
  potf2_block<3>(S, colbuf, dsv, tx, ty, bad);
  potf2_block<4>(S, colbuf, dsv, tx, ty, bad);
  potf2_block<5>(S, colbuf, dsv, tx, ty, bad);
  potf2_block<6>(S, colbuf, dsv, tx, ty, bad);
  potf2_block<7>(S, colbuf, dsv, tx, ty, bad);
  __syncthreads();

  // deferred column scaling; write L (lower, zeros above) and stage L^-1 through LDS in two
  // 64-row halves (66 KB: leaves room for a GEMM workgroup on the same CU during look-ahead)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int b = half * 4 + bb;
      const int i = tx + 16 * b;
      const double piv = sqrt(dsv[i]);
      const double ip = 1.0 / piv;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int r = ty + 16 * a;
        double lval, xval;
        if (r > i) {
          lval = S[a][b] * ip;
          xval = 0.0;
        } else if (r == i) {
          lval = piv;
          xval = ip;
        } else {
          lval = 0.0;
          xval = S[a][b] * ip; // = Linv[i][r]
        }
        A[(int64_t)r * lda + i] = lval;
        T[(i - 64 * half) * PT_LD + r] = xval;
      }
    }
    __syncthreads();
    {
      const int col = tid & 127;
      for (int row = tid >> 7; row < 64; row += 2)
        Linv[(row + 64 * half) * PB + col] = T[row * PT_LD + col];
    }
    __syncthreads();
  }
  if (tid == 0 && bad != 0 && info != nullptr) {
    if (*info == 0) *info = info_base + bad;
  }
}

} // namespace gpx

// =================================================================================================
// potf2_tile_kernel — the same 128x128 factor + inverse, re-blocked at 16x16 tiles so that all
// O(n^3) work runs on v_mfma_f64_16x16x4_f64 and only the 16x16 diagonal tiles are factored with
// scalar code (one wave, 16 column steps each).  8 panel steps x 4 barriers instead of 128 column
// steps; ~37 KB LDS (co-resident with a GEMM workgroup under look-ahead).
//
// Tiles live in REGISTERS as MFMA accumulators, owned statically by the 4 waves: the 36 lower
// Cholesky tiles C(i,j) and the 28 strictly-lower residual tiles R(i,c) of the forward substitution
// L X = I (R(i,i) = I implicit).  Panel p:
//   A  owners dump column-p tiles C(i,p), i >= p, and row-p residual tiles R(p,c), c < p, to LDS
//   B  wave 0 factors the diagonal tile (fused factor + inverse, as potf2_inv_kernel at 16x16)
//   C  TRSM tiles L(i,p) = C(i,p) Linv_pp^T (i > p) and inverse row X(p,c) = Linv_pp R(p,c) (c < p)
//   D  C(i,j) -= L(i,p) L(j,p)^T  (i >= j > p);   R(i,c) -= L(i,p) X(p,c)  (i > p, c <= p)
// =================================================================================================
namespace gpx {

typedef double pd4_t __attribute__((ext_vector_type(4)));

constexpr int TS = 16;
constexpr int TLD = 17;
constexpr int TSZ = TS * TLD; // doubles per LDS tile
constexpr size_t POTF2_TILE_LDS = (size_t)(17 * TSZ + 160) * sizeof(double); // + scratch of the diagonal-tile factor

__device__ __forceinline__ void lower_tile(int idx, int& i, int& j) { // idx = i (i + 1) / 2 + j
  i = (idx >= 28) ? 7 : (idx >= 21) ? 6 : (idx >= 15) ? 5 : (idx >= 10) ? 4 : (idx >= 6) ? 3 : (idx >= 3) ? 2 : (idx >= 1) ? 1 : 0;
  j = idx - i * (i + 1) / 2;
}
__device__ __forceinline__ void strict_tile(int idx, int& i, int& c) { // idx = i (i - 1) / 2 + c, i > c
  i = (idx >= 21) ? 7 : (idx >= 15) ? 6 : (idx >= 10) ? 5 : (idx >= 6) ? 4 : (idx >= 3) ? 3 : (idx >= 1) ? 2 : 1;
  c = idx - i * (i - 1) / 2;
}

__device__ __forceinline__ void acc_to_lds(const pd4_t& a, double* T, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) T[((lane >> 4) + 4 * r) * TLD + (lane & 15)] = a[r];
}

// acc += sgn * TA * TB^T   (TA[m][k], TB[n][k], 16x16 row-major tiles with leading dimension TLD)
__device__ __forceinline__ pd4_t mma_nt(pd4_t acc, const double* TA, const double* TB, int lane, double sgn) {
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double a = sgn * TA[fr * TLD + fk + 4 * kk];
    const double b = TB[fr * TLD + fk + 4 * kk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// acc += sgn * TA * TB     (TA[m][k], TB[k][n])
__device__ __forceinline__ pd4_t mma_nn(pd4_t acc, const double* TA, const double* TB, int lane, double sgn) {
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double a = sgn * TA[fr * TLD + fk + 4 * kk];
    const double b = TB[(fk + 4 * kk) * TLD + fr];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// One wave: factor the 16x16 tile D (lower read) in place -> L (zeros above), inverse -> Dinv.
// Lane (r = lane & 15, q = lane >> 4) owns S[r][4q .. 4q+3]; fused update rule as potf2_block.
__device__ __forceinline__ void diag16(double* D, double* Dinv, double* col /* 2 x 16 */, int lane, int& bad,
                                       int base) {
  const int r = lane & 15, q = lane >> 4;
  double e[4], mypiv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = 4 * q + t;
    e[t] = (r >= i) ? D[r * TLD + i] : 0.0;
    mypiv[t] = 1.0;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double* cb = col + (j & 1) * 16;
    if (q == (j >> 2)) cb[r] = e[j & 3];
    // one wave: the LDS queue is in order, so the reads below see the write; the asm only stops the
    // compiler from caching / reordering the accesses (no volatile: the six reads share one wait)
    asm volatile("" ::: "memory");
    const double dj = cb[j];
    const double crr = cb[r];
    const double c0 = cb[4 * q + 0], c1 = cb[4 * q + 1], c2 = cb[4 * q + 2], c3 = cb[4 * q + 3];
    asm volatile("" ::: "memory");
    const double cr = (r == j) ? 1.0 : crr;
    double ip2 = __builtin_amdgcn_rcp(dj);
    ip2 = fma(fma(-dj, ip2, 1.0), ip2, ip2);
    ip2 = fma(fma(-dj, ip2, 1.0), ip2, ip2);
    if (q == (j >> 2)) mypiv[j & 3] = dj;
    if (!(dj > 0.0) && bad == 0) bad = base + j + 1;
    const double cc[4] = {c0 * ip2, c1 * ip2, c2 * ip2, c3 * ip2};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = 4 * q + t;
      const bool on = (i > j) && (r >= i || r <= j);
      if (on) e[t] = fma(-cr, cc[t], e[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = 4 * q + t;
    const double piv = sqrt(mypiv[t]);
    const double ip = 1.0 / piv;
    double lval, xval;
    if (r > i) {
      lval = e[t] * ip;
      xval = 0.0;
    } else if (r == i) {
      lval = piv;
      xval = ip;
    } else {
      lval = 0.0;
      xval = e[t] * ip; // = Linv[i][r]
    }
    D[r * TLD + i] = lval;
    Dinv[i * TLD + r] = xval;
  }
}

// diag16, BLOCKED BY 4 COLUMNS (round 2) — the same arithmetic, entry by entry and in the same order, as diag16 above
// (every S[r][i] still receives  S[r][i] = fma(-u_j[r], v_j[i], S[r][i])  for j = 0, 1, ... with u_j = column j,
// v_j = column j x refined 1 / d_j: bit-identical results, tests/test_gpu_edges.py), but with TWO LDS round trips per
// four columns instead of four.  Panel P = columns 4P .. 4P+3, owned by the lanes with q == P (one row each):
//   1. its 4 x 4 diagonal block goes through LDS to every lane, which replays the block's four elimination steps in
//      registers (pivots, reciprocals, multipliers v) — redundantly, so no further exchange is needed to bring the
//      owners' own rows of the panel up to date;
//   2. the owners publish U (their column entries, 1 on the pivot row) and V (entries x 1 / d); the lanes right of the
//      panel apply the four rank-1 updates to their 4 entries, in column order.
// MEASURED SLOWER AND LEFT OFF (GPX_POTF2_DIAG=blocked enables it; tools/exp/potf2_phase.hip): the diagonal-tile phase
// takes 7050 cycles per panel against 5740 — the four column steps every lane replays cost ~60 fp64 VALU operations
// at 8 cycles each, more than the two LDS round trips they save (a column step is ~360 cycles, of which the round
// trip is about a third); 54 against 51 us per block.
__device__ __forceinline__ double refined_rcp(double d) {
  double ip = __builtin_amdgcn_rcp(d);
  ip = fma(fma(-d, ip, 1.0), ip, ip);
  ip = fma(fma(-d, ip, 1.0), ip, ip);
  return ip;
}

__device__ __forceinline__ void diag16_blk(double* D, double* Dinv, double* scr /* 16 + 64 + 64 */, int lane, int& bad,
                                           int base) {
  const int r = lane & 15, q = lane >> 4;
  double* blk = scr;
  double* Ub = scr + 16;
  double* Vb = scr + 80;
  double e[4], mypiv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = 4 * q + t;
    e[t] = (r >= i) ? D[r * TLD + i] : 0.0;
    mypiv[t] = 1.0;
  }
#pragma unroll
  for (int P = 0; P < 4; ++P) {
    const int j0 = 4 * P;
    const bool owner = (q == P);
    if (owner && r >= j0 && r < j0 + 4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) blk[(r - j0) * 4 + t] = e[t];
    }
    asm volatile("" ::: "memory"); // one wave: the LDS queue is in order (see diag16)
    double b[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) b[a][c] = blk[a * 4 + c];
    asm volatile("" ::: "memory");
    double ipv[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const int j = j0 + s_;
      const double dj = b[s_][s_];
      const double ip2 = refined_rcp(dj);
      ipv[s_] = ip2;
      if (owner) mypiv[s_] = dj;
      if (!(dj > 0.0) && bad == 0) bad = base + j + 1;
#pragma unroll
      for (int t = s_ + 1; t < 4; ++t) {
        const double vv = b[t][s_] * ip2; // S[j0 + t][j] / d_j
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (a >= t || a <= s_) b[a][t] = fma(-((a == s_) ? 1.0 : b[a][s_]), vv, b[a][t]);
        }
        if (owner) {
          const bool on = (r >= j0 + t) || (r <= j);
          const double u = (r == j) ? 1.0 : e[s_];
          if (on) e[t] = fma(-u, vv, e[t]);
        }
      }
    }
    if (P < 3) {
      if (owner) {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          Ub[r * 4 + s_] = (r == j0 + s_) ? 1.0 : e[s_];
          Vb[r * 4 + s_] = e[s_] * ipv[s_];
        }
      }
      asm volatile("" ::: "memory");
      double u[4], vv[4][4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) u[s_] = Ub[r * 4 + s_];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) vv[t][s_] = Vb[((4 * q + t) & 15) * 4 + s_];
      asm volatile("" ::: "memory");
      if (q > P) {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int i = 4 * q + t;
            const bool on = (r >= i) || (r <= j0 + s_);
            if (on) e[t] = fma(-u[s_], vv[t][s_], e[t]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = 4 * q + t;
    const double piv = sqrt(mypiv[t]);
    const double ip = 1.0 / piv;
    double lval, xval;
    if (r > i) {
      lval = e[t] * ip;
      xval = 0.0;
    } else if (r == i) {
      lval = piv;
      xval = ip;
    } else {
      lval = 0.0;
      xval = e[t] * ip; // = Linv[i][r]
    }
    D[r * TLD + i] = lval;
    Dinv[i * TLD + r] = xval;
  }
}

// Phase tracing for tools/exp/potf2_phase.hip (compiled only with -DGPX_POTF2_TRACE): shader-clock stamps at
// every barrier.  Measured per panel: dump 1650, diagonal factor 5750 (16 dependent column steps, each an LDS
// write -> read round trip + the reciprocal), TRSM 1800, update 4300 .. 960 cycles; the diagonal factor is 45 %
// of the 46 us.  Variants measured and rejected: gathering the column with ds_bpermute / v_readlane instead of
// LDS (7150 cycles per factor); a wave-specialised pipeline that overlaps the diagonal factor of panel p+1
// with the trailing update of panel p (three worker waves; 53 vs 49 us — the factor stays the critical path);
// and a "chain wave" version (wave 0 runs only diag(p) -> L(p+1,p) -> C(p+1,p+1) -> diag(p+1) out of LDS, two
// barriers per panel, bit-identical results): its chain costs 7750 cycles per panel as planned, but the three
// worker waves need 9700 - 12700 (each 16x16 tile update is 4 dependent MFMAs behind their own LDS operand
// reads, ~480 cycles per tile), so the chain wave waits for them and the block still takes ~50 us.  The next
// step for this kernel is software-pipelining the workers' operand reads, not more overlap.
#ifdef GPX_POTF2_TRACE
__device__ long long gpx_potf2_trace[64];
#define GPX_TRACE(slot)                                                     \
  do {                                                                      \
    if (threadIdx.x == 0) gpx_potf2_trace[(slot)] = (long long)clock64();   \
  } while (0)
#else
#define GPX_TRACE(slot) do { } while (0)
#endif

// PRE-UPDATE (Kpre > 0): the block first receives  A -= P P^T  with P = the 128 x Kpre strip `Ppre` (this block's rows of
// the panel columns that have just been solved) — the update the panel chain would otherwise apply to this diagonal
// tile with a GEMM launch of its own before the factorisation can start.  Same arithmetic as that launch: ascending
// k in MFMA groups of 4, acc - a b (the GEMM kernels run -(-acc + a b): rounding is symmetric), so not a bit changes.
template <bool BLK>
__global__ __launch_bounds__(256, 1) void potf2_tile_kernel(double* A, int64_t lda, double* Linv, int* info,
                                                            int info_base, int64_t a_bs, int64_t linv_bs,
                                                            const double* Ppre, int Kpre) {
  A += (int64_t)blockIdx.x * a_bs; // one workgroup per batch entry
  if (Ppre != nullptr) Ppre += (int64_t)blockIdx.x * a_bs; // the strip lives in the same matrix
  Linv += (int64_t)blockIdx.x * linv_bs;
  if (info != nullptr) info += blockIdx.x;
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Pbuf = lds;            // 8 tiles: column-p panel (raw -> L)
  double* Rrow = lds + 8 * TSZ;  // 8 tiles: residual row p (raw -> inverse row X(p, c))
  double* Dinv = lds + 16 * TSZ; // inverse of the diagonal tile
  double* col = lds + 17 * TSZ;  // 16 + 16 scratch doubles of the diagonal-tile factor
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int crow = lane >> 4, ccol = lane & 15; // accumulator layout: rows crow + 4 r, column ccol

  // zero the strictly-upper 16x16 tiles of both outputs (the diagonal tiles are written whole later)
  for (int idx = w; idx < 28; idx += 4) {
    int i, c;
    strict_tile(idx, i, c); // tile (c, i) is above the diagonal
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      A[(int64_t)(c * TS + crow + 4 * r) * lda + i * TS + ccol] = 0.0;
      Linv[(c * TS + crow + 4 * r) * PB + i * TS + ccol] = 0.0;
    }
  }

  pd4_t C[9], R[7];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    int i, j;
    lower_tile(4 * t + w, i, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[t][r] = A[(int64_t)(i * TS + crow + 4 * r) * lda + j * TS + ccol];
  }
#pragma unroll
  for (int u = 0; u < 7; ++u) R[u] = pd4_t{0.0, 0.0, 0.0, 0.0};
  int bad = 0;

  if (Ppre != nullptr && Kpre > 0) {
    // 16 columns of the strip at a time: 8 tiles (row tile i = rows 16 i ..) staged in LDS, double-buffered in
    // Pbuf / Rrow (both idle until the factorisation starts); every wave updates its own 9 lower tiles
    const int srow = tid >> 1, scol = (tid & 1) * 8; // 256 threads x 8 doubles = 128 rows x 16 columns
    double stage[8];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        const double2 v = *reinterpret_cast<const double2*>(Ppre + (int64_t)srow * lda + k0 + scol + q);
        stage[q] = v.x;
        stage[q + 1] = v.y;
      }
    };
    auto put = [&](double* buf) {
      double* T = buf + (srow >> 4) * TSZ + (srow & 15) * TLD + scol;
#pragma unroll
      for (int q = 0; q < 8; ++q) T[q] = stage[q];
    };
    fetch(0);
    put(Pbuf);
    __syncthreads();
    const int nchunk = Kpre / 16;
    for (int c = 0; c < nchunk; ++c) {
      double* cur = (c & 1) ? Rrow : Pbuf;
      double* nxt = (c & 1) ? Pbuf : Rrow;
      if (c + 1 < nchunk) fetch((c + 1) * 16);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        int i, j;
        lower_tile(4 * t + w, i, j);
        C[t] = mma_nt(C[t], cur + i * TSZ, cur + j * TSZ, lane, -1.0);
      }
      if (c + 1 < nchunk) put(nxt);
      __syncthreads();
    }
  }

  GPX_TRACE(0);
  for (int p = 0; p < 8; ++p) {
    // ---- A: dump column p of C and row p of R -----------------------------------------------
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      int i, j;
      lower_tile(4 * t + w, i, j);
      if (j == p) acc_to_lds(C[t], Pbuf + i * TSZ, lane);
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      int i, c;
      strict_tile(4 * u + w, i, c);
      if (i == p) acc_to_lds(R[u], Rrow + c * TSZ, lane);
    }
    __syncthreads();
    GPX_TRACE(1 + 4 * p);
    // ---- B: diagonal tile -----------------------------------------------------------------------
    if (w == 0) {
      if (BLK) diag16_blk(Pbuf + p * TSZ, Dinv, col, lane, bad, p * TS);
      else diag16(Pbuf + p * TSZ, Dinv, col, lane, bad, p * TS);
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * q + t;
        A[(int64_t)(p * TS + r) * lda + p * TS + i] = Pbuf[p * TSZ + r * TLD + i];
        Linv[(p * TS + r) * PB + p * TS + i] = Dinv[r * TLD + i];
      }
    }
    __syncthreads();
    GPX_TRACE(2 + 4 * p);
    // ---- C: panel TRSM and inverse row ----------------------------------------------------------
    for (int m = w; m < 7; m += 4) {
      if (m < 7 - p) { // L(i,p) = C(i,p) Linv_pp^T
        const int i = p + 1 + m;
        pd4_t x = mma_nt(pd4_t{0.0, 0.0, 0.0, 0.0}, Pbuf + i * TSZ, Dinv, lane, 1.0);
        acc_to_lds(x, Pbuf + i * TSZ, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(int64_t)(i * TS + crow + 4 * r) * lda + p * TS + ccol] = x[r];
      } else { // X(p,c) = Linv_pp R(p,c)
        const int c = m - (7 - p);
        pd4_t x = mma_nn(pd4_t{0.0, 0.0, 0.0, 0.0}, Dinv, Rrow + c * TSZ, lane, 1.0);
        acc_to_lds(x, Rrow + c * TSZ, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Linv[(p * TS + crow + 4 * r) * PB + c * TS + ccol] = x[r];
      }
    }
    __syncthreads();
    GPX_TRACE(3 + 4 * p);
    // ---- D: trailing updates --------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      int i, j;
      lower_tile(4 * t + w, i, j);
      if (j > p) C[t] = mma_nt(C[t], Pbuf + i * TSZ, Pbuf + j * TSZ, lane, -1.0);
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      int i, c;
      strict_tile(4 * u + w, i, c);
      if (i > p && c <= p) R[u] = mma_nn(R[u], Pbuf + i * TSZ, (c == p) ? Dinv : Rrow + c * TSZ, lane, -1.0);
    }
    __syncthreads();
    GPX_TRACE(4 + 4 * p);
  }
  if (tid == 0 && bad != 0 && info != nullptr) {
    if (*info == 0) *info = info_base + bad;
  }
}

} // namespace gpx

namespace gpx {
int launch_potf2_inv(gpx_ctx* ctx, double* dA, int64_t lda, double* dLinv, int* dInfo, int info_base,
                     int batch, int64_t a_bs, int64_t linv_bs, const double* dPre, int Kpre) {
  const int nb = batch > 1 ? batch : 1;
  // GPX_POTF2=column selects the column-by-column kernel (its > 64 KB of dynamic LDS is a per-device function
  // attribute: set once per context, i.e. on every device a process opens)
  constexpr unsigned ATTR_POTF2_COLUMN = 1u << 31;
  const bool use_tile = !ctx->potf2_column; // GPX_POTF2=column (gpx_init)
  if (!use_tile && !(ctx->func_attr_mask & ATTR_POTF2_COLUMN)) {
    GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTF2_LDS_BYTES));
    ctx->func_attr_mask |= ATTR_POTF2_COLUMN;
  }
  if (!use_tile && dPre != nullptr) return bad_arg(ctx, "the column-by-column potf2 kernel has no pre-update");
  // Few workgroups (the single-theta pipeline, small batches): run on the reserved CUs through `rstream`, fenced
  // by events into the stream the chain lives on, so the block factorisation has a CU to itself.
  hipStream_t chain = ctx->s;
  const bool reserved = ctx->rstream != nullptr && nb <= ctx->cu_reserved;
  if (reserved) {
    GPX_HIP(ctx, hipEventRecord(ctx->evR0, chain));
    GPX_HIP(ctx, hipStreamWaitEvent(ctx->rstream, ctx->evR0, 0));
    ctx->s = ctx->rstream;
  }
  {
    // algorithmic flops: factor n^3/3 + triangular inverse n^3/3
    ProfScope ps(ctx, GPX_PROF_POTF2, nb * 2.0 * PB * (double)PB * PB / 3.0);
    if (use_tile && ctx->potf2_diag_blocked)
      potf2_tile_kernel<true><<<nb, 256, POTF2_TILE_LDS, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, a_bs, linv_bs, dPre, Kpre);
    else if (use_tile)
      potf2_tile_kernel<false><<<nb, 256, POTF2_TILE_LDS, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, a_bs, linv_bs, dPre, Kpre);
    else
      potf2_inv_kernel<<<nb, 256, POTF2_LDS_BYTES, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, a_bs, linv_bs);
  }
  ctx->s = chain;
  GPX_HIP(ctx, hipGetLastError());
  if (reserved) {
    GPX_HIP(ctx, hipEventRecord(ctx->evR1, ctx->rstream));
    GPX_HIP(ctx, hipStreamWaitEvent(chain, ctx->evR1, 0));
  }
  return 0;
}
} // namespace gpx
