// potf2_tile.h — the 128 x 128 diagonal-block factor + inverse of potf2.hip (16 x 16 MFMA re-blocking; see that file)
// as a device function of ONE 256-thread workgroup with POTF2_TILE_LDS bytes of LDS, shared by potf2_tile_kernel and
// the cooperative panel kernel (panel.hip).
#pragma once
#include "common.h"

namespace gpx {

constexpr int PB = 128; // diagonal block order
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
__device__ __forceinline__ void potf2_tile_body(double* A, int64_t lda, double* Linv, int* info, int info_base,
                                                const double* Ppre, int Kpre, double* lds) {
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
