// (round 5: left the product library — VERDICT r4 weak #11: ONE bit-identity reference, potf2_tile.h, stays compiled in.
// Last commit that built it into libgpx.so: 1a89af6.  The kernel wrapper and the launch branch that used it are quoted at the
// end of this file.)
// potf2_chain.h — the 128 x 128 diagonal-block factor + inverse (potf2.hip) as a WAVE-SPECIALISED kernel body:
// wave 0 runs nothing but the dependent chain  diag16(p) -> L(p+1,p) -> D(p+1) = C(p+1,p+1) - L(p+1,p) L(p+1,p)^T ->
// diag16(p+1)  out of LDS, waves 1 - 3 own all 36 Cholesky and 28 inverse-residual 16 x 16 tiles as MFMA accumulators
// and do every other TRSM and update beside it.  Round 3's default; since round 4 GPX_POTF2=chain — the default is the
// same structure with memory-resident tiles (potf2_slim.h), which fits beside two resident trailing-update workgroups;
// this one (344 VGPRs, 46 KB) needs a drained CU and stays as a reference of the bit-identity test.
//
// Why: potf2 is the serial kernel of the blocked Cholesky (one launch per 128 columns, on the critical path of the whole
// panel chain: gpax/models/gp.py:160-164 -> jnp.linalg.cholesky).  In potf2_tile_body all four waves step through four
// phases per 16-column panel (dump, diagonal tile, TRSM, update) and three of them idle while wave 0 factors the diagonal
// tile: 8 x (1650 + 5750 + 1800 + ~2600) cycles = 46 us.  The diagonal tile's 16 dependent column steps (5750 cycles) are
// inherent; everything else can run beside them.  Here a panel costs the chain 5750 + ~900 cycles and the workers
// (whose tile indices are compile-time constants: their MFMAs interleave over tiles instead of running as 4-long
// dependent chains behind per-tile branches) stay under that.
//
// Per panel p, two workgroup barriers:
//   chain:   diag16(p): Dg -> L(p,p), Dinv[p & 1]            | B1 | T: L(p+1,p) = S1 Dinv^T -> Lcol[p+1]           | B2 |
//                                                                   U: Dg = S2 - L(p+1,p) L(p+1,p)^T
//   workers: updates of panel p-1, then S1 <- C(p+1,p),      | B1 | own column-p tiles -> L(i,p) (i > p+1) -> Lcol  | B2 |
//            S2 <- C(p+1,p+1) for the chain                         own row-p residual tiles -> X(p,c) -> Xrow
// LDS: Lcol[8] | Xrow[8] | Dinv[2] (by panel parity: the workers still read Dinv[p & 1] as X(p,p) while the chain
// writes the next one) | Dg | S1 | S2 = 21 tiles of 16 x 17 doubles = 45.7 KB.
// Measured (profiles/r03/potf2_chain_time.txt): 49.5 -> 33.5 us per launch.  Of the ~28 us inside the kernel ~21 are the eight
// diagonal tiles.  A diag16 with the pivot one step ahead (the owner of S[j+1][j+1] publishes it next to column j, every
// lane forms d_{j+1} = fma(-c, c ip2_j, s) itself and refines its reciprocal beside the updates, taking the reciprocal
// out of the per-column chain; bit-identical) was built and measured SLOWER — 40.3 us: two more LDS reads and seven more
// fp64 operations per column step on every lane cost more than the five dependent ones they hide — and removed.
// Arithmetic: tile for tile the MFMA sequences of potf2_tile_body (k ascending, the same operands, the same signs), so
// L, L^-1 and the pivots are bit-identical to it (tests/test_gpu_edges.py).
#pragma once
#include "../../gpax_amd/csrc/potf2_tile.h"

namespace gpx {

constexpr int PC_LCOL = 0, PC_XROW = 8, PC_DINV = 16, PC_DG = 18, PC_S1 = 19, PC_S2 = 20, PC_TILES = 21;
constexpr size_t POTF2_CHAIN_LDS = (size_t)(PC_TILES * TSZ + 64) * sizeof(double); // + scratch of the diagonal-tile factor

__host__ __device__ constexpr int lt_i(int idx) { // idx = i (i + 1) / 2 + j
  return (idx >= 28) ? 7 : (idx >= 21) ? 6 : (idx >= 15) ? 5 : (idx >= 10) ? 4 : (idx >= 6) ? 3 : (idx >= 3) ? 2 : (idx >= 1) ? 1 : 0;
}
__host__ __device__ constexpr int lt_j(int idx) { return idx - lt_i(idx) * (lt_i(idx) + 1) / 2; }
__host__ __device__ constexpr int st_i(int idx) { // idx = i (i - 1) / 2 + c, i > c
  return (idx >= 21) ? 7 : (idx >= 15) ? 6 : (idx >= 10) ? 5 : (idx >= 6) ? 4 : (idx >= 3) ? 3 : (idx >= 1) ? 2 : 1;
}
__host__ __device__ constexpr int st_c(int idx) { return idx - st_i(idx) * (st_i(idx) - 1) / 2; }

constexpr int PC_NC = 12, PC_NR = 10; // tiles per worker: C idx = 3 t + W - 1 (t < 12), R idx = 3 u + W - 1 (< 28)


// ---- workers -----------------------------------------------------------------------------------------------------------
// window B1(P) .. B2(P): this worker's column-P tiles become L(i,P) (i > P + 1; the chain makes i = P + 1), its row-P
// residual tiles become X(P,c) = Dinv_P R(P,c)
template <int W, int P>
__device__ __forceinline__ void worker_solve(pd4_t (&C)[PC_NC], pd4_t (&R)[PC_NR], double* lds, double* A, int64_t lda,
                                             double* Linv, int lane) {
  double* Lcol = lds + PC_LCOL * TSZ;
  double* Xrow = lds + PC_XROW * TSZ;
  const double* Dinv = lds + (PC_DINV + (P & 1)) * TSZ;
  const int fr = lane & 15, fk = lane >> 4;
  const int crow = lane >> 4, ccol = lane & 15;
#pragma unroll
  for (int t = 0; t < PC_NC; ++t) {
    const int idx = 3 * t + W - 1, i = lt_i(idx), j = lt_j(idx);
    if (j == P && i > P + 1) acc_to_lds(C[t], Lcol + i * TSZ, lane);
  }
#pragma unroll
  for (int u = 0; u < PC_NR; ++u) {
    const int idx = 3 * u + W - 1;
    if (idx < 28 && st_i(idx) == P) acc_to_lds(R[u], Xrow + st_c(idx) * TSZ, lane);
  }
  pd4_t tl[PC_NC], tx[PC_NR];
#pragma unroll
  for (int t = 0; t < PC_NC; ++t) tl[t] = pd4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < PC_NR; ++u) tx[u] = pd4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
    for (int t = 0; t < PC_NC; ++t) {
      const int idx = 3 * t + W - 1, i = lt_i(idx), j = lt_j(idx);
      if (j == P && i > P + 1) { // L(i,P) = C(i,P) Dinv^T
        const double a = Lcol[i * TSZ + fr * TLD + fk + 4 * kk];
        const double b = Dinv[fr * TLD + fk + 4 * kk];
        tl[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, tl[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < PC_NR; ++u) {
      const int idx = 3 * u + W - 1;
      if (idx < 28 && st_i(idx) == P) { // X(P,c) = Dinv R(P,c)
        const double a = Dinv[fr * TLD + fk + 4 * kk];
        const double b = Xrow[st_c(idx) * TSZ + (fk + 4 * kk) * TLD + fr];
        tx[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, tx[u], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < PC_NC; ++t) {
    const int idx = 3 * t + W - 1, i = lt_i(idx), j = lt_j(idx);
    if (j == P && i > P + 1) {
      acc_to_lds(tl[t], Lcol + i * TSZ, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) A[(int64_t)(i * TS + crow + 4 * r) * lda + P * TS + ccol] = tl[t][r];
    }
  }
#pragma unroll
  for (int u = 0; u < PC_NR; ++u) {
    const int idx = 3 * u + W - 1;
    if (idx < 28 && st_i(idx) == P) {
      const int c = st_c(idx);
      acc_to_lds(tx[u], Xrow + c * TSZ, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) Linv[(P * TS + crow + 4 * r) * PB + c * TS + ccol] = tx[u][r];
    }
  }
}

// window B2(P) .. B1(P+1): C(i,j) -= L(i,P) L(j,P)^T (i >= j > P, the next diagonal tile excepted: the chain made it),
// R(i,c) -= L(i,P) X(P,c) (i > P, c <= P, X(P,P) = Dinv_P); then the two tiles the chain reads next go to S1 / S2
template <int W, int P>
__device__ __forceinline__ void worker_update(pd4_t (&C)[PC_NC], pd4_t (&R)[PC_NR], double* lds, int lane) {
  const double* Lcol = lds + PC_LCOL * TSZ;
  const double* Xrow = lds + PC_XROW * TSZ;
  const double* Dinv = lds + (PC_DINV + (P & 1)) * TSZ;
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
    for (int t = 0; t < PC_NC; ++t) {
      const int idx = 3 * t + W - 1, i = lt_i(idx), j = lt_j(idx);
      if (j > P && !(i == P + 1 && j == P + 1)) {
        const double a = -1.0 * Lcol[i * TSZ + fr * TLD + fk + 4 * kk];
        const double b = Lcol[j * TSZ + fr * TLD + fk + 4 * kk];
        C[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < PC_NR; ++u) {
      const int idx = 3 * u + W - 1;
      if (idx < 28 && st_i(idx) > P && st_c(idx) <= P) {
        const int i = st_i(idx), c = st_c(idx);
        const double a = -1.0 * Lcol[i * TSZ + fr * TLD + fk + 4 * kk];
        const double* X = (c == P) ? Dinv : Xrow + c * TSZ;
        const double b = X[(fk + 4 * kk) * TLD + fr];
        R[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, R[u], 0, 0, 0);
      }
    }
  }
  if (P + 2 <= 7) {
#pragma unroll
    for (int t = 0; t < PC_NC; ++t) {
      const int idx = 3 * t + W - 1, i = lt_i(idx), j = lt_j(idx);
      if (i == P + 2 && j == P + 1) acc_to_lds(C[t], lds + PC_S1 * TSZ, lane);
      if (i == P + 2 && j == P + 2) acc_to_lds(C[t], lds + PC_S2 * TSZ, lane);
    }
  }
}

template <int W, int P>
__device__ __forceinline__ void worker_panels(pd4_t (&C)[PC_NC], pd4_t (&R)[PC_NR], double* lds, double* A, int64_t lda,
                                              double* Linv, int lane) {
  __syncthreads(); // B1(P): Dinv_P is there
  worker_solve<W, P>(C, R, lds, A, lda, Linv, lane);
  __syncthreads(); // B2(P): every L(i,P), X(P,c) is there
  if constexpr (P < 7) {
    worker_update<W, P>(C, R, lds, lane);
    worker_panels<W, P + 1>(C, R, lds, A, lda, Linv, lane);
  }
}

template <int W>
__device__ __forceinline__ void potf2_worker(double* A, int64_t lda, double* Linv, double* lds, int lane) {
  const int crow = lane >> 4, ccol = lane & 15;
  // zero the strictly-upper 16 x 16 tiles of both outputs (the diagonal tiles are written whole by the chain)
#pragma unroll
  for (int u = 0; u < PC_NR; ++u) {
    const int idx = 3 * u + W - 1;
    if (idx < 28) {
      const int i = st_i(idx), c = st_c(idx); // tile (c, i) is above the diagonal
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        A[(int64_t)(c * TS + crow + 4 * r) * lda + i * TS + ccol] = 0.0;
        Linv[(c * TS + crow + 4 * r) * PB + i * TS + ccol] = 0.0;
      }
    }
  }
  pd4_t C[PC_NC], R[PC_NR];
#pragma unroll
  for (int t = 0; t < PC_NC; ++t) {
    const int idx = 3 * t + W - 1, i = lt_i(idx), j = lt_j(idx);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[t][r] = A[(int64_t)(i * TS + crow + 4 * r) * lda + j * TS + ccol];
  }
#pragma unroll
  for (int u = 0; u < PC_NR; ++u) R[u] = pd4_t{0.0, 0.0, 0.0, 0.0};
  worker_panels<W, 0>(C, R, lds, A, lda, Linv, lane);
}

// ---- the chain wave ------------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ void chain_panels(double* A, int64_t lda, double* Linv, double* lds, int lane, int& bad) {
  double* Dg = lds + PC_DG * TSZ;
  double* Dinv = lds + (PC_DINV + (P & 1)) * TSZ;
  double* col = lds + PC_TILES * TSZ;
  diag16(Dg, Dinv, col, lane, bad, P * TS);
  {
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = 4 * q + t;
      A[(int64_t)(P * TS + r) * lda + P * TS + i] = Dg[r * TLD + i];
      Linv[(P * TS + r) * PB + P * TS + i] = Dinv[r * TLD + i];
    }
  }
  __syncthreads(); // B1(P)
  if constexpr (P < 7) {
    const int crow = lane >> 4, ccol = lane & 15;
    double* Lnext = lds + (PC_LCOL + P + 1) * TSZ;
    // T: L(P+1,P) = C(P+1,P) Dinv^T
    const pd4_t x = mma_nt(pd4_t{0.0, 0.0, 0.0, 0.0}, lds + PC_S1 * TSZ, Dinv, lane, 1.0);
    acc_to_lds(x, Lnext, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) A[(int64_t)((P + 1) * TS + crow + 4 * r) * lda + P * TS + ccol] = x[r];
    // U: the next diagonal tile
    pd4_t d = lds_to_acc(lds + PC_S2 * TSZ, lane);
    d = mma_nt(d, Lnext, Lnext, lane, -1.0);
    acc_to_lds(d, Dg, lane);
  }
  __syncthreads(); // B2(P)
  if constexpr (P < 7) chain_panels<P + 1>(A, lda, Linv, lds, lane, bad);
}

__device__ __forceinline__ void potf2_chain_wave(double* A, int64_t lda, double* Linv, int* info, int info_base, double* lds,
                                                 int lane) {
  const int crow = lane >> 4, ccol = lane & 15;
  // the three tiles the chain starts from: C(0,0) -> Dg, C(1,0) -> S1, C(1,1) -> S2 (acc layout in, LDS tile out)
  {
    pd4_t t00, t10, t11;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      t00[r] = A[(int64_t)(crow + 4 * r) * lda + ccol];
      t10[r] = A[(int64_t)(TS + crow + 4 * r) * lda + ccol];
      t11[r] = A[(int64_t)(TS + crow + 4 * r) * lda + TS + ccol];
    }
    acc_to_lds(t00, lds + PC_DG * TSZ, lane);
    acc_to_lds(t10, lds + PC_S1 * TSZ, lane);
    acc_to_lds(t11, lds + PC_S2 * TSZ, lane);
  }
  int bad = 0;
  chain_panels<0>(A, lda, Linv, lds, lane, bad);
  if (lane == 0 && bad != 0 && info != nullptr) {
    if (*info == 0) *info = info_base + bad;
  }
}

// 256 threads: wave 0 = chain, waves 1 - 3 = workers.  Every wave passes the same 16 barriers.
__device__ __forceinline__ void potf2_chain_body(double* A, int64_t lda, double* Linv, int* info, int info_base, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (w == 0) potf2_chain_wave(A, lda, Linv, info, info_base, lds, lane);
  else if (w == 1) potf2_worker<1>(A, lda, Linv, lds, lane);
  else if (w == 2) potf2_worker<2>(A, lda, Linv, lds, lane);
  else potf2_worker<3>(A, lda, Linv, lds, lane);
}

} // namespace gpx

// ---- what potf2.hip held for this body until round 4 -------------------------------------------------------------------
// __global__ __launch_bounds__(256, 1) void potf2_chain_kernel(double* A, int64_t lda, double* Linv, int* info, int info_base,
//                                                              int64_t a_bs, int64_t linv_bs) {
//   A += (int64_t)blockIdx.x * a_bs; Linv += (int64_t)blockIdx.x * linv_bs; if (info != nullptr) info += blockIdx.x;
//   __builtin_amdgcn_s_setprio(3);
//   extern __shared__ __attribute__((aligned(16))) double lds[];
//   potf2_chain_body(A, lda, Linv, info, info_base, lds);
// }
// launch: potf2_chain_kernel<<<nb, 256, POTF2_CHAIN_LDS, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, a_bs, linv_bs);
