// potf2_tile.h — the 128 x 128 diagonal-block factor + inverse of potf2.hip (16 x 16 MFMA re-blocking; see that file)
// as a device function of ONE 256-thread workgroup with POTF2_TILE_LDS bytes of LDS.
#pragma once
#include "common.h"

namespace gpx {

constexpr int PB = 128; // diagonal block order
typedef double pd4_t __attribute__((ext_vector_type(4)));

constexpr int TS = 16;
constexpr int TLD = 17;
constexpr int TSZ = TS * TLD; // doubles per LDS tile
constexpr size_t POTF2_TILE_LDS = (size_t)(17 * TSZ + 64) * sizeof(double); // + scratch of the diagonal-tile factor

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
__device__ __forceinline__ pd4_t lds_to_acc(const double* T, int lane) {
  pd4_t a;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = T[((lane >> 4) + 4 * r) * TLD + (lane & 15)];
  return a;
}

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
// Lane (r = lane & 15, q = lane >> 4) owns S[r][4q .. 4q+3]; fused update rule as the column kernel of round 1:
//     S[r][i] -= S[r][j] * S[i][j] / d_j      for i > j and (r >= i  or  r <= j)
// (the strict upper triangle holds the forward-substitution residual of L X = I, transposed).
//
// Round 4: WHICH lanes take part in an update is a function of (j, t) and the lane number only — a compile-time 64-bit
// lane mask.  The updates are therefore issued as ONE v_fma_f64 each under `s_mov_b64 exec, <constant>` (scalar unit)
// instead of an unconditional fma + two v_cndmask_b32 per value behind lane masks that the compiler computed up front,
// kept in ~100 SGPRs, spilled to VGPR lanes and fetched back with v_readlane inside the loop: 35 vector-ALU
// instructions per column step became 16.  That matters beyond the instruction count: on gfx950 an fp64 MFMA occupies
// the vector ALU of its SIMD for 64 cycles, so next to two resident trailing-update waves (where this wave runs in the
// GEMM-bound part of a factorisation) EVERY vector instruction of this wave waits for an MFMA slot — the in-pipeline
// time of the diagonal-block kernel is proportional to the vector instructions on its chain, not to their latency
// (profiles/r04/potf2_phase_trace.json: diag16 2.4 us per tile alone, 10 us beside the GEMM).
// Same operands, same operations, same order per element: bit-identical to the round-2/3 form.
__host__ __device__ constexpr unsigned long long diag16_mask_on(int j, int t) {
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    const int r = l & 15, i = 4 * (l >> 4) + t;
    if (i > j && (r >= i || r <= j)) m |= 1ull << l;
  }
  return m;
}
__host__ __device__ constexpr unsigned long long diag16_mask_row(int j) { // lanes with r == j
  return (1ull << j) | (1ull << (16 + j)) | (1ull << (32 + j)) | (1ull << (48 + j));
}
__host__ __device__ constexpr unsigned long long diag16_mask_group(int q) { // lanes with lane >> 4 == q
  return 0xffffull << (16 * q);
}
// The mask is written into EXEC as two 32-bit IMMEDIATES of the instruction stream: handed over in SGPRs, the compiler
// hoists all ~100 of them out of the panel loop, spills them to VGPR lanes and fetches them back with v_readlane — a
// vector instruction again.
// e = fma(-cr, cc, e) on the lanes of the mask M, the others untouched
template <unsigned long long M>
__device__ __forceinline__ void masked_fnma(double& e, double cr, double cc) {
  asm volatile("s_mov_b32 exec_lo, %3\n\ts_mov_b32 exec_hi, %4\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_mov_b64 exec, -1"
               : "+v"(e)
               : "v"(cr), "v"(cc), "i"((int)(unsigned)(M & 0xffffffffull)), "i"((int)(unsigned)(M >> 32)));
}
// dst = src on the lanes of M
template <unsigned long long M>
__device__ __forceinline__ void masked_mov(double& dst, double src) {
  asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %3\n\tv_mov_b64 %0, %1\n\ts_mov_b64 exec, -1"
               : "+v"(dst)
               : "v"(src), "i"((int)(unsigned)(M & 0xffffffffull)), "i"((int)(unsigned)(M >> 32)));
}
template <unsigned long long M>
__device__ __forceinline__ void masked_one(double& dst) {
  asm volatile("s_mov_b32 exec_lo, %1\n\ts_mov_b32 exec_hi, %2\n\tv_mov_b64 %0, 1.0\n\ts_mov_b64 exec, -1"
               : "+v"(dst)
               : "i"((int)(unsigned)(M & 0xffffffffull)), "i"((int)(unsigned)(M >> 32)));
}

template <int J>
__device__ __forceinline__ void diag16_step(double (&e)[4], double (&mypiv)[4], double* col, int lane, int& bad, int base) {
  const int r = lane & 15, q = lane >> 4;
  double* cb = col + (J & 1) * 16;
  if (q == (J >> 2)) cb[r] = e[J & 3];
  // one wave: the LDS queue is in order, so the reads below see the write; the asm only stops the
  // compiler from caching / reordering the accesses (no volatile: the six reads share one wait)
  asm volatile("" ::: "memory");
  const double dj = cb[J];
  double cr = cb[r];
  const double c0 = cb[4 * q + 0], c1 = cb[4 * q + 1], c2 = cb[4 * q + 2], c3 = cb[4 * q + 3];
  asm volatile("" ::: "memory");
  constexpr unsigned long long mrow = diag16_mask_row(J), mgrp = diag16_mask_group(J >> 2);
  masked_one<mrow>(cr); // row r == j of the X half
  double ip2 = __builtin_amdgcn_rcp(dj);
  ip2 = fma(fma(-dj, ip2, 1.0), ip2, ip2);
  ip2 = fma(fma(-dj, ip2, 1.0), ip2, ip2);
  masked_mov<mgrp>(mypiv[J & 3], dj);
  if (!(dj > 0.0) && bad == 0) bad = base + J + 1;
  // (the masks must be constant expressions: evaluated at run time they are 64-iteration scalar loops)
  constexpr unsigned long long m0 = diag16_mask_on(J, 0), m1 = diag16_mask_on(J, 1), m2 = diag16_mask_on(J, 2),
                               m3 = diag16_mask_on(J, 3);
  if constexpr (m0 != 0) masked_fnma<m0>(e[0], cr, c0 * ip2);
  if constexpr (m1 != 0) masked_fnma<m1>(e[1], cr, c1 * ip2);
  if constexpr (m2 != 0) masked_fnma<m2>(e[2], cr, c2 * ip2);
  if constexpr (m3 != 0) masked_fnma<m3>(e[3], cr, c3 * ip2);
}

template <int J>
__device__ __forceinline__ void diag16_steps(double (&e)[4], double (&mypiv)[4], double* col, int lane, int& bad, int base) {
  diag16_step<J>(e, mypiv, col, lane, bad, base);
  if constexpr (J < 15) diag16_steps<J + 1>(e, mypiv, col, lane, bad, base);
}

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
  diag16_steps<0>(e, mypiv, col, lane, bad, base);
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

// The round-2/3 form of diag16 (an unconditional fma + selects behind lane masks the compiler keeps in SGPRs): what
// potf2_tile_kernel — the REFERENCE of the bit-identity tests — keeps running, so that the tests compare the round-4
// form above (slim and chain kernels) against independently generated code, operation for operation.
__device__ __forceinline__ void diag16_select(double* D, double* Dinv, double* col /* 2 x 16 */, int lane, int& bad,
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

__device__ __forceinline__ void potf2_tile_body(double* A, int64_t lda, double* Linv, int* info, int info_base, double* lds) {
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
      diag16_select(Pbuf + p * TSZ, Dinv, col, lane, bad, p * TS);
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
