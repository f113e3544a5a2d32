// Round-2 A/B harness for the fp64 MFMA NT GEMM main loop (not part of libgpx).
//   C = beta C + alpha A B^T, A (M x K), B (N x K) row-major, k contiguous.
// Variants:
//   base      : the round-1 loop (register staging, LDS rows padded to 17 doubles, 128x128, 4 waves)
//   glds<..>  : LDS-direct staging (global_load_lds_dwordx4), unpadded 128-B LDS rows with an XOR
//               swizzle applied on the SOURCE address (chunk' = chunk ^ ((row >> 1) & 7)), fragments
//               read with one conflict-free ds_read_b64 each.  WR x WC waves of 64x64 each.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_r2.hip -o gemm_r2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---------------------------------------------------------------------------------------------
template <int LDT>
__global__ __launch_bounds__(256, 2) void gemm_base(const double* A, long lda, const double* B, long ldb, double* C,
                                                    long ldc, int K, double alpha, double beta) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int BK = 16;
  constexpr int TD = 128 * LDT;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fk = lane >> 4;
  constexpr int TPR = BK / 2, RPP = 256 / TPR, NP = 128 / RPP;
  const int lr = tid / TPR, lc = (tid % TPR) * 2;
  const double* Ap = A + ((long)by * 128 + lr) * lda + lc;
  const double* Bp = B + ((long)bx * 128 + lr) * ldb + lc;
  double* sA0 = smem; double* sB0 = smem + TD; double* sA1 = smem + 2 * TD; double* sB1 = smem + 3 * TD;
  d4_t acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = d4_t{0, 0, 0, 0};
  const int nk = K / BK;
  double2 ra[NP], rb[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    ra[i] = *(const double2*)(Ap + (long)i * RPP * lda);
    rb[i] = *(const double2*)(Bp + (long)i * RPP * ldb);
  }
  const int st = lr * LDT + lc;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    sA0[st + i * RPP * LDT] = ra[i].x; sA0[st + i * RPP * LDT + 1] = ra[i].y;
    sB0[st + i * RPP * LDT] = rb[i].x; sB0[st + i * RPP * LDT + 1] = rb[i].y;
  }
  __syncthreads();
  const int aoff = (wr * 64 + fr) * LDT + fk, boff = (wc * 64 + fr) * LDT + fk;
  for (int kt = 0; kt < nk; ++kt) {
    const double* cA = (kt & 1) ? sA1 : sA0;
    const double* cB = (kt & 1) ? sB1 : sB0;
    double* nA = (kt & 1) ? sA0 : sA1;
    double* nB = (kt & 1) ? sB0 : sB1;
    const int koff = ((kt + 1 < nk) ? kt + 1 : kt) * BK;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      ra[i] = *(const double2*)(Ap + (long)i * RPP * lda + koff);
      rb[i] = *(const double2*)(Bp + (long)i * RPP * ldb + koff);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = cA[aoff + m * 16 * LDT + kk * 4];
#pragma unroll
      for (int n = 0; n < 4; ++n) bf[n] = cB[boff + n * 16 * LDT + kk * 4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      nA[st + i * RPP * LDT] = ra[i].x; nA[st + i * RPP * LDT + 1] = ra[i].y;
      nB[st + i * RPP * LDT] = rb[i].x; nB[st + i * RPP * LDT + 1] = rb[i].y;
    }
    __syncthreads();
  }
  double* Cw = C + ((long)by * 128 + wr * 64 + fk) * ldc + (long)bx * 128 + wc * 64 + fr;
  if (beta != 0.0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double cv[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) cv[r][n] = Cw[(long)(m * 16 + 4 * r) * ldc + n * 16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = fma(beta, cv[r][n], alpha * acc[m][n][r]);
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = alpha * acc[m][n][r];
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-direct variant.  Tile (64 WR) x (64 WC), BK = 16, NBUF LDS buffers of (BM + BN) rows x 128 B.
// MODE 0: plain (compiler schedule)   1: s_setprio(1) around the MFMA block
//      2: fragments of kk + 1 are read before the MFMAs of kk (register double buffer)
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int WR, int WC, int NBUF, int MODE>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC == 4) ? 2 : 2) void gemm_glds(const double* A, long lda,
                                                                                  const double* B, long ldb,
                                                                                  double* C, long ldc, int K,
                                                                                  double alpha, double beta) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int BK = 16, NW = WR * WC, BM = 64 * WR, BN = 64 * WC;
  constexpr int ROWS = BM + BN;          // rows staged per k-tile (A rows then B rows)
  constexpr int TD = ROWS * BK;          // doubles per buffer
  constexpr int GROUPS = ROWS / 8;       // 8-row groups (one wave-instruction = 1 KiB = 8 rows)
  constexpr int GPW = GROUPS / NW;       // groups per wave
  static_assert(GROUPS % NW == 0, "groups must divide");
  const int bx = blockIdx.x, by = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC, fr = lane & 15, fk = lane >> 4;
  // staging map: group g = wave * GPW + j covers rows [8 g, 8 g + 8); lane -> row 8 g + (lane >> 3), LDS chunk' = lane & 7
  const double* src[GPW];
#pragma unroll
  for (int j = 0; j < GPW; ++j) {
    const int row = (wave * GPW + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    src[j] = (row < BM) ? A + ((long)by * BM + row) * lda + chunk * 2
                        : B + ((long)bx * BN + (row - BM)) * ldb + chunk * 2;
  }
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
      double* dst = smem + buf * TD + (wave * GPW + j) * 8 * BK; // wave-uniform base; HW adds lane * 16 B
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kt * BK), (lptr_t)dst, 16, 0, 0);
    }
  };
  d4_t acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = d4_t{0, 0, 0, 0};
  const int nk = K / BK;
  double* Cw0 = C + ((long)by * BM + wr * 64 + fk) * ldc + (long)bx * BN + wc * 64 + fr;
  if (MODE == 6) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n][r] = -Cw0[(long)(m * 16 + 4 * r) * ldc + n * 16];
  }
  // fragment offsets (doubles) inside a buffer: row * 16 + ((chunk ^ swz(row)) * 2) + (k & 1)
  int aoff[4], boff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int ra = wr * 64 + m * 16 + fr, rb = BM + wc * 64 + m * 16 + fr;
    aoff[m] = ra * BK + (((fk >> 1) ^ ((ra >> 1) & 7)) * 2) + (fk & 1);
    boff[m] = rb * BK + (((fk >> 1) ^ ((rb >> 1) & 7)) * 2) + (fk & 1);
  }
  // chunk for kk: (2 kk + (fk >> 1)) ^ s = ((fk >> 1) ^ s) ^ (2 kk)  (2 kk only touches bits 1..2) -> offset ^ (4 kk) in doubles
  issue(0, 0);
  if (NBUF == 3 && nk > 1) issue(1, 1);
  if (NBUF == 3) {
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt % NBUF;
    if (NBUF == 2) {
      if (kt + 1 < nk && MODE != 3 && MODE != 5) issue(kt + 1, (kt + 1) & 1);
    } else {
      if (kt + 2 < nk) issue(kt + 2, (kt + 2) % 3);
    }
    const double* cb = smem + cur * TD;
    if (MODE == 1) __builtin_amdgcn_s_setprio(1);
    if (MODE == 2) {
      double af[2][4], bf[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { af[0][m] = cb[aoff[m]]; bf[0][m] = cb[boff[m]]; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            af[(kk + 1) & 1][m] = cb[aoff[m] ^ (4 * (kk + 1))];
            bf[(kk + 1) & 1][m] = cb[boff[m] ^ (4 * (kk + 1))];
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk & 1][m], bf[kk & 1][n], acc[m][n], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double af[4], bf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { af[m] = cb[aoff[m] ^ (4 * kk)]; bf[m] = cb[boff[m] ^ (4 * kk)]; }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
      }
    }
    if (MODE == 1) __builtin_amdgcn_s_setprio(0);
    if (NBUF == 2) {
      if (MODE != 3 && MODE != 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    } else {
      // tile kt + 1 must have landed (its loads were issued one iteration ago); tile kt + 2 may stay in flight
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  double* Cw = C + ((long)by * BM + wr * 64 + fk) * ldc + (long)bx * BN + wc * 64 + fr;
  if (MODE == 6) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = -acc[m][n][r];
  } else if (beta != 0.0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double cv[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) cv[r][n] = Cw[(long)(m * 16 + 4 * r) * ldc + n * 16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = fma(beta, cv[r][n], alpha * acc[m][n][r]);
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = alpha * acc[m][n][r];
  }
}


// ---------------------------------------------------------------------------------------------
// buffer_load ... lds variant: 128x128, 4 waves, BK = 16, 2 LDS buffers.  Per-lane 32-bit voffsets are
// loop-invariant, the k advance is an SGPR soffset, the LDS destination is M0 (SALU only).
// MODE 7: loads bunched at the top of the k-step   8: two loads per kk block (source order)
//      9: as 8, pinned with sched_group_barrier     10: as 9 + odd workgroups start late (diagnostic)
//      8/9: 4 + 4 loads in kk = 0, 1;  11: 3 + 3 + 2 in kk = 0, 1, 2 (pinned)
// CACC: accumulators start from -C and the epilogue stores -acc (alpha = -1, beta = 1)
// tile order experiment: ORDER_SKEW s > 0 maps workgroup (x, y) of the 2-D grid to tile column (x + s y) mod gridDim.x, so
// that XCD w (workgroup id % 8; gridDim.x = 64 here) no longer sees the same tile columns in every tile row
__constant__ int ORDER_SKEW;
template <int MODE, bool CACC>
__global__ __launch_bounds__(256, 2) void gemm_bl(const double* A, long lda, const double* B, long ldb, double* C,
                                                  long ldc, int K, double alpha, double beta) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int BK = 16, BM = 128, BN = 128, ROWS = BM + BN, TD = ROWS * BK, GPW = 8;
  const int by = blockIdx.y;
  const int bx = ORDER_SKEW > 0 ? (int)((blockIdx.x + ORDER_SKEW * blockIdx.y) % gridDim.x) : (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fk = lane >> 4;
  if (MODE == 10 && ((bx + by) & 1)) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 5000) __builtin_amdgcn_s_sleep(32); // ~50 us at 100 MHz
  }
  // waves 0,1 stage the A rows, waves 2,3 the B rows: 64 rows = 8 groups of 8 rows each
  const bool isA = wave < 2;
  const double* base = isA ? A + (long)by * BM * lda : B + (long)bx * BN * ldb;
  const long ldx = isA ? lda : ldb;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  int voff[GPW];
#pragma unroll
  for (int j = 0; j < GPW; ++j) {
    const int row = (wave & 1) * 64 + j * 8 + (lane >> 3); // row inside this operand's 128-row tile
    const int lrow = (isA ? 0 : BM) + row;                 // row inside the LDS buffer
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);
    voff[j] = (int)((row * ldx + chunk * 2) * 8);
  }
  const int lds_wave = ((isA ? 0 : BM) + (wave & 1) * 64) * BK; // doubles
  d4_t acc[4][4];
  double* Cw = C + ((long)by * BM + wr * 64 + fk) * ldc + (long)bx * BN + wc * 64 + fr;
  if (CACC) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n][r] = -Cw[(long)(m * 16 + 4 * r) * ldc + n * 16];
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = d4_t{0, 0, 0, 0};
  }
  const int nk = K / BK;
  int aoff[4], boff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int ra = wr * 64 + m * 16 + fr, rb = BM + wc * 64 + m * 16 + fr;
    aoff[m] = ra * BK + (((fk >> 1) ^ ((ra >> 1) & 7)) * 2) + (fk & 1);
    boff[m] = rb * BK + (((fk >> 1) ^ ((rb >> 1) & 7)) * 2) + (fk & 1);
  }
#define GLOAD(j, buf, soff)                                                                                      \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (buf) * TD + lds_wave + (j) * 8 * BK), 16, voff[j], \
                                           (soff), 0, 0)
#pragma unroll
  for (int j = 0; j < GPW; ++j) GLOAD(j, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const int soff = ((kt + 1 < nk) ? kt + 1 : kt) * BK * 8; // the last step re-reads its own tile (unused)
    const double* cb = smem + cur * TD;
    if (MODE == 7) {
#pragma unroll
      for (int j = 0; j < GPW; ++j) GLOAD(j, nxt, soff);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (MODE == 12) {
          typedef const volatile double __attribute__((address_space(3)))* lcvd_t;
          af[m] = *((lcvd_t)cb + (aoff[m] ^ (4 * kk)));
          bf[m] = *((lcvd_t)cb + (boff[m] ^ (4 * kk)));
        } else {
          af[m] = cb[aoff[m] ^ (4 * kk)];
          bf[m] = cb[boff[m] ^ (4 * kk)];
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
        if (MODE >= 8 && MODE < 11 && kk < 2) GLOAD(4 * kk + m, nxt, soff);
        if (MODE >= 11 && kk < 2 && m < 3) GLOAD(3 * kk + m, nxt, soff);
        if (MODE >= 11 && kk == 2 && m < 2) GLOAD(6 + m, nxt, soff);
      }
      if ((MODE == 9 || MODE == 10) && kk < 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      }
      if (MODE >= 11 && kk < 3) {
#pragma unroll
        for (int q = 0; q < (kk < 2 ? 3 : 2); ++q) {
          __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
        if (kk < 2) __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
        else __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef GLOAD
  if (CACC) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = -acc[m][n][r];
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = alpha * acc[m][n][r];
  }
}

// ---------------------------------------------------------------------------------------------
// 8-WAVE variant of gemm_bl: the same 128x128 tile, BK = 16, 2 LDS buffers (64 KB), but 512 threads = 2 x 4 waves of
// 64 x 32 each (4 x 2 accumulators = 64 VGPRs): 2 workgroups / CU give FOUR waves per SIMD instead of two, so a wave
// that waits (barrier, LDS, the vmcnt before the barrier) is covered by three others.  Costs 6 fragment reads per
// 8 MFMAs instead of 8 per 16.  Each wave stages 4 x 1 KiB per k-step (waves 0-3: A rows, 4-7: B rows), one load
// behind the MFMAs of each kk block.
template <bool CACC, int OCC>
__global__ __launch_bounds__(512, OCC) void gemm_bl8(const double* A, long lda, const double* B, long ldb, double* C,
                                                     long ldc, int K, double alpha, double beta) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int BK = 16, BM = 128, BN = 128, ROWS = BM + BN, TD = ROWS * BK, GPW = 4;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fk = lane >> 4;
  const bool isA = wave < 4;
  const double* base = isA ? A + (long)by * BM * lda : B + (long)bx * BN * ldb;
  const long ldx = isA ? lda : ldb;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  int voff[GPW];
#pragma unroll
  for (int j = 0; j < GPW; ++j) {
    const int row = (wave & 3) * 32 + j * 8 + (lane >> 3);
    const int lrow = (isA ? 0 : BM) + row;
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);
    voff[j] = (int)((row * ldx + chunk * 2) * 8);
  }
  const int lds_wave = ((isA ? 0 : BM) + (wave & 3) * 32) * BK;
  d4_t acc[4][2];
  double* Cw = C + ((long)by * BM + wr * 64 + fk) * ldc + (long)bx * BN + wc * 32 + fr;
  if (CACC) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n][r] = -Cw[(long)(m * 16 + 4 * r) * ldc + n * 16];
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = d4_t{0, 0, 0, 0};
  }
  const int nk = K / BK;
  int aoff[4], boff[2];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int ra = wr * 64 + m * 16 + fr;
    aoff[m] = ra * BK + (((fk >> 1) ^ ((ra >> 1) & 7)) * 2) + (fk & 1);
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int rb = BM + wc * 32 + n * 16 + fr;
    boff[n] = rb * BK + (((fk >> 1) ^ ((rb >> 1) & 7)) * 2) + (fk & 1);
  }
#define GLOAD8(j, buf, soff)                                                                                      \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (buf) * TD + lds_wave + (j) * 8 * BK), 16, voff[j], \
                                           (soff), 0, 0)
#pragma unroll
  for (int j = 0; j < GPW; ++j) GLOAD8(j, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  typedef const volatile double __attribute__((address_space(3)))* lcvd_t;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const int soff = ((kt + 1 < nk) ? kt + 1 : kt) * BK * 8;
    const double* cb = smem + cur * TD;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double af[4], bf[2];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = *((lcvd_t)cb + (aoff[m] ^ (4 * kk)));
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = *((lcvd_t)cb + (boff[n] ^ (4 * kk)));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
      GLOAD8(kk, nxt, soff);
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef GLOAD8
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = CACC ? -acc[m][n][r] : alpha * acc[m][n][r];
}

// ---------------------------------------------------------------------------------------------
struct Prob {
  const double *A, *B;
  double* C;
  int M, N, K;
  long ld, ldc;
  const std::vector<double>* hA;
  double beta;
  double alpha = 1.0;
};

static void verify(const Prob& p, const char* name, double c0) {
  // C was c0 everywhere before the launch (beta path) ; check 32 sampled entries
  std::vector<double> row(p.N);
  double worst = 0.0;
  srand(7);
  for (int t = 0; t < 32; ++t) {
    const int i = rand() % p.M, j = rand() % p.N;
    double h;
    CK(hipMemcpy(&h, p.C + (long)i * p.ldc + j, 8, hipMemcpyDeviceToHost));
    double ref = p.beta * c0;
    for (int k = 0; k < p.K; ++k) ref += p.alpha * (*p.hA)[(long)i * p.ld + k] * (*p.hA)[(long)j * p.ld + k];
    worst = fmax(worst, fabs(h - ref) / (fabs(ref) + 1e-30));
  }
  if (worst > 1e-11) printf("   !!! %s MISMATCH rel err %.3e\n", name, worst);
}

template <typename F>
static double bench(const Prob& p, F launch, const char* name, bool check) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double c0 = 0.5;
  if (check) {
    std::vector<double> hc((size_t)p.M * p.ldc, c0);
    CK(hipMemcpy(p.C, hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
    launch();
    CK(hipDeviceSynchronize());
    verify(p, name, c0);
  }
  launch();
  double best = 1e30, sum = 0.0;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    sum += ms;
  }
  const double fl = 2.0 * p.M * p.N * p.K;
  printf("%-34s best %8.3f ms %6.2f TF | mean %6.2f TF\n", name, best, fl / (best * 1e-3) / 1e12, fl / (sum / reps * 1e-3) / 1e12);
  fflush(stdout);
  return best;
}

template <int WR, int WC, int NBUF, int MODE>
static void run_glds(Prob p, const char* name, bool check) {
  if (MODE == 6) p.alpha = -1.0;
  constexpr int BM = 64 * WR, BN = 64 * WC;
  const size_t lds = (size_t)NBUF * (BM + BN) * 16 * 8;
  CK(hipFuncSetAttribute((const void*)gemm_glds<WR, WC, NBUF, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(p.N / BN, p.M / BM);
  bench(p, [&]() { gemm_glds<WR, WC, NBUF, MODE><<<grid, 64 * WR * WC, lds>>>(p.A, p.ld, p.B, p.ld, p.C, p.ldc, p.K, p.alpha, p.beta); }, name, check);
}

template <int MODE, bool CACC>
static void run_bl(Prob p, const char* name, bool check) {
  if (CACC) p.alpha = -1.0;
  const size_t lds = (size_t)2 * 256 * 16 * 8;
  CK(hipFuncSetAttribute((const void*)gemm_bl<MODE, CACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(p.N / 128, p.M / 128);
  bench(p, [&]() { gemm_bl<MODE, CACC><<<grid, 256, lds>>>(p.A, p.ld, p.B, p.ld, p.C, p.ldc, p.K, p.alpha, p.beta); }, name, check);
}

template <bool CACC, int OCC>
static void run_bl8(Prob p, const char* name, bool check) {
  if (CACC) p.alpha = -1.0;
  const size_t lds = (size_t)2 * 256 * 16 * 8;
  CK(hipFuncSetAttribute((const void*)gemm_bl8<CACC, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(p.N / 128, p.M / 128);
  bench(p, [&]() { gemm_bl8<CACC, OCC><<<grid, 512, lds>>>(p.A, p.ld, p.B, p.ld, p.C, p.ldc, p.K, p.alpha, p.beta); }, name, check);
}

static void run_base(const Prob& p, const char* name, bool check) {
  const size_t lds = (size_t)4 * 128 * 17 * 8;
  CK(hipFuncSetAttribute((const void*)gemm_base<17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(p.N / 128, p.M / 128);
  bench(p, [&]() { gemm_base<17><<<grid, 256, lds>>>(p.A, p.ld, p.B, p.ld, p.C, p.ldc, p.K, 1.0, p.beta); }, name, check);
}

int main(int argc, char** argv) {
  const int M = 8192, N = 8192, KMAX = 2048;
  const long ld = KMAX + 16, ldc = N + 16;
  double *A, *C;
  CK(hipMalloc(&A, (size_t)M * ld * 8)); CK(hipMalloc(&C, (size_t)M * ldc * 8));
  std::vector<double> h((size_t)M * ld);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) { srand(1); for (auto& v : h) v = (double)rand() / RAND_MAX * 2 - 1; }
    else { for (auto& v : h) v = 0.0; }
    CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    for (int cfg = 0; cfg < 2; ++cfg) {
      Prob p{A, A, C, M, N, cfg == 0 ? 2048 : 512, ld, ldc, &h, cfg == 0 ? 0.0 : 1.0};
      printf("--- %s operands, K = %d, beta = %g ---\n", pass == 0 ? "random" : "zero", p.K, p.beta);
      const bool check = (pass == 0);
      if (argc > 1 && argv[1][0] == 's') { // tile-order skew experiment on the shipped loop
        for (int skew : {0, 1, 3, 5, 0}) {
          CK(hipMemcpyToSymbol(HIP_SYMBOL(ORDER_SKEW), &skew, sizeof(int)));
          char nm[64];
          snprintf(nm, sizeof nm, "bl shipped loop, column skew %d", skew);
          if (cfg == 0) run_bl<12, false>(p, nm, check);
          else run_bl<12, true>(p, nm, check);
        }
        const int zero = 0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(ORDER_SKEW), &zero, sizeof(int)));
        continue;
      }
      if (argc > 1 && argv[1][0] == '8') { // the 8-wave experiment against the shipped loop only
        if (cfg == 0) {
          run_bl<12, false>(p, "bl 3+3+2 + sgb, volatile ds_read", check);
          run_bl8<false, 2>(p, "bl8: 8 waves of 64x32, 2 wg/CU", check);
        } else {
          run_bl<12, true>(p, "bl 3+3+2 + sgb, C-in-acc, volatile ds_read", check);
          run_bl8<true, 2>(p, "bl8: 8 waves of 64x32, C-in-acc", check);
        }
        continue;
      }
      run_base(p, "base 128x128 regstage LDT17", check);
      run_glds<2, 2, 2, 0>(p, "glds 128x128 2buf", check);
      if (cfg == 0) {
        run_bl<7, false>(p, "bl loads at top", check);
        run_bl<8, false>(p, "bl loads spread (source order)", check);
        run_bl<9, false>(p, "bl loads spread + sgb", check);
        run_bl<11, false>(p, "bl loads 3+3+2 + sgb", check);
        run_bl<12, false>(p, "bl 3+3+2 + sgb, volatile ds_read", check);
        run_bl<10, false>(p, "bl spread + sgb + stagger(diag)", check);
      } else {
        run_glds<2, 2, 2, 6>(p, "glds C-in-acc", check);
        run_bl<7, true>(p, "bl loads at top, C-in-acc", check);
        run_bl<8, true>(p, "bl loads spread, C-in-acc", check);
        run_bl<9, true>(p, "bl loads spread + sgb, C-in-acc", check);
        run_bl<11, true>(p, "bl loads 3+3+2 + sgb, C-in-acc", check);
        run_bl<12, true>(p, "bl 3+3+2 + sgb, C-in-acc, volatile ds_read", check);
        run_bl<10, true>(p, "bl spread+sgb+C-in-acc+stagger(diag)", check);
      }
    }
  }
  return 0;
}
