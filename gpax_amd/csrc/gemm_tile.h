// gemm_tile.h — the two fp64 MFMA tile bodies of gemm_f64.hip as device functions (see that file's header for the
// arithmetic contract: every C element accumulates its k range in ascending k from the same start value with the same
// epilogue in every shape).  Shared with panel.hip, whose cooperative panel kernel runs the same bodies on its strips
// and tiles — so a result bit never depends on which kernel computed it.
#pragma once
#include "common.h"

namespace gpx {

typedef double d4_t __attribute__((ext_vector_type(4)));


template <int MT, int NT, int BK, bool DBUF>
constexpr size_t gemm_lds_bytes() {
  return size_t(DBUF ? 2 : 1) * (32 * MT + 32 * NT) * (BK + 1) * sizeof(double); // buffers x (A tile + B tile)
}

// Element offsets of batch entry bb.  One level: bb * *_bs.  Two levels (batch2 > 1: `batch` counts outer x inner entries, the
// inner index runs fastest — the block pairs of one level of the L^-T tree inside every sample of a batched fit step).
__device__ __forceinline__ void batch_offsets(const GemmArgs& g, const int bb, int64_t& oa, int64_t& ob, int64_t& oc) {
  if (g.batch2 > 1) {
    const int b1 = bb / g.batch2, b2 = bb - b1 * g.batch2;
    oa = (int64_t)b1 * g.a_bs + (int64_t)b2 * g.a_bs2;
    ob = (int64_t)b1 * g.b_bs + (int64_t)b2 * g.b_bs2;
    oc = (int64_t)b1 * g.c_bs + (int64_t)b2 * g.c_bs2;
  } else {
    oa = (int64_t)bb * g.a_bs;
    ob = (int64_t)bb * g.b_bs;
    oc = (int64_t)bb * g.c_bs;
  }
}

// TAG only gives the Cholesky trailing update (TAG = 1) its own symbol, so that rocprofv3 kernel
// statistics separate the dominant kernel from the small panel GEMMs (TAG = 0).
// MT x NT = 16x16 MFMA tiles per wave; the workgroup (2x2 waves) covers a (32 MT) x (32 NT) tile:
//   <4,4> 128x128  throughput shape (trailing updates, big sweeps)
//   <2,4>  64x128  in-place panel TRSM (needs the full 128-column width in one workgroup)
//   <2,2>  64x64   latency shape: small grids on the critical chain of the look-ahead — 4x more
//                  workgroups, each with a 4x shorter K loop
// BK = k-step: 16 for the throughput shape; 32 for the latency shapes (their k-step time is one
// global-load latency + barrier, not MFMA time, so fewer and fatter k-steps cut the launch latency).
// LDS rows are padded to BK + 1 doubles (odd => ds_read2_b64 fragment reads are conflict-free).
// DBUF = false: single LDS buffer (two barriers per k-step) — 17 KB for the 64x64 shape, which fits
// in the LDS two resident trailing-update workgroups leave free, so a chain launch is placed at once.
// One (32 MT) x (32 NT) tile of the latency shapes: the body of gemm_nt_kernel as a device function, so that the
// cooperative panel kernel (panel.hip) runs the very same arithmetic on its strips.
// EPI: the epilogue form, uniform over a launch and chosen by the host (as in lat_tile / nt128_tile below): 0 beta == 0 |
// 1 alpha == -1, beta == 1 | 2 generic.  (Rounds 1 - 5 chose it at run time inside the kernel: the 32 x 128 strip instantiation
// then carried the generic read-modify-write epilogue — 22 - 26 spilled VGPRs, a 92-byte private segment — for code its
// launches never run.)  Staging, k order and arithmetic are untouched: the same bits.
template <int MT, int NT, int BK, bool DBUF, int EPI>
__device__ __forceinline__ void gemm_nt_tile(const GemmArgs& g, double* smem, const int bx, const int by, const int bzz) {
  constexpr int BM = 32 * MT, BN = 32 * NT, LDT = BK + 1;
  constexpr int TA = BM * LDT, TB = BN * LDT;
  constexpr int TPR = BK / 2;    // threads per staged row (one double2 each)
  constexpr int RPP = 256 / TPR; // rows staged per pass
  constexpr int PA = BM / RPP, PB_ = BN / RPP;
  // grid.z = batch entry * nsplit + split-K slab
  const int bb = (g.batch > 1) ? bzz / g.nsplit : 0;
  const int bz = bzz - bb * g.nsplit;
  int64_t off_a, off_b, off_c;
  batch_offsets(g, bb, off_a, off_b, off_c);
  // element coordinates of this tile in the caller's global tile frame (offsets given in 128-tiles)
  const int row0 = g.ti_off * 128 + by * BM, col0 = g.tj_off * 128 + bx * BN;
  if (g.lower && col0 > row0 + BM - 1) return; // entirely above the diagonal

  int kb = 0, ke = g.K;
  if (g.ktri) kb = row0 & ~(BK - 1);            // A rows are zero left of the diagonal (upper-triangular operand)
  if (g.kcol) kb = max(kb, col0 & ~(BK - 1));   // B rows are zero left of the diagonal (upper-triangular right factor)
  if (g.kupper) ke = min(ke, col0 + BN);        // B rows are zero right of the diagonal (lower-triangular factor)
  double* C = g.C + off_c;  // may alias A (in-place panel TRSM): no restrict here
  if (g.kchunk > 0) {
    kb = max(kb, bz * g.kchunk);
    ke = min(ke, (bz + 1) * g.kchunk);
    C += (int64_t)bz * g.c_split_stride;
  }
  const int nk = (ke > kb) ? (ke - kb) / BK : 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;

  // staging map: thread -> (row lr + RPP i, cols lc, lc + 1)
  const int lr = tid / TPR, lc = (tid % TPR) * 2;
  const double* Ap = g.A + off_a + ((int64_t)by * BM + lr) * g.lda + kb + lc;
  const double* Bp = g.B + off_b + ((int64_t)bx * BN + lr) * g.ldb + kb + lc;
  const int64_t a_step = (int64_t)RPP * g.lda, b_step = (int64_t)RPP * g.ldb;

  double* sA0 = smem;
  double* sB0 = smem + TA;
  double* sA1 = DBUF ? smem + TA + TB : sA0;
  double* sB1 = DBUF ? smem + 2 * TA + TB : sB0;

  // epilogue form (uniform over the launch) — the same three forms as gemm_nt128_kernel, see the file header
  const double alpha = g.alpha, beta = g.beta;
  constexpr bool cacc = (EPI == 1);
  double* Cw = C + ((int64_t)by * BM + wr * 16 * MT + fk) * g.ldc + (int64_t)bx * BN + wc * 16 * NT + fr;
  d4_t acc[MT][NT];
  if (cacc) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n][r] = -Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16];
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = d4_t{0.0, 0.0, 0.0, 0.0};
  }

  const int a_frag_off = (wr * 16 * MT + fr) * LDT + fk;
  const int b_frag_off = (wc * 16 * NT + fr) * LDT + fk;
  const int st_off = lr * LDT + lc;

  if (nk > 0) {
    double2 ra[PA], rb[PB_];
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const double2*>(Ap + i * a_step);
#pragma unroll
    for (int i = 0; i < PB_; ++i) rb[i] = *reinterpret_cast<const double2*>(Bp + i * b_step);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      sA0[st_off + RPP * i * LDT] = ra[i].x;
      sA0[st_off + RPP * i * LDT + 1] = ra[i].y;
    }
#pragma unroll
    for (int i = 0; i < PB_; ++i) {
      sB0[st_off + RPP * i * LDT] = rb[i].x;
      sB0[st_off + RPP * i * LDT + 1] = rb[i].y;
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const double* cA = (kt & 1) ? sA1 : sA0;
      const double* cB = (kt & 1) ? sB1 : sB0;
      double* nA = (kt & 1) ? sA0 : sA1;
      double* nB = (kt & 1) ? sB0 : sB1;
      // prefetch the next k-tile (the last iteration re-reads its own tile: in bounds, unused)
      const int koff = ((kt + 1 < nk) ? (kt + 1) : kt) * BK;
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const double2*>(Ap + i * a_step + koff);
#pragma unroll
      for (int i = 0; i < PB_; ++i) rb[i] = *reinterpret_cast<const double2*>(Bp + i * b_step + koff);
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        double af[MT], bf[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = cA[a_frag_off + m * 16 * LDT + kk * 4];
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = cB[b_frag_off + n * 16 * LDT + kk * 4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0); // keep the LDS refill (and its vmcnt wait) behind the MFMAs
      if (!DBUF) __syncthreads();          // single buffer: every wave is done reading before the refill
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        nA[st_off + RPP * i * LDT] = ra[i].x;
        nA[st_off + RPP * i * LDT + 1] = ra[i].y;
      }
#pragma unroll
      for (int i = 0; i < PB_; ++i) {
        nB[st_off + RPP * i * LDT] = rb[i].x;
        nB[st_off + RPP * i * LDT + 1] = rb[i].y;
      }
      __syncthreads();
    }
  }

  // epilogue: D[row = (lane >> 4) + 4 r][col = lane & 15] per 16x16 accumulator.  A wave
  // store covers 4 rows x 16 contiguous doubles (full 128-B lines).  The generic beta path batches the
  // C loads of one accumulator row-block ahead of their use (one latency, not sixteen).
  if (cacc) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = -acc[m][n][r];
  } else if (EPI == 2) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      double cv[4][NT];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) cv[r][n] = Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = fma(beta, cv[r][n], alpha * acc[m][n][r]);
    }
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = alpha * acc[m][n][r];
  }
}

typedef void __attribute__((address_space(3)))* lds_ptr_t;
typedef const volatile double __attribute__((address_space(3)))* lds_cvd_t;

// ---- latency shapes, round 5 ---------------------------------------------------------------------------
// The same (32 MT) x (32 NT) tiles as gemm_nt_tile above (64 x 64, and 32 x 128 strips for the in-place panel TRSM) on the
// staging of the throughput shape: LDS-direct loads (`buffer_load_dwordx4 ... lds`: no staging VGPRs, no ds_write),
// unpadded swizzled rows, ONE barrier per k-slice, the epilogue form a template parameter (the generic-beta epilogue no
// longer rides in every instantiation: the strip kernel spilled 22 - 26 VGPRs for code its launches never run).
// A k-slice is BK = 8 doubles = 64-B rows in LDS, a ring of NST slices: the loads of slice t + NST - 1 are issued at the
// top of slice t (NST = 3: two slices = 2 x 8 KB in flight per workgroup while the third is multiplied), so a K = 128
// update is 16 slices behind ONE exposed load latency instead of 8 (BK = 16) exposed global -> VGPR -> LDS round trips
// with two barriers each.  (32 MT + 32 NT) x 64 B x NST: 24 KB for 64 x 64 with three slices, 20 KB for the strips with two —
// both fit in what two resident trailing-update workgroups leave free on a CU (32 KB, 80 VGPRs per SIMD; tests/test_abi.py).
// One wave-instruction = 64 lanes x 16 B = 16 LDS rows of 64 B; lane -> row (lane >> 2), LDS chunk (lane & 3), which holds
// SOURCE chunk (lane & 3) ^ ((row >> 2) & 3): a fragment read (16 rows x one chunk per half-wave) then touches all 16
// chunk positions of the 256-B bank row once — conflict-free ds_read_b64, as in nt128_tile.
// Arithmetic: every C element accumulates its k range in ascending k from the same start value with the same epilogue as
// in every other shape — bit-identical to gemm_nt_tile and nt128_tile.
template <int MT, int NT, int NST, int EPI>
__device__ __forceinline__ void lat_tile(const GemmArgs& g, double* smem, const int bx, const int by, const int bzz) {
  constexpr int BK = 8, BM = 32 * MT, BN = 32 * NT, SD = (BM + BN) * BK; // doubles per slice buffer
  constexpr int NIA = BM / 16, NIB = BN / 16;                             // wave-instructions per slice: A rows, B rows
  constexpr int LA = NIA >= 4 ? NIA / 4 : 1, LB = NIB >= 4 ? NIB / 4 : 1; // ... per wave (NIA < 4: waves share them)
  static_assert(NST == 2 || NST == 3, "ring of 2 or 3 slices");
  const int bb = (g.batch > 1) ? bzz / g.nsplit : 0;
  const int bz = bzz - bb * g.nsplit;
  int64_t off_a, off_b, off_c;
  batch_offsets(g, bb, off_a, off_b, off_c);
  const int row0 = g.ti_off * 128 + by * BM, col0 = g.tj_off * 128 + bx * BN;
  if (g.lower && col0 > row0 + BM - 1) return; // entirely above the diagonal

  int kb = 0, ke = g.K;
  if (g.ktri) kb = row0 & ~15;
  if (g.kcol) kb = max(kb, col0 & ~15);
  if (g.kupper) ke = min(ke, col0 + BN);
  double* C = g.C + off_c; // may alias A (in-place panel TRSM): the whole k range is read before the first store
  if (g.kchunk > 0) {
    kb = max(kb, bz * g.kchunk);
    ke = min(ke, (bz + 1) * g.kchunk);
    C += (int64_t)bz * g.c_split_stride;
  }
  const int nk = (ke > kb) ? (ke - kb) / BK : 0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + off_a + (int64_t)by * BM * g.lda), 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + off_b + (int64_t)bx * BN * g.ldb), 0, 0x7fffffff, 0x00020000);
  // staging map: instruction ia covers LDS rows [16 ia, 16 ia + 16) of the A part, ib likewise of the B part
  int voffA[LA], voffB[LB], ldsA[LA], ldsB[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int ia = (wave * LA + i) % NIA;
    const int row = ia * 16 + (lane >> 2);
    voffA[i] = (int)(((int64_t)row * g.lda + (((lane & 3) ^ ((row >> 2) & 3)) * 2)) * 8);
    ldsA[i] = ia * 16 * BK;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int ib = (wave * LB + i) % NIB;
    const int row = ib * 16 + (lane >> 2);
    voffB[i] = (int)(((int64_t)row * g.ldb + (((lane & 3) ^ ((row >> 2) & 3)) * 2)) * 8);
    ldsB[i] = (BM + ib * 16) * BK;
  }

  double* Cw = C + ((int64_t)by * BM + wr * 16 * MT + fk) * g.ldc + (int64_t)bx * BN + wc * 16 * NT + fr;
  d4_t acc[MT][NT];
  if (EPI == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n][r] = -Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16];
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = d4_t{0.0, 0.0, 0.0, 0.0};
  }

  // fragment offsets (doubles) in a slice: row * 8 + ((k >> 1) ^ ((row >> 2) & 3)) * 2 + (k & 1), k = 4 kk + fk: block kk = 1 is
  // `offset ^ 4` (2 kk only touches bit 1 of the chunk index)
  int aoff[MT], boff[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ra = wr * 16 * MT + m * 16 + fr;
    aoff[m] = ra * BK + (((fk >> 1) ^ ((ra >> 2) & 3)) * 2) + (fk & 1);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int rb = BM + wc * 16 * NT + n * 16 + fr;
    boff[n] = rb * BK + (((fk >> 1) ^ ((rb >> 2) & 3)) * 2) + (fk & 1);
  }

#define GPX_LAT_ISSUE(slice, buf)                                                                                       \
  do {                                                                                                                  \
    const int soff_ = (kb + (slice) * BK) * 8;                                                                          \
    _Pragma("unroll") for (int i = 0; i < LA; ++i)                                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + (buf) * SD + ldsA[i]), 16, voffA[i], soff_, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < LB; ++i)                                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(smem + (buf) * SD + ldsB[i]), 16, voffB[i], soff_, 0, 0); \
  } while (0)
#define GPX_LAT_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

  // one k-slice out of buffer `cur`: all fragments of both 4-deep blocks first (8 - 10 ds_read_b64), then the MFMAs
#define GPX_LAT_COMPUTE(cur)                                                                      \
  do {                                                                                            \
    const double* cb_ = smem + (cur) * SD;                                                        \
    double af_[2][MT], bf_[2][NT];                                                                \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                            \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) af_[kk][m] = *((lds_cvd_t)cb_ + (aoff[m] ^ (4 * kk))); \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) bf_[kk][n] = *((lds_cvd_t)cb_ + (boff[n] ^ (4 * kk))); \
    }                                                                                             \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                              \
      _Pragma("unroll") for (int m = 0; m < MT; ++m)                                              \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                            \
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af_[kk][m], bf_[kk][n], acc[m][n], 0, 0, 0); \
  } while (0)
  // workgroup barrier WITHOUT the fence of __syncthreads (which drains vmcnt: the slice in flight behind the next one
  // would be waited for at every barrier).  What has to hold at the barrier is stated explicitly: this wave's LDS-DMA
  // loads of the NEXT slice have landed (GPX_LAT_WAIT), and its own ds_reads of the CURRENT slice have RETURNED
  // (lgkmcnt(0)) — the compiler is free to sink the MFMAs that consume them below the barrier, and the buffer they read
  // is the one the loads issued right after the barrier overwrite.
#define GPX_LAT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  if (nk > 0) {
    GPX_LAT_ISSUE(0, 0);
    if (NST == 3 && nk > 1) {
      GPX_LAT_ISSUE(1, 1);
      GPX_LAT_WAIT(LA + LB);
    } else {
      GPX_LAT_WAIT(0);
    }
    GPX_LAT_BARRIER();
    int cur = 0, t = 0;
    // steady state: the loads of slice t + NST - 1 go out, slice t is multiplied, slice t + 1 has landed at the barrier
    for (; t + NST - 1 < nk; ++t) {
      int nb = cur + NST - 1;
      if (nb >= NST) nb -= NST;
      GPX_LAT_ISSUE(t + NST - 1, nb);
      GPX_LAT_COMPUTE(cur);
      if (NST == 3) GPX_LAT_WAIT(LA + LB);
      else GPX_LAT_WAIT(0);
      GPX_LAT_BARRIER();
      cur = (cur + 1 == NST) ? 0 : cur + 1;
    }
    // the last NST - 1 slices: nothing left to issue
    for (; t < nk; ++t) {
      GPX_LAT_COMPUTE(cur);
      if (t + 1 < nk) {
        GPX_LAT_WAIT(0);
        GPX_LAT_BARRIER();
      }
      cur = (cur + 1 == NST) ? 0 : cur + 1;
    }
  }
#undef GPX_LAT_COMPUTE
#undef GPX_LAT_BARRIER
#undef GPX_LAT_ISSUE
#undef GPX_LAT_WAIT

  const double alpha = g.alpha, beta = g.beta;
  if (EPI == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = -acc[m][n][r];
  } else if (EPI == 2) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      double cv[4][NT];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) cv[r][n] = Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = fma(beta, cv[r][n], alpha * acc[m][n][r]);
    }
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = alpha * acc[m][n][r];
  }
}

// ---- throughput shape -------------------------------------------------------------------------------

// EPI: 0 beta == 0 | 1 alpha == -1, beta == 1 (accumulators start from -C) | 2 generic read-modify-write
// TAG only gives the Cholesky trailing update (TAG = 1) its own symbol, so that rocprofv3 kernel statistics
// separate the dominant kernel from the other GEMM launches (TAG = 0).
template <int EPI>
__device__ __forceinline__ void nt128_tile(const GemmArgs& g, double* smem, const int bx, const int by, const int bzz) {
  constexpr int BK = 16, BM = 128, BN = 128, TD = (BM + BN) * BK, GPW = 8;
  const int bb = (g.batch > 1) ? bzz / g.nsplit : 0;
  const int bz = bzz - bb * g.nsplit;
  int64_t off_a, off_b, off_c;
  batch_offsets(g, bb, off_a, off_b, off_c);
  const int row0 = g.ti_off * 128 + by * BM, col0 = g.tj_off * 128 + bx * BN;
  if (g.lower && col0 > row0 + BM - 1) return; // entirely above the diagonal

  int kb = 0, ke = g.K;
  if (g.ktri) kb = row0 & ~(BK - 1);
  if (g.kcol) kb = max(kb, col0 & ~(BK - 1));
  if (g.kupper) ke = min(ke, col0 + BN);
  double* C = g.C + off_c;
  if (g.kchunk > 0) {
    kb = max(kb, bz * g.kchunk);
    ke = min(ke, (bz + 1) * g.kchunk);
    C += (int64_t)bz * g.c_split_stride;
  }
  const int nk = (ke > kb) ? (ke - kb) / BK : 0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;

  // staging: waves 0, 1 bring in the A rows, waves 2, 3 the B rows (64 rows = 8 groups of 8 rows per wave).
  // One wave-instruction = 64 lanes x 16 B = 1 KiB = 8 LDS rows; lane -> row (lane >> 3), LDS chunk (lane & 7),
  // which holds source chunk (lane & 7) ^ ((row >> 1) & 7) of that row.
  const bool isA = wave < 2;
  const double* src = isA ? g.A + off_a + (int64_t)by * BM * g.lda
                          : g.B + off_b + (int64_t)bx * BN * g.ldb;
  const int64_t ldx = isA ? g.lda : g.ldb;
  // 2 GiB window from the tile's first row: offsets stay below 128 rows x ld x 8 B + K x 8 B
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  int voff[GPW];
#pragma unroll
  for (int j = 0; j < GPW; ++j) {
    const int row = (wave & 1) * 64 + j * 8 + (lane >> 3);
    const int lrow = (isA ? 0 : BM) + row;
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);
    voff[j] = (int)((row * ldx + chunk * 2) * 8);
  }
  const int lds_wave = ((isA ? 0 : BM) + (wave & 1) * 64) * BK;

  double* Cw = C + ((int64_t)by * BM + wr * 64 + fk) * g.ldc + (int64_t)bx * BN + wc * 64 + fr;
  d4_t acc[4][4];
  if (EPI == 1) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n][r] = -Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16];
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = d4_t{0.0, 0.0, 0.0, 0.0};
  }

  // fragment offsets (doubles) in a buffer: row * 16 + ((k >> 1) ^ ((row >> 1) & 7)) * 2 + (k & 1), k = 4 kk + fk;
  // 2 kk only touches bits 1..2 of the chunk index, so block kk is `offset ^ (4 kk)`
  int aoff[4], boff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int ra = wr * 64 + m * 16 + fr, rb = BM + wc * 64 + m * 16 + fr;
    aoff[m] = ra * BK + (((fk >> 1) ^ ((ra >> 1) & 7)) * 2) + (fk & 1);
    boff[m] = rb * BK + (((fk >> 1) ^ ((rb >> 1) & 7)) * 2) + (fk & 1);
  }

#define GPX_GLOAD(j, buf, soff)                                                                                    \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (buf) * TD + lds_wave + (j) * 8 * BK), 16, voff[j], \
                                           (soff), 0, 0)
  if (nk > 0) {
#pragma unroll
    for (int j = 0; j < GPW; ++j) GPX_GLOAD(j, 0, kb * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1, nxt = cur ^ 1;
      // the last step re-reads its own k-tile into the idle buffer (in bounds, unused): no branch in the body
      const int soff = (kb + ((kt + 1 < nk) ? kt + 1 : kt) * BK) * 8;
      const double* cb = smem + cur * TD;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double af[4], bf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          // volatile LDS-address-space loads: one ds_read_b64 each.  Left alone the compiler pairs fragments 2 KB
          // apart into ds_read2st64_b64, whose 32-bank, 16-lane-group banking the swizzle is not made for
          // (SQ_LDS_BANK_CONFLICT 2.6e7 per launch, profiles/r02/sq.md) and which moves half the bytes per LDS cycle
          af[m] = *((lds_cvd_t)cb + (aoff[m] ^ (4 * kk)));
          bf[m] = *((lds_cvd_t)cb + (boff[m] ^ (4 * kk)));
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
          // next k-tile: 3 + 3 + 2 loads behind the MFMAs of blocks 0, 1, 2; block 3 gives them time to land
          if (kk < 2 && m < 3) GPX_GLOAD(3 * kk + m, nxt, soff);
          if (kk == 2 && m < 2) GPX_GLOAD(6 + m, nxt, soff);
        }
        if (kk < 3) {
#pragma unroll
          for (int q = 0; q < (kk < 2 ? 3 : 2); ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);  // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); // 1 VMEM read
          }
          if (kk < 2) __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
          else __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
#undef GPX_GLOAD

  const double alpha = g.alpha, beta = g.beta;
  if (EPI == 1) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = -acc[m][n][r];
  } else if (EPI == 2) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double cv[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) cv[r][n] = Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = fma(beta, cv[r][n], alpha * acc[m][n][r]);
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) Cw[(int64_t)(m * 16 + 4 * r) * g.ldc + n * 16] = alpha * acc[m][n][r];
  }
}

} // namespace gpx
