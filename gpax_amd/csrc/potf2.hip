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
#include "potf2_tile.h"
#include "potf2_chain.h"

#include <cstdlib>

namespace gpx {

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
  potf2_tile_body<BLK>(A, lda, Linv, info, info_base, Ppre, Kpre, lds);
}

// the wave-specialised form (potf2_chain.h): default; bit-identical to potf2_tile_kernel<false>
__global__ __launch_bounds__(256, 1) void potf2_chain_kernel(double* A, int64_t lda, double* Linv, int* info, int info_base,
                                                             int64_t a_bs, int64_t linv_bs) {
  A += (int64_t)blockIdx.x * a_bs; // one workgroup per batch entry
  Linv += (int64_t)blockIdx.x * linv_bs;
  if (info != nullptr) info += blockIdx.x;
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  potf2_chain_body(A, lda, Linv, info, info_base, lds);
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
    if (use_tile && ctx->potf2_chain && !ctx->potf2_diag_blocked && dPre == nullptr)
      potf2_chain_kernel<<<nb, 256, POTF2_CHAIN_LDS, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, a_bs, linv_bs);
    else if (use_tile && ctx->potf2_diag_blocked)
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
