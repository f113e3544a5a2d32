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
                                                           int* info, int info_base) {
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

int launch_potf2_inv(gpx_ctx* ctx, double* dA, int64_t lda, double* dLinv, int* dInfo,
                     int info_base) {
  static bool attr_set = false;
  if (!attr_set) {
    GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)POTF2_LDS_BYTES));
    attr_set = true;
  }
  // algorithmic flops: factor n^3/3 + triangular inverse n^3/3
  ProfScope ps(ctx, GPX_PROF_POTF2, 2.0 * PB * (double)PB * PB / 3.0);
  potf2_inv_kernel<<<1, 256, POTF2_LDS_BYTES, ctx->s>>>(dA, lda, dLinv, dInfo, info_base);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

} // namespace gpx
