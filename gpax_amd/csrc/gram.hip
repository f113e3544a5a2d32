// gram.hip — RBF / Matern-5/2 Gram-matrix build for gfx950 (HBM-write-bound).
//
// Replaces gpax/kernels/kernels.py:28-91 (square_scaled_distance + RBFKernel / MaternKernel).
// The reference forms r2 = |x/l|^2 - 2 (x/l).(z/l) + |z/l|^2 through a tiny-K matmul and clips
// at 0; with d <= 16 a GEMM is the wrong shape, so r2 is accumulated directly as
// sum_k ((x_k - z_k)/l_k)^2 (>= 0 by construction; agrees with the expansion to ~1e-16 |x/l|^2).
//
// Mapping: one 256-thread workgroup writes a 32-row x 512-column tile.  Each thread owns two
// adjacent columns (z held in registers), walks the 32 rows whose scaled x are staged in LDS
// (read as a broadcast), and writes one 16-byte store per row: a wave writes 1 KiB contiguous
// per row, i.e. full 128-B lines.  exp/sqrt, the k_scale multiply and the (noise + jitter)
// diagonal add are fused; nothing but the output touches HBM.
#include "common.h"

namespace gpx {

constexpr int GT_ROWS = 32;
constexpr int GT_COLS = 512;

template <int KIND, int D>
__global__ __launch_bounds__(256) void gram_kernel(KernelParams kpv, const double* __restrict__ X,
                                                   int n, int n_pad,
                                                   const double* __restrict__ Z, int m, int m_pad,
                                                   double diag_add, int add_diag, int lower_only,
                                                   double* __restrict__ out, int64_t ld,
                                                   int64_t out_bs, const ThetaDev* __restrict__ th,
                                                   int diag_sel, TaskStride ts,
                                                   const double* __restrict__ diag_vec) {
  if (ts.mod > 0) { // per-task inputs: batch entry z belongs to task z % mod
    const int task = blockIdx.z % ts.mod;
    X += task * ts.x_bs;
    Z += task * ts.z_bs;
  }
  int bx = blockIdx.x, by = blockIdx.y;
  if (lower_only == 2) {
    // Only the tiles a lower-triangular build writes are launched (round 5; rounds 1 - 4 launched the full grid and half of
    // it returned at once): linear id -> column tile c (512 wide) and row tile r (32 high); column c holds the row tiles
    // 16 c .. R - 1 (the tiles with j0 <= i0 + 31), consecutive ids walk down a column.
    const int R = (n_pad + GT_ROWS - 1) / GT_ROWS;
    int id = blockIdx.x, c = 0;
    while (id >= R - 16 * c) {
      id -= R - 16 * c;
      ++c;
    }
    bx = c;
    by = 16 * c + id;
  }
  const int j0 = bx * GT_COLS;
  const int i0 = by * GT_ROWS;
  if (lower_only && j0 > i0 + GT_ROWS - 1) return;
  // batched launch: sample blockIdx.z reads its hyper-parameters from the device table
  const ThetaDev* t = (th != nullptr) ? th + blockIdx.z : nullptr;
  const double k_scale = t ? t->kp.scale : kpv.scale;
  const double pi_over_p = t ? t->kp.pi_over_p : kpv.pi_over_p;
  auto inv_ell = [&](int c) -> double { return t ? t->kp.inv_ell[c] : kpv.inv_ell[c]; };
  if (t != nullptr) {
    diag_add = (diag_sel == 1) ? t->diag_train : (diag_sel == 2 ? t->diag_pred : 0.0);
    out += (int64_t)blockIdx.z * out_bs;
  }
  __shared__ double sx[GT_ROWS * GPX_MAX_DIM];
  const int tid = threadIdx.x;
  const int d = (D > 0) ? D : kpv.d; // structural: the same for every entry of a batch
  for (int idx = tid; idx < GT_ROWS * d; idx += 256) {
    const int r = idx / d, c = idx - r * d;
    const int i = i0 + r;
    sx[idx] = (i < n) ? X[(int64_t)i * d + c] * (KIND == GPX_KERNEL_PERIODIC ? 1.0 : inv_ell(c)) : 0.0;
  }
  const int j = j0 + 2 * tid;
  double z0[(D > 0) ? D : 1], z1[(D > 0) ? D : 1];
  if (D > 0) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const double zs = (KIND == GPX_KERNEL_PERIODIC) ? 1.0 : inv_ell(c);
      z0[c] = (j < m) ? Z[(int64_t)j * D + c] * zs : 0.0;
      z1[c] = (j + 1 < m) ? Z[(int64_t)(j + 1) * D + c] * zs : 0.0;
    }
  }
  __syncthreads();
  if (j >= m_pad) return;
  const int rows = min(GT_ROWS, n_pad - i0);
  for (int r = 0; r < rows; ++r) {
    const int i = i0 + r;
    double r20 = 0.0, r21 = 0.0;
    if (D > 0) {
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const double x = sx[r * D + c];
        double a = x - z0[c], b = x - z1[c];
        if (KIND == GPX_KERNEL_PERIODIC) {
          a = sin(a * pi_over_p) * inv_ell(c);
          b = sin(b * pi_over_p) * inv_ell(c);
        }
        r20 = fma(a, a, r20);
        r21 = fma(b, b, r21);
      }
    } else {
      for (int c = 0; c < d; ++c) {
        const double x = sx[r * d + c];
        const double zs = (KIND == GPX_KERNEL_PERIODIC) ? 1.0 : inv_ell(c);
        const double za = (j < m) ? Z[(int64_t)j * d + c] * zs : 0.0;
        const double zb = (j + 1 < m) ? Z[(int64_t)(j + 1) * d + c] * zs : 0.0;
        double a = x - za, b = x - zb;
        if (KIND == GPX_KERNEL_PERIODIC) {
          a = sin(a * pi_over_p) * inv_ell(c);
          b = sin(b * pi_over_p) * inv_ell(c);
        }
        r20 = fma(a, a, r20);
        r21 = fma(b, b, r21);
      }
    }
    double v0 = kernel_value<KIND>(r20, k_scale);
    double v1 = kernel_value<KIND>(r21, k_scale);
    if (add_diag) { // (noise + jitter) I, plus a per-point variance when given (measured noise, mngp.py:96)
      if (i == j) v0 += diag_add + (diag_vec != nullptr && i < n ? diag_vec[i] : 0.0);
      if (i == j + 1) v1 += diag_add + (diag_vec != nullptr && i < n ? diag_vec[i] : 0.0);
    }
    if (i >= n) v0 = v1 = 0.0;
    if (j >= m) v0 = 0.0;
    if (j + 1 >= m) v1 = 0.0;
    double* p = out + (int64_t)i * ld + j;
    if (j + 1 < m_pad) {
      *reinterpret_cast<double2*>(p) = make_double2(v0, v1);
    } else {
      p[0] = v0;
    }
  }
}

template <int KIND>
static void gram_dispatch(const KernelParams& kp, dim3 grid, hipStream_t s, const double* X, int n,
                          int n_pad, const double* Z, int m, int m_pad, double diag_add,
                          int add_diag, int lower_only, double* out, int64_t ld, int64_t out_bs,
                          const ThetaDev* th, int diag_sel, TaskStride ts, const double* diag_vec) {
  switch (kp.d) {
    case 1:
      gram_kernel<KIND, 1><<<grid, 256, 0, s>>>(kp, X, n, n_pad, Z, m, m_pad, diag_add, add_diag,
                                                 lower_only, out, ld, out_bs, th, diag_sel, ts, diag_vec);
      break;
    case 2:
      gram_kernel<KIND, 2><<<grid, 256, 0, s>>>(kp, X, n, n_pad, Z, m, m_pad, diag_add, add_diag,
                                                 lower_only, out, ld, out_bs, th, diag_sel, ts, diag_vec);
      break;
    case 3:
      gram_kernel<KIND, 3><<<grid, 256, 0, s>>>(kp, X, n, n_pad, Z, m, m_pad, diag_add, add_diag,
                                                 lower_only, out, ld, out_bs, th, diag_sel, ts, diag_vec);
      break;
    case 4:
      gram_kernel<KIND, 4><<<grid, 256, 0, s>>>(kp, X, n, n_pad, Z, m, m_pad, diag_add, add_diag,
                                                 lower_only, out, ld, out_bs, th, diag_sel, ts, diag_vec);
      break;
    default:
      gram_kernel<KIND, 0><<<grid, 256, 0, s>>>(kp, X, n, n_pad, Z, m, m_pad, diag_add, add_diag,
                                                 lower_only, out, ld, out_bs, th, diag_sel, ts, diag_vec);
  }
}

// Writes the n_pad x m_pad extent: kernel values inside n x m, zeros in the padding.
// n_pad / m_pad are passed through dOut's caller as the padded extents via ld-sized rows.
int launch_gram_padded(gpx_ctx* ctx, const KernelParams& kp, const double* dX, int n, int n_pad,
                       const double* dZ, int m, int m_pad, double diag_add, int add_diag,
                       int lower_only, double* dOut, int64_t ld, int batch, int64_t out_bs,
                       const ThetaDev* th, int diag_sel, TaskStride ts, const double* diag_vec) {
  if (n_pad <= 0 || m_pad <= 0) return 0;
  if (batch > 1 && th == nullptr) return bad_arg(ctx, "batched Gram needs a device theta table");
  dim3 grid((m_pad + GT_COLS - 1) / GT_COLS, (n_pad + GT_ROWS - 1) / GT_ROWS, batch > 1 ? batch : 1);
  if (lower_only) { // launch the lower tiles alone: column tile c of the 512-wide grid has R - 16 c row tiles of 32
    const int R = (int)grid.y;
    int64_t tiles = 0;
    for (int c = 0; c < (int)grid.x && R - 16 * c > 0; ++c) tiles += R - 16 * c;
    if (tiles > 0 && tiles < (int64_t)grid.x * grid.y) {
      grid.x = (unsigned)tiles;
      grid.y = 1;
      lower_only = 2;
    }
  }
  // algorithmic bytes: 8*n*m written (+ inputs); SURVEY 8(d): a symmetric build may claim the
  // full 8 n m.
  ProfScope ps(ctx, GPX_PROF_GRAM, 8.0 * (double)n * (double)m * (batch > 1 ? batch : 1));
  if (kp.kind == GPX_KERNEL_R2)
    gram_dispatch<GPX_KERNEL_R2>(kp, grid, ctx->s, dX, n, n_pad, dZ, m, m_pad, diag_add,
                                 add_diag, lower_only, dOut, ld, out_bs, th, diag_sel, ts, diag_vec);
  else if (kp.kind == GPX_KERNEL_RBF)
    gram_dispatch<GPX_KERNEL_RBF>(kp, grid, ctx->s, dX, n, n_pad, dZ, m, m_pad, diag_add,
                                  add_diag, lower_only, dOut, ld, out_bs, th, diag_sel, ts, diag_vec);
  else if (kp.kind == GPX_KERNEL_PERIODIC)
    gram_dispatch<GPX_KERNEL_PERIODIC>(kp, grid, ctx->s, dX, n, n_pad, dZ, m, m_pad, diag_add,
                                       add_diag, lower_only, dOut, ld, out_bs, th, diag_sel, ts, diag_vec);
  else
    gram_dispatch<GPX_KERNEL_MATERN52>(kp, grid, ctx->s, dX, n, n_pad, dZ, m, m_pad, diag_add,
                                       add_diag, lower_only, dOut, ld, out_bs, th, diag_sel, ts, diag_vec);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

int launch_gram(gpx_ctx* ctx, const KernelParams& kp, const double* dX, int n, const double* dZ,
                int m, double diag_add, int add_diag, int lower_only, double* dOut, int64_t ld) {
  return launch_gram_padded(ctx, kp, dX, n, n, dZ, m, m, diag_add, add_diag, lower_only, dOut, ld);
}

// Rows N .. Np-1 of the augmented matrix  [[K, y], [y^T, BIG]] (+) I :
//   row N      = [ y_0 .. y_{N-1} | BIG | 0 ... ]
//   row N + t  = unit vector e_{N+t}
// Factoring the augmented matrix leaves w = L^-1 y in row N of the factor, so the forward
// solve of the lml (NumPyro MVN log_prob's solve_triangular) costs no extra launch.
// info != nullptr: entry blockIdx.z of the factorisation's pivot report is cleared here (one launch and one stream gap
// less than a hipMemsetAsync of its own in front of the Cholesky: 15 us of a 400 us fit step at N = 512).
__global__ __launch_bounds__(256) void augment_kernel(double* __restrict__ K, int64_t ld, int N,
                                                      int Np, const double* __restrict__ y,
                                                      int64_t k_bs, int64_t y_bs, int y_mod, int* __restrict__ info) {
  if (info != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) info[blockIdx.z] = 0;
  K += (int64_t)blockIdx.z * k_bs;
  y += (int64_t)(y_mod > 0 ? blockIdx.z % y_mod : blockIdx.z) * y_bs;
  const int i = N + blockIdx.y;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < Np; j += gridDim.x * 256) {
    double v;
    if (i == N)
      v = (j < N) ? y[j] : (j == N ? AUG_BIG : 0.0);
    else
      v = (j == i) ? 1.0 : 0.0;
    K[(int64_t)i * ld + j] = v;
  }
}

int launch_augment(gpx_ctx* ctx, double* dK, int64_t ld, int N, int Np, const double* dy, int batch,
                   int64_t k_bs, int64_t y_bs, int y_mod, int* dInfo) {
  dim3 grid(min(64, (Np + 255) / 256), Np - N, batch > 1 ? batch : 1);
  augment_kernel<<<grid, 256, 0, ctx->s>>>(dK, ld, N, Np, dy, k_bs, y_bs, y_mod, dInfo);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// Identity padding of rows/cols n .. np-1 of a square matrix (lower part is what matters).
__global__ __launch_bounds__(256) void pad_identity_kernel(double* __restrict__ A, int64_t ld,
                                                           int n, int np) {
  const int i = blockIdx.y;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < np; j += gridDim.x * 256) {
    if (i >= n || j >= n) A[(int64_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
  }
}

int launch_pad_identity(gpx_ctx* ctx, double* dA, int64_t ld, int n, int np) {
  if (np == n) return 0;
  dim3 grid(min(64, (np + 255) / 256), np);
  pad_identity_kernel<<<grid, 256, 0, ctx->s>>>(dA, ld, n, np);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

} // namespace gpx
