// potf2.hip — 128x128 diagonal-block Cholesky fused with the inverse of the factor, one workgroup (gfx950).
//
// Role on the path: the serial kernel of the right-looking blocked Cholesky that replaces
// jnp.linalg.cholesky inside NumPyro's MultivariateNormal (gpax/models/gp.py:160-164,292).
// The inverse of the diagonal block turns every panel TRSM into an MFMA GEMM.
// A non-positive pivot makes the factor NaN from there on (propagates, like JAX) and sets *info.
//
// Two kernels, one arithmetic (bit-identical outputs, tests/test_gpu_edges.py):
//   potf2_slim_kernel  (potf2_slim.h, round 4, default)  94 VGPRs, 28 KB LDS: placed at once beside two resident
//                      trailing-update workgroups; tiles memory-resident, visited in chunks
//   potf2_tile_kernel  (potf2_tile.h, round 2, GPX_POTF2=tile)    the four-phase form, the reference of the tests
// (Round 3's wave-specialised kernel with every tile in registers — 344 VGPRs, 46 KB, needs a drained CU — left the
// library in round 5: tools/exp/potf2_chain.h.)
// (Round 1's column-by-column kernel and round 2's blocked 16 x 16 diagonal factor were measured slower and removed:
// profiles/r02/chain_experiments.md.)
#include "common.h"
#include "potf2_tile.h"
#include "potf2_slim.h"

// =================================================================================================
// potf2_tile_kernel — the 128x128 factor + inverse blocked at 16x16 tiles so that all
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

__global__ __launch_bounds__(256, 1) void potf2_tile_kernel(double* A, int64_t lda, double* Linv, int* info,
                                                            int info_base, int64_t a_bs, int64_t linv_bs) {
  A += (int64_t)blockIdx.x * a_bs; // one workgroup per batch entry
  Linv += (int64_t)blockIdx.x * linv_bs;
  if (info != nullptr) info += blockIdx.x;
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  potf2_tile_body(A, lda, Linv, info, info_base, lds);
}

// the placeable form (potf2_slim.h): <= 112 VGPRs, 28.2 KB LDS — fits beside two resident trailing-update workgroups;
// bit-identical to the kernel above.  amdgpu_num_vgpr keeps the allocator out of the AGPR half of the unified
// register file (without it: ~90 VGPRs + 48 AGPRs allocated); tests/test_abi.py checks the emitted counts.
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(112))) void potf2_slim_kernel(double* A, int64_t lda, double* Linv,
                                                                                           int* info, int info_base,
                                                                                           int64_t a_bs, int64_t linv_bs) {
  A += (int64_t)blockIdx.x * a_bs; // one workgroup per batch entry
  Linv += (int64_t)blockIdx.x * linv_bs;
  if (info != nullptr) info += blockIdx.x;
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  potf2_slim_body(A, lda, Linv, info, info_base, lds);
}

} // namespace gpx

namespace gpx {
int launch_potf2_inv(gpx_ctx* ctx, double* dA, int64_t lda, double* dLinv, int* dInfo, int info_base,
                     int batch, int64_t a_bs, int64_t linv_bs) {
  const int nb = batch > 1 ? batch : 1;
  {
    // algorithmic flops: factor n^3/3 + triangular inverse n^3/3
    ProfScope ps(ctx, GPX_PROF_POTF2, nb * 2.0 * PB * (double)PB * PB / 3.0);
    if (ctx->potf2_mode == GPX_POTF2_SLIM)
      potf2_slim_kernel<<<nb, 256, POTF2_SLIM_LDS, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, a_bs, linv_bs);
    else
      potf2_tile_kernel<<<nb, 256, POTF2_TILE_LDS, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, a_bs, linv_bs);
  }
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}
} // namespace gpx

#ifdef GPX_POTF2_TRACE
// debug build only (make trace): copies the phase-trace ring of potf2_slim.h to the host; returns the number of launches traced
extern "C" int gpx_debug_slim_trace(long long* out, int cap_records, unsigned* counts) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  unsigned c[4];
  if (hipMemcpyFromSymbol(c, HIP_SYMBOL(gpx::gpx_slim_trace_count), sizeof c) != hipSuccess) return -1;
  if (counts) for (int i = 0; i < 4; ++i) counts[i] = c[i];
  const int n = cap_records < gpx::SLIM_TRACE_RING ? cap_records : gpx::SLIM_TRACE_RING;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpx::gpx_slim_trace), (size_t)n * 4 * gpx::SLIM_TRACE_STAMPS * sizeof(long long)) != hipSuccess) return -1;
  return (int)c[0];
}
#endif
