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
#include "gemm_tile.h"
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

// Diagonal block AND panel TRSM of one step of the chain in ONE launch (round 6; a step of the one-outer-block
// factorisation is then potf2+trsm -> update: two launches instead of three).  Workgroup 0 is potf2_slim_kernel; the others
// own one 16-row strip of the panel below the diagonal block each, are dispatched with it, and wait — one lane polling a
// flag in device memory — until workgroup 0 has published L^-1 of the block: agent-scope release behind its last store,
// agent-scope acquire in every strip workgroup before its first load of L^-1 (the strips run on other XCDs, whose L2 is
// not the one workgroup 0 wrote through).  Workgroup 0 is dispatched first (workgroup ids are handed out in order), so the
// strips never wait for a workgroup that is not resident.  `epoch`: the value this launch publishes — the flag is reused
// by every step of a context, launches of one stream are ordered, and each publishes a value no earlier launch has.
//
// What the strips do while they wait is what makes this worth a kernel (the same strips as a plain rider — lat_tile behind
// the flag — measured level with the three-launch step: profiles/r06/potf2_trsm.md): the strip's own operand, 16 rows x 128
// of the panel, is read BEFORE the flag, straight into registers in MFMA fragment order (lane l: row l & 15, k = 4 kk +
// (l >> 4), 32 doubles); after the flag only L^-1 is missing, and it comes in ONE piece — 128 rows x 1 KB by LDS-direct
// loads, all issued at once, one exposed latency (the stand-alone strip kernel walks 16 k-slices through a ring of two:
// 16 exposed latencies, 10 us of the step).  LDS rows are unpadded 1-KB lines with the 16-B chunk index XOR-ed with
// (row & 15), applied to the SOURCE address (LDS-DMA writes lane-linearly) and to the fragment reads: the 16 rows a
// fragment read touches sit at 16 different chunk positions of the 256-B bank row — conflict-free ds_read_b64.
// Arithmetic: every element accumulates k = 0 .. 127 ascending from 0 in v_mfma_f64_16x16x4_f64 steps and is stored as
// alpha * acc — what lat_tile<1, 4, 2, 0> and every other shape do: the same bits.
#ifndef GPX_CHAIN_ACQUIRE
#define GPX_CHAIN_ACQUIRE 0
#endif
constexpr size_t POTF2_TRSM_LDS = (size_t)PB * PB * sizeof(double); // L^-1 whole (128 KB) >= POTF2_SLIM_LDS

#ifdef GPX_POTF2_TRACE
// trace build (make trace; tools/potf2_trsm_trace.py): 100 MHz wall-clock stamps per launch, slot epoch % ring:
// [0] workgroup 0 starts  [1] its factorisation is done  [2] flag published  [3] / [4] first / last strip sees the flag
// [5] last strip has L^-1 in LDS  [6] last strip has stored  [7] strips
constexpr int CHAIN_TRACE_RING = 1024;
__device__ long long gpx_chain_trace[CHAIN_TRACE_RING * 8];
#define GPX_CHAIN_STAMP(k) do { if (threadIdx.x == 0) trc_[(k)] = (long long)wall_clock64(); } while (0)
#define GPX_CHAIN_MIN(k) do { if (threadIdx.x == 0) atomicMin((unsigned long long*)&trc_[(k)], (unsigned long long)wall_clock64()); } while (0)
#define GPX_CHAIN_MAX(k) do { if (threadIdx.x == 0) atomicMax((unsigned long long*)&trc_[(k)], (unsigned long long)wall_clock64()); } while (0)
#else
#define GPX_CHAIN_STAMP(k) do { } while (0)
#define GPX_CHAIN_MIN(k) do { } while (0)
#define GPX_CHAIN_MAX(k) do { } while (0)
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void potf2_trsm_kernel(double* A, int64_t lda, double* Linv,
                                                                                           int* info, int info_base,
                                                                                           GemmArgs g, unsigned* flag,
                                                                                           unsigned epoch) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double lds[];
#ifdef GPX_POTF2_TRACE
  long long* trc_ = gpx_chain_trace + (size_t)(epoch % CHAIN_TRACE_RING) * 8;
#endif
  if (blockIdx.x == 0) {
    GPX_CHAIN_STAMP(0);
    potf2_slim_body(A, lda, Linv, info, info_base, lds);
    __syncthreads(); // every wave's stores of L and L^-1 are issued and counted (the barrier's workgroup-scope fence)
    GPX_CHAIN_STAMP(1);
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    GPX_CHAIN_STAMP(2);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  // in place (C == A): this workgroup's 16 rows, all of k, before the flag
  double* P = g.C + ((int64_t)(blockIdx.x - 1) * 16) * g.ldc;
  double af[32];
  {
    const double* Ap = P + (int64_t)fr * g.ldc + fk;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) af[kk] = Ap[4 * kk];
  }
  // every wave holds the strip's rows in registers before any wave may store into them (the only barrier of a strip
  // workgroup, and it is passed while workgroup 0 still factors)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // every wave polls for itself and goes on alone: no barrier behind the flag
  if (lane == 0)
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
  GPX_CHAIN_MIN(3);
  GPX_CHAIN_MAX(4);
#if GPX_CHAIN_ACQUIRE
  // the textbook form: every wave acquires at agent scope in front of its loads of L^-1.  On this chip that is a
  // `buffer_inv sc1` per wave, i.e. each of up to ~1000 waves wipes the L2 of its XCD, and every strip workgroup then
  // fetches its 128 KB of L^-1 from memory — the same 1024 lines for all of them: 12 us for 248 strips, level with the
  // three-launch step (profiles/r06/potf2_trsm.md).
  (void)__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#else
  // No cache invalidate here, and none is needed: the L2 of this XCD cannot hold a line of THIS block's L^-1.  Every L2 is
  // invalidated when a kernel starts (what makes the panel written by one launch visible to the next on another XCD),
  // nothing in this launch reads L^-1 before the flag, and workgroup 0's release has written its lines back to memory —
  // the first strip of an XCD misses and fetches them, the others hit.  (The poll itself is an agent-scope atomic load: it
  // bypasses the L2.)  The loads below are issued behind the poll that saw the flag (a GPU does not speculate past the
  // branch), so nothing is left to order but the compiler:
  asm volatile("" ::: "memory");
#endif
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, 0x7fffffff, 0x00020000);
  // wave w computes columns [32 w, 32 w + 32) and needs rows 32 w .. 32 w + 31 of L^-1, nothing else: it brings them in
  // itself (one wave-instruction = 64 lanes x 16 B = one row) and waits for nobody — its first 16 x 16 accumulator starts
  // when the first 16 rows have landed, behind them the other 16 are still in flight
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int row = 32 * wave + r;
    const int voff = (row * PB + ((lane ^ (row & 15)) * 2)) * 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(lds + row * PB), 16, voff, 0, 0, 0);
  }
  d4_t acc[2] = {d4_t{0.0, 0.0, 0.0, 0.0}, d4_t{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    if (n == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (n == 1) GPX_CHAIN_MAX(5);
    const int j = wave * 32 + n * 16 + fr;
    const int jb = j * PB + (fk & 1), jx = j & 15;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      const double bf = *((lds_cvd_t)lds + (jb + ((2 * kk + (fk >> 1)) ^ jx) * 2));
      acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf, acc[n], 0, 0, 0);
    }
  }
  const double alpha = g.alpha;
  double* Cw = P + (int64_t)fk * g.ldc + wave * 32 + fr;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int n = 0; n < 2; ++n) Cw[(int64_t)(4 * r) * g.ldc + n * 16] = alpha * acc[n][r];
#ifdef GPX_POTF2_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  GPX_CHAIN_MAX(6);
}

} // namespace gpx

namespace gpx {
// potf2 of the diagonal block + the in-place panel TRSM of the `below` 128-row tiles under it (g: the TRSM as
// launch_gemm_nt would get it — A = C = the panel, B = L^-1 of the block, K = 128, alpha = 1, beta = 0), one launch
int launch_potf2_trsm(gpx_ctx* ctx, double* dA, int64_t lda, double* dLinv, int* dInfo, int info_base, const GemmArgs& g0,
                      int below) {
  // one flag (and epoch counter) per stream of the context: launches of ONE stream are ordered, so each finds the flag at a
  // value only earlier launches of that stream have published; two chains of one context on two streams (the sparse path
  // can run them so) must not publish through the same word
  constexpr int SLOTS = 3, SLOT_STRIDE = 64; // bytes
  if (!ctx->chain_flag_zeroed) { // once per context
    if (ctx->chain_flag.ensure(SLOTS * SLOT_STRIDE) != hipSuccess) return bad_arg(ctx, "chain flag");
    GPX_HIP(ctx, hipMemsetAsync(ctx->chain_flag.p, 0, SLOTS * SLOT_STRIDE, ctx->s));
    GPX_HIP(ctx, hipStreamSynchronize(ctx->s)); // done before any other stream of the context uses it
    ctx->chain_flag_zeroed = true;
  }
  if (!ctx->chain_attr_set) { // 128 KB of dynamic LDS: above the default limit, per device (this context's)
    GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_trsm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)POTF2_TRSM_LDS));
    ctx->chain_attr_set = true;
  }
  const int slot = (ctx->s == ctx->stream) ? 0 : ((ctx->s == ctx->pstream) ? 1 : 2);
  GemmArgs g = g0;
  g.nsplit = 1;
  g.batch = 1;
  ctx->chain_epoch[slot] += 1;
  if (ctx->chain_epoch[slot] == 0) ctx->chain_epoch[slot] = 1; // (0 is the cleared flag)
  const unsigned epoch = ctx->chain_epoch[slot];
  unsigned* flag = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->chain_flag.p) + slot * SLOT_STRIDE);
#ifdef GPX_POTF2_TRACE
  {
    long long init[8] = {0, 0, 0, 0x7fffffffffffffffLL, 0, 0, 0, 8LL * below};
    GPX_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(gpx_chain_trace), init, sizeof init,
                                        (size_t)(epoch % CHAIN_TRACE_RING) * 8 * sizeof(long long), hipMemcpyHostToDevice, ctx->s));
  }
#endif
  {
    ProfScope ps(ctx, GPX_PROF_POTF2, 2.0 * PB * (double)PB * PB / 3.0 + 2.0 * below * TILE * (double)TILE * TILE);
    potf2_trsm_kernel<<<1 + 8 * below, 256, POTF2_TRSM_LDS, ctx->s>>>(dA, lda, dLinv, dInfo, info_base, g,
                                                                     flag, epoch);
  }
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

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
// debug build only (make trace): the stamps of the last `cap_records` potf2_trsm launches (8 long long each, slot = epoch % ring)
extern "C" int gpx_debug_chain_trace(long long* out, int cap_records) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  const int n = cap_records < gpx::CHAIN_TRACE_RING ? cap_records : gpx::CHAIN_TRACE_RING;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpx::gpx_chain_trace), (size_t)n * 8 * sizeof(long long)) != hipSuccess) return -1;
  return n;
}
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
