// panel.hip — ONE cooperative kernel for the panel chain of an outer block of the blocked Cholesky (linalg.hip):
// the diagonal-block factorisations, the panel TRSMs and the updates inside the block's own columns, which the driver
// otherwise issues as 3 dependent launches per diagonal block (potf2 -> TRSM -> inner update; 12 per 512-column block).
//
// Role on the path: jnp.linalg.cholesky underneath NumPyro's MultivariateNormal (gpax/models/gp.py:160-164,292).  In
// the chain-bound tail of a factorisation (profiles/r02/timeline_c3.md: the last ~15 outer blocks at N = 16384, 23 %
// of the time for 10 % of the flops) the trailing update no longer hides that chain; what bounds it is the latency of
// launches and stream-to-stream events between ~70 us kernels.  Here the dependencies are flags in device memory.
//
// Work decomposition for outer block [ob, oe) (nd = oe - ob diagonal blocks), tile rows ob .. nrows-1:
//   role 0                      diagonal row ob: P(ob)  (potf2_tile_body: L_kk and its inverse)
//   roles 1 + 4 (q - 1) + s     diagonal row r = ob + q (q = 1 .. nd-1), 32-row strip s = 0 .. 3 of that tile row:
//                               for k = ob .. r-1:  wait P(k);  T: strip(r,k) <- strip(r,k) Linv_k^T;
//                                                   for j = k+1 .. r:  wait L(j,k);  U: strip(r,j) -= strip(r,k) L(j,k)^T
//                               strip 0 then waits for its three siblings and runs P(r)
//   roles nd_roles + f          row r = oe + f below the block (the k_pX ride-along rows included): the same sweep on
//                               whole 128 x 128 tiles (nt128_tile), k = ob .. oe-1, j = k+1 .. oe-1
// The critical path P(k) -> T(k+1,k) -> U(k+1,k+1,k) -> P(k+1) runs on FOUR workgroups per diagonal row (a 128^3
// product is 15 us of MFMA time on one CU, 4 us on four), everything else fills in beside it.
// Arithmetic: the very tile bodies of the launched kernels (gemm_tile.h, potf2_tile.h) with K = 128 per update and
// accumulators that start from -C: a tile's updates form the same fma chain, so not a bit differs from the launches
// (tests/test_gpu_edges.py).
//
// Coherence.  The 8 XCDs of an MI355X have private L2s that are made coherent at kernel boundaries only, so workgroups
// that exchange tiles INSIDE a kernel must share one L2.  Workgroup b of a dispatch runs on XCD b % 8 (checked once per
// device by panel_probe; a CU mask cannot confine a queue to one XCD: an empty per-XCD mask means "all CUs",
// profiles/r02/cumask.log).  The kernel is launched with 8 G workgroups; those that find themselves on an XCD other
// than 0 return at once, the G on XCD 0 take roles from a ticket counter.  Producer: stores, s_waitcnt vmcnt(0) (the
// vector L1 is write-through: the data is in L2), barrier, relaxed agent-scope atomic store of the flag.  Consumer:
// relaxed agent-scope atomic loads until the flag shows this launch's epoch, barrier, buffer_inv sc1 (drop the L1).
// No L2 write-back or invalidate is ever needed — which is what an agent-scope fence would cost on this chip.
// Progress: roles are taken in an order in which every wait targets an EARLIER role, except between the four strips of
// one diagonal row; G >= 13 workgroups arrive on XCD 0 by the dispatch rule and all fit there (2 per CU), so every
// role that is waited for is held by a running workgroup.  Every spin is bounded: on time-out the fail flag is raised,
// every workgroup leaves, and the factorisation reports a failed pivot (NaN outputs) instead of hanging the GPU.
#include "common.h"
#include "gemm_tile.h"
#include "potf2_tile.h"

#include <cstdlib>

namespace gpx {

namespace {

constexpr int PS_TICKETS = 256;         // ring of ticket counters, one per launch epoch
constexpr int PS_FAIL = PS_TICKETS;     // raised on a spin time-out / a workgroup on the wrong XCD
constexpr int PS_RAN = PS_TICKETS + 1;  // launches that completed role 0 (tests)
constexpr int PS_FLAGP = PS_TICKETS + 8;             // 8 ints: P(ob + c) done
constexpr int PS_FLAGT = PS_FLAGP + 8;               // [q][s][c]: strip s of diagonal row q holds L(ob+q, ob+c)
constexpr int PS_FLAGD = PS_FLAGT + 8 * 4 * 8;       // [q][s]: strip s of diagonal row q has applied all its updates
constexpr int PS_INTS = PS_FLAGD + 8 * 4;
constexpr long long SPIN_LIMIT = 1ll << 22;          // x ~0.3 us sleep: ~1.5 s, then fail

struct PanelArgs {
  double* A;
  int64_t lda;
  double* Linv; // diagonal-block inverses, block kb at Linv + kb * 128 * 128
  int* info;
  int* sync;
  int epoch;
  int ob, oe, nrows; // tile indices: block columns [ob, oe), tile rows ob .. nrows-1 take part
};

__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Everything this workgroup has stored is in L2; the next reads do not come from a stale L1 line.
__device__ __forceinline__ void wg_fence() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  asm volatile("buffer_inv sc1" ::: "memory");
}

__device__ __forceinline__ void wg_release(int* flag, int epoch) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) st_flag(flag, epoch);
}

// Waits until flags[0 .. n) (n <= 4, stride `stride` ints) all show `epoch`.  Returns false when the launch has failed.
__device__ __forceinline__ bool wg_acquire(int* flags, int n, int stride, int epoch, int* sync, int* s_ok) {
  const int t = threadIdx.x;
  if (t == 0) *s_ok = 1;
  __syncthreads();
  if (t < n) {
    const int* p = flags + t * stride;
    bool ok = false;
    for (long long it = 0; it < SPIN_LIMIT; ++it) {
      if (ld_flag(p) == epoch) {
        ok = true;
        break;
      }
      if ((it & 255) == 255 && ld_flag(sync + PS_FAIL) != 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) {
      st_flag(sync + PS_FAIL, 1);
      *s_ok = 0;
    }
  }
  __syncthreads();
  const bool ok = *s_ok != 0;
  asm volatile("buffer_inv sc1" ::: "memory");
  return ok;
}

// a failed launch reports a failed pivot in this outer block: the factorisation's outputs are NaN-filled downstream
__device__ __forceinline__ void mark_failed(const PanelArgs& a) {
  if (threadIdx.x == 0 && a.info != nullptr) atomicCAS(a.info, 0, a.ob * TILE + 1);
}

__device__ __forceinline__ GemmArgs op_args(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                                            double alpha, double beta) {
  GemmArgs g{};
  g.A = A;
  g.lda = lda;
  g.B = B;
  g.ldb = ldb;
  g.C = C;
  g.ldc = ldc;
  g.K = TILE;
  g.alpha = alpha;
  g.beta = beta;
  g.nsplit = 1;
  g.batch = 1;
  return g;
}

// The three tile bodies as real (non-inlined) device functions: inlined into one kernel their live ranges pile up and
// the allocator spills (228 VGPRs to scratch at the 256-register budget of two workgroups per CU); called, each keeps
// the allocation it has in its own kernel.  They address the kernel's dynamic LDS themselves (ds_* instructions, not
// flat ones through a generic pointer).
// A non-inlined function that took generic pointers would address the matrix with flat_* instructions (which also tie
// up the LDS counter): the parameters are typed as GLOBAL pointers, and the address space propagates from there.
typedef __attribute__((address_space(1))) double* gdp_t;
typedef __attribute__((address_space(1))) const double* gcdp_t;
typedef __attribute__((address_space(1))) int* gip_t;

__device__ __noinline__ void panel_potf2(gdp_t A, int64_t lda, gdp_t Linv, gip_t info, int info_base) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  potf2_tile_body<false>((double*)A, lda, (double*)Linv, (int*)info, info_base, nullptr, 0, smem);
}
__device__ __noinline__ void panel_strip(gcdp_t A, int64_t lda, gcdp_t B, int64_t ldb, gdp_t C, int64_t ldc, double alpha,
                                         double beta) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const GemmArgs g = op_args((const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, alpha, beta);
  gemm_nt_tile<1, 4, 16, false>(g, smem, 0, 0, 0);
}
__device__ __noinline__ void panel_tile_trsm(gcdp_t A, int64_t lda, gcdp_t B, int64_t ldb, gdp_t C, int64_t ldc) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const GemmArgs g = op_args((const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, 1.0, 0.0);
  nt128_tile<0>(g, smem, 0, 0, 0);
}
__device__ __noinline__ void panel_tile_update(gcdp_t A, int64_t lda, gcdp_t B, int64_t ldb, gdp_t C, int64_t ldc) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const GemmArgs g = op_args((const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, -1.0, 1.0);
  nt128_tile<1>(g, smem, 0, 0, 0);
}

} // namespace

__global__ __launch_bounds__(256, 2) void panel_chain_kernel(PanelArgs a) {
  __shared__ int s_role, s_ok;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((xcc & 0xf) != 0) return; // tiles are exchanged through ONE L2: only the workgroups of XCD 0 take part
  __builtin_amdgcn_s_setprio(3);
  const int nd = a.oe - a.ob;
  const int nd_roles = 1 + 4 * (nd - 1);
  const int nroles = nd_roles + (a.nrows - a.oe);
  int* const sync = a.sync;
  int* const ticket = sync + (a.epoch & (PS_TICKETS - 1));
  int* const flagP = sync + PS_FLAGP;
  const int64_t lda = a.lda;
  const int epoch = a.epoch;
  for (;;) {
    if (threadIdx.x == 0) s_role = atomicAdd(ticket, 1);
    __syncthreads();
    const int role = s_role;
    __syncthreads();
    if (role >= nroles) return;
    if (role == 0) {
      // the ticket slot of the launch half a ring ahead is ours to clear: its previous user finished long ago
      if (threadIdx.x == 0) st_flag(sync + ((a.epoch + PS_TICKETS / 2) & (PS_TICKETS - 1)), 0);
      panel_potf2((gdp_t)(a.A + (int64_t)a.ob * TILE * lda + (int64_t)a.ob * TILE), lda, (gdp_t)(a.Linv + (int64_t)a.ob * TILE * TILE),
                  (gip_t)a.info, a.ob * TILE);
      wg_release(flagP + 0, epoch);
      if (threadIdx.x == 0) atomicAdd(sync + PS_RAN, 1);
      continue;
    }
    if (role < nd_roles) {
      // ---- strip s of diagonal row q ------------------------------------------------------------------------------
      const int q = 1 + (role - 1) / 4, s = (role - 1) & 3, r = a.ob + q;
      double* const rowbase = a.A + ((int64_t)r * TILE + 32 * s) * lda; // first matrix row of this strip
      bool ok = true;
      for (int c = 0; c < q && ok; ++c) {
        const int k = a.ob + c;
        ok = wg_acquire(flagP + c, 1, 1, epoch, sync, &s_ok);
        if (!ok) break;
        double* const Lrk = rowbase + (int64_t)k * TILE;
        // T: strip(r,k) <- strip(r,k) Linv_k^T, in place (one workgroup owns the whole row width of the strip)
        panel_strip((gcdp_t)Lrk, lda, (gcdp_t)(a.Linv + (int64_t)k * TILE * TILE), TILE, (gdp_t)Lrk, lda, 1.0, 0.0);
        wg_release(sync + PS_FLAGT + (q * 4 + s) * 8 + c, epoch);
        for (int cj = c + 1; cj <= q && ok; ++cj) {
          // B = L(ob + cj, k): the four strips of diagonal row cj (for cj == q: this row's own siblings)
          ok = wg_acquire(sync + PS_FLAGT + (cj * 4) * 8 + c, 4, 8, epoch, sync, &s_ok);
          if (!ok) break;
          const double* Ljk = a.A + (int64_t)(a.ob + cj) * TILE * lda + (int64_t)k * TILE;
          double* Crj = rowbase + (int64_t)(a.ob + cj) * TILE;
          panel_strip((gcdp_t)Lrk, lda, (gcdp_t)Ljk, lda, (gdp_t)Crj, lda, -1.0, 1.0);
          wg_fence(); // the strip just written is read again by this workgroup's next step
        }
      }
      if (!ok) {
        mark_failed(a);
        return;
      }
      wg_release(sync + PS_FLAGD + q * 4 + s, epoch);
      if (s == 0) {
        if (!wg_acquire(sync + PS_FLAGD + q * 4 + 1, 3, 1, epoch, sync, &s_ok)) {
          mark_failed(a);
          return;
        }
        panel_potf2((gdp_t)(a.A + (int64_t)r * TILE * lda + (int64_t)r * TILE), lda, (gdp_t)(a.Linv + (int64_t)r * TILE * TILE), (gip_t)a.info,
                    r * TILE);
        wg_release(flagP + q, epoch);
      }
      continue;
    }
    // ---- a whole tile row below the block ----------------------------------------------------------------------------
    const int r = a.oe + (role - nd_roles);
    double* const rowbase = a.A + (int64_t)r * TILE * lda;
    bool ok = true;
    for (int c = 0; c < nd && ok; ++c) {
      const int k = a.ob + c;
      ok = wg_acquire(flagP + c, 1, 1, epoch, sync, &s_ok);
      if (!ok) break;
      double* const Lrk = rowbase + (int64_t)k * TILE;
      panel_tile_trsm((gcdp_t)Lrk, lda, (gcdp_t)(a.Linv + (int64_t)k * TILE * TILE), TILE, (gdp_t)Lrk, lda);
      wg_fence();
      for (int cj = c + 1; cj < nd && ok; ++cj) {
        ok = wg_acquire(sync + PS_FLAGT + (cj * 4) * 8 + c, 4, 8, epoch, sync, &s_ok);
        if (!ok) break;
        const double* Ljk = a.A + (int64_t)(a.ob + cj) * TILE * lda + (int64_t)k * TILE;
        double* Crj = rowbase + (int64_t)(a.ob + cj) * TILE;
        panel_tile_update((gcdp_t)Lrk, lda, (gcdp_t)Ljk, lda, (gdp_t)Crj, lda);
        wg_fence();
      }
    }
    if (!ok) {
      mark_failed(a);
      return;
    }
  }
}

// ---- once per device: does workgroup b of a dispatch run on XCD b % 8 ? ---------------------------------------------------
__global__ void panel_probe_kernel(int* out) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf);
}

int panel_probe(gpx_ctx* ctx) {
  constexpr int NB = 512;
  int* d = nullptr;
  GPX_HIP(ctx, hipMalloc(&d, NB * sizeof(int)));
  int h[NB];
  bool ok = true;
  for (int rep = 0; rep < 2 && ok; ++rep) {
    panel_probe_kernel<<<NB, 64, 0, ctx->pstream>>>(d);
    if (hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->pstream) != hipSuccess ||
        hipStreamSynchronize(ctx->pstream) != hipSuccess) {
      ok = false;
      break;
    }
    // the XCD of workgroup b may depend on b % 8 only, and the eight classes must sit on eight different XCDs — then
    // exactly NB / 8 workgroups of any dispatch run on XCD 0, whichever class that is
    unsigned seen = 0;
    for (int c = 0; c < 8; ++c) seen |= 1u << (h[c] & 15);
    if (seen != 0xffu) ok = false;
    for (int b = 0; b < NB; ++b)
      if (h[b] != h[b & 7]) ok = false;
    if (getenv("GPX_DEBUG"))
      fprintf(stderr, "[gpx] panel_probe rep %d: xcd of workgroups 0..15 = %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d -> %s\n", rep,
              h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], ok ? "ok" : "rule broken");
  }
  (void)hipFree(d);
  (void)hipGetLastError();
  return ok ? 1 : 0;
}

// One outer block's panel chain as ONE launch on the stream the chain lives on (ctx->s).  Caller: potrf_lower, single
// sample, block in the chain-bound tail.
int launch_panel_chain(gpx_ctx* ctx, double* dA, int64_t lda, int nrows, int ob, int oe, double* dLinv, int* dInfo) {
  const int nd = oe - ob;
  if (nd < 1 || nd > 8 || nrows < oe) return bad_arg(ctx, "panel chain shape");
  if (ctx->panel_sync.p == nullptr) {
    GPX_HIP(ctx, ctx->panel_sync.ensure(PS_INTS * sizeof(int)));
    GPX_HIP(ctx, hipMemsetAsync(ctx->panel_sync.p, 0, PS_INTS * sizeof(int), ctx->s));
    constexpr size_t lds = (size_t)2 * 256 * 16 * sizeof(double);
    GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panel_chain_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  ctx->panel_epoch += 1;
  if (ctx->panel_epoch <= 0) ctx->panel_epoch = 1;
  PanelArgs a{dA, lda, dLinv, dInfo, ctx->panel_sync.i(), ctx->panel_epoch, ob, oe, nrows};
  const int nroles = 1 + 4 * (nd - 1) + (nrows - oe);
  int G = nroles < 64 ? nroles : 64; // workgroups wanted on XCD 0 (2 per CU there)
  if (G < 16) G = 16;
  // algorithmic flops of the chain: nd factor + inverse blocks, the TRSMs and the updates inside the block's columns
  double work = nd * 2.0 * TILE * (double)TILE * TILE / 3.0;
  for (int c = 0; c < nd; ++c) {
    const double below = nrows - (ob + c) - 1;
    work += 2.0 * below * TILE * (double)TILE * TILE;
    for (int cj = c + 1; cj < nd; ++cj) work += 2.0 * (nrows - (ob + cj)) * TILE * (double)TILE * TILE;
  }
  ProfScope ps(ctx, GPX_PROF_PANEL, work);
  constexpr size_t lds = (size_t)2 * 256 * 16 * sizeof(double);
  panel_chain_kernel<<<8 * G, 256, lds, ctx->s>>>(a);
  GPX_HIP(ctx, hipGetLastError());
  ctx->panel_launches += 1;
  return 0;
}

// tests: launches issued by this context, launches whose role 0 ran on the device, and the device fail flag
int panel_stats(gpx_ctx* ctx, int64_t* launches, int* ran, int* failed) {
  if (launches) *launches = ctx->panel_launches;
  int h[2] = {0, 0};
  if (ctx->panel_sync.p != nullptr) {
    GPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GPX_HIP(ctx, hipMemcpy(h, ctx->panel_sync.i() + PS_FAIL, sizeof h, hipMemcpyDeviceToHost));
  }
  if (failed) *failed = h[0];
  if (ran) *ran = h[1];
  return 0;
}

} // namespace gpx
