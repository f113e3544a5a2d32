// potf2_slim.h — the 128 x 128 diagonal-block factor + inverse (potf2.hip) in a footprint that is PLACED AT ONCE beside two
// resident trailing-update workgroups: at most 112 VGPRs per lane and 28.2 KB of LDS.
//
// Why (round 4): the blocked Cholesky under gpax/models/gp.py:160-164 launches this kernel once per 128 columns, on the
// critical path of the panel chain, while the big trailing SYRK (gemm_nt128_kernel<1,1>: 196 -> 200 VGPRs, 64 KB LDS, two
// workgroups per CU) fills the chip.  What those two leave free on a CU is 512 - 2 x 200 = 112 registers per SIMD lane and
// 160 - 128 = 32 KB of LDS.  The wave-specialised kernel of round 3 (potf2_chain.h: 344 VGPRs, 46 KB) fits in neither, so
// the dispatcher can only place it on a CU that has drained completely: 35 us stand-alone became 115 us (builder's box)
// to 190 us (driver's box) inside the pipeline — the whole spread of the round-3 headline.  The hardware hands a freed
// slot to the queue whose next workgroup FITS; stream priority does not change that.  So the kernel is made to fit.
//
// How: the state of the algorithm is 64 tiles of 16 x 16 doubles (36 Cholesky + 28 inverse-residual) = 128 KB, more
// than 4 waves x 64 lanes x 112 registers.  Same wave specialisation and the same MFMA sequences as potf2_chain.h
// (chain wave 0: diag16(p) -> L(p+1,p) -> D(p+1) -> diag16(p+1) out of LDS; workers 1 - 3: everything else), but the
// tiles are MEMORY-resident: the Cholesky tiles stay in the block A itself, the residual tiles R(i,c) of the forward
// substitution L X = I in the lower tiles of the output Linv (128 + 128 KB, L2).  After panel P a worker VISITS each of
// its tiles that the panel touches — load (coalesced: accumulator layout = four 128-B rows per instruction; a
// residual tile's first visit starts from zero without a load), four MFMAs against the LDS column buffer, store —
// in chunks of three with the next chunk's loads in flight.  A tile is always visited by the same wave, so program
// order is all the coherence it needs.  The chain wave never touches memory between its start tiles and its outputs;
// what a worker needs right after a barrier (its column tiles in A-operand layout, its residual row in accumulator
// = B-operand layout) it loads before it.  The panel loop is a real loop and the visit lists are tables in constant
// memory: the code is small and the register allocation is what the chunks need, not what a fully unrolled schedule
// lets the scheduler hoist (the unrolled form of this design spilled 900 registers under the 112 cap).
// The inverse row X(P, c < P) shares the LDS column buffer with L(i > P, P): slot c is dead as an L tile once panel c
// is over.  13 LDS tiles instead of 21.
// Arithmetic: tile for tile the MFMA sequences of potf2_tile_body / potf2_chain_body (k ascending, the same operands,
// the same signs; a double stored to memory and reloaded is the same double), so L, L^-1 and the pivots are
// bit-identical to both (tests/test_gpu_edges.py).
#pragma once
#include <utility>

#include "potf2_tile.h"

namespace gpx {

constexpr int PSL_LCOL = 0, PSL_DINV = 8, PSL_DG = 10, PSL_S1 = 11, PSL_S2 = 12, PSL_TILES = 13;
constexpr size_t POTF2_SLIM_LDS = (size_t)(PSL_TILES * TSZ + 64) * sizeof(double); // + scratch of the diagonal-tile factor

// ---- phase trace (debug builds only: make -C gpax_amd/csrc trace -> libgpx_trace.so) -----------------------------------
// 100 MHz wall-clock stamps of every wave at its barriers, one record of 4 x 40 stamps per launch in a ring of 512:
// what tools/potf2_trace.py reads to tell placement wait, contention on the chain wave and waiting for the workers apart.
#ifdef GPX_POTF2_TRACE
constexpr int SLIM_TRACE_RING = 512, SLIM_TRACE_STAMPS = 40;
__device__ long long gpx_slim_trace[SLIM_TRACE_RING * 4 * SLIM_TRACE_STAMPS];
__device__ unsigned gpx_slim_trace_count[4];
#define GPX_SLIM_TRACE_BEGIN(w)                                                                                  \
  long long* trc_ = nullptr;                                                                                      \
  {                                                                                                               \
    unsigned id_ = 0;                                                                                             \
    if (lane == 0) id_ = atomicAdd(&gpx_slim_trace_count[(w)], 1u);                                              \
    id_ = __builtin_amdgcn_readfirstlane(id_);                                                                    \
    trc_ = gpx_slim_trace + ((size_t)(id_ % SLIM_TRACE_RING) * 4 + (w)) * SLIM_TRACE_STAMPS;                      \
  }
#define GPX_SLIM_STAMP(k)                                   \
  do {                                                      \
    if (lane == 0) trc_[(k)] = (long long)wall_clock64();   \
  } while (0)
#else
#define GPX_SLIM_TRACE_BEGIN(w) do { } while (0)
#define GPX_SLIM_STAMP(k) do { } while (0)
#endif

// ---- who visits what, when (compile-time tables) ---------------------------------------------------------------------------
// kind 0: Cholesky tile (i, j) in A;  kind 1: residual tile (i, c) in Linv.  Owner: (i + j) % 3 — the tiles of every
// window and of every solve spread evenly over the three workers, and a tile never changes hands.
struct SlimVisit {
  unsigned char kind, i, j, flags; // flags: 1 first visit (starts from zero, no load) | 2 result also to S1 | 4 to S2
};
constexpr int PSL_CHUNK = 3, PSL_MAXUPD = 12, PSL_NCHUNK = PSL_MAXUPD / PSL_CHUNK, PSL_MAXSOL = 3;
struct SlimTables {
  unsigned char nupd[3][8];
  SlimVisit upd[3][8][PSL_MAXUPD]; // window after panel P: C(i,j) -= L(i,P) L(j,P)^T, R(i,c) -= L(i,P) X(P,c)
  unsigned char nsol[3][8];
  SlimVisit sol[3][8][PSL_MAXSOL]; // panel P: kind 0: C(i,P) -> L(i,P) (i > P + 1); kind 1: R(P,c) -> X(P,c) (c < P)
  int max_upd, max_sol;
};
constexpr SlimTables make_slim_tables() {
  SlimTables t{};
  for (int P = 0; P < 8; ++P) {
    for (int pass = 0; pass < 2; ++pass) // the two tiles the chain reads next go first
      for (int j = P + 1; j < 8; ++j)
        for (int i = j; i < 8; ++i) {
          if (i == P + 1 && j == P + 1) continue; // the chain makes the next diagonal tile
          const bool s1 = (i == P + 2 && j == P + 1), s2 = (i == P + 2 && j == P + 2);
          if ((pass == 0) != (s1 || s2)) continue;
          const int w = (i + j) % 3;
          if (t.nupd[w][P] < PSL_MAXUPD)
            t.upd[w][P][t.nupd[w][P]] = SlimVisit{0, (unsigned char)i, (unsigned char)j, (unsigned char)((s1 ? 2 : 0) | (s2 ? 4 : 0))};
          t.nupd[w][P]++;
        }
    for (int c = 0; c <= P; ++c)
      for (int i = P + 1; i < 8; ++i) {
        const int w = (i + c) % 3;
        if (t.nupd[w][P] < PSL_MAXUPD)
          t.upd[w][P][t.nupd[w][P]] = SlimVisit{1, (unsigned char)i, (unsigned char)c, (unsigned char)(c == P ? 1 : 0)};
        t.nupd[w][P]++;
      }
    for (int i = P + 2; i < 8; ++i) {
      const int w = (i + P) % 3;
      if (t.nsol[w][P] < PSL_MAXSOL) t.sol[w][P][t.nsol[w][P]] = SlimVisit{0, (unsigned char)i, (unsigned char)P, 0};
      t.nsol[w][P]++;
    }
    for (int c = 0; c < P; ++c) {
      const int w = (P + c) % 3;
      if (t.nsol[w][P] < PSL_MAXSOL) t.sol[w][P][t.nsol[w][P]] = SlimVisit{1, (unsigned char)P, (unsigned char)c, 0};
      t.nsol[w][P]++;
    }
  }
  for (int w = 0; w < 3; ++w)
    for (int P = 0; P < 8; ++P) {
      if (t.nupd[w][P] > t.max_upd) t.max_upd = t.nupd[w][P];
      if (t.nsol[w][P] > t.max_sol) t.max_sol = t.nsol[w][P];
    }
  return t;
}
constexpr SlimTables SLIM_TABLES_HOST = make_slim_tables();
static_assert(SLIM_TABLES_HOST.max_upd <= PSL_MAXUPD, "update visit table overflows");
static_assert(SLIM_TABLES_HOST.max_sol <= PSL_MAXSOL, "solve visit table overflows");
__device__ __constant__ const SlimTables slim_tab = make_slim_tables();

// ---- tile <-> memory ----------------------------------------------------------------------------------------------------
// Buffer addressing: one resource for the block A, one for Linv; an access is (resource, 32-bit lane offset in a VGPR,
// scalar tile offset) — three offset VGPRs per wave in all and no address arithmetic on the vector ALU.  (With flat
// global addressing the compiler folds the constant tile offsets into 64-bit VGPR pointers: two registers per tile
// row in flight, +30 VGPRs.)
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
struct SlimOff {
  __amdgpu_buffer_rsrc_t ra, rl;
  unsigned a;   // accumulator (= B-operand) layout in A:    row lane >> 4 (+ 4 r), column lane & 15
  unsigned l;   // the same in Linv (leading dimension PB)
  unsigned aop; // A-operand layout in A: row lane & 15, column lane >> 4 (+ 4 kk)
  unsigned lda8; // bytes per row of A
};
__device__ __forceinline__ SlimOff slim_offsets(double* A, int64_t lda, double* Linv, int lane) {
  SlimOff o;
  o.ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  o.rl = __builtin_amdgcn_make_buffer_rsrc((void*)Linv, 0, PB * PB * 8, 0x00020000);
  o.lda8 = (unsigned)lda * 8u;
  o.a = (unsigned)(lane >> 4) * o.lda8 + (unsigned)(lane & 15) * 8u;
  o.l = (unsigned)((lane >> 4) * PB + (lane & 15)) * 8u;
  o.aop = (unsigned)(lane & 15) * o.lda8 + (unsigned)(lane >> 4) * 8u;
  return o;
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double x) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, x), rs, (int)voff, (int)soff, 0);
}
struct InA {    // tile (i, j) of the block A
  int i, j;
};
struct InLinv { // tile (i, j) of Linv
  int i, j;
};
__device__ __forceinline__ pd4_t mem_to_acc(const SlimOff& o, InA t) {
  pd4_t a;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = buf_ld(o.ra, o.a, (unsigned)(t.i * TS + 4 * r) * o.lda8 + (unsigned)(t.j * TS * 8));
  return a;
}
__device__ __forceinline__ pd4_t mem_to_acc(const SlimOff& o, InLinv t) {
  pd4_t a;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = buf_ld(o.rl, o.l, (unsigned)(((t.i * TS + 4 * r) * PB + t.j * TS) * 8));
  return a;
}
__device__ __forceinline__ void acc_to_mem(const pd4_t& a, const SlimOff& o, InA t) {
#pragma unroll
  for (int r = 0; r < 4; ++r) buf_st(o.ra, o.a, (unsigned)(t.i * TS + 4 * r) * o.lda8 + (unsigned)(t.j * TS * 8), a[r]);
}
__device__ __forceinline__ void acc_to_mem(const pd4_t& a, const SlimOff& o, InLinv t) {
#pragma unroll
  for (int r = 0; r < 4; ++r) buf_st(o.rl, o.l, (unsigned)(((t.i * TS + 4 * r) * PB + t.j * TS) * 8), a[r]);
}
__device__ __forceinline__ pd4_t mem_to_aop(const SlimOff& o, InA t) { // a[kk] = T[fr][fk + 4 kk]
  pd4_t a;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) a[kk] = buf_ld(o.ra, o.aop, (unsigned)(t.i * TS) * o.lda8 + (unsigned)((t.j * TS + 4 * kk) * 8));
  return a;
}

// ---- workers -----------------------------------------------------------------------------------------------------------
// Everything below is straight-line code per (worker, panel): the visit lists are constant expressions.  What keeps it
// inside the register budget is the ORDER it is written in — one chunk of three tiles in registers, the next chunk's
// loads in flight, one k-step of LDS operands ahead of the MFMAs — pinned with sched_barrier: left to itself the
// scheduler hoists every load of a window to its top.
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
#define GPX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// what solve(P) works on, loaded before the barrier it follows: column tiles C(i,P) as A operands, the residual row
// R(P,c) as B operands (both final since this wave's own stores of the window before)
template <int W, int P>
__device__ __forceinline__ void slim_prefetch(pd4_t (&pre)[PSL_MAXSOL], const double* A, int64_t lda, const double* Linv,
                                              const SlimOff& o) {
  constexpr int n = SLIM_TABLES_HOST.nsol[W - 1][P];
  static_for<n>([&](auto S) {
    constexpr int s = S;
    constexpr SlimVisit v = SLIM_TABLES_HOST.sol[W - 1][P][s];
    if constexpr (v.kind == 0) pre[s] = mem_to_aop(o, InA{v.i, P});
    else pre[s] = mem_to_acc(o, InLinv{P, v.j});
  });
}

// window B1(P) .. B2(P): this worker's column-P tiles become L(i,P) = C(i,P) Dinv^T (i > P + 1; the chain makes i = P + 1),
// its row-P residual tiles become X(P,c) = Dinv_P R(P,c)
template <int W, int P>
__device__ __forceinline__ void slim_solve(pd4_t (&pre)[PSL_MAXSOL], double* lds, double* A, int64_t lda, double* Linv,
                                           const SlimOff& o, int lane) {
  double* Lcol = lds + PSL_LCOL * TSZ; // slot i > P: L(i,P); slot c < P: X(P,c)
  const double* Dinv = lds + (PSL_DINV + (P & 1)) * TSZ;
  const int ont = (lane & 15) * TLD + (lane >> 4);
  constexpr int n = SLIM_TABLES_HOST.nsol[W - 1][P];
  pd4_t res[PSL_MAXSOL];
  static_for<n>([&](auto S) { res[S] = pd4_t{0.0, 0.0, 0.0, 0.0}; });
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double dv = Dinv[ont + 4 * kk];
    static_for<n>([&](auto S) {
      constexpr int s = S;
      constexpr SlimVisit v = SLIM_TABLES_HOST.sol[W - 1][P][s];
      if constexpr (v.kind == 0) res[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(pre[s][kk], dv, res[s], 0, 0, 0);
      else res[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(dv, pre[s][kk], res[s], 0, 0, 0);
    });
  }
  static_for<n>([&](auto S) {
    constexpr int s = S;
    constexpr SlimVisit v = SLIM_TABLES_HOST.sol[W - 1][P][s];
    if constexpr (v.kind == 0) {
      acc_to_lds(res[s], Lcol + v.i * TSZ, lane);
      acc_to_mem(res[s], o, InA{v.i, P});
    } else {
      acc_to_lds(res[s], Lcol + v.j * TSZ, lane);
      acc_to_mem(res[s], o, InLinv{P, v.j});
    }
  });
}

template <int W, int P, int C>
__device__ __forceinline__ void slim_chunk_load(pd4_t (&acc)[PSL_CHUNK], const double* A, int64_t lda, const double* Linv,
                                                const SlimOff& o) {
  constexpr int n = SLIM_TABLES_HOST.nupd[W - 1][P];
  constexpr int m = (n - C * PSL_CHUNK) < PSL_CHUNK ? (n - C * PSL_CHUNK) : PSL_CHUNK;
  static_for<m>([&](auto Q) {
    constexpr int q = Q;
    constexpr SlimVisit v = SLIM_TABLES_HOST.upd[W - 1][P][C * PSL_CHUNK + q];
    if constexpr ((v.flags & 1) != 0) acc[q] = pd4_t{0.0, 0.0, 0.0, 0.0};
    else if constexpr (v.kind == 0) acc[q] = mem_to_acc(o, InA{v.i, v.j});
    else acc[q] = mem_to_acc(o, InLinv{v.i, v.j});
  });
}

// LDS operands of chunk C, k-step kk: a = -L(i,P), b = L(j,P) (NT) or X(P,c) (NN; X(P,P) = Dinv_P)
template <int W, int P, int C>
__device__ __forceinline__ void slim_chunk_ops(double (&a)[PSL_CHUNK], double (&b)[PSL_CHUNK], int kk, const double* lds, int ont,
                                               int onn) {
  const double* Lcol = lds + PSL_LCOL * TSZ;
  const double* Dinv = lds + (PSL_DINV + (P & 1)) * TSZ;
  constexpr int n = SLIM_TABLES_HOST.nupd[W - 1][P];
  constexpr int m = (n - C * PSL_CHUNK) < PSL_CHUNK ? (n - C * PSL_CHUNK) : PSL_CHUNK;
  static_for<m>([&](auto Q) {
    constexpr int q = Q;
    constexpr SlimVisit v = SLIM_TABLES_HOST.upd[W - 1][P][C * PSL_CHUNK + q];
    a[q] = -1.0 * Lcol[v.i * TSZ + ont + 4 * kk];
    if constexpr (v.kind == 0) b[q] = Lcol[v.j * TSZ + ont + 4 * kk];
    else if constexpr (v.j == P) b[q] = Dinv[onn + 4 * kk * TLD];
    else b[q] = Lcol[v.j * TSZ + onn + 4 * kk * TLD];
  });
}

template <int W, int P, int C>
__device__ __forceinline__ void slim_chunks(pd4_t (&cur)[PSL_CHUNK], pd4_t (&nxt)[PSL_CHUNK], double* lds, double* A, int64_t lda,
                                            double* Linv, const SlimOff& o, int lane, int ont, int onn) {
  constexpr int n = SLIM_TABLES_HOST.nupd[W - 1][P];
  constexpr int nch = (n + PSL_CHUNK - 1) / PSL_CHUNK;
  constexpr int m = (n - C * PSL_CHUNK) < PSL_CHUNK ? (n - C * PSL_CHUNK) : PSL_CHUNK;
  if constexpr (C + 1 < nch) slim_chunk_load<W, P, C + 1>(nxt, A, lda, Linv, o);
  double a[2][PSL_CHUNK], b[2][PSL_CHUNK];
  slim_chunk_ops<W, P, C>(a[0], b[0], 0, lds, ont, onn);
  GPX_SCHED_FENCE();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (kk < 3) slim_chunk_ops<W, P, C>(a[(kk + 1) & 1], b[(kk + 1) & 1], kk + 1, lds, ont, onn);
    static_for<m>([&](auto Q) {
      constexpr int q = Q;
      cur[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][q], b[kk & 1][q], cur[q], 0, 0, 0);
    });
    GPX_SCHED_FENCE();
  }
  static_for<m>([&](auto Q) {
    constexpr int q = Q;
    constexpr SlimVisit v = SLIM_TABLES_HOST.upd[W - 1][P][C * PSL_CHUNK + q];
    if constexpr (v.kind == 0) acc_to_mem(cur[q], o, InA{v.i, v.j});
    else acc_to_mem(cur[q], o, InLinv{v.i, v.j});
    if constexpr ((v.flags & 2) != 0) acc_to_lds(cur[q], lds + PSL_S1 * TSZ, lane);
    if constexpr ((v.flags & 4) != 0) acc_to_lds(cur[q], lds + PSL_S2 * TSZ, lane);
  });
  GPX_SCHED_FENCE();
  if constexpr (C + 1 < nch) slim_chunks<W, P, C + 1>(nxt, cur, lds, A, lda, Linv, o, lane, ont, onn);
}

// window B2(P) .. B1(P+1): C(i,j) -= L(i,P) L(j,P)^T (i >= j > P, the next diagonal tile excepted: the chain made it),
// R(i,c) -= L(i,P) X(P,c) (i > P, c <= P, X(P,P) = Dinv_P); the two tiles the chain reads next also go to S1 / S2
template <int W, int P>
__device__ __forceinline__ void slim_update(double* lds, double* A, int64_t lda, double* Linv, const SlimOff& o, int lane) {
  const int ont = (lane & 15) * TLD + (lane >> 4), onn = (lane >> 4) * TLD + (lane & 15);
  constexpr int n = SLIM_TABLES_HOST.nupd[W - 1][P];
  if constexpr (n > 0) {
    pd4_t cur[PSL_CHUNK], nxt[PSL_CHUNK];
    slim_chunk_load<W, P, 0>(cur, A, lda, Linv, o);
    GPX_SCHED_FENCE();
    slim_chunks<W, P, 0>(cur, nxt, lds, A, lda, Linv, o, lane, ont, onn);
  }
}

#ifdef GPX_POTF2_TRACE
#define GPX_SLIM_TRC_ARG , long long* trc_
#define GPX_SLIM_TRC_PASS , trc_
#else
#define GPX_SLIM_TRC_ARG
#define GPX_SLIM_TRC_PASS
#endif
template <int W, int P>
__device__ __forceinline__ void slim_panels(pd4_t (&pre)[PSL_MAXSOL], double* lds, double* A, int64_t lda, double* Linv,
                                            const SlimOff& o, int lane GPX_SLIM_TRC_ARG) {
  GPX_SLIM_STAMP(1 + 4 * P); // arrives at B1(P): its update window (and the prefetch) is done
  __syncthreads(); // B1(P): Dinv_P is there
  GPX_SLIM_STAMP(2 + 4 * P);
  slim_solve<W, P>(pre, lds, A, lda, Linv, o, lane);
  GPX_SLIM_STAMP(3 + 4 * P); // arrives at B2(P)
  __syncthreads(); // B2(P): every L(i,P), X(P,c) is there
  GPX_SLIM_STAMP(4 + 4 * P);
  if constexpr (P < 7) {
    slim_update<W, P>(lds, A, lda, Linv, o, lane);
    slim_prefetch<W, P + 1>(pre, A, lda, Linv, o);
    slim_panels<W, P + 1>(pre, lds, A, lda, Linv, o, lane GPX_SLIM_TRC_PASS);
  }
}

template <int W>
__device__ __forceinline__ void potf2_slim_worker(double* A, int64_t lda, double* Linv, double* lds, int lane) {
  const int crow = lane >> 4, ccol = lane & 15;
  GPX_SLIM_TRACE_BEGIN(W);
  GPX_SLIM_STAMP(0);
  const SlimOff o = slim_offsets(A, lda, Linv, lane);
  pd4_t pre[PSL_MAXSOL];
  slim_prefetch<W, 0>(pre, A, lda, Linv, o);
  // zero the strictly-upper 16 x 16 tiles of both outputs (the diagonal tiles are written whole by the chain)
#pragma unroll 1
  for (int i = 1; i < 8; ++i) {
#pragma unroll 1
    for (int c = 0; c < i; ++c) {
      if ((i + c) % 3 != W - 1) continue; // tile (c, i) is above the diagonal
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        A[(int64_t)(c * TS + crow + 4 * r) * lda + i * TS + ccol] = 0.0;
        Linv[(c * TS + crow + 4 * r) * PB + i * TS + ccol] = 0.0;
      }
    }
  }
  slim_panels<W, 0>(pre, lds, A, lda, Linv, o, lane GPX_SLIM_TRC_PASS);
  GPX_SLIM_STAMP(33);
}

// ---- the chain wave (potf2_chain.h's, on this kernel's LDS map, as a loop) --------------------------------------------------
// n_active: order of the leading part that holds data — the block is the identity from there on (fit_small.hip: N + 1 of 128).
// A diagonal tile that lies wholly in the identity part factors into itself: its 16 dependent column steps (2.4 us of the
// chain wave's time per tile) are skipped; everything else — barriers, T / U, the workers — runs as always.
__device__ __forceinline__ void potf2_slim_chain_wave(double* A, int64_t lda, double* Linv, int* info, int info_base, double* lds,
                                                      int lane, int n_active = PB) {
  double* Dg = lds + PSL_DG * TSZ;
  double* col = lds + PSL_TILES * TSZ;
  GPX_SLIM_TRACE_BEGIN(0);
  GPX_SLIM_STAMP(0);
  const SlimOff o = slim_offsets(A, lda, Linv, lane);
  // the three tiles the chain starts from: C(0,0) -> Dg, C(1,0) -> S1, C(1,1) -> S2
  {
    const pd4_t t00 = mem_to_acc(o, InA{0, 0});
    const pd4_t t10 = mem_to_acc(o, InA{1, 0});
    const pd4_t t11 = mem_to_acc(o, InA{1, 1});
    acc_to_lds(t00, Dg, lane);
    acc_to_lds(t10, lds + PSL_S1 * TSZ, lane);
    acc_to_lds(t11, lds + PSL_S2 * TSZ, lane);
  }
  int bad = 0;
#pragma unroll 1
  for (int P = 0; P < 8; ++P) {
    double* Dinv = lds + (PSL_DINV + (P & 1)) * TSZ;
    if (P * TS < n_active) {
      diag16(Dg, Dinv, col, lane, bad, P * TS);
    } else { // identity tile: L = L^-1 = I
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * q + t;
        Dg[r * TLD + i] = (r == i) ? 1.0 : 0.0;
        Dinv[i * TLD + r] = (r == i) ? 1.0 : 0.0;
      }
    }
    {
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * q + t;
        A[(int64_t)(P * TS + r) * lda + P * TS + i] = Dg[r * TLD + i];
        Linv[(P * TS + r) * PB + P * TS + i] = Dinv[r * TLD + i];
      }
    }
    GPX_SLIM_STAMP(1 + 4 * P); // diag16(P) done
    __syncthreads(); // B1(P)
    GPX_SLIM_STAMP(2 + 4 * P);
    if (P < 7) {
      double* Lnext = lds + (PSL_LCOL + P + 1) * TSZ;
      // T: L(P+1,P) = C(P+1,P) Dinv^T
      const pd4_t x = mma_nt(pd4_t{0.0, 0.0, 0.0, 0.0}, lds + PSL_S1 * TSZ, Dinv, lane, 1.0);
      acc_to_lds(x, Lnext, lane);
      acc_to_mem(x, o, InA{P + 1, P});
      // U: the next diagonal tile
      pd4_t d = lds_to_acc(lds + PSL_S2 * TSZ, lane);
      d = mma_nt(d, Lnext, Lnext, lane, -1.0);
      acc_to_lds(d, Dg, lane);
    }
    GPX_SLIM_STAMP(3 + 4 * P); // T / U done
    __syncthreads(); // B2(P)
    GPX_SLIM_STAMP(4 + 4 * P);
  }
  GPX_SLIM_STAMP(33);
  if (lane == 0 && bad != 0 && info != nullptr) {
    if (*info == 0) *info = info_base + bad;
  }
}

// 256 threads: wave 0 = chain, waves 1 - 3 = workers.  Every wave passes the same 16 barriers.
__device__ __forceinline__ void potf2_slim_body(double* A, int64_t lda, double* Linv, int* info, int info_base, double* lds,
                                                int n_active = PB) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (w == 0) potf2_slim_chain_wave(A, lda, Linv, info, info_base, lds, lane, n_active);
  else if (w == 1) potf2_slim_worker<1>(A, lda, Linv, lds, lane);
  else if (w == 2) potf2_slim_worker<2>(A, lda, Linv, lds, lane);
  else potf2_slim_worker<3>(A, lda, Linv, lds, lane);
}

} // namespace gpx
