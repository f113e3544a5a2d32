// gemm_f64.hip — fp64 MFMA "NT" GEMM tile kernels for gfx950:
//     C[i][j] = beta * C[i][j] + alpha * sum_k A[i][k] * B[j][k]        (row-major everywhere)
//
// These kernels carry every dense contraction on the exact-GP path (the work JAX hands to
// LAPACK/cuSOLVER underneath gpax/models/gp.py:160-164,271-273,292): the Cholesky trailing
// update (SYRK form, lower tiles only), the panel TRSM (multiplication by the inverted 128x128
// diagonal block), the right-looking TRSM sweeps of the posterior, the split-K SYRK of the
// posterior covariance, L^-T L^-1 for the gradient, and the MVN draw.
//
// Two kernels, one arithmetic:
//   gemm_nt128_kernel  THROUGHPUT shape, 128x128 block tile, BK = 16, 256 threads = 2x2 waves, each wave a
//       64x64 sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators.  LDS-direct staging: every wave issues
//       eight `buffer_load_dwordx4 ... lds` per k-step (1 KiB = 8 rows x 128 B each; loop-invariant 32-bit
//       voffsets, the k advance in an SGPR soffset, destination in M0 — no VALU in the loop), SPREAD over the
//       first three of the four 16-MFMA blocks of the step (3 + 3 + 2, pinned with sched_group_barrier):
//       bunched at the top of the step the same loads cost 6 % of the loop (r02 harness, tools/exp/gemm_r2.hip:
//       68.2 -> 75.1 TFLOP/s on zero operands at K = 2048 = 95 % of the MFMA peak).  LDS rows are unpadded
//       128-B lines with an XOR swizzle on the 16-B chunk index, chunk' = chunk ^ ((row >> 1) & 7), applied to
//       the SOURCE address (LDS-DMA writes lane-linearly) and to the fragment reads: each fragment is ONE
//       conflict-free ds_read_b64.  2 buffers x 32 KB = 64 KB + < 256 VGPRs => 2 workgroups / CU.
//   gemm_nt_kernel     LATENCY shapes (64x64, 64x128) for grids too small to fill the chip; register staging,
//       LDS rows padded to 17 doubles (round 1).
// Both accumulate every C element over its k range in ascending k, from the same start value, with the same
// epilogue — so the choice of shape never changes a result bit (batched == single-sample launches):
//   beta == 0               acc starts at 0,  C = alpha * acc
//   alpha == -1, beta == 1  acc starts at -C (the tile is read in the prologue, while the first k-tile is in
//                           flight, instead of read-modify-written in the epilogue), C = -acc
//   otherwise               acc starts at 0,  C = fma(beta, C, alpha * acc)
//
// f64 MFMA fragment layout (differs from the f32 forms!):
//   A: lane l holds A[i = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][j = l & 15];
//   D: lane l, register r holds D[row = (l >> 4) + 4 r][col = l & 15].
#include "common.h"
#include "gemm_tile.h"

#include <cstdlib>

namespace gpx {

// Batched launches: which XCD works on which batch entry.  The hardware deals workgroups to the 8 XCDs in linear id order
// (id % 8), x fastest: with the batch in grid.z the tiles of ONE entry are consecutive ids, i.e. spread over all eight L2s,
// and every XCD fetches that entry's A / B panels for itself — at N = 512 a K = 128 update of 1024 samples moved ~7.7 GB
// (C 2.7 GB + the panels up to eight times) at 4.7 TB/s: HBM-bound on re-fetched operands.  Entries are independent and
// identical, so the order is free: within each group of eight entries, id -> (entry = id % 8, tile = id / 8) puts ALL tiles
// of an entry on one XCD, back to back (its panels are fetched once into that L2), and keeps the eight XCDs exactly
// balanced.  A tile's arithmetic does not depend on where it runs.  (Entries beyond the last full group of eight keep the
// plain order.)
__device__ __forceinline__ void batch_xcd_order(int& bx, int& by, int& bz) {
  const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  if (gz < 8) return;
  const int per = gx * gy;
  const int lin = bx + gx * (by + gy * bz);
  const int grp = lin / (8 * per);
  if ((grp + 1) * 8 > gz) return; // ragged tail
  const int loc = lin - grp * 8 * per;
  const int t = loc >> 3;
  bz = grp * 8 + (loc & 7);
  by = t / gx;
  bx = t - by * gx;
}

// Tile enumeration of a persistent launch: only the tiles a launch really has (lower: tj_off + bx <= ti_off + by),
// row-major (consecutive ids share their A row panel), one slab per (batch entry, split-K slab).
struct TileMap {
  int tiles_m, tiles_n, per_slab, total;
  int a, f0, b, tri; // rows a .. b-1 hold f0, f0 + 1, ... tiles (the triangular part, `tri` tiles), rows >= b tiles_n each
  int col_desc;      // full grid, column by column from the LAST column (1): the order that hands out the long tiles first when
                     // the k range ends at the column tile (kupper) — row-major order would start the longest tiles last;
                     // 2: from the FIRST column (kcol: the k range starts at the column tile)
};

// slab-local tile id -> (by, bx)
__host__ __device__ __forceinline__ void decode_tile(const TileMap& tm, int lower, int l, int& by, int& bx) {
  if (!lower) {
    if (tm.col_desc) {
      const int c = l / tm.tiles_m;
      by = l - c * tm.tiles_m;
      bx = tm.col_desc == 2 ? c : tm.tiles_n - 1 - c;
    } else {
      by = l / tm.tiles_n;
      bx = l - by * tm.tiles_n;
    }
  } else if (l < tm.tri) { // n = rows before `by` in the triangular part: n f0 + n (n - 1) / 2 <= l
    const double q = 2.0 * tm.f0 - 1.0;
    int n = (int)((-q + sqrt(q * q + 8.0 * l)) * 0.5);
    while ((int64_t)(n + 1) * tm.f0 + (int64_t)(n + 1) * n / 2 <= l) ++n;
    while ((int64_t)n * tm.f0 + (int64_t)n * (n - 1) / 2 > l) --n;
    by = tm.a + n;
    bx = l - (int)((int64_t)n * tm.f0 + (int64_t)n * (n - 1) / 2);
  } else {
    l -= tm.tri;
    by = tm.b + l / tm.tiles_n;
    bx = l - (l / tm.tiles_n) * tm.tiles_n;
  }
}

// grid.x of a live-tiles launch -> (bx, by), wave-uniform
__device__ __forceinline__ void live_tile(const TileMap& tm, int lower, int& bx, int& by) {
  int ty, tx;
  decode_tile(tm, lower, bx, ty, tx);
  bx = __builtin_amdgcn_readfirstlane(tx);
  by = __builtin_amdgcn_readfirstlane(ty);
}

template <int TAG, int MT, int NT, int BK, bool DBUF, int EPI>
__global__ __launch_bounds__(256, (MT * NT >= 16) ? 2 : ((MT * NT >= 8) ? 3 : (BK == 32 ? 4 : 5))) void gemm_nt_kernel(GemmArgs g, TileMap tm) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (TAG == 0) __builtin_amdgcn_s_setprio(2); // panel / small GEMMs sit on the critical path of the look-ahead
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.nsplit == 1) batch_xcd_order(bx, by, bz);
  if (tm.total > 0) live_tile(tm, g.lower, bx, by); // (as in gemm_lat_kernel below)
  gemm_nt_tile<MT, NT, BK, DBUF, EPI>(g, smem, bx, by, bz);
}

// round-5 latency shapes (gemm_tile.h lat_tile): <2,2> 64 x 64 (ring of 3 k-slices, 24 KB), <1,4> 32 x 128 strips for the
// in-place panel TRSM (ring of 2, 20 KB); <= 80 VGPRs: six waves per SIMD alone, one beside two trailing-update workgroups
// tm.total > 0: blockIdx.x runs over a LIST of the launch's tiles (plain_launch_map below, TileMap in units of this shape's
// tiles).  Lower launches list only their live tiles.  Not because the workgroups above the diagonal cost time — they return
// at once, and the chip hands out > 2000 empty workgroups per us (tools/exp/dispatch_probe.hip) — but because they take part
// in the deal: workgroup id -> XCD id % 8, so in a square grid of even width the live tiles of column bx all land on XCD
// bx % 8, and a triangle has more tiles in its low columns: XCD 0 holds 21 % more live tiles than the average at 16 x 16
// tile rows (41 % at 8 x 8), 17 % more work in K^-1 = L^-T L^-1 at 32 x 32 — and a launch lasts as long as its fullest
// XCD.  A list deals the live tiles themselves round.
template <int MT, int NT, int NST, int EPI>
__global__ __launch_bounds__(256, 6) void gemm_lat_kernel(GemmArgs g, TileMap tm) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __builtin_amdgcn_s_setprio(2); // chain launches sit on the critical path of the look-ahead
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.nsplit == 1) batch_xcd_order(bx, by, bz);
  if (tm.total > 0) live_tile(tm, g.lower, bx, by);
  lat_tile<MT, NT, NST, EPI>(g, smem, bx, by, bz);
}

template <int TAG, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt128_kernel(GemmArgs g, TileMap tm) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (TAG == 0) __builtin_amdgcn_s_setprio(2); // non-trailing launches sit on the critical path of the look-ahead
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.nsplit == 1) batch_xcd_order(bx, by, bz);
  if (tm.total > 0) live_tile(tm, g.lower, bx, by); // (as in gemm_lat_kernel)
  nt128_tile<EPI>(g, smem, bx, by, bz);
}

// PERSISTENT, dynamically scheduled variant: the grid is two workgroups per CU; each takes the next tile from an atomic
// counter until none is left.  A tile's arithmetic does not depend on who computes it, so results are those of
// gemm_nt128_kernel bit for bit.  Used where a launch's tiles differ widely in length (k ranges trimmed to a triangle)
// and where the panel chain of the driver holds no kernel that needs a drained CU (ctx->persist_scope).
typedef const __attribute__((address_space(4))) char* kernarg_ptr_t;

template <int TAG, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt128_persist_kernel(GemmArgs g_by_value, TileMap tm_by_value,
                                                                    int* __restrict__ counter) {
  constexpr int TD = 256 * 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (TAG == 0) __builtin_amdgcn_s_setprio(2);
  int* s_tile = reinterpret_cast<int*>(smem + 2 * TD); // 16 B behind the two k-tile buffers (one LDS object only)
  // The argument block (50 + 10 dwords) is re-read from the kernarg segment for every tile — scalar loads out of the
  // constant cache, nothing against a tile of >= 10 us — instead of living in SGPRs across the tile loop: held there it took
  // 33 - 39 of them to VGPR lanes (v_writelane / v_readlane around every use; rounds 2 - 5).  The pointer is made opaque per
  // iteration so that the loads are not hoisted back out of the loop.
  (void)g_by_value;
  (void)tm_by_value;
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass of hipcc parses kernel bodies too and has no constant address space)
  kernarg_ptr_t ka = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  for (;;) {
    if (threadIdx.x == 0) *s_tile = atomicAdd(counter, 1);
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(*s_tile);
    asm volatile("" : "+s"(ka));
    const TileMap tm = *reinterpret_cast<const __attribute__((address_space(4))) TileMap*>(ka + sizeof(GemmArgs));
    if (t >= tm.total) return;
    const GemmArgs g = *reinterpret_cast<const __attribute__((address_space(4))) GemmArgs*>(ka);
    const int bzz = t / tm.per_slab;
    int by, bx;
    decode_tile(tm, g.lower, t - bzz * tm.per_slab, by, bx);
    nt128_tile<EPI>(g, smem, __builtin_amdgcn_readfirstlane(bx), __builtin_amdgcn_readfirstlane(by), bzz);
    __syncthreads(); // every wave is done with the k-tile buffers and has read *s_tile
  }
#endif
}

// tiles of a (tiles_m x tiles_n) grid; lower: row by holds clamp(by + delta + 1, 0, tiles_n) tiles (delta = ti_off - tj_off)
static TileMap make_tile_map2(int lower, int delta, int tiles_m, int tiles_n, int slabs) {
  TileMap tm{};
  tm.tiles_m = tiles_m;
  tm.tiles_n = tiles_n;
  if (tiles_n <= 0 || tiles_m <= 0) {
    tm.per_slab = tm.total = 0;
    return tm;
  }
  if (!lower) {
    tm.per_slab = tiles_m * tiles_n;
  } else {
    int a = delta < 0 ? -delta : 0;
    if (a > tiles_m) a = tiles_m;
    int b = tiles_n - delta - 1;
    if (b < a) b = a;
    if (b > tiles_m) b = tiles_m;
    tm.a = a;
    tm.b = b;
    tm.f0 = a + delta + 1;
    const int64_t n = b - a;
    tm.tri = (int)(n * tm.f0 + n * (n - 1) / 2);
    tm.per_slab = tm.tri + (tiles_m - b) * tiles_n;
  }
  tm.total = tm.per_slab * slabs;
  return tm;
}
// The grid of a plain (non-persistent) launch as a list: lower launches list only their live tiles (gemm_lat_kernel above:
// the deal to the XCDs then is even); launches whose k range
// ends (kupper) or starts (kcol) at the column tile list their tiles column by column from the longest column.  The second
// is about WHICH tiles share a CU, not about order in time (a chain-size launch is resident all at once): workgroup id ->
// XCD id % 8, then the XCD's CUs in turn, so in a row-major grid of 32 tile columns the four workgroups of a CU are tiles
// of the same eight columns — four long k ranges on one CU, four short ones on another (the K = 2048 product of the top
// L^-T tree level at N = 4096: 249 us against 164 for its row-trimmed twin of the same flop).  Column-major, a CU's tiles
// are eight columns apart.  unit: rows of the launch's tile per 128 (2 for the 64 x 64 shapes).  total == 0: plain grid.
static TileMap plain_launch_map(const gpx_ctx* ctx, const GemmArgs& g, int tiles_m, int tiles_n, int unit) {
  TileMap tm{};
  if (!ctx->lat_lin || (int64_t)tiles_m * tiles_n * g.nsplit * g.batch < 16) return tm;
  if (g.lower) {
    tm = make_tile_map2(1, unit * (g.ti_off - g.tj_off), unit * tiles_m, unit * tiles_n, 1);
  } else if ((g.kupper || g.kcol) && !g.ktri) {
    tm = make_tile_map2(0, 0, unit * tiles_m, unit * tiles_n, 1);
    tm.col_desc = g.kupper ? 1 : 2;
  }
  return tm;
}

static TileMap make_tile_map(const GemmArgs& g, int tiles_m, int tiles_n) {
  TileMap tm = make_tile_map2(g.lower, g.ti_off - g.tj_off, tiles_m, tiles_n, g.nsplit * g.batch);
  tm.col_desc = (!g.lower && g.kupper && !g.ktri) ? 1 : ((!g.lower && g.kcol && !g.ktri) ? 2 : 0);
  return tm;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a function: every context sets it once
// for each variant it launches (a process may hold contexts on several GPUs; contexts are also driven from
// different host threads, so the "done" bits live in the context, not in a function-local static).
enum : unsigned { ATTR_BIG_BASE = 2 /* + 3 * TAG + EPI */, ATTR_PERSIST_BASE = 8 /* + 3 * TAG + EPI */,
                  ATTR_SMALL_22 = 14 /* + EPI */, ATTR_SMALL_24 = 17, ATTR_SMALL_22_BK32 = 22, ATTR_SMALL_24_BK32 = 25 };

template <int TAG, int MT, int NT, int BK, bool DBUF, int EPI>
static int launch_variant_epi(gpx_ctx* ctx, const GemmArgs& g, int tiles_m, int tiles_n) {
  constexpr size_t lds = gemm_lds_bytes<MT, NT, BK, DBUF>();
  constexpr unsigned bit = 1u << (EPI + (BK == 32 ? ((NT == 2) ? ATTR_SMALL_22_BK32 : ATTR_SMALL_24_BK32) : ((NT == 2) ? ATTR_SMALL_22 : ATTR_SMALL_24)));
  if (!(ctx->func_attr_mask & bit)) {
    GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<TAG, MT, NT, BK, DBUF, EPI>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->func_attr_mask |= bit;
  }
  // tiles_m / tiles_n are given in 128-tiles
  dim3 grid(tiles_n * (4 / NT), tiles_m * (4 / MT), g.nsplit * g.batch);
  TileMap tm{};
  if (MT == 2 && NT == 2) tm = plain_launch_map(ctx, g, tiles_m, tiles_n, 2);
  if (tm.total > 0) grid = dim3(tm.total, 1, g.nsplit * g.batch);
  else if (g.lower && tm.per_slab == 0 && tm.tiles_m > 0) return 0; // (a lower launch without a live tile)
  gemm_nt_kernel<TAG, MT, NT, BK, DBUF, EPI><<<grid, 256, lds, ctx->s>>>(g, tm);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

// the epilogue form is a template parameter of the kernel (gemm_tile.h gemm_nt_tile): chosen here, per launch
template <int TAG, int MT, int NT, int BK, bool DBUF>
static int launch_variant(gpx_ctx* ctx, const GemmArgs& g0, int tiles_m, int tiles_n, int splits) {
  GemmArgs g = g0;
  g.nsplit = splits > 0 ? splits : 1;
  if (g.batch < 1) g.batch = 1;
  if (g.beta == 0.0) return launch_variant_epi<TAG, MT, NT, BK, DBUF, 0>(ctx, g, tiles_m, tiles_n);
  if (g.alpha == -1.0 && g.beta == 1.0) return launch_variant_epi<TAG, MT, NT, BK, DBUF, 1>(ctx, g, tiles_m, tiles_n);
  return launch_variant_epi<TAG, MT, NT, BK, DBUF, 2>(ctx, g, tiles_m, tiles_n);
}

template <int MT, int NT, int NST, int EPI>
static int launch_lat_epi(gpx_ctx* ctx, const GemmArgs& g, int tiles_m, int tiles_n) {
  constexpr size_t lds = (size_t)NST * (32 * MT + 32 * NT) * 8 * sizeof(double); // < 48 KB: no attribute needed
  dim3 grid(tiles_n * (4 / NT), tiles_m * (4 / MT), g.nsplit * g.batch);
  TileMap tm{};
  if (MT == 2 && NT == 2) tm = plain_launch_map(ctx, g, tiles_m, tiles_n, 2); // (a grid of a few dozen workgroups is placed at once either way)
  if (tm.total > 0) grid = dim3(tm.total, 1, g.nsplit * g.batch);
  else if (g.lower && tm.per_slab == 0 && tm.tiles_m > 0) return 0;
  gemm_lat_kernel<MT, NT, NST, EPI><<<grid, 256, lds, ctx->s>>>(g, tm);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

template <int MT, int NT, int NST>
static int launch_lat(gpx_ctx* ctx, const GemmArgs& g0, int tiles_m, int tiles_n, int splits) {
  GemmArgs g = g0;
  g.nsplit = splits > 0 ? splits : 1;
  if (g.batch < 1) g.batch = 1;
  if (g.beta == 0.0) return launch_lat_epi<MT, NT, NST, 0>(ctx, g, tiles_m, tiles_n);
  if (g.alpha == -1.0 && g.beta == 1.0) return launch_lat_epi<MT, NT, NST, 1>(ctx, g, tiles_m, tiles_n);
  return launch_lat_epi<MT, NT, NST, 2>(ctx, g, tiles_m, tiles_n);
}

template <int TAG, int EPI>
static int launch_big_epi(gpx_ctx* ctx, const GemmArgs& g, int tiles_m, int tiles_n) {
  constexpr size_t lds = (size_t)2 * 256 * 16 * sizeof(double); // 2 buffers x (128 A rows + 128 B rows) x 128 B
  constexpr unsigned bit = 1u << (ATTR_BIG_BASE + 3 * TAG + EPI);
  if (ctx->persist_scope > 0) {
    constexpr unsigned pbit = 1u << (ATTR_PERSIST_BASE + 3 * TAG + EPI);
    if (!(ctx->func_attr_mask & pbit)) {
      GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt128_persist_kernel<TAG, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 16)));
      ctx->func_attr_mask |= pbit;
    }
    const TileMap tm = make_tile_map(g, tiles_m, tiles_n);
    if (tm.total <= 0) return 0;
    // one counter per launch out of a ring: zeroed in stream order right before the launch that uses it
    if (ctx->tile_counters.ensure(GPX_TILE_COUNTERS * sizeof(int)) != hipSuccess) return bad_arg(ctx, "tile counters");
    // A slot comes round again after GPX_TILE_COUNTERS launches — possibly on ANOTHER stream of this context, where
    // stream order says nothing about the previous user: wait for the event that user recorded behind its kernel.
    const unsigned slot = ctx->tile_counter_seq++ % GPX_TILE_COUNTERS;
    if (ctx->tile_counter_ev.empty()) {
      ctx->tile_counter_ev.assign(GPX_TILE_COUNTERS, nullptr);
      ctx->tile_counter_stream.assign(GPX_TILE_COUNTERS, nullptr);
    }
    if (ctx->tile_counter_ev[slot] != nullptr && ctx->tile_counter_stream[slot] != ctx->s)
      GPX_HIP(ctx, hipStreamWaitEvent(ctx->s, ctx->tile_counter_ev[slot], 0));
    int* counter = ctx->tile_counters.i() + slot;
    GPX_HIP(ctx, hipMemsetAsync(counter, 0, sizeof(int), ctx->s));
    const int slots = 2 * ctx->prop.multiProcessorCount;
    const int grid = tm.total < slots ? tm.total : slots;
    gemm_nt128_persist_kernel<TAG, EPI><<<grid, 256, lds + 16, ctx->s>>>(g, tm, counter);
    GPX_HIP(ctx, hipGetLastError());
    if (ctx->tile_counter_ev[slot] == nullptr)
      GPX_HIP(ctx, hipEventCreateWithFlags(&ctx->tile_counter_ev[slot], hipEventDisableTiming));
    GPX_HIP(ctx, hipEventRecord(ctx->tile_counter_ev[slot], ctx->s));
    ctx->tile_counter_stream[slot] = ctx->s;
    return 0;
  }
  if (!(ctx->func_attr_mask & bit)) {
    GPX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt128_kernel<TAG, EPI>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->func_attr_mask |= bit;
  }
  dim3 grid(tiles_n, tiles_m, g.nsplit * g.batch);
  TileMap tm = plain_launch_map(ctx, g, tiles_m, tiles_n, 1);
  if (tm.total > 0) grid = dim3(tm.total, 1, g.nsplit * g.batch);
  else if (g.lower && tm.per_slab == 0 && tm.tiles_m > 0) return 0;
  gemm_nt128_kernel<TAG, EPI><<<grid, 256, lds, ctx->s>>>(g, tm);
  GPX_HIP(ctx, hipGetLastError());
  return 0;
}

template <int TAG>
static int launch_big(gpx_ctx* ctx, const GemmArgs& g0, int tiles_m, int tiles_n, int splits) {
  GemmArgs g = g0;
  g.nsplit = splits > 0 ? splits : 1;
  if (g.batch < 1) g.batch = 1;
  if (g.beta == 0.0) return launch_big_epi<TAG, 0>(ctx, g, tiles_m, tiles_n);
  if (g.alpha == -1.0 && g.beta == 1.0) return launch_big_epi<TAG, 1>(ctx, g, tiles_m, tiles_n);
  return launch_big_epi<TAG, 2>(ctx, g, tiles_m, tiles_n);
}

int launch_gemm_nt(gpx_ctx* ctx, const GemmArgs& g, int tiles_m, int tiles_n, int splits,
                   int prof_cls, double work) {
  if (tiles_m <= 0 || tiles_n <= 0) return 0;
  // In place (C == A: the panel TRSM A <- A B^T): every workgroup tile must own the rows it reads — ONE column tile and a k
  // range no wider than it (K <= 128) — and read its whole k range before its first store.  All three tile bodies do
  // (gemm_tile.h: the accumulators are complete before the epilogue), so whichever shape the tile count below selects —
  // strips, 64 x 64 never (tiles_n == 1 keeps C == A off it), the 128 x 128 or the persistent kernel for batched chains —
  // is safe; anything else in place is refused here rather than left to a race.
  if (g.C == g.A && (tiles_n != 1 || g.K > TILE)) return bad_arg(ctx, "in-place GEMM: one column tile, K <= 128");
  // algorithmic bytes: `work` = 2 K per updated entry => entries = work / (2 K), each read and written once (beta != 0)
  const double bmul_p = (g.batch > 1 ? g.batch : 1);
  ProfScope ps(ctx, prof_cls, work * bmul_p, (g.K > 0 ? work / (2.0 * g.K) : 0.0) * (g.beta != 0.0 ? 16.0 : 8.0) * bmul_p);
  if (prof_cls == GPX_PROF_GEMM_TRAILING) return launch_big<1>(ctx, g, tiles_m, tiles_n, splits);
  // latency-bound launches (too few 128x128 tiles to fill 256 CUs x 2): smaller workgroup tiles
  // Workgroups the launch would have with 128x128 tiles (batch entries included).  The choice of shape does
  // not change results: every C element accumulates its k range in the same order in all shapes (the k ranges
  // trimmed by ktri / kupper only drop structural zeros), so batched and single-sample launches stay
  // bit-identical even when they pick different shapes.
  // Batch entries count towards the tile total whatever K where the chain has the chip to itself (round 5).  Rounds 1 - 4 kept the 64 x 64 shapes for the K = 128
  // panel steps of a batched sweep (measured in round 1, against that round's register-staged 128 x 128 kernel: 126 000 vs
  // 50 000 posteriors/s at N = 512); with the LDS-direct kernel the big tile halves the operand traffic per flop of
  // those updates (a 64 x 64 tile of K = 128 reads 128 KB of panels for 64 KB of C): batched sweep N = 256 349 -> 368 k,
  // N = 512 145 -> 160 k, N = 1024 / 2048 +2 % posteriors/s (profiles/r05/README.md).
  const int nsplit = splits > 0 ? splits : 1;
  // (Inside the blocked two-stream sweeps — lat_now == 1 — the K = 128 chain launches keep the shapes that are placed at
  // once beside two resident trailing-update workgroups: C4, batches of 4 - 7 at N = 8192, loses 2 - 3 % otherwise.)
  const int bmul = (g.batch > 1 && (g.K >= 512 || ctx->lat_now == 5)) ? g.batch : 1;
  const double tiles = (double)tiles_m * tiles_n * nsplit * (g.lower ? 0.55 : 1.0) * bmul;
  // persistent scope: the bulk update's workgroups hold their slots until its queue is dry, so a big-shape launch on
  // the panel stream would find no room: everything there takes the shapes that fit next to two resident workgroups
  const bool on_panel = ctx->persist_scope > 0 && ctx->s != ctx->stream;
  // launches with fewer 128x128 tiles take the latency shapes (profiles/r02/chain_experiments.md).  The rank-128 lower update
  // of a chain that has the chip to itself keeps the round-5 64 x 64 shape up to the largest one-block factorisation: that
  // launch is bound by the read-modify-write of C, and on live tiles only the small shape is ahead at every size
  // (t x t tile rows, us: t = 28 37 / 44, t = 31 42 / 76, t = 36 57 / 65, t = 40 65 / 84; profiles/r05/update_shape_sweep.json)
  const bool rank128_alone = ctx->lat_lin && g.lower && g.K <= TILE && g.batch <= 1 && g.C != g.A && nsplit == 1 &&
                             (ctx->lat_gemm != 0 ? ctx->lat_gemm : ctx->lat_now) == 5;
  // Any other launch of such a chain: the 64 x 64 shape up to 900 tiles (30 x 30 full, 40 x 40 lower).  On live tiles it
  // is level with or ahead of the throughput shape at every size but the exact fits of 256 / 1024 tiles
  // (profiles/r05/shape_sweep.json), and it takes K^-1 = L^-T L^-1 of the gradient half off the persistent 128 x 128 launch
  // that lasts as long as its longest tile: fit step N = 4096 3.30 -> 3.03 ms, N = 5120 4.79 -> 4.69; beyond (N = 8192:
  // 1024 tiles of K = 4096 at the top of the L^-T tree) the small shape loses (12.9 -> 13.3 ms at 1400) —
  // profiles/r05/small_max.md.
  const bool alone = ctx->lat_lin && g.batch <= 1 && (ctx->lat_gemm != 0 ? ctx->lat_gemm : ctx->lat_now) == 5;
  const double small_max = rank128_alone ? 1400.0 : (alone ? 900.0 : 400.0);
  if (g.big_shape && g.C != g.A && !on_panel) return launch_big<0>(ctx, g, tiles_m, tiles_n, splits);
  if (tiles < small_max || on_panel || g.latency_shape) {
    if ((ctx->lat_gemm != 0 ? ctx->lat_gemm : ctx->lat_now) == 5) { // round-5 latency shapes; else the register-staged kernels below
      if (g.C == g.A) {
        if (tiles_n == 1) return launch_lat<1, 4, 2>(ctx, g, tiles_m, tiles_n, splits);
      } else {
        return launch_lat<2, 2, 3>(ctx, g, tiles_m, tiles_n, splits);
      }
    } else if (g.C == g.A) { // in-place (panel TRSM): one workgroup must own the whole row width
      // 32x128 strip, single LDS buffer: 22 KB and < 80 VGPRs, i.e. it fits in what two resident trailing-update
      // workgroups leave free on a CU (32 KB, 80 registers per SIMD) and is placed at once; the round-1 64x128 shape
      // (52 KB, 125 VGPRs) had to wait for a trailing workgroup to retire — half a tile time (~50 us) per panel step
      if (tiles_n == 1)
        return (ctx->small_bk_now == 32 && g.K % 32 == 0) ? launch_variant<0, 1, 4, 32, false>(ctx, g, tiles_m, tiles_n, splits)
                                                      : launch_variant<0, 1, 4, 16, false>(ctx, g, tiles_m, tiles_n, splits);
    } else {
      return (ctx->small_bk_now == 32 && g.K % 32 == 0 && !g.ktri && !g.kupper && !g.kcol && g.kchunk % 32 == 0)
                 ? launch_variant<0, 2, 2, 32, false>(ctx, g, tiles_m, tiles_n, splits)
                 : launch_variant<0, 2, 2, 16, false>(ctx, g, tiles_m, tiles_n, splits);
    }
  }
  return launch_big<0>(ctx, g, tiles_m, tiles_n, splits);
}

} // namespace gpx

// The tile list of a plain launch, on the host: entry i of the grid -> (by, bx) as the kernels decode it (include/gpx.h).
extern "C" int gpx_debug_tile_list(int lower, int delta, int tiles_m, int tiles_n, int order, int cap, int* by_bx) {
  using namespace gpx;
  if (tiles_m < 0 || tiles_n < 0 || order < 0 || order > 2 || cap < 0 || (cap > 0 && !by_bx)) return -1;
  TileMap tm = make_tile_map2(lower ? 1 : 0, delta, tiles_m, tiles_n, 1);
  tm.col_desc = lower ? 0 : order;
  for (int l = 0; l < tm.total && l < cap; ++l) decode_tile(tm, lower ? 1 : 0, l, by_bx[2 * l], by_bx[2 * l + 1]);
  return tm.total;
}

namespace gpx {

// ---- raw MFMA issue-rate microbenchmark ----------------------------------------------------
// out[0] = shader cycles (s_memtime) spent by wave 0 of block 0 in the loop,
// out[1] = constant-rate wall clock ticks (100 MHz) over the same span.
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters) {
  d4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4_t{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const long long c0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // MFMA results -> VALU readers (no auto-padding for asm)
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const long long c1 = clock64();
  const long long w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = (double)(c1 - c0);
    out[1] = (double)(w1 - w0);
  }
  if (s == 12345.678) out[2] = s; // keep the chain live
}

// tflops[0] = best sustained TFLOP/s over 1/2/4 waves per SIMD; [1] = shader cycles per MFMA
// (per wave) in that configuration; [2] = effective shader clock in MHz during the run.
int mfma_peak(gpx_ctx* ctx, double* tflops) {
  GPX_TRY(ctx->scal.ensure(4096) == hipSuccess ? 0 : -2);
  const int iters = 16384;
  double best = 0.0, best_cyc = 0.0, best_mhz = 0.0;
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    const int blocks = ctx->prop.multiProcessorCount * bpc;
    double* probe = ctx->scal.d() + 64;
    mfma_peak_kernel<<<blocks, 256, 0, ctx->s>>>(probe, iters); // warm-up / clock ramp
    GPX_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    mfma_peak_kernel<<<blocks, 256, 0, ctx->s>>>(probe, iters);
    GPX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    GPX_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    GPX_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    double h[2];
    GPX_HIP(ctx, hipMemcpy(h, probe, sizeof h, hipMemcpyDeviceToHost));
    const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * 2048.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (tf > best) {
      best = tf;
      best_cyc = h[0] / ((double)iters * 8.0);
      best_mhz = h[1] > 0 ? h[0] / (h[1] / 100.0) : 0.0; // wall clock ticks at 100 MHz
    }
  }
  tflops[0] = best;
  tflops[1] = best_cyc;
  tflops[2] = best_mhz;
  return 0;
}

} // namespace gpx
