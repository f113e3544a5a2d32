// rccl_bind.h — run-time binding of RCCL and the sharding helpers shared by the two multi-GPU launch models of the
// predictive sweep: multi.hip (one process owns the node, gpx_node_*) and rank.hip (one process per GPU, gpx_rank_*).
//
// RCCL is bound with dlopen / dlsym (whichever librccl.so.1 the process already holds, else ROCm's), so libgpx has no
// link-time and no HEADER dependency on it: the few types and enumerators used are restated here from the public
// NCCL 2.x ABI (rccl.h: ncclComm_t opaque, ncclUniqueId = 128 opaque bytes passed BY VALUE to ncclCommInitRank,
// ncclDouble = 8, ncclMax = 2, ncclSuccess = 0).  A ROCm install without the RCCL development package still builds
// libgpx; a 1-GPU user never loads RCCL at all.
#pragma once
#include <dlfcn.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "common.h"

namespace gpx {

typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t; // ncclSuccess = 0
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclDouble = 8;
constexpr ncclRedOp_t ncclMax = 2;
struct ncclUniqueId {
  char internal[GPX_UNIQUE_ID_BYTES];
};

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

inline bool load_rccl(RcclApi& r, std::string& err) {
  if (r.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) {
    const char* de = dlerror();
    err = std::string("cannot load RCCL (librccl.so.1): ") + (de ? de : "dlopen failed");
    return false;
  }
#define GPX_SYM(field, name)                                            \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name)); \
  if (!r.field) {                                                       \
    err = std::string("RCCL symbol missing: ") + name;                  \
    r.handle = nullptr;                                                 \
    return false;                                                       \
  }
  GPX_SYM(CommInitAll, "ncclCommInitAll")
  GPX_SYM(CommInitRank, "ncclCommInitRank")
  GPX_SYM(GetUniqueId, "ncclGetUniqueId")
  GPX_SYM(CommDestroy, "ncclCommDestroy")
  GPX_SYM(GetErrorString, "ncclGetErrorString")
  GPX_SYM(Broadcast, "ncclBroadcast")
  GPX_SYM(AllReduce, "ncclAllReduce")
  GPX_SYM(Send, "ncclSend")
  GPX_SYM(Recv, "ncclRecv")
  GPX_SYM(GroupStart, "ncclGroupStart")
  GPX_SYM(GroupEnd, "ncclGroupEnd")
  GPX_SYM(GetVersion, "ncclGetVersion")
#undef GPX_SYM
  // The enumerators and the by-value unique id above are restated from the NCCL 2.x ABI (RCCL 2.27 here): refuse a
  // library that reports another major version instead of calling it with constants that may have moved.
  int ver = 0;
  if (r.GetVersion(&ver) != ncclSuccess || ver <= 0) {
    err = "ncclGetVersion failed: cannot check the RCCL ABI";
    r.handle = nullptr;
    return false;
  }
  const int major = ver >= 10000 ? ver / 10000 : ver / 1000; // NCCL_VERSION_CODE: X*10000 + Y*100 + Z from 2.9 on
  if (major != 2) {
    err = "unsupported RCCL major version " + std::to_string(major) + " (version code " + std::to_string(ver) +
          "): gpax_amd/csrc/rccl_bind.h restates the NCCL 2.x ABI (ncclDouble = 8, ncclMax = 2, 128-byte unique id by value)";
    r.handle = nullptr;
    return false;
  }
  return true;
}

// contiguous block [lo, hi) of part `r` out of `parts` over S items, sizes differing by at most 1
inline void shard_range(int S, int r, int parts, int* lo, int* hi) {
  const int base = S / parts, rem = S % parts;
  *lo = r * base + (r < rem ? r : rem);
  *hi = *lo + base + (r < rem ? 1 : 0);
}

// contiguous blocks [lo[r], hi[r]) over S items with sizes in proportion to the weights w[r] > 0 (largest-remainder rounding,
// ties to the lower part): the same on every process that holds the same weights.  w == nullptr: shard_range.
inline void shard_ranges_weighted(int S, const double* w, int parts, int* lo, int* hi) {
  std::vector<int> cnt((size_t)parts, 0);
  double sum = 0.0;
  bool ok = w != nullptr;
  for (int r = 0; ok && r < parts; ++r) {
    if (!(w[r] > 0.0) || !(w[r] < 1e300)) ok = false;
    else sum += w[r];
  }
  if (!ok) {
    for (int r = 0; r < parts; ++r) shard_range(S, r, parts, lo + r, hi + r);
    return;
  }
  std::vector<double> frac((size_t)parts);
  int given = 0;
  for (int r = 0; r < parts; ++r) {
    const double share = (double)S * (w[r] / sum);
    cnt[(size_t)r] = (int)share;
    frac[(size_t)r] = share - cnt[(size_t)r];
    given += cnt[(size_t)r];
  }
  for (; given < S; ++given) {
    int best = 0;
    for (int r = 1; r < parts; ++r)
      if (frac[(size_t)r] > frac[(size_t)best]) best = r;
    cnt[(size_t)best] += 1;
    frac[(size_t)best] = -1.0;
  }
  int at = 0;
  for (int r = 0; r < parts; ++r) {
    lo[r] = at;
    at += cnt[(size_t)r];
    hi[r] = at;
  }
}

// result block of c samples, in doubles: [means c*M | draws c*n*M | vars c*M | pivots: 2c ints in c doubles]
struct BlockLayout {
  int64_t means, draws, vars, infos, total;
  BlockLayout(int c, int n, int M) {
    means = 0;
    draws = (int64_t)c * M;
    vars = draws + (int64_t)c * n * M;
    infos = vars + (int64_t)c * M;
    total = infos + c;
  }
};

// payload of one sharded sweep, in doubles: what the root hands to every GPU.  The node model keeps the theta table on
// the host (same process); the rank model appends it (with_theta) because the other processes do not have it.
struct PayloadLayout {
  int64_t X, Xn, y, eps, ells, scales, noises, total;
  PayloadLayout(int N, int d, int M, int yres_rows, int S, int n, int ne, bool with_theta) {
    X = 0;
    Xn = X + (int64_t)N * d;
    y = Xn + (int64_t)M * d;
    eps = y + (int64_t)yres_rows * N;
    ells = eps + (int64_t)S * n * M;
    scales = ells + (with_theta ? (int64_t)S * ne : 0);
    noises = scales + (with_theta ? S : 0);
    total = noises + (with_theta ? S : 0);
  }
};

// Copy `cnt` samples of a gathered block (host image `blk`, laid out for `cap` samples) from position `pos` of the block
// to global samples g0 .. g0 + cnt - 1 of the caller's arrays, decoding the pivots exactly as gpx_predict_sweep does:
// train-factor failure -> means / vars / draws NaN, covariance failure -> draws NaN, infos = pivot (train) or -pivot (cov).
inline void scatter_chunk(const double* blk, int cap, int pos, int cnt, int g0, int N, int M, int n, int cM, double* means,
                          double* samples, int* infos, double* vars) {
  const BlockLayout bl(cap, n, M);
  std::memcpy(means + (int64_t)g0 * M, blk + bl.means + (int64_t)pos * M, (size_t)cnt * M * sizeof(double));
  if (n > 0)
    std::memcpy(samples + (int64_t)g0 * n * M, blk + bl.draws + (int64_t)pos * n * M, (size_t)cnt * n * M * sizeof(double));
  if (vars) std::memcpy(vars + (int64_t)g0 * M, blk + bl.vars + (int64_t)pos * M, (size_t)cnt * M * sizeof(double));
  const int* hin = reinterpret_cast<const int*>(blk + bl.infos) + 2 * (int64_t)pos;
  for (int s = 0; s < cnt; ++s) {
    int it = hin[2 * s], ic = hin[2 * s + 1];
    if (it > N) it = 0;
    if (ic > cM) ic = 0;
    const int code = it != 0 ? it : (ic != 0 ? -ic : 0);
    const int gs = g0 + s;
    if (infos) infos[gs] = code;
    if (it != 0)
      for (int a = 0; a < M; ++a) {
        means[(int64_t)gs * M + a] = NAN;
        if (vars) vars[(int64_t)gs * M + a] = NAN;
      }
    if (code != 0 && n > 0)
      for (int64_t t = 0; t < (int64_t)n * M; ++t) samples[(int64_t)gs * n * M + t] = NAN;
  }
}

// a whole block of c_r consecutive samples starting at global sample g0
inline void scatter_block(const double* blk, int c_r, int g0, int N, int M, int n, int cM, double* means, double* samples,
                          int* infos, double* vars) {
  scatter_chunk(blk, c_r, 0, c_r, g0, N, M, n, cM, means, samples, infos, vars);
}

} // namespace gpx

#include <array>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace gpx {

// What every GPU of a sharded sweep is told: sizes, flags and where the broadcast payload sits on ITS device.
struct ShardJob {
  int kind, N, d, M, n, yres_rows, noiseless, m_slice, ne;
  double jitter;
  bool want_vars;
  const double *ells, *scales, *noises; // HOST theta tables of all S samples (indexed by global sample)
};

// The block of c_r samples starting at global sample g_lo on ONE GPU, worked off by (up to) per_gpu of the contexts in
// flight there, one host thread each (ctypes / the caller holds no lock; every context has its own streams).  The
// contexts take CHUNKS from a shared cursor — half of a context's fair share of what is left, never less than one
// launch batch B (what the library reports after a context's first chunk) — so they all stop within one batch of each
// other.  (Fixed sub-blocks of c_r / per_gpu, rounds 1 - 3: the context the hardware favoured finished early and the
// rest of the block ran with fewer samples in flight; bench.py saw +-4 % from that alone.)  Results do not depend on
// the split.  `payload` / `blk`: the device copies of the inputs and of this GPU's result block.  rc / cov-block
// slots: per_gpu each.
struct ShardCursor {
  std::mutex mu;
  // min_chunk: the launch batch B once a context has reported it (1 until then: at N >= 8192 a launch holds B <= 8
  // samples and a chunk rounded up to 8 would leave a short sweep to the first one or two contexts); no chunk exceeds the
  // fair share ceil(total / parts), so every context gets work whenever there are at least `parts` samples
  int next = 0, total = 0, parts = 1, min_chunk = 1;
  bool take(int* lo, int* hi) {
    std::lock_guard<std::mutex> g(mu);
    if (next >= total) return false;
    const int left = total - next;
    int c = (left + 2 * parts - 1) / (2 * parts);
    if (c < min_chunk) c = min_chunk;
    c = (c + min_chunk - 1) / min_chunk * min_chunk; // whole launch batches
    const int fair = (total + parts - 1) / parts;
    if (c > fair) c = fair;
    *lo = next;
    *hi = next + c < total ? next + c : total;
    next = *hi;
    return true;
  }
  void saw_batch(int b) {
    std::lock_guard<std::mutex> g(mu);
    if (b > min_chunk) min_chunk = b;
  }
};

// Where ONE GPU put the chunks it took from a cursor it shares with other GPUs (the node model, multi.hip): a GPU's result
// block is filled in the order its contexts take chunks, and the root scatters by this list.
struct ChunkLog {
  std::mutex mu;
  int placed = 0;                         // samples placed so far = next free position in this GPU's block
  std::vector<std::array<int, 3>> chunks; // (first global sample, count, position in the block)
};

// cur == nullptr: the block [g_lo, g_lo + c_r) belongs to this GPU alone (the rank model: static blocks per rank); a
// cursor of its own deals it to the contexts and sample g_lo + i sits at position i of `blk` (laid out for c_r samples).
// cur != nullptr (the node model): the cursor runs over c_r samples starting at g_lo that SEVERAL GPUs work off; `blk` is
// laid out for `cap` samples, this GPU's chunks are placed one behind the other and recorded in `log`.
// slow_us > 0 (tests: GPX_NODE_SLOW): every context of this GPU sleeps that long after each chunk — an artificially slow GPU.
inline void spawn_shard_sweep(std::vector<std::thread>& threads, const std::vector<gpx_ctx*>& ctxs, int per_gpu, int g_lo,
                              int c_r, const ShardJob& jb, const PayloadLayout& pl, const double* payload, double* blk,
                              int* rc_slots, int* cb_slots, std::shared_ptr<ShardCursor> cur = nullptr,
                              ChunkLog* log = nullptr, int cap = 0, int slow_us = 0) {
  if (c_r <= 0) return;
  const bool shared = cur != nullptr;
  const BlockLayout bl(shared ? cap : c_r, jb.n, jb.M);
  const int parts = shared ? per_gpu : (per_gpu < c_r ? per_gpu : c_r);
  if (!shared) {
    cur = std::make_shared<ShardCursor>();
    cur->total = c_r;
    cur->parts = parts;
  }
  for (int c = 0; c < parts; ++c) {
    gpx_ctx* ctx = ctxs[(size_t)c];
    int* rc_slot = rc_slots + c;
    int* cb_slot = cb_slots + c;
    const ShardJob j = jb;
    const PayloadLayout p = pl;
    *rc_slot = 0;
    *cb_slot = j.M;
    threads.emplace_back([=]() {
      int slo = 0, shi = 0, rc = 0;
      while (rc == 0 && cur->take(&slo, &shi)) {
        const int g0 = g_lo + slo; // first global sample of this chunk
        const int cnt = shi - slo;
        int pos = slo;
        if (log) {
          std::lock_guard<std::mutex> g(log->mu);
          pos = log->placed;
          log->placed += cnt;
          log->chunks.push_back({g0, cnt, pos});
        }
        rc = sweep_device_io(
            ctx, j.kind, cnt, j.ells + (int64_t)g0 * j.ne, j.scales + g0, j.noises + g0, payload + p.X, j.N, j.d,
            payload + p.y + (j.yres_rows == 1 ? 0 : (int64_t)g0 * j.N), j.yres_rows == 1 ? 1 : cnt, payload + p.Xn, j.M,
            j.noiseless, j.jitter, j.n > 0 ? payload + p.eps + (int64_t)g0 * j.n * j.M : nullptr, j.n,
            blk + bl.means + (int64_t)pos * j.M, j.n > 0 ? blk + bl.draws + (int64_t)pos * j.n * j.M : nullptr,
            reinterpret_cast<int*>(blk + bl.infos) + 2 * pos, j.want_vars ? blk + bl.vars + (int64_t)pos * j.M : nullptr,
            j.m_slice);
        if (rc == 0) rc = gpx_synchronize(ctx);
        int last = 0;
        if (rc == 0 && gpx_sweep_stats(ctx, nullptr, nullptr, &last) == 0) cur->saw_batch(last);
        *cb_slot = ctx_cov_block(ctx);
        if (slow_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(slow_us));
      }
      *rc_slot = rc;
    });
  }
}

} // namespace gpx
