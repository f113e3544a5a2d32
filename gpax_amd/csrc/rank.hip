// rank.hip — the predictive sweep over the GPUs of one node with ONE PROCESS PER GPU (include/gpx.h: gpx_rank_*).
//
// Reference seam: the vmap over posterior samples in ExactGP.predict (gpax/models/gp.py:392-395), the only axis of the
// exact-GP path that shards; y_means.mean(0) (gp.py:399) is the only cross-sample reduction.  This is
// the launch model of `torch.distributed.run` / `mpirun` style launchers (RANK, LOCAL_RANK, WORLD_SIZE), without
// PyTorch: every process owns one GPU (`inflight` libgpx contexts on it) and one RCCL communicator rank
// (ncclCommInitRank; the 128-byte unique id travels from rank 0 to the others through the launcher's rendezvous,
// gpax_amd/launch.py).  gpx_rank_predict_sweep is a COLLECTIVE call (same scalar arguments on every rank, array data
// on rank 0 only):
//
//   rank 0     : one H2D upload of [X | X_new | y_res | eps | theta table]            (page-locked staging)
//   RCCL / xGMI: ncclBroadcast of that payload                                         (KBs .. a few MB)
//   every rank : its contiguous block of the S samples — sized by its GPU's measured speed once the ranks have calibrated
//                (gpx_rank_calibrate) — through the batched device pipeline (sweep_core, api.hip), dealt to the
//                contexts in flight on its GPU as they go, one host thread per context
//   RCCL / xGMI: ncclSend of the [means | draws | vars | pivots] block to rank 0, ncclRecv there (one group)
//   rank 0     : one D2H download
// No collective sits inside the sweep.  gpx_rank_barrier / gpx_rank_allreduce_max (ncclAllReduce of a few doubles)
// give a launcher what it needs to time a collective region (bench.py: barrier, max over ranks).
//
// Transports.  "rccl" is the product path.  "file" replaces the three RCCL steps by files in a directory all ranks
// share (tmp + rename = atomic publish, polling readers): it exists so that the multi-process control flow can run on a
// box where RCCL cannot span the ranks — several ranks on ONE GPU in the tests (RCCL refuses duplicate devices) — and
// as the fallback of bench.py when RCCL fails to initialise; it is chosen explicitly (gpx_rank_init's `file_dir`).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>

#include "rccl_bind.h"

using namespace gpx;

struct gpx_rank {
  int device = -1, rank = 0, nranks = 1;
  std::vector<gpx_ctx*> ctxs; // contexts in flight on this rank's GPU
  hipStream_t cs = nullptr;   // communication / staging stream
  DevBuf payload, out, red;   // broadcast payload, result block (rank 0: every block), small reduction buffer
  PinBuf pin_in, pin_out, pin_red;
  std::vector<double> theta;  // host copy of the theta table on ranks > 0
  RcclApi rccl;
  ncclComm_t comm = nullptr;
  bool use_rccl = true;
  int rccl_version = 0;
  std::string file_dir; // "file" transport: directory shared by the ranks
  uint64_t seq = 0;     // collective sequence number (file names)
  double file_timeout_s = 600.0;
  std::string err;
  int64_t sweeps = 0;
  // GPX_RANK_FORCE_COLLECTIVES=1 (diagnostic): a communicator of ONE rank still issues every RCCL call of the path —
  // the flag all-reduce, the payload broadcast, a send / receive group (to itself) — instead of skipping them.  On a
  // 1-GPU box this is as close as the code gets to its first multi-GPU run: argument validity, datatypes, counts, stream
  // use and group nesting are all exercised (NCCL_DEBUG=INFO log: profiles/r04/rank1_rccl_nccl_debug.log).
  bool force_coll = false;
  int64_t coll_calls = 0; // RCCL collective / p2p calls issued so far
  // relative speed of every rank's GPU (gpx_rank_calibrate; empty: equal blocks) — what sizes the ranks' blocks of a sweep
  std::vector<double> speeds;
};

namespace {

int rank_fail(gpx_rank* rk, const std::string& msg) {
  rk->err = msg;
  return -2;
}
int rank_bad_arg(gpx_rank* rk, const char* msg) {
  rk->err = std::string("bad argument: ") + msg;
  return -1;
}

#define RANK_HIP(rk, expr)                                                                        \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) return rank_fail((rk), std::string(#expr ": ") + hipGetErrorString(_e)); \
  } while (0)
#define RANK_NCCL(rk, expr)                                                                              \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) return rank_fail((rk), std::string(#expr ": ") + (rk)->rccl.GetErrorString(_r)); \
  } while (0)
#define RANK_TRY(expr)       \
  do {                       \
    int _rc = (expr);        \
    if (_rc != 0) return _rc; \
  } while (0)

// ---- "file" transport -------------------------------------------------------------------------------------------
std::string file_name(const gpx_rank* rk, const char* tag, uint64_t seq, int who) {
  return rk->file_dir + "/" + tag + "." + std::to_string(seq) + "." + std::to_string(who);
}

int file_put(gpx_rank* rk, const std::string& name, const void* data, size_t bytes) {
  const std::string tmp = name + ".tmp";
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  if (fd < 0) return rank_fail(rk, "file transport: cannot create " + tmp);
  const char* p = static_cast<const char*>(data);
  size_t left = bytes;
  while (left > 0) {
    const ssize_t w = write(fd, p, left);
    if (w <= 0) {
      close(fd);
      return rank_fail(rk, "file transport: short write to " + tmp);
    }
    p += w;
    left -= (size_t)w;
  }
  close(fd);
  if (rename(tmp.c_str(), name.c_str()) != 0) return rank_fail(rk, "file transport: cannot publish " + name);
  return 0;
}

// waits until `name` has been published, then reads exactly `bytes` from it
int file_get(gpx_rank* rk, const std::string& name, void* data, size_t bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  int fd = -1;
  for (;;) {
    fd = open(name.c_str(), O_RDONLY);
    if (fd >= 0) break;
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > rk->file_timeout_s) return rank_fail(rk, "file transport: timed out waiting for " + name);
    usleep(waited < 0.05 ? 100 : 1000);
  }
  char* p = static_cast<char*>(data);
  size_t left = bytes;
  while (left > 0) {
    const ssize_t r = read(fd, p, left);
    if (r <= 0) {
      close(fd);
      return rank_fail(rk, "file transport: short read from " + name);
    }
    p += r;
    left -= (size_t)r;
  }
  close(fd);
  return 0;
}

// ---- collectives on small host vectors ------------------------------------------------------------------------------
// v[0..n) <- max over ranks, elementwise (n <= 64)
int allreduce_max(gpx_rank* rk, double* v, int n) {
  if (n < 1 || n > 64) return rank_bad_arg(rk, "allreduce_max: 1..64 values");
  const uint64_t seq = rk->seq++;
  if (rk->nranks == 1 && !(rk->force_coll && rk->use_rccl)) return 0;
  if (!rk->use_rccl) {
    RANK_TRY(file_put(rk, file_name(rk, "red", seq, rk->rank), v, (size_t)n * sizeof(double)));
    double other[64];
    for (int r = 0; r < rk->nranks; ++r) {
      if (r == rk->rank) continue;
      RANK_TRY(file_get(rk, file_name(rk, "red", seq, r), other, (size_t)n * sizeof(double)));
      for (int i = 0; i < n; ++i)
        if (other[i] > v[i] || other[i] != other[i]) v[i] = other[i];
    }
    return 0;
  }
  RANK_HIP(rk, hipSetDevice(rk->device));
  RANK_HIP(rk, rk->red.ensure(64 * sizeof(double)));
  RANK_HIP(rk, rk->pin_red.ensure(64 * sizeof(double)));
  std::memcpy(rk->pin_red.p, v, (size_t)n * sizeof(double));
  RANK_HIP(rk, hipMemcpyAsync(rk->red.p, rk->pin_red.p, (size_t)n * sizeof(double), hipMemcpyHostToDevice, rk->cs));
  RANK_NCCL(rk, rk->rccl.AllReduce(rk->red.p, rk->red.p, (size_t)n, ncclDouble, ncclMax, rk->comm, rk->cs));
  rk->coll_calls += 1;
  RANK_HIP(rk, hipMemcpyAsync(rk->pin_red.p, rk->red.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, rk->cs));
  RANK_HIP(rk, hipStreamSynchronize(rk->cs));
  std::memcpy(v, rk->pin_red.p, (size_t)n * sizeof(double));
  return 0;
}

// `count` doubles from rank 0's host buffer into every rank's host buffer
int bcast_host(gpx_rank* rk, double* buf, int64_t count) {
  if (count < 0 || (count > 0 && !buf)) return rank_bad_arg(rk, "bcast: buffer");
  const uint64_t seq = rk->seq++;
  if ((rk->nranks == 1 && !(rk->force_coll && rk->use_rccl)) || count == 0) return 0;
  const size_t bytes = (size_t)count * sizeof(double);
  if (!rk->use_rccl) {
    if (rk->rank == 0) return file_put(rk, file_name(rk, "bc", seq, 0), buf, bytes);
    return file_get(rk, file_name(rk, "bc", seq, 0), buf, bytes);
  }
  RANK_HIP(rk, hipSetDevice(rk->device));
  RANK_HIP(rk, hipStreamSynchronize(rk->cs));
  RANK_HIP(rk, rk->payload.ensure(bytes));
  RANK_HIP(rk, rk->pin_in.ensure(bytes));
  if (rk->rank == 0) {
    std::memcpy(rk->pin_in.p, buf, bytes);
    RANK_HIP(rk, hipMemcpyAsync(rk->payload.p, rk->pin_in.p, bytes, hipMemcpyHostToDevice, rk->cs));
  }
  RANK_NCCL(rk, rk->rccl.Broadcast(rk->payload.p, rk->payload.p, (size_t)count, ncclDouble, 0, rk->comm, rk->cs));
  rk->coll_calls += 1;
  if (rk->rank != 0) RANK_HIP(rk, hipMemcpyAsync(rk->pin_in.p, rk->payload.p, bytes, hipMemcpyDeviceToHost, rk->cs));
  RANK_HIP(rk, hipStreamSynchronize(rk->cs));
  if (rk->rank != 0) std::memcpy(buf, rk->pin_in.p, bytes);
  return 0;
}

} // namespace

extern "C" {

int gpx_shard_range(int S, int part, int parts, int* lo, int* hi) {
  if (S < 0 || parts < 1 || part < 0 || part >= parts || !lo || !hi) return -1;
  shard_range(S, part, parts, lo, hi);
  return 0;
}

int gpx_rank_unique_id(char* id, char* errbuf, int errlen) {
  static RcclApi api; // process-wide binding for this entry (a gpx_rank binds its own copy of the same handle)
  std::string err;
  if (!id) return -1;
  if (!load_rccl(api, err)) {
    if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "%s", err.c_str());
    return -3;
  }
  ncclUniqueId uid;
  const ncclResult_t r = api.GetUniqueId(&uid);
  if (r != ncclSuccess) {
    if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "ncclGetUniqueId: %s", api.GetErrorString(r));
    return -2;
  }
  std::memcpy(id, uid.internal, GPX_UNIQUE_ID_BYTES);
  return 0;
}

int gpx_rank_init(int device, int rank, int nranks, const char* unique_id, const char* file_dir, int inflight,
                  gpx_rank** out) {
  if (!out) return -1;
  gpx_rank* rk = new gpx_rank();
  *out = rk; // returned even on failure so the caller can read gpx_rank_last_error
  if (nranks < 1 || rank < 0 || rank >= nranks) return rank_bad_arg(rk, "rank / nranks");
  rk->rank = rank;
  rk->nranks = nranks;
  rk->use_rccl = !(file_dir && file_dir[0]);
  if (!rk->use_rccl) rk->file_dir = file_dir;
  if (rk->use_rccl && !unique_id) return rank_bad_arg(rk, "the rccl transport needs the unique id of rank 0");
  if (const char* e = getenv("GPX_RANK_FILE_TIMEOUT")) rk->file_timeout_s = atof(e) > 0 ? atof(e) : rk->file_timeout_s;
  if (const char* e = getenv("GPX_RANK_FORCE_COLLECTIVES")) rk->force_coll = (e[0] == '1');
  if (inflight < 1) inflight = 1;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return rank_fail(rk, "no HIP device available");
  if (device < 0 || device >= count) return rank_bad_arg(rk, "device ordinal out of range");
  rk->device = device;
  for (int c = 0; c < inflight; ++c) {
    gpx_ctx* ctx = nullptr;
    const int rc = gpx_init(device, &ctx);
    if (rc != 0) {
      rk->err = std::string("gpx_init failed: ") + gpx_last_error(ctx);
      gpx_destroy(ctx);
      return rc;
    }
    rk->ctxs.push_back(ctx);
  }
  RANK_HIP(rk, hipSetDevice(device));
  RANK_HIP(rk, hipStreamCreateWithFlags(&rk->cs, hipStreamNonBlocking));
  if (rk->use_rccl) {
    if (!load_rccl(rk->rccl, rk->err)) return -3;
    (void)rk->rccl.GetVersion(&rk->rccl_version);
    ncclUniqueId uid;
    std::memcpy(uid.internal, unique_id, GPX_UNIQUE_ID_BYTES);
    RANK_NCCL(rk, rk->rccl.CommInitRank(&rk->comm, nranks, uid, rank)); // collective: returns when every rank has joined
  }
  return 0;
}

void gpx_rank_destroy(gpx_rank* rk) {
  if (!rk) return;
  if (rk->device >= 0) {
    (void)hipSetDevice(rk->device);
    if (rk->cs) (void)hipStreamSynchronize(rk->cs);
    if (rk->comm) (void)rk->rccl.CommDestroy(rk->comm);
    for (gpx_ctx* c : rk->ctxs) gpx_destroy(c);
    rk->payload.release();
    rk->out.release();
    rk->red.release();
    rk->pin_in.release();
    rk->pin_out.release();
    rk->pin_red.release();
    if (rk->cs) (void)hipStreamDestroy(rk->cs);
  }
  delete rk;
}

const char* gpx_rank_last_error(const gpx_rank* rk) { return rk ? rk->err.c_str() : "null rank"; }

int gpx_rank_info(const gpx_rank* rk, int* rank, int* nranks, int* inflight, int* transport_rccl, int* rccl_version) {
  if (!rk) return -1;
  if (rank) *rank = rk->rank;
  if (nranks) *nranks = rk->nranks;
  if (inflight) *inflight = (int)rk->ctxs.size();
  if (transport_rccl) *transport_rccl = rk->use_rccl ? 1 : 0;
  if (rccl_version) *rccl_version = rk->rccl_version;
  return 0;
}

int64_t gpx_rank_collective_calls(const gpx_rank* rk) { return rk ? rk->coll_calls : -1; }

int gpx_rank_device_pci(const gpx_rank* rk, int* domain, int* bus, int* dev) {
  if (!rk || rk->device < 0) return -1;
  return gpx_device_pci(rk->device, domain, bus, dev);
}

int gpx_rank_allreduce_max(gpx_rank* rk, double* v, int n) {
  if (!rk || !v) return -1;
  return allreduce_max(rk, v, n);
}

int gpx_shard_ranges_weighted(int S, const double* weights, int parts, int* lo, int* hi) {
  if (S < 0 || parts < 1 || !lo || !hi) return -1;
  shard_ranges_weighted(S, weights, parts, lo, hi);
  return 0;
}

// Collective.  The ranks of a sweep work off STATIC contiguous blocks (processes share no cursor), so a block's size should
// follow its GPU's speed: GPUs of one node differ by a few per cent in sustained clocks (+-4 % over the builder's boxes),
// and with equal blocks the slowest of eight sets the time of the sweep.  Every rank times a probe on its own GPU — the
// Cholesky trailing-update form of the fp64 MFMA GEMM (C -= A B^T, 32 x 32 tiles of 128, K = 2048: 69 GFLOP, about a
// millisecond) on resident scratch operands, best of three averages over ten launches — and the rates are exchanged with one
// all-reduce (a vector with the own slot set, zeros elsewhere, maximum over ranks).  Rates are taken relative to their mean
// and clamped to [0.85, 1.15]: a probe that caught a GPU mid clock ramp must not starve it.
int gpx_rank_calibrate(gpx_rank* rk, double* speeds) {
  if (!rk || rk->device < 0) return -1;
  if (rk->nranks > 64) return rank_bad_arg(rk, "gpx_rank_calibrate: at most 64 ranks");
  double best = 1e300;
  for (int t = 0; t < 3; ++t) {
    double ms = 0.0;
    const int rc = gpx_debug_gemm_time(rk->ctxs[0], 32, 32, 2048, 1, 0, 2, 10, &ms);
    if (rc != 0) {
      best = -1.0; // this rank cannot say: every rank falls back to equal blocks (the exchange below still runs)
      break;
    }
    if (ms < best) best = ms;
  }
  double mine = best > 0.0 ? 1.0 / best : 0.0;
  if (const char* e = getenv("GPX_RANK_SPEED_SCALE")) { // tests: an artificially slow / fast rank
    const double f = atof(e);
    if (f > 0.0) mine *= f;
  }
  double v[64];
  for (int r = 0; r < rk->nranks; ++r) v[r] = (r == rk->rank) ? mine : 0.0;
  RANK_TRY(allreduce_max(rk, v, rk->nranks));
  double sum = 0.0;
  bool ok = true;
  for (int r = 0; r < rk->nranks; ++r) {
    if (!(v[r] > 0.0)) ok = false;
    sum += v[r];
  }
  rk->speeds.clear();
  if (ok) {
    const double mean = sum / rk->nranks;
    for (int r = 0; r < rk->nranks; ++r) {
      double s = v[r] / mean;
      if (s < 0.85) s = 0.85;
      if (s > 1.15) s = 1.15;
      rk->speeds.push_back(s);
    }
  }
  if (speeds)
    for (int r = 0; r < rk->nranks; ++r) speeds[r] = ok ? rk->speeds[(size_t)r] : 1.0;
  return 0;
}

int gpx_rank_barrier(gpx_rank* rk) {
  if (!rk) return -1;
  // everything this rank has queued on its own contexts is done before it reports in
  for (gpx_ctx* c : rk->ctxs) {
    const int rc = gpx_synchronize(c);
    if (rc != 0) return rank_fail(rk, std::string("gpx_synchronize: ") + gpx_last_error(c));
  }
  double one = 1.0;
  return allreduce_max(rk, &one, 1);
}

int gpx_rank_bcast(gpx_rank* rk, double* buf, int64_t count) {
  if (!rk) return -1;
  return bcast_host(rk, buf, count);
}

int gpx_rank_predict_sweep(gpx_rank* rk, int kind, const double* X, int N, int d, int S, const double* ells,
                           const double* scales, const double* noises, const double* yres, int yres_rows,
                           const double* Xnew, int M, int noiseless, double jitter, const double* eps, int n,
                           double* means, double* samples, int* infos, double* vars, int want_vars, int m_slice) {
  if (!rk || rk->device < 0) return -1;
  const bool root = rk->rank == 0;
  // Argument checks must not let one rank leave while the others enter the broadcast (they would wait for it for
  // ever — 600 s under the file transport): every rank first learns whether ANY rank found a bad argument (one
  // all-reduce of a flag; the arrays exist on rank 0 only, so only it can judge them) and all leave together.
  const char* bad = nullptr;
  if (S < 0 || n < 0) bad = "negative count";
  else if (N < 1 || M < 1 || d < 1 || d > GPX_MAX_DIM) bad = "sizes";
  else if (yres_rows != 1 && yres_rows != S) bad = "yres_rows must be 1 or S";
  else if (root && S > 0 && (!X || !ells || !scales || !noises || !yres || !Xnew || !means)) bad = "null pointer on rank 0";
  else if (root && S > 0 && n > 0 && (!eps || !samples)) bad = "eps/samples required when n > 0";
  else if (root && want_vars && !vars) bad = "vars required when want_vars";
  {
    double flag = bad ? 1.0 : 0.0;
    const int arc = allreduce_max(rk, &flag, 1);
    if (arc != 0) return arc;
    if (bad) return rank_bad_arg(rk, bad);
    if (flag != 0.0) return rank_bad_arg(rk, "another rank rejected its arguments (the arrays live on rank 0)");
  }
  if (S == 0) return 0;
  const int G = rk->nranks;
  const int ne = d + (kind == GPX_KERNEL_PERIODIC ? 1 : 0);
  const uint64_t seq = rk->seq++;
  RANK_HIP(rk, hipSetDevice(rk->device));

  // ---- 1. payload from rank 0 to every rank -----------------------------------------------------------------------
  const PayloadLayout pl(N, d, M, yres_rows, S, n, ne, true);
  const size_t p_bytes = (size_t)pl.total * sizeof(double);
  RANK_HIP(rk, hipStreamSynchronize(rk->cs)); // previous use of the staging buffers is over
  RANK_HIP(rk, rk->pin_in.ensure(p_bytes));
  RANK_HIP(rk, rk->payload.ensure(p_bytes));
  double* hp = rk->pin_in.d();
  if (root) {
    std::memcpy(hp + pl.X, X, (size_t)N * d * sizeof(double));
    std::memcpy(hp + pl.Xn, Xnew, (size_t)M * d * sizeof(double));
    std::memcpy(hp + pl.y, yres, (size_t)yres_rows * N * sizeof(double));
    if (n > 0) std::memcpy(hp + pl.eps, eps, (size_t)S * n * M * sizeof(double));
    std::memcpy(hp + pl.ells, ells, (size_t)S * ne * sizeof(double));
    std::memcpy(hp + pl.scales, scales, (size_t)S * sizeof(double));
    std::memcpy(hp + pl.noises, noises, (size_t)S * sizeof(double));
  }
  const size_t th_bytes = (size_t)(pl.total - pl.ells) * sizeof(double);
  if (rk->use_rccl) {
    if (root) RANK_HIP(rk, hipMemcpyAsync(rk->payload.p, hp, p_bytes, hipMemcpyHostToDevice, rk->cs));
    if (G > 1 || rk->force_coll) {
      RANK_NCCL(rk, rk->rccl.Broadcast(rk->payload.p, rk->payload.p, (size_t)pl.total, ncclDouble, 0, rk->comm, rk->cs));
      rk->coll_calls += 1;
    }
    if (!root) // the theta table is host data of the sweep (every context builds its device table from it)
      RANK_HIP(rk, hipMemcpyAsync(hp + pl.ells, rk->payload.d() + pl.ells, th_bytes, hipMemcpyDeviceToHost, rk->cs));
  } else {
    if (root && G > 1) RANK_TRY(file_put(rk, file_name(rk, "in", seq, 0), hp, p_bytes));
    if (!root) RANK_TRY(file_get(rk, file_name(rk, "in", seq, 0), hp, p_bytes));
    RANK_HIP(rk, hipMemcpyAsync(rk->payload.p, hp, p_bytes, hipMemcpyHostToDevice, rk->cs));
  }
  RANK_HIP(rk, hipStreamSynchronize(rk->cs));
  const double *h_ells = ells, *h_scales = scales, *h_noises = noises;
  if (!root) {
    rk->theta.assign(hp + pl.ells, hp + pl.total);
    h_ells = rk->theta.data();
    h_scales = h_ells + (int64_t)S * ne;
    h_noises = h_scales + S;
  }

  // ---- 2. this rank's block of samples; contexts in flight split it again ------------------------------------------
  std::vector<int> lo((size_t)G), hi((size_t)G);
  std::vector<int64_t> boff((size_t)G + 1, 0); // block offsets (doubles) inside rank 0's gather buffer
  // blocks in proportion to the GPUs' measured speeds when the ranks have calibrated (gpx_rank_calibrate: every rank holds
  // the same vector), equal blocks otherwise
  shard_ranges_weighted(S, (int)rk->speeds.size() == G ? rk->speeds.data() : nullptr, G, lo.data(), hi.data());
  for (int r = 0; r < G; ++r) boff[(size_t)r + 1] = boff[(size_t)r] + BlockLayout(hi[(size_t)r] - lo[(size_t)r], n, M).total;
  const int c_me = hi[(size_t)rk->rank] - lo[(size_t)rk->rank];
  const int64_t my_total = BlockLayout(c_me, n, M).total;
  // (forced collectives with one rank: room for a copy of the block behind it — the receive side of the self send)
  const int64_t need = (root ? boff[(size_t)G] : my_total) + ((rk->force_coll && G == 1) ? my_total : 0);
  RANK_HIP(rk, rk->out.ensure((size_t)(need > 0 ? need : 1) * sizeof(double)));
  // below N ~ 3000 one context's batched sweep already fills a GPU: one context there
  const int per_gpu = (N < 3000) ? 1 : (int)rk->ctxs.size();
  const ShardJob jb{kind, N, d, M, n, yres_rows, noiseless, m_slice, ne, jitter, want_vars != 0, h_ells, h_scales, h_noises};
  std::vector<std::thread> threads;
  std::vector<int> rcs((size_t)per_gpu, 0), cblock((size_t)per_gpu, M);
  spawn_shard_sweep(threads, rk->ctxs, per_gpu, lo[(size_t)rk->rank], c_me, jb, pl, rk->payload.d(), rk->out.d(), rcs.data(),
                    cblock.data());
  for (std::thread& t : threads) t.join();
  int sweep_rc = 0;
  for (int c = 0; c < per_gpu; ++c)
    if (rcs[(size_t)c] != 0 && sweep_rc == 0) {
      sweep_rc = rcs[(size_t)c];
      rk->err = std::string("sweep failed on rank ") + std::to_string(rk->rank) + ": " + gpx_last_error(rk->ctxs[(size_t)c]);
    }
  // Every rank learns whether ANY rank's sweep failed (one all-reduce of a flag) before the gather: a failed block must
  // not reach rank 0 as if it were a result, and all ranks must leave the collective together.
  {
    double failed = sweep_rc != 0 ? 1.0 : 0.0;
    const std::string own_err = rk->err;
    const int arc = allreduce_max(rk, &failed, 1);
    if (arc != 0) return arc;
    if (failed != 0.0) {
      if (sweep_rc == 0) rk->err = "the sweep failed on another rank";
      else rk->err = own_err;
      return sweep_rc != 0 ? sweep_rc : -4;
    }
  }

  // ---- 3. gather on rank 0, one download -----------------------------------------------------------------------------
  RANK_HIP(rk, hipSetDevice(rk->device));
  if (rk->use_rccl) {
    if (G > 1 || (rk->force_coll && my_total > 0)) {
      ncclResult_t first = ncclSuccess; // the group is always closed, whatever a call inside it returns
      RANK_NCCL(rk, rk->rccl.GroupStart());
      if (G == 1) { // forced: the block goes to this rank itself, into the spare room behind it; the results stay where they are
        first = rk->rccl.Send(rk->out.p, (size_t)my_total, ncclDouble, 0, rk->comm, rk->cs);
        if (first == ncclSuccess) first = rk->rccl.Recv(rk->out.d() + my_total, (size_t)my_total, ncclDouble, 0, rk->comm, rk->cs);
        rk->coll_calls += 2;
      } else if (root) {
        for (int r = 1; r < G && first == ncclSuccess; ++r) {
          const int64_t cnt = boff[(size_t)r + 1] - boff[(size_t)r];
          if (cnt > 0) {
            first = rk->rccl.Recv(rk->out.d() + boff[(size_t)r], (size_t)cnt, ncclDouble, r, rk->comm, rk->cs);
            rk->coll_calls += 1;
          }
        }
      } else if (my_total > 0 && c_me > 0) {
        first = rk->rccl.Send(rk->out.p, (size_t)my_total, ncclDouble, 0, rk->comm, rk->cs);
        rk->coll_calls += 1;
      }
      const ncclResult_t ge = rk->rccl.GroupEnd();
      if (first == ncclSuccess) first = ge;
      if (first != ncclSuccess) return rank_fail(rk, std::string("ncclSend/ncclRecv: ") + rk->rccl.GetErrorString(first));
    }
    if (root) {
      const size_t o_bytes = (size_t)boff[(size_t)G] * sizeof(double);
      RANK_HIP(rk, rk->pin_out.ensure(o_bytes));
      RANK_HIP(rk, hipMemcpyAsync(rk->pin_out.p, rk->out.p, o_bytes, hipMemcpyDeviceToHost, rk->cs));
    }
    RANK_HIP(rk, hipStreamSynchronize(rk->cs));
  } else {
    const size_t mine = (size_t)my_total * sizeof(double);
    RANK_HIP(rk, rk->pin_out.ensure(root ? (size_t)boff[(size_t)G] * sizeof(double) : (mine ? mine : 8)));
    if (c_me > 0) RANK_HIP(rk, hipMemcpyAsync(rk->pin_out.p, rk->out.p, mine, hipMemcpyDeviceToHost, rk->cs));
    RANK_HIP(rk, hipStreamSynchronize(rk->cs));
    if (!root && c_me > 0) RANK_TRY(file_put(rk, file_name(rk, "out", seq, rk->rank), rk->pin_out.p, mine));
    if (root)
      for (int r = 1; r < G; ++r) {
        const int64_t cnt = boff[(size_t)r + 1] - boff[(size_t)r];
        if (cnt > 0 && hi[(size_t)r] > lo[(size_t)r])
          RANK_TRY(file_get(rk, file_name(rk, "out", seq, r), rk->pin_out.d() + boff[(size_t)r], (size_t)cnt * sizeof(double)));
      }
  }
  if (root) {
    const double* ho = rk->pin_out.d();
    for (int r = 0; r < G; ++r)
      if (hi[(size_t)r] > lo[(size_t)r])
        scatter_block(ho + boff[(size_t)r], hi[(size_t)r] - lo[(size_t)r], lo[(size_t)r], N, M, n, cblock[0], means, samples,
                      infos, want_vars ? vars : nullptr);
  }
  rk->sweeps += 1;
  return 0;
}

} // extern "C"
