// multi.hip — the predictive sweep over the GPUs of ONE node from ONE process (include/gpx.h: gpx_node_*,
// gpx_predict_sweep_multi).
//
// Reference seam: the vmap over posterior samples in ExactGP.predict (gpax/models/gp.py:392-395) is the only axis
// of the exact-GP path that shards: the S per-theta pipelines are independent given
// (X_train, y_res, X_new), and y_means.mean(0) (gp.py:399) is the only cross-sample reduction.
//
//   root GPU   : one H2D upload of [X | X_new | y_res | eps] (page-locked staging)
//   RCCL / xGMI: ncclBroadcast of that payload to every GPU of the node            (KBs .. a few MB)
//   every GPU  : chunks of the S samples, taken from ONE cursor all contexts of all GPUs share (dealt as they go, in whole
//                launch batches), through the batched device pipeline (sweep_core, api.hip); one host thread per context
//   RCCL / xGMI: ncclSend / ncclRecv (one group) of every GPU's means | draws | vars | pivots to the root
//   root GPU   : one D2H download
// No collective sits inside the sweep.  The theta tables stay on the host: every context builds the device table of
// its own samples from them (they are host data of this very process).
//
// RCCL is bound at run time (dlopen of librccl.so.1: whichever copy the process already holds — e.g. PyTorch's —
// or ROCm's), so libgpx has no link-time dependency on it and a 1-GPU user never loads it ... except through this
// entry.  GPX_NODE_TRANSPORT=memcpy replaces the two RCCL steps by hipMemcpyPeerAsync; it exists so that the
// sharding / threading logic can be exercised on a box with a single GPU (the same device listed twice, which RCCL
// refuses) and is never selected implicitly.
#include <cstdlib>

#include "rccl_bind.h"

using namespace gpx;

namespace {

struct NodeDev {
  int device = -1;
  std::vector<gpx_ctx*> ctxs; // contexts in flight on this GPU
  hipStream_t cs = nullptr;   // communication / staging stream
  DevBuf payload;             // [X | X_new | y_res | eps] as broadcast
  DevBuf out;                 // this GPU's result block, laid out for all S samples (it holds those it took, in taking order)
  DevBuf gather;              // this GPU's block compacted (what it sends); on the root: every GPU's block — the gather target
};

} // namespace

struct gpx_node {
  std::vector<NodeDev> devs;
  std::vector<ncclComm_t> comms;
  RcclApi rccl;
  bool use_rccl = true;
  int rccl_version = 0;
  std::string err;
  PinBuf pin_in, pin_out;
  int64_t sweeps = 0;
  std::vector<int> shares;        // samples every GPU worked off in the last sweep (gpx_node_last_shares)
  int slow_gpu = -1, slow_us = 0; // tests: GPX_NODE_SLOW="<gpu index>:<microseconds>" — that GPU's contexts sleep after every chunk
};

namespace {

int node_fail(gpx_node* nd, const std::string& msg) {
  nd->err = msg;
  return -2;
}
int node_bad_arg(gpx_node* nd, const char* msg) {
  nd->err = std::string("bad argument: ") + msg;
  return -1;
}

#define NODE_HIP(nd, expr)                                                                        \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) return node_fail((nd), std::string(#expr ": ") + hipGetErrorString(_e)); \
  } while (0)
#define NODE_NCCL(nd, expr)                                                                              \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) return node_fail((nd), std::string(#expr ": ") + (nd)->rccl.GetErrorString(_r)); \
  } while (0)

} // namespace

extern "C" {

int gpx_node_init(int ngpu, const int* devices, int inflight, gpx_node** out) {
  if (!out) return -1;
  gpx_node* nd = new gpx_node();
  *out = nd; // returned even on failure so the caller can read gpx_node_last_error
  if (ngpu < 1) {
    nd->err = "bad argument: ngpu must be >= 1";
    return -1;
  }
  if (inflight < 1) inflight = 1;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return node_fail(nd, "no HIP device available");
  const char* tr = getenv("GPX_NODE_TRANSPORT");
  nd->use_rccl = !(tr && std::string(tr) == "memcpy");
  if (const char* sl = getenv("GPX_NODE_SLOW")) {
    int gi = -1, us = 0;
    if (sscanf(sl, "%d:%d", &gi, &us) == 2 && gi >= 0 && us > 0) {
      nd->slow_gpu = gi;
      nd->slow_us = us;
    }
  }
  nd->devs.resize((size_t)ngpu);
  std::vector<int> devlist((size_t)ngpu);
  for (int r = 0; r < ngpu; ++r) {
    const int dev = devices ? devices[r] : r;
    if (dev < 0 || dev >= count) {
      nd->err = "bad argument: device ordinal out of range";
      return -1;
    }
    devlist[(size_t)r] = dev;
    NodeDev& D = nd->devs[(size_t)r];
    D.device = dev;
    for (int c = 0; c < inflight; ++c) {
      gpx_ctx* ctx = nullptr;
      const int rc = gpx_init(dev, &ctx);
      if (rc != 0) {
        nd->err = std::string("gpx_init failed: ") + gpx_last_error(ctx);
        gpx_destroy(ctx);
        return rc;
      }
      D.ctxs.push_back(ctx);
    }
    NODE_HIP(nd, hipSetDevice(dev));
    NODE_HIP(nd, hipStreamCreateWithFlags(&D.cs, hipStreamNonBlocking));
  }
  if (nd->use_rccl) {
    if (!load_rccl(nd->rccl, nd->err)) return -3;
    (void)nd->rccl.GetVersion(&nd->rccl_version);
    nd->comms.resize((size_t)ngpu);
    NODE_NCCL(nd, nd->rccl.CommInitAll(nd->comms.data(), ngpu, devlist.data()));
  } else {
    // peer copies between distinct devices: enable direct access where the topology allows it (xGMI)
    for (int a = 0; a < ngpu; ++a)
      for (int b = 0; b < ngpu; ++b)
        if (devlist[(size_t)a] != devlist[(size_t)b]) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, devlist[(size_t)a], devlist[(size_t)b]) == hipSuccess && can) {
            (void)hipSetDevice(devlist[(size_t)a]);
            (void)hipDeviceEnablePeerAccess(devlist[(size_t)b], 0); // "already enabled" is fine
            (void)hipGetLastError();
          }
        }
  }
  return 0;
}

void gpx_node_destroy(gpx_node* nd) {
  if (!nd) return;
  for (size_t r = 0; r < nd->devs.size(); ++r) {
    NodeDev& D = nd->devs[r];
    if (D.device < 0) continue;
    (void)hipSetDevice(D.device);
    if (D.cs) (void)hipStreamSynchronize(D.cs);
    if (nd->use_rccl && r < nd->comms.size() && nd->comms[r]) (void)nd->rccl.CommDestroy(nd->comms[r]);
    for (gpx_ctx* c : D.ctxs) gpx_destroy(c);
    D.payload.release();
    D.out.release();
    D.gather.release();
    if (D.cs) (void)hipStreamDestroy(D.cs);
  }
  nd->pin_in.release();
  nd->pin_out.release();
  delete nd; // the RCCL handle stays loaded for the life of the process (other users may share it)
}

const char* gpx_node_last_error(const gpx_node* nd) { return nd ? nd->err.c_str() : "null node"; }

int gpx_node_info(const gpx_node* nd, int* ngpu, int* inflight, int* transport_rccl, int* rccl_version) {
  if (!nd || nd->devs.empty()) return -1;
  if (ngpu) *ngpu = (int)nd->devs.size();
  if (inflight) *inflight = (int)nd->devs[0].ctxs.size();
  if (transport_rccl) *transport_rccl = nd->use_rccl ? 1 : 0;
  if (rccl_version) *rccl_version = nd->rccl_version;
  return 0;
}

int gpx_node_last_shares(const gpx_node* nd, int* counts, int cap) {
  if (!nd || cap < 0 || (cap > 0 && !counts)) return -1;
  for (size_t r = 0; r < nd->shares.size() && (int)r < cap; ++r) counts[r] = nd->shares[r];
  return (int)nd->shares.size();
}

int gpx_predict_sweep_multi(gpx_node* nd, int kind, const double* X, int N, int d, int S, const double* ells,
                            const double* scales, const double* noises, const double* yres, int yres_rows,
                            const double* Xnew, int M, int noiseless, double jitter, const double* eps, int n,
                            double* means, double* samples, int* infos, double* vars, int m_slice) {
  if (!nd || nd->devs.empty()) return -1;
  if (S < 0 || n < 0) return node_bad_arg(nd, "negative count");
  if (S == 0) return 0;
  if (!X || !ells || !scales || !noises || !yres || !Xnew || !means) return node_bad_arg(nd, "null pointer");
  if (N < 1 || M < 1 || d < 1 || d > GPX_MAX_DIM) return node_bad_arg(nd, "sizes");
  if (n > 0 && (!eps || !samples)) return node_bad_arg(nd, "eps/samples required when n > 0");
  if (yres_rows != 1 && yres_rows != S) return node_bad_arg(nd, "yres_rows must be 1 or S");
  const int G = (int)nd->devs.size();
  const int ne = d + (kind == GPX_KERNEL_PERIODIC ? 1 : 0);

  // ---- 1. payload: one upload to the root GPU, one broadcast over xGMI ----------------------------------------
  const PayloadLayout pl(N, d, M, yres_rows, S, n, ne, false);
  const size_t p_bytes = (size_t)pl.total * sizeof(double);
  NodeDev& root = nd->devs[0];
  NODE_HIP(nd, hipSetDevice(root.device));
  NODE_HIP(nd, hipStreamSynchronize(root.cs)); // previous use of the staging buffers is over
  NODE_HIP(nd, nd->pin_in.ensure(p_bytes));
  double* hp = nd->pin_in.d();
  std::memcpy(hp + pl.X, X, (size_t)N * d * sizeof(double));
  std::memcpy(hp + pl.Xn, Xnew, (size_t)M * d * sizeof(double));
  std::memcpy(hp + pl.y, yres, (size_t)yres_rows * N * sizeof(double));
  if (n > 0) std::memcpy(hp + pl.eps, eps, (size_t)S * n * M * sizeof(double));
  for (int r = 0; r < G; ++r) {
    NODE_HIP(nd, hipSetDevice(nd->devs[(size_t)r].device));
    NODE_HIP(nd, nd->devs[(size_t)r].payload.ensure(p_bytes));
  }
  NODE_HIP(nd, hipSetDevice(root.device));
  NODE_HIP(nd, hipMemcpyAsync(root.payload.p, hp, p_bytes, hipMemcpyHostToDevice, root.cs));
  if (nd->use_rccl) {
    // errors inside a group are collected; the group is ALWAYS closed before returning (an open group would poison
    // every later RCCL call of this process)
    ncclResult_t first = ncclSuccess;
    NODE_NCCL(nd, nd->rccl.GroupStart());
    for (int r = 0; r < G && first == ncclSuccess; ++r) {
      NodeDev& D = nd->devs[(size_t)r];
      first = nd->rccl.Broadcast(D.payload.p, D.payload.p, (size_t)pl.total, ncclDouble, 0, nd->comms[(size_t)r], D.cs);
    }
    const ncclResult_t ge = nd->rccl.GroupEnd();
    if (first == ncclSuccess) first = ge;
    if (first != ncclSuccess) return node_fail(nd, std::string("ncclBroadcast: ") + nd->rccl.GetErrorString(first));
  } else {
    for (int r = 1; r < G; ++r) {
      NodeDev& D = nd->devs[(size_t)r];
      NODE_HIP(nd, hipMemcpyPeerAsync(D.payload.p, D.device, root.payload.p, root.device, p_bytes, root.cs));
    }
  }
  for (int r = 0; r < G; ++r) {
    NODE_HIP(nd, hipSetDevice(nd->devs[(size_t)r].device));
    NODE_HIP(nd, hipStreamSynchronize(nd->devs[(size_t)r].cs));
  }
  NODE_HIP(nd, hipSetDevice(root.device));
  NODE_HIP(nd, hipStreamSynchronize(root.cs));

  // ---- 2. the samples are DEALT to the GPUs as they go: one cursor over all S (rccl_bind.h ShardCursor: guided
  // self-scheduling in whole launch batches) that every context of every GPU takes its chunks from — this process owns
  // them all, so a shared cursor costs one mutex.  (Static contiguous blocks per GPU, rounds 1 - 5: GPUs of one node differ
  // by a few per cent in sustained clocks — the builder's boxes spread +-4 % — and the slowest one set the time of the
  // sweep.)  A sample's values do not depend on the batch it rides in, so the deal never changes a result bit.  Every
  // GPU fills its result block in the order its contexts take chunks (ChunkLog) — laid out for S samples, the most one
  // GPU can end up with.
  const BlockLayout capL(S, n, M);
  for (int r = 0; r < G; ++r) {
    NodeDev& D = nd->devs[(size_t)r];
    NODE_HIP(nd, hipSetDevice(D.device));
    NODE_HIP(nd, D.out.ensure((size_t)capL.total * sizeof(double)));
  }
  // below N ~ 3000 one context's batched sweep already fills a GPU: one context per GPU there
  const int per_gpu = (N < 3000) ? 1 : (int)root.ctxs.size();
  const ShardJob jb{kind, N, d, M, n, yres_rows, noiseless, m_slice, ne, jitter, vars != nullptr, ells, scales, noises};
  std::vector<std::thread> threads;
  std::vector<int> rcs((size_t)G * per_gpu, 0);
  std::vector<int> cblock((size_t)G * per_gpu, M);
  std::vector<ChunkLog> logs((size_t)G);
  auto cur = std::make_shared<ShardCursor>();
  cur->total = S;
  cur->parts = G * per_gpu;
  for (int r = 0; r < G; ++r) {
    NodeDev& D = nd->devs[(size_t)r];
    spawn_shard_sweep(threads, D.ctxs, per_gpu, 0, S, jb, pl, D.payload.d(), D.out.d(), &rcs[(size_t)r * per_gpu],
                      &cblock[(size_t)r * per_gpu], cur, &logs[(size_t)r], S, r == nd->slow_gpu ? nd->slow_us : 0);
  }
  for (std::thread& t : threads) t.join();
  int cM = M; // covariance block of the sweep (the same on every context that did work; idle ones still say M)
  for (int r = 0; r < G; ++r)
    for (int c = 0; c < per_gpu; ++c) {
      if (rcs[(size_t)r * per_gpu + c] != 0) {
        nd->err = std::string("sweep failed on device ") + std::to_string(nd->devs[(size_t)r].device) + ": " +
                  gpx_last_error(nd->devs[(size_t)r].ctxs[(size_t)c]);
        return rcs[(size_t)r * per_gpu + c];
      }
      if (cblock[(size_t)r * per_gpu + c] < cM) cM = cblock[(size_t)r * per_gpu + c];
    }
  nd->shares.assign((size_t)G, 0);
  std::vector<int64_t> boff((size_t)G + 1, 0); // block offsets (doubles) inside the root's gather buffer
  for (int r = 0; r < G; ++r) {
    nd->shares[(size_t)r] = logs[(size_t)r].placed;
    boff[(size_t)r + 1] = boff[(size_t)r] + BlockLayout(logs[(size_t)r].placed, n, M).total;
  }

  // ---- 3. gather on the root GPU: every GPU's block, compacted to the samples it holds (means | draws | vars | pivots),
  // ONE message per GPU in one RCCL group, one download --------------------------------------------------
  NODE_HIP(nd, hipSetDevice(root.device));
  NODE_HIP(nd, root.gather.ensure((size_t)(boff[(size_t)G] > 0 ? boff[(size_t)G] : 1) * sizeof(double)));
  struct Seg {
    int64_t src, dst, cnt;
  };
  auto segments = [&](int r, Seg* sg) -> int {
    const int c = logs[(size_t)r].placed;
    if (c <= 0) return 0;
    const BlockLayout bl(c, n, M);
    int k = 0;
    sg[k++] = Seg{capL.means, boff[(size_t)r] + bl.means, (int64_t)c * M};
    if (n > 0) sg[k++] = Seg{capL.draws, boff[(size_t)r] + bl.draws, (int64_t)c * n * M};
    if (vars) sg[k++] = Seg{capL.vars, boff[(size_t)r] + bl.vars, (int64_t)c * M};
    sg[k++] = Seg{capL.infos, boff[(size_t)r] + bl.infos, (int64_t)c};
    return k;
  };
  Seg sg[4];
  { // the root's own block: device-to-device on its staging stream
    const int k = segments(0, sg);
    for (int i = 0; i < k; ++i)
      NODE_HIP(nd, hipMemcpyAsync(root.gather.d() + sg[i].dst, root.out.d() + sg[i].src, (size_t)sg[i].cnt * sizeof(double),
                                  hipMemcpyDeviceToDevice, root.cs));
  }
  // every other GPU compacts its block on ITS device first (the same <= 4 device-to-device copies), so that what crosses xGMI
  // is ONE contiguous message per GPU — the send / receive pattern of rounds 1 - 5
  for (int r = 1; r < G; ++r) {
    NodeDev& D = nd->devs[(size_t)r];
    const int k = segments(r, sg);
    if (k == 0) continue;
    const int64_t tot = boff[(size_t)r + 1] - boff[(size_t)r];
    NODE_HIP(nd, hipSetDevice(D.device));
    NODE_HIP(nd, D.gather.ensure((size_t)tot * sizeof(double)));
    for (int i = 0; i < k; ++i)
      NODE_HIP(nd, hipMemcpyAsync(D.gather.d() + (sg[i].dst - boff[(size_t)r]), D.out.d() + sg[i].src,
                                  (size_t)sg[i].cnt * sizeof(double), hipMemcpyDeviceToDevice, D.cs));
  }
  if (nd->use_rccl) {
    ncclResult_t first = ncclSuccess;
    NODE_NCCL(nd, nd->rccl.GroupStart());
    for (int r = 1; r < G && first == ncclSuccess; ++r) {
      const int64_t cnt = boff[(size_t)r + 1] - boff[(size_t)r];
      if (cnt <= 0 || logs[(size_t)r].placed <= 0) continue;
      NodeDev& D = nd->devs[(size_t)r];
      first = nd->rccl.Recv(root.gather.d() + boff[(size_t)r], (size_t)cnt, ncclDouble, r, nd->comms[0], root.cs);
      if (first == ncclSuccess) first = nd->rccl.Send(D.gather.p, (size_t)cnt, ncclDouble, 0, nd->comms[(size_t)r], D.cs);
    }
    const ncclResult_t ge = nd->rccl.GroupEnd();
    if (first == ncclSuccess) first = ge;
    if (first != ncclSuccess) return node_fail(nd, std::string("ncclSend/ncclRecv: ") + nd->rccl.GetErrorString(first));
    for (int r = 1; r < G; ++r) {
      NODE_HIP(nd, hipSetDevice(nd->devs[(size_t)r].device));
      NODE_HIP(nd, hipStreamSynchronize(nd->devs[(size_t)r].cs));
    }
  } else {
    for (int r = 1; r < G; ++r) {
      const int64_t cnt = boff[(size_t)r + 1] - boff[(size_t)r];
      if (cnt <= 0 || logs[(size_t)r].placed <= 0) continue;
      NodeDev& D = nd->devs[(size_t)r];
      NODE_HIP(nd, hipSetDevice(D.device));
      NODE_HIP(nd, hipStreamSynchronize(D.cs)); // the compaction on the source device is done before the root pulls
      NODE_HIP(nd, hipSetDevice(root.device));
      NODE_HIP(nd, hipMemcpyPeerAsync(root.gather.d() + boff[(size_t)r], root.device, D.gather.p, D.device,
                                      (size_t)cnt * sizeof(double), root.cs));
    }
  }
  NODE_HIP(nd, hipSetDevice(root.device));
  const size_t o_bytes = (size_t)boff[(size_t)G] * sizeof(double);
  NODE_HIP(nd, nd->pin_out.ensure(o_bytes));
  NODE_HIP(nd, hipMemcpyAsync(nd->pin_out.p, root.gather.p, o_bytes, hipMemcpyDeviceToHost, root.cs));
  NODE_HIP(nd, hipStreamSynchronize(root.cs));
  const double* ho = nd->pin_out.d();
  for (int r = 0; r < G; ++r)
    for (const std::array<int, 3>& ch : logs[(size_t)r].chunks) // (first global sample, count, position in the block)
      scatter_chunk(ho + boff[(size_t)r], logs[(size_t)r].placed, ch[2], ch[1], ch[0], N, M, n, cM, means, samples, infos, vars);
  nd->sweeps += 1;
  return 0;
}

} // extern "C"
