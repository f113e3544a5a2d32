// Feasibility probe: eager launch chain vs hipGraph replay of the same chain, 1..T host threads.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_probe tools/exp/graph_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
__global__ void small(double* p, int n, int spin) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = (i < n) ? p[i] : 0.0;
  for (int k = 0; k < spin; ++k) v = v * 1.0000001 + 1e-9;
  if (i < n) p[i] = v;
}
struct Ctx { hipStream_t s, ps; hipEvent_t ev[64]; double* buf; hipGraphExec_t exec = nullptr; };
static int g_noevents = 0;
static void chain(Ctx& c, int nk, int spin) {
  if (g_noevents) {
    for (int k = 0; k < nk; ++k) small<<<(k % 4 == 0) ? 1 : 8, 256, 0, c.s>>>(c.buf + 4096, 2048, spin);
    return;
  }
  // mimics the potrf structure: every 4th kernel forks to the panel stream and joins back
  CK(hipEventRecord(c.ev[0], c.s));
  CK(hipStreamWaitEvent(c.ps, c.ev[0], 0));
  int e = 1;
  for (int k = 0; k < nk; ++k) {
    if (k % 4 == 0) {
      small<<<1, 256, 0, c.ps>>>(c.buf, 256, spin);
      CK(hipEventRecord(c.ev[e], c.ps));
      CK(hipStreamWaitEvent(c.s, c.ev[e], 0));
      e = e % 62 + 1;
    } else {
      small<<<8, 256, 0, c.s>>>(c.buf + 4096, 2048, spin);
    }
  }
}
int main(int argc, char** argv) {
  int nk = argc > 1 ? atoi(argv[1]) : 32, spin = argc > 2 ? atoi(argv[2]) : 200, reps = 500;
  int maxT = 8;
  g_noevents = argc > 3 ? atoi(argv[3]) : 0;
  std::vector<Ctx> cs(maxT);
  for (auto& c : cs) {
    CK(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c.ps, hipStreamNonBlocking));
    for (auto& e : c.ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipMalloc(&c.buf, 1 << 20));
    CK(hipMemset(c.buf, 0, 1 << 20));
    chain(c, nk, spin);
    CK(hipStreamSynchronize(c.s));
    hipGraph_t g;
    CK(hipStreamBeginCapture(c.s, hipStreamCaptureModeThreadLocal));
    chain(c, nk, spin);
    CK(hipStreamEndCapture(c.s, &g));
    CK(hipGraphInstantiate(&c.exec, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    CK(hipGraphLaunch(c.exec, c.s));
    CK(hipStreamSynchronize(c.s));
  }
  for (int mode = 0; mode < 2; ++mode)
    for (int T : {1, 2, 4, 8}) {
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
          Ctx& c = cs[t];
          for (int r = 0; r < reps; ++r) {
            if (mode == 0) chain(c, nk, spin); else CK(hipGraphLaunch(c.exec, c.s));
          }
          CK(hipStreamSynchronize(c.s));
        });
      for (auto& x : th) x.join();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%s nk=%d spin=%d threads=%d: %.1f us per chain per thread, %.0f chains/s total, %.2f us/kernel\n", mode ? "graph" : "eager", nk, spin, T,
             us / reps, T * reps / (us * 1e-6), us / reps / nk);
    }
  if (g_noevents) {  // one host thread feeding T streams round-robin
    for (int T : {1, 2, 4, 8}) {
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r)
        for (int t = 0; t < T; ++t) chain(cs[t], nk, spin);
      for (int t = 0; t < T; ++t) CK(hipStreamSynchronize(cs[t].s));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("1 host thread, %d streams: %.1f us per chain-round, %.0f chains/s total\n", T, us / reps, T * reps / (us * 1e-6));
    }
  }
  return 0;
}
