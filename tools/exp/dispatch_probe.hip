// dispatch_probe.hip — how fast does the chip hand out workgroups?  (DESIGN.md "what is open" item 2)
// A kernel whose workgroups each hold `lds` bytes of LDS and stay for `spin` x ~0.64 us (s_sleep) and nothing else:
// duration of a launch of G workgroups of T threads, G = 256 ... 8192, T = 64 ... 1024, by HIP events over 20 launches.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/exp/dispatch_probe tools/exp/dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void probe(int spin, int* out) {
  extern __shared__ int sm[];
  if (threadIdx.x == 0) sm[0] = blockIdx.x;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127); // ~127 x 64 clocks
  __syncthreads();
  if (out != nullptr && threadIdx.x == 0 && sm[0] < 0) out[0] = sm[0];
}

int main() {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int lds_sizes[] = {0, 24 * 1024};
  const int spins[] = {0, 1};
  const int threads[] = {64, 128, 256, 512, 1024};
  const int grids[] = {256, 512, 1024, 2048, 4096, 8192};
  for (int lds : lds_sizes)
    for (int spin : spins)
      for (int t : threads) {
        printf("lds %5d spin %d threads %4d :", lds, spin, t);
        for (int g : grids) {
          for (int r = 0; r < 3; ++r) probe<<<g, t, lds, 0>>>(spin, nullptr);
          hipDeviceSynchronize();
          hipEventRecord(a, 0);
          for (int r = 0; r < 20; ++r) probe<<<g, t, lds, 0>>>(spin, nullptr);
          hipEventRecord(b, 0);
          hipEventSynchronize(b);
          float ms = 0.f;
          hipEventElapsedTime(&ms, a, b);
          printf("  G=%d %.1f us", g, 1e3 * ms / 20);
        }
        printf("\n");
      }
  return 0;
}
