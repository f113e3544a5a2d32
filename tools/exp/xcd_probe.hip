// probe: which XCD does workgroup b of a launch run on?  (the rule the XCD-aware tile orders of gemm_f64.hip assume:
// b % 8.)  Workgroups with 64 KB of LDS and 256 threads (2 per CU, like the big-tile GEMM); a fraction of them exits
// at once (like the tiles above the diagonal), the others spin for `work_us`.  Prints how many workgroups ran on XCD
// (linear id % 8) and, if not all, the histogram of (xcc - id) mod 8, for 1-D and 2-D grids.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_probe.hip -o xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int exit_mod, int work_ticks) {
  extern __shared__ double smem[];
  const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[lin] = xcc & 0xf;
  if (exit_mod > 0 && (int)(blockIdx.x % (unsigned)exit_mod) >= exit_mod / 2) return; // "above the diagonal"
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0 && work_ticks < 0) smem[0] = 1.0;
}

static int run(const char* name, dim3 grid, int exit_mod, int work_us) {
  const size_t n = (size_t)grid.x * grid.y;
  unsigned* d;
  CK(hipMalloc(&d, n * sizeof(unsigned)));
  CK(hipMemset(d, 0xff, n * sizeof(unsigned)));
  CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  probe<<<grid, 256, 65536>>>(d, exit_mod, work_us * 100);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> h(n);
  CK(hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost));
  size_t ok = 0;
  size_t hist[8] = {0}, per[8] = {0};
  for (size_t b = 0; b < n; ++b) {
    if (h[b] == (b & 7)) ++ok;
    hist[(h[b] + 8 - (b & 7)) & 7]++;
    per[h[b] & 7]++;
  }
  printf("%-44s %6zu workgroups, on XCD id %% 8: %6zu (%.1f %%) | (xcc - id) mod 8:", name, n, ok, 100.0 * ok / n);
  for (int i = 0; i < 8; ++i) printf(" %zu", hist[i]);
  printf(" | per XCD:");
  for (int i = 0; i < 8; ++i) printf(" %zu", per[i]);
  printf("\n");
  CK(hipFree(d));
  return 0;
}

int main() {
  run("1-D 4096, all work 50 us", dim3(4096), 0, 50);
  run("1-D 12288, half exits at once", dim3(12288), 2, 50);
  run("1-D 12288, 3 of 4 exit in runs of 16", dim3(12288), 64, 50);
  run("2-D 77 x 77, all work", dim3(77, 77), 0, 50);
  run("2-D 77 x 77, upper part of each row exits", dim3(77, 77), 77, 50);
  run("2-D 80 x 77, all work", dim3(80, 77), 0, 50);
  run("1-D 3528 (swizzle-like), all work 170 us", dim3(3528), 0, 170);
  return 0;
}
