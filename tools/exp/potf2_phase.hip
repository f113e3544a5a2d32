// Phase timing of potf2_tile_kernel (s_memtime at every barrier).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPX_POTF2_TRACE -I gpax_amd/csrc -o tools/exp/potf2_phase tools/exp/potf2_phase.hip
#include "../../gpax_amd/csrc/potf2.hip"
#include <vector>
int main() {
  const int n = 128;
  std::vector<double> h(n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? n : 0.0) + 1.0 / (1.0 + abs(i - j));
  double *dA, *dL;
  hipMalloc(&dA, n * n * 8);
  hipMalloc(&dL, n * n * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gpx::potf2_tile_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)gpx::POTF2_TILE_LDS);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gpx::potf2_tile_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)gpx::POTF2_TILE_LDS);
  long long t[64];
  for (int variant = 0; variant < 2; ++variant) {
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, h.data(), n * n * 8, hipMemcpyHostToDevice);
    if (variant == 0) gpx::potf2_tile_kernel<false><<<1, 256, gpx::POTF2_TILE_LDS>>>(dA, n, dL, nullptr, 0, 0, 0, nullptr, 0);
    else gpx::potf2_tile_kernel<true><<<1, 256, gpx::POTF2_TILE_LDS>>>(dA, n, dL, nullptr, 0, 0, 0, nullptr, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(t, HIP_SYMBOL(gpx::gpx_potf2_trace), sizeof t);
  }
  printf("%s\n", variant == 0 ? "diagonal tiles column by column (round 1)" : "diagonal tiles blocked by 4 columns");
  printf("shader-clock cycles per phase\n p:   A(dump)  B(diag)  C(trsm)  D(update)\n");
  long long tot[4] = {0, 0, 0, 0};
  for (int p = 0; p < 8; ++p) {
    long long a = t[1 + 4 * p] - t[4 * p], b = t[2 + 4 * p] - t[1 + 4 * p], c = t[3 + 4 * p] - t[2 + 4 * p],
              d = t[4 + 4 * p] - t[3 + 4 * p];
    printf(" %d: %8lld %8lld %8lld %8lld\n", p, a, b, c, d);
    tot[0] += a; tot[1] += b; tot[2] += c; tot[3] += d;
  }
  printf("sum: %7lld %8lld %8lld %8lld   total %lld\n", tot[0], tot[1], tot[2], tot[3], t[32] - t[0]);
  }
  return 0;
}
