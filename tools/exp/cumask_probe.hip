// probe: hipExtStreamCreateWithCUMask bit -> (XCC, CU) mapping on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__global__ void where(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
int run(hipStream_t s, int blocks, const char* name) {
  unsigned* d; CK(hipMalloc(&d, blocks * 8));
  where<<<blocks, 256, 0, s>>>(d, 2000);
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(2 * blocks); CK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
  std::set<unsigned> xs; std::set<unsigned long long> cus;
  for (int b = 0; b < blocks; ++b) {
    unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    xs.insert(xcc); cus.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu);
  }
  printf("%-28s blocks %5d -> %zu XCCs, %zu distinct (xcc,se,sh,cu)\n", name, blocks, xs.size(), cus.size());
  if (cus.size() <= 16) { for (auto c : cus) printf("   xcc %llu se %llu sh %llu cu %llu\n", c >> 16, (c >> 8) & 0xff, (c >> 4) & 0xf, c & 0xf); }
  hipFree(d);
  return 0;
}
int main() {
  hipStream_t all, res, rest;
  CK(hipStreamCreate(&all));
  uint32_t m_res[8] = {0xff, 0, 0, 0, 0, 0, 0, 0};             // bits 0..7
  uint32_t m_rest[8]; for (int i = 0; i < 8; ++i) m_rest[i] = 0xffffffffu; m_rest[0] = 0xffffff00u;
  CK(hipExtStreamCreateWithCUMask(&res, 8, m_res));
  CK(hipExtStreamCreateWithCUMask(&rest, 8, m_rest));
  run(all, 4096, "unmasked");
  run(res, 256, "mask bits 0..7");
  run(rest, 4096, "mask all but bits 0..7");
  uint32_t m8[8] = {0x01010101u, 0x01010101u, 0, 0, 0, 0, 0, 0}; // bits 0,8,16,...,56
  hipStream_t s8; CK(hipExtStreamCreateWithCUMask(&s8, 8, m8));
  run(s8, 256, "mask bits 0,8,..,56");
  return 0;
}
