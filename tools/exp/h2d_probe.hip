// Probe: does hipFree of a large device buffer slow down later pageable H2D hipMemcpyAsync calls?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(double* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0; }
int main(int argc, char** argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0: pageable source, 1: pinned source
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  size_t nb = 1638400;
  std::vector<double> host(nb / 8, 1.0);
  double* pinned; CK(hipHostMalloc((void**)&pinned, nb));
  double* dst; CK(hipMalloc((void**)&dst, nb));
  auto h2d = [&](const char* tag) {
    for (int r = 0; r < 3; ++r) {
      double t0 = now();
      CK(hipMemcpyAsync(dst, mode ? pinned : host.data(), nb, hipMemcpyHostToDevice, s));
      double t1 = now();
      CK(hipStreamSynchronize(s));
      printf("%s: enqueue %.3f ms, complete %.3f ms\n", tag, t1 - t0, now() - t0);
    }
  };
  h2d("fresh");
  std::vector<double*> bufs;
  for (int i = 0; i < 6; ++i) { double* p; CK(hipMalloc((void**)&p, (size_t)100 << 20)); touch<<<(100u << 20) / 8 / 256, 256, 0, s>>>(p, ((size_t)100 << 20) / 8); bufs.push_back(p); }
  CK(hipStreamSynchronize(s));
  h2d("after 6 x 100 MB mallocs");
  for (auto p : bufs) CK(hipFree(p));
  bufs.clear();
  h2d("after freeing them");
  for (int i = 0; i < 6; ++i) { double* p; CK(hipMalloc((void**)&p, (size_t)400 << 20)); touch<<<(400u << 20) / 8 / 256, 256, 0, s>>>(p, ((size_t)400 << 20) / 8); bufs.push_back(p); }
  CK(hipStreamSynchronize(s));
  h2d("after 6 x 400 MB mallocs");
  return 0;
}
