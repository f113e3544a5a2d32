// probe: can a stream wait (hipStreamWaitValue32) on a counter that a RUNNING kernel of another stream increments?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__global__ void producer(unsigned* sig, long long* out, int early, long long work_ticks, long long early_ticks) {
  const long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t0;
  bool done = false;
  while (wall_clock64() - t0 < work_ticks) {
    if (!done && (int)blockIdx.x < early && wall_clock64() - t0 >= early_ticks) {
      __syncthreads();
      if (threadIdx.x == 0) {
        out[8 + blockIdx.x] = wall_clock64();          // a payload the consumer checks
        __threadfence_system();
        __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      done = true;
    }
    __builtin_amdgcn_s_sleep(16);
  }
  if (threadIdx.x == 0) atomicMax((unsigned long long*)&out[1], (unsigned long long)wall_clock64());
}
__global__ void stamp(long long* out, int early) {
  out[2] = wall_clock64();
  long long ok = 1;
  for (int i = 0; i < early; ++i) ok &= (out[8 + i] != 0);
  out[3] = ok;
}
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("CanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  unsigned* sig; long long* out;
  CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
  CK(hipMalloc(&out, 1024));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(out, 0, 1024));
    CK(hipStreamWriteValue32(a, sig, 0, 0));
    CK(hipStreamSynchronize(a));
    const int early = 8;
    producer<<<512, 256, 0, a>>>(sig, out, early, 20000 /*200 us*/, 2000 /*20 us*/);
    CK(hipStreamWaitValue32(b, sig, early, hipStreamWaitValueGte, 0xffffffffu));
    stamp<<<1, 1, 0, b>>>(out, early);
    CK(hipStreamSynchronize(b)); CK(hipStreamSynchronize(a));
    long long h[4]; CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    printf("rep %d: producer ran %.1f us; consumer kernel started %.1f us after the producer's start (signals at 20 us), payload visible: %lld\n",
           rep, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, h[3]);
  }
  return 0;
}
