// Standalone A/B harness for the fp64 MFMA NT GEMM main loop (not part of libgpx).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_variants.hip -o gemm_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// V: 0 baseline (LDT=18, compiler-chosen reads), 1 = explicit ds_read_b64 via asm (no read2 merge),
//    2 = LDT 17 with b64 stores, 3 = baseline without sched_barrier, 4 = setprio around MFMA,
//    5 = BK 32
template <int V, int BK, int LDT>
__global__ __launch_bounds__(256, 2) void gemm_var(const double* A, long lda, const double* B, long ldb, double* C,
                                                   long ldc, int K) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int TD = 128 * LDT;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fk = lane >> 4;
  constexpr int TPR = BK / 2;           // threads per row (2 doubles each)
  constexpr int RPP = 256 / TPR;        // rows per pass
  constexpr int NP = 128 / RPP;         // passes
  const int lr = tid / TPR, lc = (tid % TPR) * 2;
  const double* Ap = A + ((long)by * 128 + lr) * lda + lc;
  const double* Bp = B + ((long)bx * 128 + lr) * ldb + lc;
  double* sA0 = smem; double* sB0 = smem + TD; double* sA1 = smem + 2 * TD; double* sB1 = smem + 3 * TD;
  d4_t acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = d4_t{0, 0, 0, 0};
  const int nk = K / BK;
  double2 ra[NP], rb[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    ra[i] = *(const double2*)(Ap + (long)i * RPP * lda);
    rb[i] = *(const double2*)(Bp + (long)i * RPP * ldb);
  }
  const int st = lr * LDT + lc;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (LDT % 2 == 0) {
      *(double2*)(sA0 + st + i * RPP * LDT) = ra[i];
      *(double2*)(sB0 + st + i * RPP * LDT) = rb[i];
    } else {
      sA0[st + i * RPP * LDT] = ra[i].x; sA0[st + i * RPP * LDT + 1] = ra[i].y;
      sB0[st + i * RPP * LDT] = rb[i].x; sB0[st + i * RPP * LDT + 1] = rb[i].y;
    }
  }
  __syncthreads();
  const int aoff = (wr * 64 + fr) * LDT + fk, boff = (wc * 64 + fr) * LDT + fk;
  for (int kt = 0; kt < nk; ++kt) {
    const double* cA = (kt & 1) ? sA1 : sA0;
    const double* cB = (kt & 1) ? sB1 : sB0;
    double* nA = (kt & 1) ? sA0 : sA1;
    double* nB = (kt & 1) ? sB0 : sB1;
    const int koff = ((kt + 1 < nk) ? kt + 1 : kt) * BK;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      ra[i] = *(const double2*)(Ap + (long)i * RPP * lda + koff);
      rb[i] = *(const double2*)(Bp + (long)i * RPP * ldb + koff);
    }
    if (V == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double af[4], bf[4];
      if (V == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          unsigned a1 = (unsigned)(size_t)(cA + aoff + m * 16 * LDT + kk * 4);
          unsigned b1 = (unsigned)(size_t)(cB + boff + m * 16 * LDT + kk * 4);
          asm volatile("ds_read_b64 %0, %1" : "=v"(af[m]) : "v"(a1));
          asm volatile("ds_read_b64 %0, %1" : "=v"(bf[m]) : "v"(b1));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = cA[aoff + m * 16 * LDT + kk * 4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bf[n] = cB[boff + n * 16 * LDT + kk * 4];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    if (V == 4) __builtin_amdgcn_s_setprio(0);
    if (V != 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (LDT % 2 == 0) {
        *(double2*)(nA + st + i * RPP * LDT) = ra[i];
        *(double2*)(nB + st + i * RPP * LDT) = rb[i];
      } else {
        nA[st + i * RPP * LDT] = ra[i].x; nA[st + i * RPP * LDT + 1] = ra[i].y;
        nB[st + i * RPP * LDT] = rb[i].x; nB[st + i * RPP * LDT + 1] = rb[i].y;
      }
    }
    __syncthreads();
  }
  double* Cw = C + ((long)by * 128 + wr * 64 + fk) * ldc + (long)bx * 128 + wc * 64 + fr;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < 4; ++n) Cw[(long)(m * 16 + 4 * r) * ldc + n * 16] = acc[m][n][r];
}

template <int V, int BK, int LDT>
double run(const double* A, const double* B, double* C, int M, int N, int K, long ld, const char* name) {
  size_t lds = (size_t)4 * 128 * LDT * sizeof(double);
  CK(hipFuncSetAttribute((const void*)gemm_var<V, BK, LDT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(N / 128, M / 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gemm_var<V, BK, LDT><<<grid, 256, lds>>>(A, ld, B, ld, C, N, K);
  double best = 1e30;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0));
    gemm_var<V, BK, LDT><<<grid, 256, lds>>>(A, ld, B, ld, C, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double tf = 2.0 * M * N * K / (best * 1e-3) / 1e12;
  double h[4]; CK(hipMemcpy(h, C + 12345, sizeof h, hipMemcpyDeviceToHost));
  printf("%-28s %8.3f ms %6.1f TF   (C[12345]=%g)\n", name, best, tf, h[0]);
  return tf;
}

int main(int argc, char** argv) {
  const int M = 8192, N = 8192, K = 2048; const long ld = K + 16;
  double *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * ld * 8)); CK(hipMalloc(&B, (size_t)N * ld * 8)); CK(hipMalloc(&C, (size_t)M * N * 8));
  std::vector<double> h((size_t)M * ld);
  for (int pass = 0; pass < 1; ++pass) {
    if (pass == 0) { srand(1); for (auto& v : h) v = (double)rand() / RAND_MAX * 2 - 1; }
    else { for (auto& v : h) v = 0.0; }
    CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    printf("--- %s operands ---\n", pass == 0 ? "random" : "zero");
    run<0, 16, 18>(A, B, C, M, N, K, ld, "V0 baseline LDT18");
    run<1, 16, 18>(A, B, C, M, N, K, ld, "V1 asm ds_read_b64");
    run<2, 16, 17>(A, B, C, M, N, K, ld, "V2 LDT17 b64 stores");
    run<3, 16, 18>(A, B, C, M, N, K, ld, "V3 no sched_barrier");
    run<4, 16, 18>(A, B, C, M, N, K, ld, "V4 setprio");
    run<5, 32, 34>(A, B, C, M, N, K, ld, "V5 BK32 LDT34");
    run<5, 32, 33>(A, B, C, M, N, K, ld, "V6 BK32 LDT33");
    run<3, 16, 17>(A, B, C, M, N, K, ld, "V7 LDT17 no sched_barrier");
    run<3, 32, 33>(A, B, C, M, N, K, ld, "V8 BK32 LDT33 no sched");
    run<3, 16, 19>(A, B, C, M, N, K, ld, "V9 LDT19 no sched");
    run<4, 16, 17>(A, B, C, M, N, K, ld, "V10 LDT17 setprio");
    run<2, 16, 21>(A, B, C, M, N, K, ld, "V11 LDT21");
  }
  return 0;
}
