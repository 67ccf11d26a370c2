// Probe: can a one-workgroup kernel on a CU-masked stream run concurrently with a chip-filling kernel on another stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256, 2) k_busy(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  for (int i = 0; i < iters; ++i) a = a * b + 1e-9;
  if (a == 12345.0) out[0] = a;
}
__global__ void __launch_bounds__(512) k_small(double* out, int iters, long long* t) {
  extern __shared__ double lds[];
  long long t0 = wall_clock64();
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  for (int i = 0; i < iters; ++i) a = a * b + 1e-9;
  lds[threadIdx.x] = a;
  if (a == 12345.0) out[1] = lds[0];
  if (threadIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("CUs %d\n", cus);
  double* out; long long* t; CK(hipMalloc(&out, 64)); CK(hipMalloc(&t, 64));
  const int words = (cus + 31) / 32;
  for (int mode = 0; mode < 3; ++mode) {
    hipStream_t A, B;
    if (mode == 0) { CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking)); }
    else if (mode == 1) { int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); printf("prio range %d %d\n", lo, hi); CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi)); }
    else {
      std::vector<uint32_t> ma(words, 0xffffffffu), mb(words, 0u);
      ma[0] &= ~1u; mb[0] = 1u;   // CU bit 0 reserved for B
      CK(hipExtStreamCreateWithCUMask(&A, words, ma.data())); CK(hipExtStreamCreateWithCUMask(&B, words, mb.data()));
    }
    CK(hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
    hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a0, A));
      k_busy<<<4 * cus * 4, 256, 0, A>>>(out, 200000);   // ~8 rounds of 2 WG/CU
      CK(hipEventRecord(a1, A));
      CK(hipEventRecord(b0, B));
      k_small<<<1, 512, 130 * 1024, B>>>(out, 20000, t);
      CK(hipEventRecord(b1, B));
      CK(hipDeviceSynchronize());
      float ta, tb, tab; CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&tab, a0, b1));
      printf("mode %d rep %d: busy %.3f ms, small %.3f ms (event span), small done %.3f ms after busy start\n", mode, rep, ta, tb, tab);
    }
    // small alone
    CK(hipEventRecord(b0, B)); k_small<<<1, 512, 130 * 1024, B>>>(out, 20000, t); CK(hipEventRecord(b1, B)); CK(hipDeviceSynchronize());
    float tb; CK(hipEventElapsedTime(&tb, b0, b1)); printf("mode %d small alone %.3f ms\n", mode, tb);
    CK(hipStreamDestroy(A)); CK(hipStreamDestroy(B));
  }
  return 0;
}
