#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_add(int n, const double* a, double* b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i] + 1.0; }
extern "C" int mini_run() {
  double *a, *b; hipStream_t st;
  if (hipMalloc(&a, 800) != hipSuccess || hipMalloc(&b, 800) != hipSuccess) return -1;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipMemsetAsync(a, 0, 800, st);
  k_add<<<1, 128, 0, st>>>(100, a, b);
  hipError_t e = hipStreamSynchronize(st);
  double h[100]; hipMemcpy(h, b, 800, hipMemcpyDeviceToHost);
  printf("mini: %s b[5]=%f\n", hipGetErrorString(e), h[5]);
  return 0;
}
