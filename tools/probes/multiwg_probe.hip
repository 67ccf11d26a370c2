// Probe: G workgroups on a CU-masked stream — where do they land (XCC / SE / CU) and how long does a flag round trip
// between two of them take (sc1 store + relaxed poll), for different mask bit sets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(512) k_probe(int* flags, unsigned* ids, long long* ticks, double* data, int rounds) {
  extern __shared__ double lds[];
  const int w = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) {
    ids[2 * w] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
    ids[2 * w + 1] = __builtin_amdgcn_s_getreg((3 << 11) | 20); // XCC_ID
  }
  lds[threadIdx.x] = 0.0;
  __syncthreads();
  // token ring: WG w waits for flags[w] == r, writes 64 doubles of payload (sc1), then sets flags[(w+1)%G] = r (+1 when wrapping)
  long long t0 = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (threadIdx.x < 64) {
      const int want = (w == 0) ? r - 1 : r;
      if (threadIdx.x == 0) {
        if (!(w == 0 && r == 1)) while (__hip_atomic_load(&flags[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
        if (w == 0 && r == 1) t0 = wall_clock64();
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const double v = (r > 1 || w > 0) ? data[((w + G - 1) % G) * 64 + threadIdx.x] : 0.0;   // what the predecessor wrote
      __hip_atomic_store(&data[w * 64 + threadIdx.x], v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      if (threadIdx.x == 0) __hip_atomic_store(&flags[(w + 1) % G], (w == G - 1) ? r : r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (w == 0 && threadIdx.x == 0) {
    while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != rounds) __builtin_amdgcn_s_sleep(1);
    ticks[0] = wall_clock64() - t0;
  }
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  int* flags; unsigned* ids; long long* ticks; double* data;
  CK(hipMalloc(&flags, 64 * 4)); CK(hipMalloc(&ids, 64 * 4)); CK(hipMalloc(&ticks, 64)); CK(hipMalloc(&data, 64 * 64 * 8));
  CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const int G = 4, rounds = 200;
  int sets[8][4] = {{0, 1, 2, 3}, {0, 8, 16, 24}, {0, 32, 64, 96}, {0, 2, 4, 6}, {0, 4, 8, 12}, {0, 16, 32, 48}, {-1, -1, -1, -1}, {0, 0, 0, 0}};
  for (int s = 0; s < 8; ++s) {
    std::vector<uint32_t> m(words, sets[s][0] < 0 ? 0xffffffffu : 0u);
    for (int k = 0; k < G; ++k) if (sets[s][k] >= 0) m[sets[s][k] / 32] |= 1u << (sets[s][k] % 32);
    fflush(stdout);
    hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, words, m.data()));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(flags, 0, 64 * 4, st)); CK(hipMemsetAsync(data, 0, 64 * 64 * 8, st));
      k_probe<<<G, 512, 100 * 1024, st>>>(flags, ids, ticks, data, rounds);
      CK(hipStreamSynchronize(st));
    }
    unsigned h[8]; long long t; double d[64];
    CK(hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(d, data, sizeof(d), hipMemcpyDeviceToHost));
    printf("mask bits {%d,%d,%d,%d}:", sets[s][0], sets[s][1], sets[s][2], sets[s][3]);
    for (int w = 0; w < G; ++w) printf("  wg%d xcc %u se %u cu %u", w, h[2 * w + 1] & 15, (h[2 * w] >> 13) & 7, (h[2 * w] >> 8) & 15);
    printf("  | hop %.3f us (ring of %d, %d rounds, data[0]=%.0f)\n", t / 100.0 / (rounds * G), G, rounds, d[0]);
    CK(hipStreamDestroy(st));
  }
  return 0;
}
