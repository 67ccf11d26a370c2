// What the pair tiles' per-hit arithmetic (PairHit, ba_schur_tiled.hpp) costs on the fp64 pipe, alone: no LDS, no memory — the
// two cameras' constants and the accumulators in registers, one or two wavefronts per SIMD.  Beside it: 36 independent FMAs per
// iteration (the pipe's issue rate) and a chain of dependent ones.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I realsensecalibration_amd/csrc tools/probes/hit_probe.hip -o build/hit_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ba_schur_tiled.hpp"

using namespace rsba;
template <int K, int W>
__global__ void __launch_bounds__(256, 2) k_probe(int iters, const double* __restrict__ in, double* __restrict__ out, long long* __restrict__ clk) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  if (iters < 0) lds[tid] = 0.0;   // (keeps the allocation)
  double ca[15], cb[15], acc[36];
  for (int i = 0; i < 15; ++i) { ca[i] = in[i] + 1e-6 * (tid & 15); cb[i] = in[16 + i] + 1e-6 * (tid >> 4); }
  ca[14] = cb[14] = 0.0;
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
  double X[3] = {in[32] + 1e-3 * tid, in[33], in[34]};
  const double v0 = in[35], v1 = in[36], v2 = in[37], v3 = in[38], v4 = in[39], v5 = in[40];
  const double step = in[41];
  const long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
  for (int n = 0; n < iters; ++n) {
    if (K == 0) PairHit<false>(ca, cb, X, v0, v1, v2, v3, v4, v5, 1.0, 1.0, acc);
    if (K == 1) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = fma(X[0], ca[i % 14], acc[i]);   // 180 independent FMAs
    }
    if (K == 2) {
#pragma unroll
      for (int r = 0; r < 180; ++r) acc[0] = fma(acc[0], X[0], step);   // 180 dependent FMAs
    }
    X[0] += step; X[1] -= step;
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 7 && tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  double s = 0.0;
  for (int i = 0; i < 36; ++i) s += acc[i];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int K, int W> int Run(const char* name) {
  const int wgs_per_cu = W;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, grid = cus * wgs_per_cu, iters = 20000;
  const size_t lds = wgs_per_cu == 1 ? 150 * 1024 : 79 * 1024;
  CK(hipFuncSetAttribute((const void*)k_probe<K, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<double> h(64, 0.0);
  const double R[9] = {0.9, -0.1, 0.2, 0.12, 0.95, -0.05, -0.18, 0.07, 0.93};
  for (int s = 0; s < 2; ++s) { for (int i = 0; i < 9; ++i) h[16 * s + i] = R[i]; h[16 * s + 9] = 0.1; h[16 * s + 10] = -0.2; h[16 * s + 11] = 3.0; h[16 * s + 12] = 630; h[16 * s + 13] = 631; }
  h[32] = 0.1; h[33] = -0.2; h[34] = 0.3; h[35] = 1e-3; h[36] = 1e-5; h[37] = -2e-5; h[38] = 2e-3; h[39] = 3e-5; h[40] = 1.5e-3; h[41] = 1e-9;
  double *in, *out; long long* clk; CK(hipMalloc(&clk, 16)); CK(hipMalloc(&in, 64 * 8)); CK(hipMalloc(&out, (size_t)grid * 256 * 8));
  CK(hipMemcpy(in, h.data(), 64 * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_probe<K, W><<<grid, 256, lds>>>(2000, in, out, clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); k_probe<K, W><<<grid, 256, lds>>>(iters, in, out, clk); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double cyc = ms * 1e-3 * 2.4e9 / iters;   // cycles per iteration of a wavefront (2.4 GHz)
  long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
  printf("%-34s %d wavefront(s) per SIMD: %7.1f cycles (at 2.4 GHz) per iteration and wavefront = %7.1f per SIMD; clock64 / wall clock: %.0f MHz, %.1f clock64 ticks per iteration\n", name, wgs_per_cu, cyc, cyc / wgs_per_cu,
         100.0 * hc[0] / hc[1], (double)hc[0] / iters);
  (void)hipFree(in); (void)hipFree(out); return 0;
}
int main() {
  if (Run<1, 1>("180 independent v_fma_f64")) return 1;
  if (Run<2, 1>("180 dependent v_fma_f64")) return 1;
  if (Run<0, 1>("PairHit (one hit, ~180 fp64)")) return 1;
  if (Run<1, 2>("180 independent v_fma_f64")) return 1;
  if (Run<2, 2>("180 dependent v_fma_f64")) return 1;
  if (Run<0, 2>("PairHit (one hit, ~180 fp64)")) return 1;
  return 0;
}
