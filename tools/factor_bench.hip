// Microbenchmark of the 32 x 32 diagonal factorisation + inverse (one wavefront), the critical path of the reduced
// system's panels: ns per call for DiagFactorInverse and for experimental variants, alone and beside busy MFMA waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/factor_bench.hip -o build/factor_bench && build/factor_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../realsensecalibration_amd/csrc/ba_cholesky.hpp"
#include "factor_variants.hpp"

using namespace rsba;

template <int V>
__global__ void __launch_bounds__(512) k_bench(const double* __restrict__ M, double* __restrict__ out, long long* __restrict__ ticks, int reps, int busy) {
  __shared__ double Src[RSBA_PB * RSBA_PLD], Pre[RSBA_PB * RSBA_PLD], T[RSBA_PB * RSBA_PLD], Lt[RSBA_PB * RSBA_PLD], invd[RSBA_PB];
  __shared__ double Bst[64 * RSBA_PLD];
  __shared__ int s_done;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < RSBA_PB * RSBA_PB; e += blockDim.x) Src[(e >> 5) * RSBA_PLD + (e & 31)] = M[e];
  for (int e = tid; e < 64 * RSBA_PLD; e += blockDim.x) Bst[e] = 1e-3 * (e % 17);
  if (tid == 0) s_done = 0;
  __syncthreads();
  if (wave == 0) {
    bool good = true;
    const long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
      for (int e = lane; e < RSBA_PB * RSBA_PB; e += 64) Pre[(e >> 5) * RSBA_PLD + (e & 31)] = Src[(e >> 5) * RSBA_PLD + (e & 31)];
      __builtin_amdgcn_wave_barrier();
      if (V == 0) good = DiagFactorInverseCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 1) good = FactorV1Call((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 2) good = FactorV2Call((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 3) good = FactorV3Call((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 4) good = FactorV4Call((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 5) good = FactorV5Call((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 6) good = FactorV6Call((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 10) good = FactorK_2_1_1((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 11) good = FactorK_2_1_0((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 12) good = FactorK_1_1_0((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 13) good = FactorK_0_1_0((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 14) good = FactorK_2_0_0((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 15) good = FactorK_0_0_0((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = wall_clock64();
    if (lane == 0) { ticks[0] = t1 - t0; ticks[1] = good; __hip_atomic_store(&s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  } else if ((busy == 1 && wave == 4) || busy == 2 || (busy == 3 && wave != 4)) {
    // busy == 1: only the wave that shares wave 0's SIMD; busy == 2: all seven; busy == 3: the six on the other SIMDs
    const int mi = lane & 15, kk = lane >> 4;
    d4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    while (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double b0 = Bst[(8 * kk + u) * RSBA_PLD + mi], b1 = Bst[(8 * kk + u) * RSBA_PLD + 16 + mi];
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b0, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b1, a1, 0, 0, 0);
      }
    }
    if (a0[0] + a1[1] == 12345.0) out[4096 + tid] = a0[0];
  }
  __syncthreads();
  for (int e = tid; e < RSBA_PB * RSBA_PB; e += blockDim.x) {
    out[e] = Lt[(e >> 5) * RSBA_PLD + (e & 31)];
    out[1024 + e] = T[(e >> 5) * RSBA_PLD + (e & 31)];
  }
  if (tid < RSBA_PB) out[2048 + tid] = invd[tid];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int V>
static int Run(const char* name, const double* dM, const std::vector<double>& Lref, const std::vector<double>& Tref) {
  double* dout; long long* dt;
  CK(hipMalloc(&dout, 8192 * sizeof(double))); CK(hipMalloc(&dt, 2 * sizeof(long long)));
  const int reps = 2000;
  for (int busy = 0; busy < 4; busy += (V == 0 ? 1 : 3)) {
    k_bench<V><<<1, 512>>>(dM, dout, dt, 10, busy);   // warm the code
    k_bench<V><<<1, 512>>>(dM, dout, dt, reps, busy);
    CK(hipDeviceSynchronize());
    long long t[2]; std::vector<double> o(8192);
    CK(hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), dout, 8192 * sizeof(double), hipMemcpyDeviceToHost));
    double eL = 0, eT = 0, mL = 0, mT = 0;
    for (int i = 0; i < 1024; ++i) { eL = std::fmax(eL, std::fabs(o[i] - Lref[i])); mL = std::fmax(mL, std::fabs(Lref[i])); eT = std::fmax(eT, std::fabs(o[1024 + i] - Tref[i])); mT = std::fmax(mT, std::fabs(Tref[i])); }
    printf("%-10s busy=%d  %7.1f ns per call  good=%lld  |L-Lref|/|L| %.2e  |T-Tref|/|T| %.2e\n", name, busy, 10.0 * t[0] / reps, t[1], eL / mL, eT / mT);
  }
  (void)hipFree(dout); (void)hipFree(dt);
  return 0;
}

int main() {
  const int n = 32;
  std::vector<double> B(n * n), M(n * n, 0.0), L(n * n, 0.0), T(n * n, 0.0);
  unsigned long long s = 88172645463325252ULL;
  for (auto& b : B) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (double)(s % 20001) / 10000.0 - 1.0; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += B[i * n + k] * B[j * n + k]; M[i * n + j] = a + (i == j ? 8.0 : 0.0); }
  for (int j = 0; j < n; ++j) {   // reference in long double
    long double d = M[j * n + j]; for (int k = 0; k < j; ++k) d -= (long double)L[j * n + k] * L[j * n + k];
    L[j * n + j] = (double)sqrtl(d);
    for (int i = j + 1; i < n; ++i) { long double v = M[i * n + j]; for (int k = 0; k < j; ++k) v -= (long double)L[i * n + k] * L[j * n + k]; L[i * n + j] = (double)(v / sqrtl(d)); }
  }
  for (int c = 0; c < n; ++c) for (int i = c; i < n; ++i) { long double v = i == c ? 1.0L : 0.0L; for (int q = c; q < i; ++q) v -= (long double)L[i * n + q] * T[q * n + c]; T[i * n + c] = (double)(v / L[i * n + i]); }
  double* dM; CK(hipMalloc(&dM, n * n * sizeof(double))); CK(hipMemcpy(dM, M.data(), n * n * sizeof(double), hipMemcpyHostToDevice));
  if (Run<99>("null (bench overhead)", dM, L, T)) return 1;
  if (Run<0>("current", dM, L, T)) return 1;
  if (Run<1>("variant1", dM, L, T)) return 1;
  if (Run<2>("variant2", dM, L, T)) return 1;
  if (Run<3>("v3 pipelined", dM, L, T)) return 1;
  { long long z[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_v3_phase), z, sizeof(z))); }
  if (Run<4>("v3 no inverse", dM, L, T)) return 1;
  { long long z[8]; CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_v3_phase), sizeof(z))); printf("v3 phases (cycles per call): load %.0f half1 %.0f mid %.0f half2 %.0f store %.0f inverse %.0f\n", z[0] / 2010.0, z[1] / 2010.0, z[2] / 2010.0, z[3] / 2010.0, z[4] / 2010.0, z[5] / 2010.0); }
  if (Run<5>("v5 lds bulk", dM, L, T)) return 1;
  if (Run<6>("v5 no inv/pan", dM, L, T)) return 1;
  if (Run<10>("k N2 B1 S1", dM, L, T)) return 1;
  if (Run<11>("k N2 B1 S0", dM, L, T)) return 1;
  if (Run<12>("k N1 B1 S0", dM, L, T)) return 1;
  if (Run<13>("k N0 B1 S0", dM, L, T)) return 1;
  if (Run<14>("k N2 B0 S0", dM, L, T)) return 1;
  if (Run<15>("k N0 B0 S0", dM, L, T)) return 1;
  return 0;
}
