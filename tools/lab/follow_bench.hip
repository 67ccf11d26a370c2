// Micro-benchmark + check of the follower factorisation (csrc/ba_cholesky_follow.hpp) against the product's routines:
//   V0  DiagFactorInverseCall (factor + inverse)                     V1  DiagFactorOnlyCall (factor alone)
//   V2  DiagFactorFollowACall (factor + 32 follower rows)            V3  DiagFactorFollowABCall (factor + 64 follower rows)
//   V4  DiagFactorOnlyCall + TrsmRowsQuad of 32 rows by wavefronts 0, 1 (what the tiled factorisation does for its rows 32..63)
// alone on its CU and beside seven wavefronts of matrix-core work.  Reference: long double on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/lab/follow_bench.hip -o build/follow_bench && build/follow_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../realsensecalibration_amd/csrc/ba_cholesky_follow.hpp"

using namespace rsba;

template <int V>
__global__ void __launch_bounds__(512) k_bench(const double* __restrict__ M /* 96 x 96 */, double* __restrict__ out, long long* __restrict__ ticks, int reps, int busy) {
  __shared__ double Src[3 * RSBA_PB * RSBA_PLD], Pre[RSBA_PB * RSBA_PLD], FA[RSBA_PB * RSBA_PLD], FB[RSBA_PB * RSBA_PLD], T[RSBA_PB * RSBA_PLD], Lt[RSBA_PB * RSBA_PLD], invd[RSBA_PB];
  __shared__ double LtT[RSBA_PB * RSBA_PLD], Xq[RSBA_PB * RSBA_PLD];
  __shared__ double Bst[64 * RSBA_PLD];
  __shared__ int s_done;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 3 * RSBA_PB * RSBA_PB; e += blockDim.x) { const int b = e >> 10, r = (e >> 5) & 31, c = e & 31; Src[b * RSBA_PB * RSBA_PLD + r * RSBA_PLD + c] = M[(32 * b + r) * 96 + c]; }
  for (int e = tid; e < 64 * RSBA_PLD; e += blockDim.x) Bst[e] = 1e-3 * (e % 17);
  if (tid == 0) s_done = 0;
  __syncthreads();
  if (wave == 0 || (V == 4 && wave == 1)) {
    bool good = true;
    const long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
      for (int e = lane + 64 * wave; e < RSBA_PB * RSBA_PB; e += (V == 4 ? 128 : 64)) {
        const int o = (e >> 5) * RSBA_PLD + (e & 31);
        Pre[o] = Src[o]; FA[o] = Src[RSBA_PB * RSBA_PLD + o]; FB[o] = Src[2 * RSBA_PB * RSBA_PLD + o];
      }
      if (V == 4) { __builtin_amdgcn_s_barrier(); } else __builtin_amdgcn_wave_barrier();
      if (V == 0) good = DiagFactorInverseCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 1) good = DiagFactorOnlyCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
      if (V == 2) good = DiagFactorFollowACall((lds_double*)Pre, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, (lds_double*)FA, lane) && good;
      if (V == 3) good = DiagFactorFollowABCall((lds_double*)Pre, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, (lds_double*)FA, (lds_double*)FB, lane) && good;
      if (V == 4) {
        if (wave == 0) good = DiagFactorOnlyCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && good;
        __builtin_amdgcn_s_barrier();
        for (int e = tid; e < RSBA_PB * RSBA_PB; e += 128) { const int j = e >> 5, c = e & 31; LtT[j * RSBA_PLD + c] = c > j ? Lt[c * RSBA_PLD + j] : (c == j ? invd[j] : 0.0); }
        __builtin_amdgcn_s_barrier();
        TrsmRowsQuad(FA, RSBA_PLD, LtT, Xq, tid, 32);
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = wall_clock64();
    if (tid == 0) { ticks[0] = t1 - t0; ticks[1] = good; __hip_atomic_store(&s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  } else if (busy == 2 && wave >= 2) {
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
    if (a0[0] + a1[1] == 12345.0) out[8000 + tid] = a0[0];
  }
  __syncthreads();
  for (int e = tid; e < RSBA_PB * RSBA_PB; e += blockDim.x) {
    const int o = (e >> 5) * RSBA_PLD + (e & 31);
    out[e] = Lt[o]; out[1024 + e] = V == 4 ? Xq[o] : FA[o]; out[2048 + e] = FB[o];
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int V>
static int Run(const char* name, const double* dM, const std::vector<double>& L, int nfol) {
  double* dout; long long* dt;
  CK(hipMalloc(&dout, 8600 * sizeof(double))); CK(hipMalloc(&dt, 2 * sizeof(long long)));
  const int reps = 2000;
  for (int busy = 0; busy <= (V == 4 ? 0 : 2); busy += 2) {
    // (V4 synchronises its two wavefronts with the workgroup barrier: a workgroup of exactly those two, nothing beside them)
    k_bench<V><<<1, V == 4 ? 128 : 512>>>(dM, dout, dt, 10, busy);
    k_bench<V><<<1, V == 4 ? 128 : 512>>>(dM, dout, dt, reps, busy);
    CK(hipDeviceSynchronize());
    long long t[2]; std::vector<double> o(8600);
    CK(hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), dout, 8600 * sizeof(double), hipMemcpyDeviceToHost));
    double e[3] = {0, 0, 0}, m[3] = {0, 0, 0};
    for (int b = 0; b < 3; ++b) for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
      const double ref = (b == 0 && c > r) ? 0.0 : L[(32 * b + r) * 96 + c];
      e[b] = std::fmax(e[b], std::fabs(o[1024 * b + 32 * r + c] - ref)); m[b] = std::fmax(m[b], std::fabs(ref));
    }
    printf("%-34s busy=%d  %7.1f ns per call  good=%lld  |L11 err| %.1e", name, busy, 10.0 * t[0] / reps, t[1], e[0] / m[0]);
    if (nfol >= 1) printf("  |X(1,0) err| %.1e", e[1] / m[1]);
    if (nfol >= 2) printf("  |X(2,0) err| %.1e", e[2] / m[2]);
    printf("\n");
  }
  (void)hipFree(dout); (void)hipFree(dt);
  return 0;
}

int main() {
  const int n = 96;
  std::vector<double> B(n * n), M(n * n, 0.0), L(n * n, 0.0);
  unsigned long long s = 88172645463325252ULL;
  for (auto& b : B) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (double)(s % 20001) / 10000.0 - 1.0; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += B[i * n + k] * B[j * n + k]; M[i * n + j] = a + (i == j ? 8.0 : 0.0); }
  // reference: the first 32 columns of the Cholesky factor of M (rows 0..95) in long double = L11, X(1,0), X(2,0)
  for (int j = 0; j < 32; ++j) {
    long double d = M[j * n + j]; for (int k = 0; k < j; ++k) d -= (long double)L[j * n + k] * L[j * n + k];
    L[j * n + j] = (double)sqrtl(d);
    for (int i = j + 1; i < n; ++i) { long double v = M[i * n + j]; for (int k = 0; k < j; ++k) v -= (long double)L[i * n + k] * L[j * n + k]; L[i * n + j] = (double)(v / sqrtl(d)); }
  }
  double* dM; CK(hipMalloc(&dM, n * n * sizeof(double))); CK(hipMemcpy(dM, M.data(), n * n * sizeof(double), hipMemcpyHostToDevice));
  if (Run<0>("V0 factor + inverse", dM, L, 0)) return 1;
  if (Run<1>("V1 factor only", dM, L, 0)) return 1;
  if (Run<2>("V2 factor + 32 followers", dM, L, 1)) return 1;
  if (Run<3>("V3 factor + 64 followers", dM, L, 2)) return 1;
  if (Run<4>("V4 factor, then TrsmRowsQuad(32)", dM, L, 1)) return 1;
  return 0;
}
