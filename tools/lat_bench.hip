// Dependent-instruction latencies on one wavefront of gfx950 (what a step of the diagonal factorisation is made of).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lat_bench.hip -o build/lat_bench && build/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double ReadLaneD(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

#define N 512
template <int K>
__global__ void __launch_bounds__(64) k_lat(double* out, long long* t, double a, double b) {
  __shared__ double sh[128];
  const int lane = threadIdx.x;
  double x = 1.0 + 1e-9 * lane;
  sh[lane] = x; sh[64 + lane] = x;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (K == 0) { x = __builtin_fma(x, a, b); }
    if (K == 1) { x = __builtin_amdgcn_rsq(x); }
    if (K == 2) { const double s = ReadLaneD(x, i & 31); x = __builtin_fma(x, a, s); }                 // VALU -> readlane -> SGPR -> VALU
    if (K == 3) { sh[lane] = x; __builtin_amdgcn_wave_barrier(); x = __builtin_fma(sh[i & 31], a, b); __builtin_amdgcn_wave_barrier(); }   // LDS write -> broadcast read
    if (K == 4) { const int lo = __builtin_amdgcn_ds_bpermute((i & 31) * 4, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute((i & 31) * 4, __double2hiint(x)); x = __builtin_fma(__hiloint2double(hi, lo), a, b); }
    if (K == 5) { x = x * a; }
    if (K == 6) { float f = (float)x; f = __builtin_amdgcn_rsqf(f); x = (double)f; }
    if (K == 7) { const double s = ReadLaneD(x, i & 31); x = s * a; }   // readlane -> v_mul with SGPR (result again per lane)
    if (K == 8) { x = __builtin_amdgcn_rcp(x); }
    // what "acquire" costs: an agent-scope acquire fence (buffer_inv sc1) alone; the fence + a dependent plain load of a line
    // another kernel wrote; the same load with agent scope (sc1) and no fence
    if (K == 10) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); x = __builtin_fma(x, a, b); }
    if (K == 11) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); x = __builtin_fma(out[64 + ((i * 16) & 1023) + (lane & 1)], a, x); }
    if (K == 12) { x = __builtin_fma(__hip_atomic_load(out + 64 + ((i * 16) & 1023) + (lane & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, x); }
    if (K == 13) { x = __builtin_fma(out[64 + ((i * 16) & 1023) + (lane & 1)], a, x); }
    if (K == 9) { x = __builtin_fma(x, a, b); x = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x130 /* row_shr? */, 0xf, 0xf, false) == 12345 ? 0.0 : x; }
    asm volatile("" : "+v"(x));
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  out[lane] = x;
  if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int K> int Run(const char* name) {
  double* o; long long* t; CK(hipMalloc(&o, (64 + 2048) * 8)); CK(hipMemset(o, 0, (64 + 2048) * 8)); CK(hipMalloc(&t, 16));
  k_lat<K><<<1, 64>>>(o, t, 0.999, 1e-3); k_lat<K><<<1, 64>>>(o, t, 0.999, 1e-3); CK(hipDeviceSynchronize());
  long long h[2]; CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
  printf("%-44s %7.1f clock64 ticks, %6.2f ns per link\n", name, (double)h[0] / N, 10.0 * h[1] / N);
  (void)hipFree(o); (void)hipFree(t); return 0;
}
int main() {
  if (Run<0>("v_fma_f64 chain")) return 1;
  if (Run<5>("v_mul_f64 chain")) return 1;
  if (Run<1>("v_rsq_f64 chain")) return 1;
  if (Run<8>("v_rcp_f64 chain")) return 1;
  if (Run<6>("cvt f32 + v_rsq_f32 + cvt f64 chain")) return 1;
  if (Run<2>("readlane pair -> fma(vgpr, vgpr, sgpr) chain")) return 1;
  if (Run<7>("readlane pair -> mul(sgpr) chain")) return 1;
  if (Run<3>("ds_write -> broadcast ds_read -> fma chain")) return 1;
  if (Run<4>("ds_bpermute pair -> fma chain")) return 1;
  if (Run<10>("acquire fence (agent) + fma chain")) return 1;
  if (Run<11>("acquire fence (agent) + dependent plain load")) return 1;
  if (Run<12>("dependent agent-scope load, no fence")) return 1;
  if (Run<13>("dependent plain load (L2 hit after the first pass)")) return 1;
  return 0;
}
