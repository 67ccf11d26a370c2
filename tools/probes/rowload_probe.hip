// Probe: how long does one workgroup's fetch of the rows of a 32-row block (the A operand of the panel update in
// ba_cholesky_diag.hpp, ~90 KB through one CU) take, by the lane -> address pattern of the loads?
//   pattern 0: as DiagUpdateWave / RowUpdateHalf issue them — lane (mi, kk) of the MFMA layout takes doubles 8 kk .. 8 kk + 7 of a
//              32-column slab in four 16-byte loads: every instruction touches both 128-byte lines of the slab in all 16 rows,
//              every line is touched by four instructions;
//   pattern 1: lane (mi, kk) takes doubles 4 kk .. 4 kk + 3 of each 16-column half slab in two 16-byte loads back to back: an
//              instruction touches one line per row, a line is touched by two instructions;
//   pattern 2: pattern 0's addresses, all first touches first (v2-major);
//   pattern 3: row-wise (8 lanes per line, every line touched once): what the memory system can do for this volume;
//   pattern 4: the rows stored k-major, fetched in the lane layout of the matrix cores (four whole lines per instruction).
// Optionally beside a chip-filling streaming kernel on another stream (argv[1] = 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2_t __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256, 2) k_stream(const double* __restrict__ src, size_t nelem, double* out, int rounds) {
  double acc = 0.0;
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nelem; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc == 12345.678) out[0] = acc;
}

template <int kPattern>
__global__ void __launch_bounds__(512) k_rows(const double* __restrict__ mats, int nmat, int n, int p, long long* t, double* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mi = lane & 15, kk = lane >> 4;
  const bool worker = wave != 0 && wave != 4;
  const int wk = wave < 4 ? wave - 1 : wave - 2, h = wk & 1, ks = wk >> 1;
  const int qper = (p + 2) / 3, sa = ks * qper, sb = min(p - 1, (ks + 1) * qper);
  double sink = 0.0;
  for (int m = 0; m < nmat; ++m) {
    const double* A = mats + (size_t)m * n * n;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (worker) {
      double pf[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) pf[i][u] = 0.0;
      const int nb0 = 32 * (p + 1);
      if (kPattern == 0) {
        const double* arow = A + (size_t)(nb0 + h * 16 + mi) * n + 8 * kk;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (sa + i < sb) {
#pragma unroll
            for (int v2 = 0; v2 < 4; ++v2) { const d2_t x = *reinterpret_cast<const d2_t*>(arow + (sa + i) * 32 + 2 * v2); pf[i][2 * v2] = x[0]; pf[i][2 * v2 + 1] = x[1]; }
          }
      } else if (kPattern == 1) {
        const double* arow = A + (size_t)(nb0 + h * 16 + mi) * n + 4 * kk;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (sa + i < sb) {
#pragma unroll
            for (int v2 = 0; v2 < 4; ++v2) { const d2_t x = *reinterpret_cast<const d2_t*>(arow + (sa + i) * 32 + 16 * (v2 >> 1) + 2 * (v2 & 1)); pf[i][2 * v2] = x[0]; pf[i][2 * v2 + 1] = x[1]; }
          }
      } else if (kPattern == 2) {
        const double* arow = A + (size_t)(nb0 + h * 16 + mi) * n + 8 * kk;
#pragma unroll
        for (int v2 = 0; v2 < 4; ++v2)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (sa + i < sb) { const d2_t x = *reinterpret_cast<const d2_t*>(arow + (sa + i) * 32 + 2 * v2); pf[i][2 * v2] = x[0]; pf[i][2 * v2 + 1] = x[1]; }
      } else if (kPattern == 4) {
        // the rows stored k-major (AT[k][row]): lane (mi, kk) takes k = 8 kk + u of row mi in eight 8-byte loads — an instruction
        // touches four whole lines (sixteen consecutive rows of four k)
        const double* acol = A + (size_t)(8 * kk) * n + nb0 + h * 16 + mi;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (sa + i < sb) {
#pragma unroll
            for (int u = 0; u < 8; ++u) pf[i][u] = acol[(size_t)((sa + i) * 32 + u) * n];
          }
      } else {
        // 16 rows x (sb - sa) slabs x 256 bytes: an instruction takes 8 rows x one line (lane & 7: the 16-byte chunk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (sa + i < sb) {
#pragma unroll
            for (int v2 = 0; v2 < 4; ++v2) {
              const int row = (lane >> 3) + 8 * (v2 & 1), line = v2 >> 1;
              const d2_t x = *reinterpret_cast<const d2_t*>(A + (size_t)(nb0 + h * 16 + row) * n + (sa + i) * 32 + 16 * line + 2 * (lane & 7));
              pf[i][2 * v2] = x[0]; pf[i][2 * v2 + 1] = x[1];
            }
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) sink += pf[i][u];
    }
    __syncthreads();
    if (tid == 0) t[m] = wall_clock64() - t0;
  }
  if (sink == 12345.678) out[tid] = sink;
}

int main(int argc, char** argv) {
  const int noise = argc > 1 ? atoi(argv[1]) : 0;
  const int n = 384, nmat = 200;
  double* mats; CK(hipMalloc(&mats, (size_t)nmat * n * n * 8)); CK(hipMemset(mats, 0, (size_t)nmat * n * n * 8));
  long long* t; CK(hipMalloc(&t, nmat * 8));
  double* out; CK(hipMalloc(&out, 4096));
  const size_t big = (size_t)1 << 28;   // 2 GiB of doubles for the background stream
  double* src = nullptr;
  hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  if (noise) { CK(hipMalloc(&src, big * 8)); CK(hipMemset(src, 0, big * 8)); }
  std::vector<long long> ht(nmat);
  for (int p : {4, 8, 10}) {
    for (int pat = 0; pat < 5; ++pat) {
      CK(hipDeviceSynchronize());
      if (noise) k_stream<<<512, 256, 0, A>>>(src, big, out, noise);
      if (pat == 0) k_rows<0><<<1, 512, 0, B>>>(mats, nmat, n, p, t, out);
      if (pat == 1) k_rows<1><<<1, 512, 0, B>>>(mats, nmat, n, p, t, out);
      if (pat == 2) k_rows<2><<<1, 512, 0, B>>>(mats, nmat, n, p, t, out);
      if (pat == 3) k_rows<3><<<1, 512, 0, B>>>(mats, nmat, n, p, t, out);
      if (pat == 4) k_rows<4><<<1, 512, 0, B>>>(mats, nmat, n, p, t, out);
      CK(hipStreamSynchronize(B));
      CK(hipMemcpy(ht.data(), t, nmat * 8, hipMemcpyDeviceToHost));
      CK(hipDeviceSynchronize());
      double sum = 0; long long mn = 1 << 30, mx = 0;
      for (int m = 20; m < nmat; ++m) { sum += ht[m]; mn = ht[m] < mn ? ht[m] : mn; mx = ht[m] > mx ? ht[m] : mx; }
      const double kb = 32.0 * 32 * (p - 1) * 8 / 1024;
      printf("noise %d  panel %2d (%5.1f KB)  pattern %d: mean %.2f us  min %.2f  max %.2f\n", noise, p, kb, pat, sum / (nmat - 20) / 100.0, mn / 100.0, mx / 100.0);
    }
  }
  return 0;
}
