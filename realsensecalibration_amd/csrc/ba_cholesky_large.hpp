// Dense SPD solve of the reduced camera system for n > RSBA_CHOL_MAXN (more than 64 cameras; 1536 x 1536 for the
// 256-camera configuration): the matrix no longer fits one workgroup's LDS, so the factorisation is right-looking and
// spread over the whole chip, one launch per 32-wide panel:
//
//   k_sys_build   <<<n+1 rows>>>   W = scaled, damped, mirrored system + rhs row (and the optional copies)
//   k_chol_step   <<<tiles>>>      per panel kb: every workgroup factors the 32 x 32 diagonal block itself (redundantly:
//                                  9 us of one wavefront, cheaper than another launch and a round trip through HBM),
//                                  solves its own two 64-row strips of the panel against it on the matrix cores
//                                  (X = Rows T', T = L11^-1) and applies W[I,J] -= X_I X_J' to its 64 x 64 tile of the
//                                  trailing matrix (lower triangle + the rhs row, which rides along as row n).
//                                  Diagonal tiles also store their X strip into the factor F, tile 0 the diagonal block.
//   k_chol_finish <<<1>>>          blocked back-substitution with the stored T blocks + the camera-step epilogue
//
// The panel columns of W are only read during a step (the update touches columns >= kb+32) and F is only written, so
// no tile depends on another inside a launch; summation order is fixed by the tile shape: bitwise reproducible.
// F has the layout CholeskySolvePanelLDS leaves in A: L in the lower triangle, T in the strict upper triangle of
// each diagonal block, y in row n, inverse pivots in row n+1.
#pragma once
#include "ba_cholesky.hpp"

namespace rsba {

#define RSBA_CT 64   // trailing-update tile

__global__ void __launch_bounds__(256)
k_sys_build(const double* __restrict__ red, RedLayout L, double* __restrict__ W /* (n+1) x n */, double* __restrict__ S_copy,
            double* __restrict__ rhs_copy, double* __restrict__ scale_c, IterParams ip, int sym_full, int* __restrict__ ok_flag) {
  const int n = L.nc, tid = threadIdx.x;
  // at the first iteration the scale is defined here (from diagU) and nobody may read scale_c yet
  auto sc = [&](int i) { return ip.first ? (ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(red[L.diagU() + i])) : 1.0) : scale_c[i]; };
  if (blockIdx.x == 0 && tid == 0) *ok_flag = 1;
  for (int i = blockIdx.x; i <= n; i += gridDim.x) {
    if (i == n) {
      for (int j = tid; j < n; j += blockDim.x) {
        const double v = sc(j) * (red[L.gc() + j] + red[L.corr() + j]);
        W[(size_t)n * n + j] = v;
        if (rhs_copy) rhs_copy[j] = v;
      }
      continue;
    }
    const int bi = i / 6;
    const double si = sc(i);
    if (ip.first && tid == 0) scale_c[i] = si;
    for (int j = tid; j < n; j += blockDim.x) {
      const int bj = j / 6;
      const bool upper = sym_full || (bi < bj) || (bi == bj && i <= j);
      const double raw = upper ? red[L.S() + (size_t)i * n + j] : red[L.S() + (size_t)j * n + i];
      double v = raw * (si * sc(j));
      if (i == j) v += fmin(fmax(si * si * red[L.diagU() + i], ip.min_lm_diagonal), ip.max_lm_diagonal) / ip.radius;
      W[(size_t)i * n + j] = v;
      if (S_copy) S_copy[(size_t)i * n + j] = v;
    }
  }
}

__host__ __device__ inline size_t CholStepLdsDoubles() { return 3 * RSBA_PB * RSBA_PLD + 64 + 2 * RSBA_CT * RSBA_PLD; }

__global__ void __launch_bounds__(256)
k_chol_step(int n, int kb, double* __restrict__ W, double* __restrict__ F, int* __restrict__ ok_flag) {
  extern __shared__ double lds[];
  double* Pan = lds;                          // 32 x 33 diagonal block
  double* T = Pan + RSBA_PB * RSBA_PLD;
  double* Lt = T + RSBA_PB * RSBA_PLD;
  double* invd = Lt + RSBA_PB * RSBA_PLD;     // 32 (+32 spare)
  double* XI = invd + 64;                     // 64 x 33
  double* XJ = XI + RSBA_CT * RSBA_PLD;       // 64 x 33
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = min(RSBA_PB, n - kb);
  const int r0 = kb + nb;                     // first trailing row/column
  // tile (I, J), I >= J, from the linear index
  const int t = blockIdx.x;
  int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const int ri = r0 + I * RSBA_CT, rj = r0 + J * RSBA_CT;

  // diagonal block and the two strips of the panel (rows past n and columns past nb read as zero)
  for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) {
    const int r = e >> 5, c = e & 31;
    Pan[r * RSBA_PLD + c] = (r < nb && c < nb) ? W[(size_t)(kb + r) * n + kb + c] : 0.0;
  }
  for (int e = tid; e < RSBA_CT * RSBA_PB; e += 256) {
    const int r = e >> 5, c = e & 31;
    XI[r * RSBA_PLD + c] = (ri + r <= n && c < nb) ? W[(size_t)(ri + r) * n + kb + c] : 0.0;
    if (I != J) XJ[r * RSBA_PLD + c] = (rj + r <= n && c < nb) ? W[(size_t)(rj + r) * n + kb + c] : 0.0;
  }
  __syncthreads();
  if (wave == 0) {
    const bool good = DiagFactorInverse(Pan, nb, T, Lt, invd, lane);
    if (!good && t == 0 && lane == 0) *ok_flag = 0;
  }
  __syncthreads();
  if (t == 0) {
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) {
      const int r = e >> 5, c = e & 31;
      if (r < nb && c < nb) F[(size_t)(kb + r) * n + kb + c] = (c > r) ? T[c * RSBA_PLD + r] : Pan[r * RSBA_PLD + c];
    }
    if (tid < nb) F[(size_t)(n + 1) * n + kb + tid] = invd[tid];
  }
  // strips: X = Rows * T' in place, one wave per 16 rows (A-op[i][k] = X[16w+i][k], B-op[k][j] = T[j][k])
  const int i = lane & 15, kk = lane >> 4;
  for (int which = 0; which < (I != J ? 2 : 1); ++which) {
    double* X = which ? XJ : XI;
    d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int qs = 0; qs < RSBA_PB; qs += 4) {
      const double a = X[(16 * wave + i) * RSBA_PLD + qs + kk];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[i * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(16 + i) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();  // the wave's 16 rows are all read before any is overwritten
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int r = 16 * wave + kk + 4 * tt;
      X[r * RSBA_PLD + i] = acc0[tt];
      X[r * RSBA_PLD + 16 + i] = acc1[tt];
    }
  }
  __syncthreads();
  const double* XJr = (I != J) ? XJ : XI;
  if (I == J) {
    // this strip of the factor (the rhs row n included: its entries are y)
    for (int e = tid; e < RSBA_CT * RSBA_PB; e += 256) {
      const int r = e >> 5, c = e & 31;
      if (ri + r <= n && c < nb) F[(size_t)(ri + r) * n + kb + c] = XI[r * RSBA_PLD + c];
    }
  }
  // trailing update of the 64 x 64 tile: wave w owns rows 16w..16w+15, four 16-column blocks
  d4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int qs = 0; qs < RSBA_PB; qs += 4) {
    const double a = XI[(16 * wave + i) * RSBA_PLD + qs + kk];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, XJr[(16 * jb + i) * RSBA_PLD + qs + kk], acc[jb], 0, 0, 0);
  }
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int r = ri + 16 * wave + kk + 4 * tt, c = rj + 16 * jb + i;
      if (r <= n && c < n && (c <= r)) W[(size_t)r * n + c] -= acc[jb][tt];
    }
  }
}

__global__ void __launch_bounds__(1024)
k_chol_finish(int C, const double* __restrict__ red, RedLayout L, double* __restrict__ F, const double* __restrict__ scale_c,
              const double* __restrict__ cam_x, double* __restrict__ cam_c, const double* __restrict__ intr, double* __restrict__ camc_c,
              double* __restrict__ dcam, const double* __restrict__ gmax_p, double* __restrict__ res, const int* __restrict__ ok_flag,
              const double* __restrict__ cam_free) {
  extern __shared__ double lds[];
  const int n = L.nc, tid = threadIdx.x, nt = blockDim.x;
  double* y = BackSubstituteBlocks(n, F, lds);
  double* ysol = F + (size_t)n * n;
  for (int i = tid; i < n; i += nt) ysol[i] = y[i];
  __threadfence_block();
  __syncthreads();
  CameraStepEpilogue(C, red, L, scale_c, ysol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, *ok_flag, lds, cam_free);
}

}  // namespace rsba
