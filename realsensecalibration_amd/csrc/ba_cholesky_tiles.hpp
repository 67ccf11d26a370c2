// Dense SPD factorisation of the reduced camera system for n > 384 (more than 64 cameras) in ONE launch.
//
// ba_cholesky_large.hpp spreads the right-looking factorisation over the chip with one launch per 32-wide panel and lets
// every workgroup refactor the 32 x 32 diagonal block itself (48 launches of ~26 us at 256 cameras).  Here every 64 x 64
// tile (I >= J) of the lower triangle is one RESIDENT workgroup that keeps its tile in LDS from the first panel to the
// last that touches it, and the panels are chained with flags in global memory (the protocol of ba_cholesky_multi.hpp:
// agent-scope stores behind s_waitcnt, relaxed polls, one acquire fence; ~1.2 us per hop):
//
//   panel p (tile column Jp = p / 2, half hf = p & 1):
//     diagonal tile (Jp, Jp)   factors the 32 x 32 block (one wavefront, DiagFactorInverse), stores L11 / T = L11^-1,
//                              raises tdone[p]; for hf = 0 it also solves X for its rows 32..63
//     column tiles (I, Jp)     wait for tdone[p], X = Rows T' on the matrix cores, store X as L, raise xdone[p][I];
//                              for hf = 0 their columns 32..63 then take -X_I X_Jp[32..63]'
//     trailing tiles (I, J>Jp) wait for xdone[p][I] and xdone[p][J], load the two 64 x 32 strips of X, tile -= X_I X_J'
//
// A tile retires after panel 2J + 1.  The matrix is padded with identity to whole panels (as in ba_cholesky_multi.hpp);
// the right-hand side rides along as row m.  F gets the layout BackSubstituteBlocks reads (L in the lower triangle, T in
// the strict upper triangle of the diagonal blocks, y in row n, inverse pivots in row n + 1); k_chol_finish does the rest.
// All tiles must be resident at once (two workgroups per CU): the host checks the tile count and falls back to the
// multi-launch path otherwise; every wait has the budget of WaitReady and gives up instead of hanging.
#pragma once
#include "ba_cholesky_large.hpp"
#include "ba_cholesky_multi.hpp"

namespace rsba {

#define RSBA_TL 65   // leading dimension of the resident tile (64 x 64, odd stride)

struct TileCholFlags {
  int* tdone;    // [np]            == tag: L11 / T of panel p are in F
  int* xdone;    // [np][nrt]       == tag: X of panel p for tile row I is in F
  int* error;    // != 0: somebody gave up
  int nrt;
  // The chain from one tile column's last factorisation to the next one's first: the sub-diagonal tile (J + 1, J) hands its
  // rows' columns of the column's SECOND panel over as soon as they are final (after the first panel's update) — unsolved,
  // ah[J + 1] (64 x 32), flag adone[J + 1] — and the next diagonal tile forms X = Ahat T' itself the moment T is published,
  // instead of waiting for the sub-diagonal tile to load T, solve, store X, have the stores acknowledged and raise its flag
  // (the lesson of ba_cholesky_diag.hpp: a block is handed over unsolved).  nullptr: everybody waits for xdone.
  int* adone = nullptr;      // [nrt] == tag
  double* ah = nullptr;      // [nrt][64 * 32]
};

__host__ __device__ inline size_t TileCholLdsDoubles() { return (size_t)64 * RSBA_TL + 64 * RSBA_PLD + 3 * RSBA_PB * RSBA_PLD + 64; }

// The next diagonal tile's last update (TileCholFlags::adone): the sub-diagonal tile's rows as handed over -> XJ, T of the
// panel -> T, X = Ahat T' -> XI, with the sub-diagonal tile's own sequence of matrix-core operations on the same operands:
// the same bits.  Out of line: its own register allocation (inlined, the kernel went from 254 registers to 256 with 32 spilled).
// Whole workgroup; false: a flag did not come.
static __device__ __noinline__ bool DiagTileFormsX(const int* adone_flag, const int* tdone_flag, const int* error, int tag, long long budget,
                                                   const double* __restrict__ ah, const double* __restrict__ F, int n, int kb,
                                                   double* XJ, double* T, double* XI) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mi = lane & 15, kk = lane >> 4;
  if (!WaitFlagWG(adone_flag, tag, error, budget)) return false;
  {
    double av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) av[u] = ah[tid + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int e = tid + u * 256; XJ[(e >> 5) * RSBA_PLD + (e & 31)] = av[u]; }
  }
  if (!WaitFlagWG(tdone_flag, tag, error, budget)) return false;
  {
    double tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * 256, r = e >> 5, c = e & 31;   // T[r][c], r >= c
      tv[u] = (kb + r < n && kb + c < n) ? (r > c ? F[(size_t)(kb + c) * n + kb + r] : (r == c ? F[(size_t)(n + 1) * n + kb + c] : 0.0))
                                          : (r == c ? 1.0 : 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; T[(e >> 5) * RSBA_PLD + (e & 31)] = tv[u]; }
  }
  __syncthreads();
  d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const int row = 16 * wave + mi;
#pragma unroll
  for (int qs = 0; qs < RSBA_PB; qs += 4) {
    const double a = XJ[row * RSBA_PLD + qs + kk];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[mi * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(16 + mi) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) { const int r = 16 * wave + kk + 4 * tt; XI[r * RSBA_PLD + mi] = acc0[tt]; XI[r * RSBA_PLD + 16 + mi] = acc1[tt]; }
  return true;
}

__global__ void __launch_bounds__(256, 2)
k_chol_tiles_persistent(int n, const double* __restrict__ W /* (n + 1) x n: scaled, damped system + rhs row (k_sys_build) */,
                        double* __restrict__ F /* (n + 2) x n */, int* __restrict__ ok_flag, TileCholFlags f, int tag,
                        double* __restrict__ res) {
  extern __shared__ double lds[];
  double* Tl = lds;                              // the tile, 64 x 65
  double* XI = Tl + 64 * RSBA_TL;                // 64 x 33: X of this tile's rows for the current panel
  double* T = XI + 64 * RSBA_PLD;                // 32 x 33
  double* Pan = T + RSBA_PB * RSBA_PLD;          // 32 x 33 (diagonal tiles)   |  together: XJ, 64 x 33 (the others)
  double* Lt = Pan + RSBA_PB * RSBA_PLD;         // 32 x 33 (diagonal tiles)   |
  double* XJ = Pan;
  double* invd = Lt + RSBA_PB * RSBA_PLD;        // 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = lane & 15, kk = lane >> 4;
  const int m = (n + RSBA_PB - 1) / RSBA_PB * RSBA_PB, np = m / RSBA_PB;
  const long long budget = RSBA_STALL_TICKS;
  __shared__ int s_good;
  // tile (I, J), I >= J, from the linear index
  const int t = blockIdx.x;
  int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const int r0 = 64 * I, c0 = 64 * J;
  if (c0 >= m) return;   // (the tile row of the rhs row reaches one column past the matrix when m is a multiple of 64)
  // entry (gi, gj) of the padded system; row m is the right-hand side
  auto sysv = [&](int gi, int gj) {
    if (gi > m || gj >= m) return 0.0;
    if (gi == m) return gj < n ? W[(size_t)n * n + gj] : 0.0;
    if (gi >= n || gj >= n) return gi == gj ? 1.0 : 0.0;
    return W[(size_t)gi * n + gj];
  };
  for (int e = tid; e < 64 * 64; e += 256) { const int r = e >> 6, c = e & 63; Tl[r * RSBA_TL + c] = sysv(r0 + r, c0 + c); }
  if (t == 0 && tid == 0) { __hip_atomic_store(ok_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); res[RES_STALL] = 0.0; }   // (a stall, half a second later, sets it)
  __syncthreads();
  // store a value of L / y into F (real entries only; the rhs row m lands in row n)
  auto storeF = [&](int gi, int gj, double v) {
    if (gj >= n) return;
    if (gi < n) StoreShared(&F[(size_t)gi * n + gj], v);
    else if (gi == m) StoreShared(&F[(size_t)n * n + gj], v);
  };
  auto loadF = [&](int gi, int gj) -> double {   // L entry (gi, gj) of the padded factor, gi > gj's panel
    if (gj >= n) return 0.0;                     // padded columns: identity, nothing below the diagonal
    if (gi < n) return F[(size_t)gi * n + gj];
    if (gi == m) return F[(size_t)n * n + gj];
    return 0.0;
  };
  bool stalled = false;
  const int plast = min(2 * J + 1, np - 1);
  for (int p = 0; p <= plast && !stalled; ++p) {
    const int Jp = p >> 1, hf = p & 1, kb = p * RSBA_PB, lc = 32 * hf;   // lc: the panel's first column inside a tile of column Jp
    if (J == Jp) {
      if (I == J) {
        // ---- diagonal tile: factor the block at [lc, lc + 32)^2
        for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) { const int r = e >> 5, c = e & 31; Pan[r * RSBA_PLD + c] = Tl[(lc + r) * RSBA_TL + lc + c]; }
        __syncthreads();
        if (wave == 0) {
          const bool good = DiagFactorInverseCall((lds_double*)Pan, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane);
          if (lane == 0) s_good = good ? 1 : 0;
        }
        __syncthreads();
        for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) {
          const int r = e >> 5, c = e & 31;
          if (kb + r < n && kb + c < n) StoreShared(&F[(size_t)(kb + r) * n + kb + c], c > r ? T[c * RSBA_PLD + r] : Pan[r * RSBA_PLD + c]);
        }
        if (tid < RSBA_PB && kb + tid < n) StoreShared(&F[(size_t)(n + 1) * n + kb + tid], invd[tid]);
        if (tid == 0 && !s_good) __hip_atomic_store(ok_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // second half: T is what the next tile column waits for — out at once; nothing of this tile lies below the block, so
        // there is no X to form and nobody waits for this tile's xdone.  First half: the column tiles have the whole second
        // factorisation's time to pick T up, so it is published together with X below (one wait for the stores' acknowledgements
        // instead of two on this tile's own chain)
        if (hf == 1 || 2 * J + 1 >= np) { PublishFlagWG(f.tdone + p, tag); if (hf == 1) continue; }
      } else {
        if (!WaitFlagWG(f.tdone + p, tag, f.error, budget)) { stalled = true; break; }
        {
          double tv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = tid + u * 256, r = e >> 5, c = e & 31;   // T[r][c], r >= c
            tv[u] = (kb + r < n && kb + c < n) ? (r > c ? F[(size_t)(kb + c) * n + kb + r] : (r == c ? F[(size_t)(n + 1) * n + kb + c] : 0.0))
                                                : (r == c ? 1.0 : 0.0);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; T[(e >> 5) * RSBA_PLD + (e & 31)] = tv[u]; }
        }
        __syncthreads();
      }
      // ---- X = Rows T' for this tile's rows below the diagonal block: wave w takes rows 16 w .. 16 w + 15
      {
        const int rlo = (I == J) ? lc + RSBA_PB : 0;     // first tile row that is below the diagonal block
        d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        const int row = 16 * wave + mi;
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4) {
          const double a = Tl[row * RSBA_TL + lc + qs + kk];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[mi * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(16 + mi) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int r = 16 * wave + kk + 4 * tt;
          const bool below = r >= rlo;
          const double x0 = below ? acc0[tt] : 0.0, x1 = below ? acc1[tt] : 0.0;
          XI[r * RSBA_PLD + mi] = x0; XI[r * RSBA_PLD + 16 + mi] = x1;
          if (below) {
            Tl[r * RSBA_TL + lc + mi] = x0; Tl[r * RSBA_TL + lc + 16 + mi] = x1;
            storeF(r0 + r, kb + mi, x0); storeF(r0 + r, kb + 16 + mi, x1);
          }
        }
      }
      if (I == J && hf == 0 && 2 * J + 1 < np) {
        // (T and X of the first half together)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) { __hip_atomic_store(f.tdone + p, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(f.xdone + (size_t)p * f.nrt + I, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      } else {
        PublishFlagWG(f.xdone + (size_t)p * f.nrt + I, tag);
      }
      // ---- first half done: this tile's columns 32..63 take -X_I X_Jp[rows 32..63]'
      if (hf == 0 && 2 * J + 1 < np) {
        if (I != J) {
          if (!WaitFlagWG(f.xdone + (size_t)p * f.nrt + J, tag, f.error, budget)) { stalled = true; break; }
          double xv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int e = tid + u * 256, r = 32 + (e >> 5), c = e & 31; xv[u] = loadF(c0 + r, kb + c); }
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; XJ[(32 + (e >> 5)) * RSBA_PLD + (e & 31)] = xv[u]; }
          __syncthreads();
        }
        const double* XB = (I == J) ? XI : XJ;
        d4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        const int row = 16 * wave + mi;
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4) {
          const double a = XI[row * RSBA_PLD + qs + kk];
          a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, XB[(32 + mi) * RSBA_PLD + qs + kk], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, XB[(48 + mi) * RSBA_PLD + qs + kk], a1, 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int r = 16 * wave + kk + 4 * tt;
          Tl[r * RSBA_TL + 32 + mi] -= a0[tt];
          Tl[r * RSBA_TL + 48 + mi] -= a1[tt];
        }
        __syncthreads();
        if (f.adone != nullptr && I == J + 1) {
          // the sub-diagonal tile: its rows' columns of the second panel are final — hand them over unsolved
          double* ah = f.ah + (size_t)I * 64 * 32;
          for (int e = tid; e < 64 * 32; e += 256) StoreShared(&ah[e], Tl[(e >> 5) * RSBA_TL + 32 + (e & 31)]);
          PublishFlagWG(f.adone + I, tag);
        }
      }
    } else {
      // ---- trailing tile: tile -= X_I X_J'
      const bool own_x = f.adone != nullptr && I == J && Jp == J - 1 && hf == 1;   // the diagonal tile's last update: it forms X itself
      if (own_x) {
        if (!DiagTileFormsX(f.adone + I, f.tdone + p, f.error, tag, budget, f.ah + (size_t)I * 64 * 32, F, n, kb, XJ, T, XI)) { stalled = true; break; }
      } else {
      if (!WaitFlagWG(f.xdone + (size_t)p * f.nrt + I, tag, f.error, budget)) { stalled = true; break; }
      if (I != J && !WaitFlagWG(f.xdone + (size_t)p * f.nrt + J, tag, f.error, budget)) { stalled = true; break; }
      {
        double xi[8], xj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + u * 256, r = e >> 5, c = e & 31;
          xi[u] = loadF(r0 + r, kb + c);
          xj[u] = I != J ? loadF(c0 + r, kb + c) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + u * 256, r = e >> 5, c = e & 31;
          XI[r * RSBA_PLD + c] = xi[u];
          if (I != J) XJ[r * RSBA_PLD + c] = xj[u];
        }
      }
      }
      __syncthreads();
      const double* XB = (I == J) ? XI : XJ;
      d4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      const int row = 16 * wave + mi;
#pragma unroll
      for (int qs = 0; qs < RSBA_PB; qs += 4) {
        const double a = XI[row * RSBA_PLD + qs + kk];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, XB[(16 * jb + mi) * RSBA_PLD + qs + kk], acc[jb], 0, 0, 0);
      }
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Tl[(16 * wave + kk + 4 * tt) * RSBA_TL + 16 * jb + mi] -= acc[jb][tt];
      __syncthreads();
    }
  }
  if (stalled && tid == 0) {
    __hip_atomic_store(f.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ok_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    res[RES_STALL] = 1.0;   // the host repeats the step with the multi-launch factorisation
  }
}

// Block back-substitution L' x = y for the large system on SEVERAL workgroups (k_chol_finish does it on one: 48 dependent
// strips of L from memory, 0.3 ms at 256 cameras).  Workgroup g owns three consecutive 32-column blocks of y.  From the
// last block up: the owner of block b has applied every later block to its columns, so x_b = T_b' y_b; it publishes x_b
// (32 doubles + a flag); every workgroup with columns below b subtracts L[b, own columns]' x_b from its part of y — the strip
// it needs is loaded one block ahead (its addresses do not depend on x).  The chain is T'y + a hop only where the owner
// changes; workgroup 0 ends with block 0, has seen every flag, and runs the camera-step epilogue.
#define RSBA_BSM_BPG 3   // blocks per workgroup (96 columns)
__global__ void __launch_bounds__(256)
k_backsub_multi(int C, const double* __restrict__ red, RedLayout L, const double* __restrict__ F, double* __restrict__ xsol,
                const double* __restrict__ scale_c, const double* __restrict__ cam_x, double* __restrict__ cam_c,
                const double* __restrict__ intr, double* __restrict__ camc_c, double* __restrict__ dcam,
                const double* __restrict__ gmax_p, double* __restrict__ res, const int* __restrict__ ok_flag,
                const double* __restrict__ cam_free, int* __restrict__ xdone, int* __restrict__ error, int tag) {
  __shared__ double yown[32 * RSBA_BSM_BPG];       // this workgroup's columns of y
  __shared__ double Tb[RSBA_BSM_BPG][RSBA_PB * RSBA_PLD];   // T of this workgroup's own blocks, loaded before the chain arrives
  __shared__ double xb[RSBA_PB];
  __shared__ double part[2][128];
  __shared__ double epi[4 * 256];
  const int n = L.nc, tid = threadIdx.x, w = blockIdx.x;
  const int m = (n + RSBA_PB - 1) / RSBA_PB * RSBA_PB, nblk = m / RSBA_PB;
  const int cw = 32 * RSBA_BSM_BPG, col0 = w * cw;          // own columns [col0, col0 + cw)
  const int q = tid & 127, hp = tid >> 7;                   // column (q < cw) and half of the block's 32 rows
  const long long budget = RSBA_STALL_TICKS;
  for (int i = tid; i < cw; i += 256) yown[i] = col0 + i < n ? F[(size_t)n * n + col0 + i] : 0.0;
  // T_b (r >= c): stored at F[kb + c][kb + r] for r > c, the diagonal in row n + 1; identity on the padding
  for (int e = tid; e < RSBA_BSM_BPG * RSBA_PB * RSBA_PB; e += 256) {
    const int j = e >> 10, r = (e >> 5) & 31, c = e & 31, kb = col0 + 32 * j;
    Tb[j][r * RSBA_PLD + c] = (kb + r < n && kb + c < n) ? (r > c ? F[(size_t)(kb + c) * n + kb + r] : (r == c ? F[(size_t)(n + 1) * n + kb + c] : 0.0))
                                                         : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  // strip part of block b for this thread: L[32 b + 16 hp + c][col0 + q], c < 16 (zero outside the real lower triangle)
  auto load_strip = [&](int b, double (&d)[16]) {
    const int gq = col0 + q;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int gi = 32 * b + 16 * hp + c;
      d[c] = (b >= 0 && q < cw && gq < 32 * b && gq < n && gi < n) ? F[(size_t)gi * n + gq] : 0.0;
    }
  };
  const int btop = nblk - 1;
  // the strips of the next three blocks are in flight: an iteration is ~1 us, a load from memory 2-3 us
  double lv[16], l1[16], l2[16], l3[16];
  load_strip(btop, lv); load_strip(btop - 1, l1); load_strip(btop - 2, l2);
  bool stalled = false;
  int pending = -1;   // a block of this workgroup whose x is stored but not flagged yet
  // wave 0 stored x with agent-scope stores; the flag follows once they are performed — not right after the stores (that
  // wait is a memory round trip on the owner's own chain) but one block later, behind the update
  auto publish_pending = [&]() {
    if (pending >= 0 && tid < 64) {
      __builtin_amdgcn_s_waitcnt(0);
      if (tid == 0) __hip_atomic_store(xdone + pending, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pending = -1;
  };
  __shared__ double xg[32 * RSBA_BSM_BPG];   // the x of another workgroup's three blocks, fetched at once
  for (int b = btop; b >= w * RSBA_BSM_BPG && !stalled; --b) {   // blocks below this workgroup's columns do not touch them
    const int owner = b / RSBA_BSM_BPG, kb = 32 * b;
    // (the flag first: its wait for the x stores' acknowledgement is a wait for EVERYTHING this wavefront has in flight, and
    //  behind the strip loads of this iteration that was a round trip to memory per block of the owner's own chain)
    publish_pending();
    load_strip(b - 3, l3);
    if (w == owner) {
      if (tid < RSBA_PB) {
        const double* Tj = Tb[b - owner * RSBA_BSM_BPG];
        double sacc = 0.0;
#pragma unroll 8
        for (int i = 0; i < RSBA_PB; ++i) sacc += Tj[i * RSBA_PLD + tid] * yown[kb - col0 + i];
        xb[tid] = sacc;
        if (kb + tid < n) StoreShared(&xsol[kb + tid], sacc);
      }
      pending = b;
      __syncthreads();
    } else {
      // another workgroup's blocks: one wait and one fetch for its whole range (its flags come in descending block order:
      // the lowest one says all three are there) — a wait and a round trip per block made the low workgroups, which see
      // every block, the slowest part of the chain
      const int bl = owner * RSBA_BSM_BPG;                         // the owner's lowest block
      if (b == min(btop, bl + RSBA_BSM_BPG - 1)) {
        if (!WaitFlagWG(xdone + bl, tag, error, budget)) { stalled = true; break; }
        if (tid < 32 * RSBA_BSM_BPG) xg[tid] = 32 * bl + tid < n ? xsol[32 * bl + tid] : 0.0;
        __syncthreads();
      }
      if (tid < RSBA_PB) xb[tid] = xg[kb - 32 * bl + tid];
      __syncthreads();
    }
    // y[own columns < kb] -= L[block b rows, column]' x_b, the two halves of the rows added in a fixed order
    {
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) sacc += lv[c] * xb[16 * hp + c];
      if (q < 128) part[hp][q] = sacc;
    }
    __syncthreads();
    if (tid < cw && col0 + tid < kb) yown[tid] -= part[0][tid] + part[1][tid];
#pragma unroll
    for (int c = 0; c < 16; ++c) { lv[c] = l1[c]; l1[c] = l2[c]; l2[c] = l3[c]; }
    __syncthreads();
  }
  publish_pending();
  if (stalled) { if (tid == 0) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); res[RES_STALL] = 1.0; } return; }
  if (w != 0) return;
  // workgroup 0: every block's x is in xsol and visible (it waited for, or produced, each of them)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  int ok = 1;
  if (tid == 0) ok = __hip_atomic_load(ok_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  CameraStepEpilogue(C, red, L, scale_c, xsol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, ok, epi, cam_free);
}

// ------------------------------------------------------------------------------------------------
// Block back-substitution with the chain in ONE workgroup (round 3; k_backsub_multi above is its predecessor and fallback).
// k_backsub_multi passes the chain from owner to owner: per owner's three blocks a hop (store x, acknowledgement, flag, poll,
// fetch), three strip updates and three solves, ~11 us — 184 us at 256 cameras.  Here workgroup 0 solves EVERY block, in
// order, with all of x in its LDS, and never waits for a store: per block it subtracts the nearest strips' blocks itself
// (L(i, b) x_i for b < i <= top of its column range + RSBA_BSC_LAG: at most LAG + 2 products of 32 x 32, their entries of L
// prefetched one block ahead, addresses known) and x_b = T_b' y_b; one wavefront sends x_b out while the others go on.
// Everything FARTHER above is the helpers' business: helper h owns 96 columns of y and applies the strip of every block i more
// than RSBA_BSC_LAG above its range as soon as x_i is published (strips prefetched three ahead, as in k_backsub_multi), then
// hands its slice of y over (ys, hdone) — LAG + 1 blocks before the chain gets there.  Same sums per entry in a fixed order:
// bitwise reproducible.
// What bounds it: the chain's own loads.  One CU pulls ~70 GB/s through its L1, whether the lines sit in its XCD's L2 or not
// (workgroups that touched the lines ahead of the chain changed nothing), so every near block costs ~0.12 us; with LAG = 3
// the chain read 40 KB per block and took 117 us, LAG = 2: 110, LAG = 1: 106, LAG = 0 (the helper's hand-over on the chain): 111.
// The loads must cover whole 128-byte lines per row (32 consecutive columns per half wavefront): with a column's eight row
// groups in neighbouring lanes — which saves two of the four barriers per block — every line is asked for twice, 164 us.
// ------------------------------------------------------------------------------------------------
#define RSBA_BSC_LAG 1
__global__ void __launch_bounds__(256)
k_backsub_chain(int C, const double* __restrict__ red, RedLayout L, const double* __restrict__ F, double* __restrict__ xsol,
                const double* __restrict__ scale_c, const double* __restrict__ cam_x, double* __restrict__ cam_c,
                const double* __restrict__ intr, double* __restrict__ camc_c, double* __restrict__ dcam,
                const double* __restrict__ gmax_p, double* __restrict__ res, const int* __restrict__ ok_flag,
                const double* __restrict__ cam_free, int* __restrict__ xdone, int* __restrict__ hdone, double* __restrict__ ys,
                int* __restrict__ error, int tag) {
  const int n = L.nc, tid = threadIdx.x, w = blockIdx.x;
  const int m = (n + RSBA_PB - 1) / RSBA_PB * RSBA_PB, nblk = m / RSBA_PB, btop = nblk - 1;
  const long long budget = RSBA_STALL_TICKS;
  __shared__ double part[8][RSBA_PB];
  __shared__ double xb[RSBA_PB];
  __shared__ double epi[4 * 256];
  if (w > 0) {
    // ---- helper: 96 columns of y, the strips of the blocks far above them
    __shared__ double yown[96];
    const int h = w - 1, c0 = 96 * h, b1 = min(btop, 3 * h + 2);
    const int q = tid & 127, hp = tid >> 7;
    for (int i = tid; i < 96; i += 256) yown[i] = c0 + i < n ? F[(size_t)n * n + c0 + i] : 0.0;
    auto load_strip = [&](int b, double (&d)[16]) {
      const int gq = c0 + q;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int gi = 32 * b + 16 * hp + c;
        d[c] = (b > b1 + RSBA_BSC_LAG && q < 96 && gq < n && gi < n) ? F[(size_t)gi * n + gq] : 0.0;
      }
    };
    double lv[16], l1[16], l2[16], l3[16];
    load_strip(btop, lv); load_strip(btop - 1, l1); load_strip(btop - 2, l2);
    __syncthreads();
    bool stalled = false;
    for (int b = btop; b > b1 + RSBA_BSC_LAG; --b) {
      load_strip(b - 3, l3);
      if (!WaitFlagWG(xdone + b, tag, error, budget)) { stalled = true; break; }
      if (tid < RSBA_PB) xb[tid] = 32 * b + tid < n ? __hip_atomic_load(&xsol[32 * b + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      __syncthreads();
      {
        double sacc = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) sacc += lv[c] * xb[16 * hp + c];
        epi[hp * 128 + q] = sacc;
      }
      __syncthreads();
      if (tid < 96) yown[tid] -= epi[tid] + epi[128 + tid];
#pragma unroll
      for (int c = 0; c < 16; ++c) { lv[c] = l1[c]; l1[c] = l2[c]; l2[c] = l3[c]; }
      __syncthreads();
    }
    if (stalled) { if (tid == 0) __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (tid < 96) StoreShared(&ys[96 * h + tid], yown[tid]);
    PublishFlagWG(hdone + h, tag);
    return;
  }
  // ---- the chain
  extern __shared__ double xl[];                       // m: every x, as it is solved
  __shared__ double yl[96];                            // the current helper range's slice of y
  __shared__ double ybl[RSBA_PB];
  const int c = tid & 31, p8 = tid >> 5;               // column of the block, eighth of its 32 rows (rows 4 p8 .. 4 p8 + 3)
  // what block b needs from memory, fetched one block ahead: its near blocks of L (at most LAG + 2) and T_b.  Checked form: any
  // block, padding rows and columns read as zero / identity.  Plain form (the loop's, whenever no padded row or column is in
  // reach): ONE per-thread byte offset and a scalar row offset per load — the checked form's index arithmetic, 28 loads of it,
  // was 1.65 us of the chain's 2.7 us per block.
  constexpr int kNear = RSBA_BSC_LAG + 2;
  auto fetch_checked = [&](int b, double (&ln)[kNear][4], double (&tn)[4]) {
    const int h = b / 3, b1 = min(btop, 3 * h + 2), ihi = min(btop, b1 + RSBA_BSC_LAG), kb = 32 * b;
#pragma unroll
    for (int u = 0; u < kNear; ++u) {
      const int i = b + 1 + u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = 32 * i + 4 * p8 + r, gj = kb + c;
        ln[u][r] = (b >= 0 && i <= ihi && gi < n && gj < n) ? F[(size_t)gi * n + gj] : 0.0;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * p8 + r;   // T[i][c], i >= c: at F[kb + c][kb + i] for i > c, the diagonal in row n + 1; identity on the padding
      tn[r] = b < 0 ? 0.0 : ((kb + i < n && kb + c < n) ? (i > c ? F[(size_t)(kb + c) * n + kb + i] : (i == c ? F[(size_t)(n + 1) * n + kb + c] : 0.0))
                                                         : (i == c ? 1.0 : 0.0));
    }
  };
  const char* Fb = reinterpret_cast<const char*>(F);
  const unsigned rowb = 8u * (unsigned)n;                 // bytes per row of F; (n + 2) n doubles stay far below 4 GB
  auto fetch_plain = [&](int b, double (&ln)[kNear][4], double (&tn)[4], double& dgn) {
    // needs 0 <= b and every row 32 i + 31 of its near blocks and 32 b + 31 below n
    const int h = b / 3, b1 = min(btop, 3 * h + 2), nu = min(btop, b1 + RSBA_BSC_LAG) - b;
    const unsigned kb = 32u * (unsigned)b;
    const unsigned o_l = (kb + 32u + 4u * p8) * rowb + 8u * (kb + c);       // L(b + 1, b)[4 p8][c]
#pragma unroll
    for (int u = 0; u < kNear; ++u) {
      if (u < nu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ln[u][r] = *reinterpret_cast<const double*>(Fb + (size_t)(o_l + (unsigned)(32 * u + r) * rowb));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) ln[u][r] = 0.0;
      }
    }
    const unsigned o_t = (kb + c) * rowb + 8u * (kb + 4u * p8);             // row kb + c of F, columns kb + 4 p8 ..
    // raw: row kb + c of F holds T[i][c] for i > c, the diagonal comes from row n + 1, the rest of the row belongs to other
    // blocks — TEntry() picks when the values are used (picking here made the compiler wait for the loads here)
    dgn = *reinterpret_cast<const double*>(Fb + (size_t)((unsigned)(n + 1) * rowb + 8u * (kb + c)));
#pragma unroll
    for (int r = 0; r < 4; ++r) tn[r] = *reinterpret_cast<const double*>(Fb + (size_t)(o_t + 8u * r));
  };
  auto fetch = [&](int b, double (&ln)[kNear][4], double (&tn)[4], double& dgn) {
    const int b1 = min(btop, 3 * (b / 3) + 2);
    if (b >= 0 && 32 * (min(btop, b1 + RSBA_BSC_LAG) + 1) <= n) { fetch_plain(b, ln, tn, dgn); return true; }
    fetch_checked(b, ln, tn);
    return false;
  };
  double lcur[kNear][4], tcur[4], lnext[kNear][4], tnext[4], dgcur = 0.0, dgnext = 0.0;
  bool raw_cur = fetch(btop, lcur, tcur, dgcur), raw_next = false;
  bool stalled = false;
  for (int b = btop; b >= 0; --b) {
    const int h = b / 3, b1 = min(btop, 3 * h + 2), ihi = min(btop, b1 + RSBA_BSC_LAG);
    if (b < btop && tid >= 192) {
      __builtin_amdgcn_s_waitcnt(0);
      if (tid == 192) __hip_atomic_store(xdone + b + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    raw_next = fetch(b - 1, lnext, tnext, dgnext);
    if (b == b1) {
      // a new range of columns: its helper has applied every strip above b1 + LAG (or there are none)
      if (!WaitFlagWG(hdone + h, tag, error, budget)) { stalled = true; break; }
      if (tid < 96) yl[tid] = __hip_atomic_load(&ys[96 * h + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
    // the near blocks: sum_i L(i, b)' x_i over b < i <= ihi
    {
      double sacc = 0.0;
#pragma unroll
      for (int u = 0; u < kNear; ++u) {
        const int i = b + 1 + u;
        if (i <= ihi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc += lcur[u][r] * xl[32 * i + 4 * p8 + r];
        }
      }
      part[p8][c] = sacc;
    }
    __syncthreads();
    if (tid < RSBA_PB) {
      double sacc = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) sacc += part[q][tid];
      ybl[tid] = yl[32 * (b - 3 * h) + tid] - sacc;
    }
    __syncthreads();
    // x_b = T_b' y_b
    {
      double sacc = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * p8 + r;
        const double tv = raw_cur ? (i > c ? tcur[r] : (i == c ? dgcur : 0.0)) : tcur[r];
        sacc += tv * ybl[i];
      }
      part[p8][c] = sacc;
    }
    __syncthreads();
    if (tid < RSBA_PB) {
      double sacc = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) sacc += part[q][tid];
      xl[32 * b + tid] = sacc;
    }
    __syncthreads();
    // x_b goes out now; its flag follows at the top of the next iteration (below): by then the store is acknowledged, and the
    // wait for that is a wait for loads this wavefront needs there anyway.  (Flag right behind the store: every block the whole
    // workgroup stood at the next barrier until the last wavefront's store AND its prefetches for the next block had come back.)
    if (tid >= 192 && tid < 224) {
      const int t = tid - 192;
      if (32 * b + t < n) StoreShared(&xsol[32 * b + t], xl[32 * b + t]);
    }
#pragma unroll
    for (int u = 0; u < kNear; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) lcur[u][r] = lnext[u][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) tcur[r] = tnext[r];
    dgcur = dgnext;
    raw_cur = raw_next;
  }
  if (stalled) { if (tid == 0) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); res[RES_STALL] = 1.0; } return; }
  // every x is in xsol (this workgroup stored them) — visible to its own plain loads behind the acknowledgements and one acquire
  __builtin_amdgcn_s_waitcnt(0);
  if (tid == 192) __hip_atomic_store(xdone + 0, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  int ok = 1;
  if (tid == 0) ok = __hip_atomic_load(ok_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  CameraStepEpilogue(C, red, L, scale_c, xsol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, ok, epi, cam_free);
}

}  // namespace rsba
