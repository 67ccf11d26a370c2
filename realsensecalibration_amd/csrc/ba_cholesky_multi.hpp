// The reduced camera system on SEVERAL workgroups (n <= 384, n a multiple of 32: 32, 48 or 64 cameras).
//
// One workgroup factors the 384 x 384 system in ~310 us, and the phase profile says why: the sequential 32 x 32
// diagonal factorisation (6-8 us per panel) runs next to a row update that one CU cannot feed — its loads are latency x
// concurrency bound (~30 GB/s), and it re-reads the factor every panel (left-looking).  Here the 32-row blocks of the
// matrix are dealt round-robin to G workgroups (G CUs, each with its own queue of loads in flight):
//
//   block b (rows 32b .. 32b+31; block n/32 is the right-hand side row) belongs to workgroup b mod G.
//   Panel p (columns 32p ..):  every workgroup loads the panel's columns of ITS blocks b >= p (scaled and damped on the
//   fly, PanelSource), the strip L[block p, 0:32p] (B operand of the update; written by block p's owner, flag
//   strip_ready[p]), and updates its blocks on the matrix cores.  The owner of block p factors the diagonal block
//   (wave 0, DiagFactorInverse) and publishes L11 / T = L11^-1 (flag tdone[p]); everybody then solves X = Rows T' for its
//   blocks and stores them.  The owner of block p+1 raises strip_ready[p+1] as soon as ITS block's X is stored.
//
// Hand-offs are flags in global memory (agent-scope relaxed stores behind s_waitcnt, relaxed polls, one acquire fence;
// ~1.2 us per hop between XCDs, tools/probes/multiwg_probe.hip).  Workgroup 0 owns the right-hand side row, runs the
// back-substitution and the camera-step epilogue.  Summation orders are fixed: bitwise reproducible, and identical on
// every rank of a multi-GPU run.  All waits carry the budget of WaitReady: a stall gives up (RES_STALL), never hangs.
#pragma once
#include "ba_cholesky.hpp"
#include "ba_point_kernels.hpp"

namespace rsba {

#define RSBA_MC_MAXG 6

struct MultiCholFlags {
  int* tdone;         // [16]  == tag when panel p's L11 / T are in global memory
  int* strip_ready;   // [16]  == tag when the rows of block p hold L for all columns < 32 p
  int* wg_done;       // [8]   == tag when workgroup w has stored its last entries
  int* error;         // != 0: somebody gave up waiting
};

// One lane polls (relaxed, sleeping), then the whole workgroup acquires.  false: budget exhausted or error raised.
__device__ __forceinline__ bool WaitFlagWG(const int* flag, int tag, const int* error, long long budget) {
  __shared__ int s_ok2;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      __builtin_amdgcn_s_sleep(2);
      if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > budget) { ok = 0; break; }
    }
    s_ok2 = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return s_ok2 != 0;
}

// After the workgroup's agent-scope stores: all of them performed, then the flag.
__device__ __forceinline__ void PublishFlagWG(int* flag, int tag) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void StoreShared(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(512)
k_reduced_system_solve_multi(int C, double* __restrict__ red, RedLayout L, double* __restrict__ A, double* __restrict__ scale_c,
                             const double* __restrict__ cam_x, double* __restrict__ cam_c, const double* __restrict__ intr,
                             double* __restrict__ camc_c, double* __restrict__ dcam, const double* __restrict__ gmax_p,
                             double* __restrict__ res, IterParams ip, int* __restrict__ chol_ok, StageGate gate, MultiCholFlags f, int tag,
                             long long* __restrict__ mtrace /* diagnostic: [G][16][8] wall-clock stamps, or nullptr */) {
  extern __shared__ double lds[];
  const int n = L.nc, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
  const int G = gridDim.x, w = blockIdx.x;
  const int np = n / RSBA_PB;              // column panels; blocks 0 .. np (block np: the rhs row alone)
  const long long budget = gate.budget > 0 ? gate.budget : RSBA_STALL_TICKS;
  __shared__ int s_ok;
  // LDS: strip (32 p rows of 33) | this workgroup's blocks of the panel (32 x 33 each) | T | Lt | invd | scale
  const int max_rows = n + RSBA_PB;       // strip + panel blocks never exceed (p + ceil((np + 1 - p) / G)) * 32 <= n + 32 rows
  double* T = lds + (size_t)max_rows * RSBA_PLD;
  double* Lt = T + RSBA_PB * RSBA_PLD;
  double* invd = Lt + RSBA_PB * RSBA_PLD;
  double* scl = invd + 64;
  if (tid == 0) s_ok = 1;
  bool stalled = false;

  // pipelined first iteration: the Jacobi scale needs the whole damping diagonal
  if (gate.ready != nullptr && ip.first) {
    for (int g = 0; g * gate.cols < n; ++g)
      if (!WaitReady(gate.ready + 1 + g, gate.tag, w == 0 ? gate.waited : nullptr, gate.budget)) { stalled = true; break; }
  }
  if (!stalled) {
    for (int i = tid; i < n; i += nt) {
      const double sc = ip.first ? (ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(red[L.diagU() + i])) : 1.0) : scale_c[i];
      scl[i] = sc;
      if (ip.first && w == 0) scale_c[i] = sc;
    }
    if (w == 0 && tid == 0) __hip_atomic_store(chol_ok, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const double* S = red + L.S();
  const double inv_radius = 1.0 / ip.radius;

#define RSBA_MC_STAMP(k) do { if (mtrace && tid == 0) mtrace[((size_t)w * 16 + p) * 8 + (k)] = wall_clock64(); } while (0)
  for (int p = 0; p < np && !stalled; ++p) {
    const int kb = p * RSBA_PB;
    RSBA_MC_STAMP(0);
    if (gate.ready != nullptr && kb % gate.cols == 0 && !ip.first) {
      if (!WaitReady(gate.ready + 1 + kb / gate.cols, gate.tag, w == 0 ? gate.waited : nullptr, gate.budget)) { stalled = true; break; }
    }
    // this workgroup's blocks b >= p: b = first, first + G, ...
    const int first = p + ((w - p % G) + G) % G;
    const int nown = first > np ? 0 : (np - first) / G + 1;
    const bool owner = first == p;
    double* Bst = lds;
    double* Pan = lds + (size_t)kb * RSBA_PLD;    // slot j: rows of block first + j G
    // 1. panel columns of the owned blocks, scaled and damped on the fly; the rhs row from gc + corr
    for (int e = tid; e < nown * RSBA_PB * (RSBA_PB / 4); e += nt) {
      const int j = e / (RSBA_PB * (RSBA_PB / 4)), rr = (e >> 3) & 31, c0 = (e & 7) * 4;
      const int b = first + j * G, gi = b * RSBA_PB + rr;
      double v[4] = {0.0, 0.0, 0.0, 0.0};
      if (b < np) {
        const double* srow = S + (size_t)gi * n + kb + c0;
        const double2 a01 = *reinterpret_cast<const double2*>(srow), a23 = *reinterpret_cast<const double2*>(srow + 2);
        v[0] = a01.x; v[1] = a01.y; v[2] = a23.x; v[3] = a23.y;
        const double si = scl[gi];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int gj = kb + c0 + u;
          v[u] *= si * scl[gj];
          if (gi == gj) v[u] += fmin(fmax(si * si * red[L.diagU() + gi], ip.min_lm_diagonal), ip.max_lm_diagonal) * inv_radius;
        }
      } else if (rr == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int gj = kb + c0 + u; v[u] = scl[gj] * (red[L.gc() + gj] + red[L.corr() + gj]); }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) Pan[(j * RSBA_PB + rr) * RSBA_PLD + c0 + u] = v[u];
    }
    RSBA_MC_STAMP(1);
    // 2. strip: rows of block p, columns 0 .. kb, transposed into Bst[q][c]
    if (p > 0) {
      if (!owner && !WaitFlagWG(f.strip_ready + p, tag, f.error, budget)) { stalled = true; break; }
      if (owner) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // its own agent-scope stores of the last panel, not a stale L1 line
      RSBA_MC_STAMP(2);
      for (int e = tid; e < (kb >> 2) * RSBA_PB; e += nt) {
        const int c = e / (kb >> 2), q0 = (e - c * (kb >> 2)) * 4;
        const double* lrow = A + (size_t)(kb + c) * n + q0;
        const double2 a01 = *reinterpret_cast<const double2*>(lrow), a23 = *reinterpret_cast<const double2*>(lrow + 2);
        Bst[(q0 + 0) * RSBA_PLD + c] = a01.x; Bst[(q0 + 1) * RSBA_PLD + c] = a01.y;
        Bst[(q0 + 2) * RSBA_PLD + c] = a23.x; Bst[(q0 + 3) * RSBA_PLD + c] = a23.y;
      }
    }
    __syncthreads();
    RSBA_MC_STAMP(3);
    // 3. update: Pan[block] -= A[block rows, 0:kb] Bst', one wave per 16-row half, the diagonal block's halves first
    if (p > 0) {
      const int i = lane & 15, kk = lane >> 4;
      for (int hb = wave; hb < 2 * nown; hb += nwave) {
        const int j = hb >> 1, b = first + j * G;
        const int prow = j * RSBA_PB + (hb & 1) * 16;          // first Pan row of this half
        const int grow = b * RSBA_PB + (hb & 1) * 16 + i;      // global row of this lane's A operand
        const bool gl = grow <= n && (b < np || (hb & 1) == 0 && i == 0);
        const double* arow = A + (size_t)(gl ? grow : 0) * n + 8 * kk;
        d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        double an[8], an2[8];
        auto fetch = [&](double (&d)[8], int q) {
          const double2* pa = reinterpret_cast<const double2*>(arow + q);
#pragma unroll
          for (int v = 0; v < 4; ++v) { const double2 t = pa[v]; d[2 * v] = gl ? t.x : 0.0; d[2 * v + 1] = gl ? t.y : 0.0; }
        };
        fetch(an, 0);
        if (RSBA_PB < kb) fetch(an2, RSBA_PB);
        for (int q0 = 0; q0 < kb; q0 += RSBA_PB) {
          double ac[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { ac[u] = an[u]; an[u] = an2[u]; }
          if (q0 + 2 * RSBA_PB < kb) fetch(an2, q0 + 2 * RSBA_PB);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const double b0 = Bst[(q0 + 8 * kk + u) * RSBA_PLD + i];
            const double b1 = Bst[(q0 + 8 * kk + u) * RSBA_PLD + 16 + i];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], b1, acc1, 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = prow + kk + 4 * t;
          Pan[r * RSBA_PLD + i] -= acc0[t];
          Pan[r * RSBA_PLD + 16 + i] -= acc1[t];
        }
      }
      __syncthreads();
    }
    RSBA_MC_STAMP(4);
    // 4. the diagonal block: factor + inverse by the owner's wave 0, published; the others fetch T
    if (owner) {
      if (wave == 0 && !DiagFactorInverseCall((lds_double*)Pan, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && lane == 0) s_ok = 0;
      __syncthreads();
      // L11 in the lower triangle, T transposed into the strict upper one, inverse pivots into row n + 1 (the layout
      // BackSubstituteBlocks reads)
      for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) {
        const int r = e >> 5, c = e & 31;
        StoreShared(&A[(size_t)(kb + r) * n + kb + c], c > r ? T[c * RSBA_PLD + r] : Pan[r * RSBA_PLD + c]);
      }
      if (tid < RSBA_PB) StoreShared(&A[(size_t)(n + 1) * n + kb + tid], invd[tid]);
      if (tid == 0 && !s_ok) __hip_atomic_store(chol_ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      PublishFlagWG(f.tdone + p, tag);
    } else if (nown > 0) {
      if (!WaitFlagWG(f.tdone + p, tag, f.error, budget)) { stalled = true; break; }
      for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) {
        const int r = e >> 5, c = e & 31;   // T[r][c], r >= c: stored at A[kb + c][kb + r] for r > c, the diagonal in row n + 1
        T[r * RSBA_PLD + c] = r > c ? A[(size_t)(kb + c) * n + kb + r] : (r == c ? A[(size_t)(n + 1) * n + kb + c] : 0.0);
      }
      __syncthreads();
    }
    RSBA_MC_STAMP(5);
    // 5. X = Rows T' for the blocks below the diagonal, stored as L
    {
      const int j0 = owner ? 1 : 0;
      const int i = lane & 15, kk = lane >> 4;
      for (int hb = 2 * j0 + wave; hb < 2 * nown; hb += nwave) {
        const int j = hb >> 1, b = first + j * G;
        const int prow = j * RSBA_PB + (hb & 1) * 16;
        d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4) {
          const double a = Pan[(prow + i) * RSBA_PLD + qs + kk];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[i * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(16 + i) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int grow = b * RSBA_PB + (hb & 1) * 16 + kk + 4 * tt;
          if (grow <= n && (b < np || grow == n)) {
            StoreShared(&A[(size_t)grow * n + kb + i], acc0[tt]);
            StoreShared(&A[(size_t)grow * n + kb + 16 + i], acc1[tt]);
          }
        }
      }
    }
    RSBA_MC_STAMP(6);
    // 6. the owner of the next diagonal block: its rows are complete through this panel
    if (p + 1 < np && (p + 1) % G == w) PublishFlagWG(f.strip_ready + p + 1, tag);
    else { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
    RSBA_MC_STAMP(7);
  }

  if (stalled) {
    if (tid == 0) { __hip_atomic_store(f.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (w == 0) res[RES_STALL] = 1.0; }
    if (w == 0) SolveDone(gate);
    return;
  }
  if (w != 0) { PublishFlagWG(f.wg_done + w, tag); return; }
  // workgroup 0: everybody's rows, then L' x = y and the camera step
  for (int o = 1; o < G; ++o)
    if (!WaitFlagWG(f.wg_done + o, tag, f.error, budget)) { if (tid == 0) res[RES_STALL] = 1.0; SolveDone(gate); return; }
  double* ysol = A + (size_t)n * n;
  double* y = BackSubstituteBlocks(n, A, lds);
  for (int i = tid; i < n; i += nt) ysol[i] = y[i];
  __threadfence_block();
  __syncthreads();
  int ok = 1;
  if (tid == 0) { res[RES_STALL] = 0.0; ok = __hip_atomic_load(chol_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  CameraStepEpilogue(C, red, L, scale_c, ysol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, ok, lds, ip.cam_free);
  SolveDone(gate);
}

__host__ __device__ inline size_t MultiCholLdsDoubles(int n) { return (size_t)(n + RSBA_PB) * RSBA_PLD + 2 * RSBA_PB * RSBA_PLD + 64 + n; }

}  // namespace rsba
