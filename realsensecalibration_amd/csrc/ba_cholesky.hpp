// Dense SPD solve of the reduced camera system on one workgroup (the Eigen LLT of Ceres'
// DenseSchurComplementSolver), for n <= RSBA_CHOL_MAXN.
//
// Left-looking blocked Cholesky, 32-wide panels, 1024 threads (16 waves).  Per panel:
//   1. the panel rows kb..n (the right-hand side rides along as row n) are loaded into LDS, scaled and
//      damped on the fly when a raw source matrix is given (PanelSource)
//   2. update with all previous panels, Pan -= L[rows, 0:kb] * L[kb:kb+32, 0:kb]', on the matrix cores
//      (v_mfma_f64_16x16x4_f64): the B operand (the panel's own 32 rows of L, all kb columns) is staged in
//      LDS once — panel + strip together always fit the same (n+1) x 33 doubles — so the k loop runs
//      without barriers; the A operand streams from L2 one step ahead of the MFMAs
//   3. 32x32 diagonal block factorised by wave 0 in registers (lane = row; column j goes through LDS once per
//      step and comes back as broadcast reads; no branches: partial panels are padded with identity),
//      1/sqrt(pivot) from v_rsq_f64 + two Newton steps; then T = L11^-1 (lane = column)
//   4. rows below: X = Rows * T' on the matrix cores
//   5. panel written back (L overwrites A); T is kept in the block's strict upper triangle (+ 1/diag in row n+1)
// then y = row n (the forward substitution came for free) and a blocked back-substitution that uses the stored
// T blocks, so no step of it is sequential.
// Everything latency-critical stays in LDS/registers.  Non-positive pivots clear *ok (Ceres:
// LINEAR_SOLVER_FAILURE -> invalid step).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>

namespace rsba {

#define RSBA_CHOL_MAXN 384
#define RSBA_PB 32                       // panel width
#define RSBA_PLD (RSBA_PB + 1)           // LDS leading dimension (bank-conflict padding)

typedef double d4_t __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DRSBA_PROFILE_PHASES): cycle totals per phase, read back with hipMemcpyFromSymbol.
#ifdef RSBA_PROFILE_PHASES
__device__ long long g_phase_cycles[16];
#define RSBA_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { long long _t = clock64(); g_phase_cycles[k] += _t - _t0; _t0 = _t; } } while (0)
#define RSBA_STAMP_INIT long long _t0 = clock64()
#else
#define RSBA_STAMP(k) do {} while (0)
#define RSBA_STAMP_INIT do {} while (0)
#endif

__device__ __forceinline__ double ReadLaneD(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// LDS doubles: (n+2) x 33 shared by the panel and the B strip, a 32 x 33 tile for T, 32 inverse pivots, scratch.
__host__ __device__ inline size_t CholeskyLdsDoubles(int n) {
  const size_t fact = (size_t)(n + 2) * RSBA_PLD + 2 * RSBA_PB * RSBA_PLD + RSBA_PB + 64 + (size_t)n;
  const size_t back = (size_t)((n + 63) & ~63) + 3 * RSBA_PB * RSBA_PLD + 64;
  return fact > back ? fact : back;
}

// Optional fused source: when src.S != nullptr the panel rows are not read from A but built on the fly from the
// raw (unscaled, undamped, full symmetric) matrix: A_ij = S_ij s_i s_j (+ clamp(s_i^2 diagU_i, lo, hi)/radius on the
// diagonal), which saves a separate pass over the matrix.  Row n (the rhs) is always taken from A.
struct PanelSource {
  const double* S;
  const double* scale;
  const double* diagU;
  double lo, hi, inv_radius;
  const double* gc;     // gated solve only: the rhs entries of a camera group, s_i (gc_i + corr_i), are written into
  const double* corr;   // row n of A when the group's gate opens (they come from the same stage of the Schur kernel)
};

// Columns of the matrix may still be in production when the factorisation starts (pipelined solve): gate.ready[1 + g]
// becomes gate.tag when the columns of camera group g (gate.cols wide) are complete; ready == nullptr: no gating.
struct StageGate {
  const int* ready;
  int tag, cols;
  long long* waited;  // += ticks (100 MHz) spent waiting, by thread 0: the kernel's duration minus this is its own work
  long long* trace;   // diagnostic (RSBA_TRACE=1): wall-clock stamps of the waits, nullptr otherwise
};

// Spin (one lane, sleeping between polls) until *flag == tag; false when the producer does not show up in
// RSBA_STALL_TICKS of the 100 MHz wall clock — the caller gives up instead of hanging the queue.
#ifndef RSBA_STALL_TICKS
#define RSBA_STALL_TICKS 5000000LL
#endif
__device__ __forceinline__ bool WaitReady(const int* flag, int tag, long long* waited_ticks) {
  __shared__ int s_wait_ok;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > RSBA_STALL_TICKS) { ok = 0; break; }
    }
    s_wait_ok = ok;
    if (waited_ticks) *waited_ticks += wall_clock64() - t0;
  }
  __syncthreads();
  __threadfence();  // every thread's later loads see what the producer published before the flag
  return s_wait_ok != 0;
}

// Factor the padded 32 x 32 diagonal block held in LDS (rows of `Pd`, stride RSBA_PLD) and invert it.  One wavefront.
// Out: Pd rows < nb = L11 (upper part zeroed), Lt = padded L11, T = L11^-1, invd = 1/diag.  Returns false on a
// non-positive pivot.  (Rows/columns >= nb are padded with identity so all 32 steps run unconditionally; lanes 32..63
// shadow lanes 0..31, so the whole sequence is one branch-free basic block.  The empty asm statements pin every updated
// value at its step: without them LLVM sinks the updates towards their uses and spills ~1300 registers.)
__device__ __forceinline__ bool DiagFactorInverse(double* __restrict__ Pan, int nb, double* __restrict__ T, double* __restrict__ Lt,
                                                  double* __restrict__ invd, int lane) {
#ifdef RSBA_PROFILE_PHASES
      long long _w0 = clock64();
#endif
      double row[RSBA_PB];
      const int lr = lane & 31;
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) row[c] = (lr < nb) ? Pan[lr * RSBA_PLD + c] : (c == lr ? 1.0 : 0.0);
      bool good = true;
      double ilv = 1.0;  // 1 / L[lr][lr]
#pragma unroll
      for (int j = 0; j < RSBA_PB; ++j) {
        const double d = ReadLaneD(row[j], j);
        if (!(d > 0.0) || !(d <= DBL_MAX)) good = false;
        const double dd = good ? d : 1.0;
        // il = 1/sqrt(d): hardware estimate + two Newton steps (full fp64); l = d * il
        double il = __builtin_amdgcn_rsq(dd);
        il = il * (1.5 - 0.5 * dd * il * il);
        il = il * (1.5 - 0.5 * dd * il * il);
        const double lij = (lr == j) ? dd * il : row[j] * il;
        row[j] = lij;
        if (lr == j) ilv = il;
        invd[j] = il;  // wave-uniform value
        // a_ic -= l_ij l_cj with l_cj read straight out of lane c's register (v_readlane -> SGPR operand): no LDS
        // round trip on the critical path.  Entries above the diagonal (c > row) pick up garbage; never read.
        // groups of four: the eight v_readlane of a group issue back to back, so the SGPR-write -> VALU-read hazard
        // of one value is covered by the next ones instead of s_nops
#pragma unroll
        for (int c0 = j + 1; c0 < RSBA_PB; c0 += 4) {
          double lc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) lc[u] = (c0 + u < RSBA_PB) ? ReadLaneD(lij, c0 + u) : 0.0;
#pragma unroll
          for (int u = 0; u < 4; ++u) if (c0 + u < RSBA_PB) row[c0 + u] -= lij * lc[u];
#pragma unroll
          for (int u = 0; u < 4; ++u) if (c0 + u < RSBA_PB) asm volatile("" : "+v"(row[c0 + u]));
        }
      }
      // padded factor -> Lt (32 x 33); the real rows also back into the panel
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) {
        const double v = (c <= lr) ? row[c] : 0.0;
        Lt[lr * RSBA_PLD + c] = v;
        if (lr < nb) Pan[lr * RSBA_PLD + c] = v;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
#ifdef RSBA_PROFILE_PHASES
      if (lane == 0) { long long _w1 = clock64(); g_phase_cycles[10] += _w1 - _w0; _w0 = _w1; }
#endif
      // T = L11^-1 in 16 x 16 blocks: T = [[T11, 0], [-T22 L21 T11, T22]].  Lanes 0..15 invert the top-left block and
      // lanes 16..31 the bottom-right one at the same time (column lr & 15 each, a 16-step chain instead of 32); the
      // off-diagonal block is two 16x16x16 products on the matrix cores.
      {
        const int hb = lr & 16;          // 0: block (0,0), 16: block (1,1)
        const int lc = lr & 15;          // column inside the block
        double t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          double sacc = (i == lc) ? 1.0 : 0.0, sacc2 = 0.0;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if (q < i) {
              // L[hb + i][hb + q] lives in lane hb + i, register row[hb + q]
              const double la = ReadLaneD(row[q], i), lb = ReadLaneD(row[16 + q], 16 + i);
              const double lv = hb ? lb : la;
              if (q & 1) sacc2 -= lv * t[q]; else sacc -= lv * t[q];
            }
          }
          const double ia_ = ReadLaneD(ilv, i), ib_ = ReadLaneD(ilv, 16 + i);
          t[i] = (sacc + sacc2) * (hb ? ib_ : ia_);
          asm volatile("" : "+v"(t[i]));
        }
        // diagonal blocks into the T tile; M1 scratch = T[0..15][16..31]
        if (lane < RSBA_PB) {
#pragma unroll
          for (int i = 0; i < 16; ++i) T[(hb + i) * RSBA_PLD + hb + lc] = t[i];
        }
        __builtin_amdgcn_wave_barrier();
        const int mi = lane & 15, mk = lane >> 4;
        // M1 = L21 T11:  A[i][k] = L[16+i][k] (Lt), B[k][j] = T11[k][j]
        d4_t m1 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[(16 + mi) * RSBA_PLD + ks + mk], T[(ks + mk) * RSBA_PLD + mi], m1, 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) T[(mk + 4 * tt) * RSBA_PLD + 16 + mi] = m1[tt];   // M1[row][col] -> scratch quadrant
        __builtin_amdgcn_wave_barrier();
        // T21 = -T22 M1:  A[i][k] = T22[i][k] = T[16+i][16+k], B[k][j] = M1[k][j]
        d4_t t21 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) t21 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(16 + mi) * RSBA_PLD + 16 + ks + mk], T[(ks + mk) * RSBA_PLD + 16 + mi], t21, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) { T[(16 + mk + 4 * tt) * RSBA_PLD + mi] = -t21[tt]; T[(mk + 4 * tt) * RSBA_PLD + 16 + mi] = 0.0; }
      }
#ifdef RSBA_PROFILE_PHASES
      if (lane == 0) g_phase_cycles[11] += clock64() - _w0;
#endif
  return good;
}

// Blocked back-substitution L' x = y with the stored block inverses; y (row n of A) is copied to LDS and holds x on
// return (first n doubles of `lds`).  All threads of the workgroup.
__device__ __forceinline__ double* BackSubstituteBlocks(int n, double* __restrict__ A, double* lds) {
  const int tid = threadIdx.x, nt = blockDim.x;
  // Back-substitution L' x = y, right-looking from the bottom: x_blk = T' y_blk with the stored T = L11^-1
  // (T'[c][i] = T[i][c], i >= c; no sequential step), then y[0:kb] -= L[kb:kb+nb, 0:kb]' x_blk, which reads the
  // block's rows of L contiguously (one round of global latency per block).
  double* y = lds;                                 // n
  double* Tb = lds + ((n + 63) & ~63);             // 32 x 33: Tb[i][c] = T[i][c]
  double* xb = Tb + RSBA_PB * RSBA_PLD;            // 32
  for (int i = tid; i < n; i += nt) y[i] = A[(size_t)n * n + i];
  __syncthreads();
  // Global loads are issued a phase early: the T block of the NEXT block row is fetched while this one is solved, and
  // the strip of L this block multiplies is fetched before x_blk exists (it does not depend on it).
  const int kb_last = ((n - 1) / RSBA_PB) * RSBA_PB;
  double tpre[2] = {0.0, 0.0};  // this thread's two entries of the next T block (nt = 512: 1024 entries)
  auto fetch_T = [&](int kb, int slot) {
    const int e = tid + slot * nt;
    const int nbk = min(RSBA_PB, n - kb);
    const int i = e >> 5, c = e & 31;
    double tv = 0.0;
    // T[i][c] for i > c sits at A[kb+c][kb+i]; the diagonal in row n+1
    if (e < RSBA_PB * RSBA_PB && i < nbk && c < nbk) tv = (i > c) ? A[(size_t)(kb + c) * n + kb + i] : (i == c ? A[(size_t)(n + 1) * n + kb + c] : 0.0);
    return tv;
  };
  for (int sl = 0; sl < 2; ++sl) tpre[sl] = (tid + sl * nt < RSBA_PB * RSBA_PB) ? fetch_T(kb_last, sl) : 0.0;
  for (int kb = kb_last; kb >= 0; kb -= RSBA_PB) {
    const int nb = min(RSBA_PB, n - kb);
    for (int sl = 0; sl < 2; ++sl) { const int e = tid + sl * nt; if (e < RSBA_PB * RSBA_PB) Tb[(e >> 5) * RSBA_PLD + (e & 31)] = tpre[sl]; }
    // strip of L for this block (rows kb..kb+nb, column q = tid): independent of x_blk, so load it now
    double lv[RSBA_PB];
    const int q = tid;  // kb <= 352 < nt
#pragma unroll
    for (int c = 0; c < RSBA_PB; ++c) lv[c] = (c < nb && q < kb) ? A[(size_t)(kb + c) * n + q] : 0.0;
    if (kb >= RSBA_PB) { for (int sl = 0; sl < 2; ++sl) tpre[sl] = (tid + sl * nt < RSBA_PB * RSBA_PB) ? fetch_T(kb - RSBA_PB, sl) : 0.0; }
    __syncthreads();
    if (tid < RSBA_PB) {
      double sacc = 0.0;
#pragma unroll 8
      for (int i = 0; i < RSBA_PB; ++i) sacc += Tb[i * RSBA_PLD + tid] * ((i < nb) ? y[kb + i] : 0.0);
      xb[tid] = (tid < nb) ? sacc : 0.0;
    }
    __syncthreads();
    if (tid < nb) y[kb + tid] = xb[tid];
    if (q < kb) {
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) sacc += lv[c] * xb[c];
      y[q] -= sacc;
    }
    for (int q2 = tid + nt; q2 < kb; q2 += nt) {  // n > 512 never reaches here (RSBA_CHOL_MAXN), kept for safety
      double sacc = 0.0;
      for (int c = 0; c < nb; ++c) sacc += A[(size_t)(kb + c) * n + q2] * xb[c];
      y[q2] -= sacc;
    }
    __syncthreads();
  }
  return y;
}

// A: (n+2) x n row-major in global memory; rows 0..n-1 the SPD matrix (lower triangle read), row n the rhs,
// row n+1 scratch (inverse pivots).  On return the lower triangle holds L, row n holds y = L^-1 rhs, x_out x.
// Only the panels kb_begin <= kb < kb_end are factored, and the back-substitution runs only when `solve` is set: the
// pipelined solver calls this once per camera group, as soon as that group's columns of the matrix exist, while the
// rest of the chip is still eliminating points for the later groups.  All state between calls lives in A.
// With a gate the panels wait for their camera group's columns; a stalled wait returns with *ok_out = -1.
__device__ void CholeskySolvePanelLDS(int n, double* __restrict__ A, double* __restrict__ x_out, int* ok_out, double* lds, PanelSource src,
                                      int kb_begin, int kb_end, bool solve, StageGate gate = StageGate{nullptr, 0, 0, nullptr, nullptr}) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
  double* T = lds + (size_t)(n + 2) * RSBA_PLD;        // 32 x 33: T = L11^-1
  double* Lt = T + RSBA_PB * RSBA_PLD;                 // 32 x 33: padded L11
  double* invd = Lt + RSBA_PB * RSBA_PLD;              // 32
  double* colb = invd + RSBA_PB;                       // 32 (+32 spare)
  double* scl = colb + 64;                             // n: LDS copy of the column scale (fused source only)
  __shared__ int s_ok;
  if (tid == 0) s_ok = 1;
  if (src.S != nullptr) for (int i = tid; i < n; i += nt) scl[i] = src.scale[i];
  __syncthreads();
  RSBA_STAMP_INIT;

  for (int kb = kb_begin; kb < kb_end; kb += RSBA_PB) {
    if (gate.ready != nullptr && kb % gate.cols == 0) {
      if (gate.trace && tid == 0) gate.trace[2 + 2 * (kb / gate.cols)] = wall_clock64();
      if (!WaitReady(gate.ready + 1 + kb / gate.cols, gate.tag, gate.waited)) { if (tid == 0) *ok_out = -1; __syncthreads(); return; }
      if (gate.trace && tid == 0) gate.trace[3 + 2 * (kb / gate.cols)] = wall_clock64();
      if (src.gc != nullptr) {
        for (int i = kb + tid; i < min(kb + gate.cols, n); i += nt) A[(size_t)n * n + i] = src.scale[i] * (src.gc[i] + src.corr[i]);
        __threadfence_block();
        __syncthreads();
      }
    }
    const int nb = min(RSBA_PB, n - kb);
    const int R = n + 1 - kb;  // panel rows, rhs row included (panel-relative row R-1)
    // LDS split of the shared (n+1) x 33 area: B strip first (kb rows of 33: Bst[q][c] = L[kb+c][q]), panel after
    double* Bst = lds;
    double* Pan = lds + (size_t)kb * RSBA_PLD;
    // 1. load the panel (scaled / damped on the fly) and the B strip
    // four consecutive doubles of a row per thread and step (two dwordx4 loads): one round of latency per panel
#pragma unroll 4
    for (int e = tid; e < R * (RSBA_PB / 4); e += nt) {
      const int r = e >> 3, c0 = (e & 7) * 4;
      const int gi = kb + r;
      double v[4] = {0.0, 0.0, 0.0, 0.0};
      const bool fused_row = src.S != nullptr && gi < n;
      const double* srow = (fused_row ? src.S : A) + (size_t)gi * n + kb + c0;
      if (c0 + 3 < nb && ((((size_t)gi * n + kb + c0) & 1) == 0)) {
        const double2 a01 = *reinterpret_cast<const double2*>(srow), a23 = *reinterpret_cast<const double2*>(srow + 2);
        v[0] = a01.x; v[1] = a01.y; v[2] = a23.x; v[3] = a23.y;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (c0 + u < nb) v[u] = srow[u];
      }
      if (fused_row) {
        const double si = scl[gi];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int gj = kb + c0 + u;
          if (c0 + u < nb) {
            v[u] *= si * scl[gj];
            if (gi == gj) v[u] += fmin(fmax(si * si * src.diagU[gi], src.lo), src.hi) * src.inv_radius;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) Pan[r * RSBA_PLD + c0 + u] = v[u];
    }
    // B strip: row kb+c of L, columns 0..kb, four at a time, transposed into Bst[q][c]
#pragma unroll 4
    for (int e = tid; e < (kb >> 2) * RSBA_PB; e += nt) {
      const int c = e / (kb >> 2), q0 = (e - c * (kb >> 2)) * 4;
      double v[4] = {0.0, 0.0, 0.0, 0.0};
      if (c < nb) {
        const double* lrow = A + (size_t)(kb + c) * n + q0;
        if (((((size_t)(kb + c) * n + q0)) & 1) == 0) {
          const double2 a01 = *reinterpret_cast<const double2*>(lrow), a23 = *reinterpret_cast<const double2*>(lrow + 2);
          v[0] = a01.x; v[1] = a01.y; v[2] = a23.x; v[3] = a23.y;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = lrow[u];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) Bst[(q0 + u) * RSBA_PLD + c] = v[u];
    }
    __syncthreads();
    RSBA_STAMP(0);
    // 2. update with previous panels (MFMA), one wave per 16-row block, no barriers inside
    // 2a. the diagonal block's own 32 rows first, K-split over all waves (both operands come from the LDS strip),
    //     partial products added into the panel in wave order;
    // 2b. then waves 1.. update the rows below while wave 0 goes straight on to factor the diagonal block (step 3):
    //     the serial factorisation hides behind the GEMM of the rest.
    if (kb > 0) {
      const int i = lane & 15, kk = lane >> 4;
      {
        d4_t a00 = {0, 0, 0, 0}, a01 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
        for (int q0 = wave * RSBA_PB; q0 < kb; q0 += nwave * RSBA_PB) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const double* bq = Bst + (size_t)(q0 + 4 * u + kk) * RSBA_PLD;
            const double x0 = bq[i], x1 = bq[16 + i];   // row i / 16+i of the diagonal block, also columns i / 16+i
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, a01, 0, 0, 0);
            a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, a10, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, a11, 0, 0, 0);
          }
        }
        // the waves add their partial products in wave order (plain read-modify-write, one barrier per wave) so the
        // factor is bitwise reproducible: every rank of a multi-GPU run must end up with identical cameras
        for (int wv = 0; wv < nwave && wv * RSBA_PB < kb; ++wv) {
          if (wave == wv) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int r = kk + 4 * t;
              Pan[r * RSBA_PLD + i] -= a00[t];
              Pan[r * RSBA_PLD + 16 + i] -= a01[t];
              Pan[(16 + r) * RSBA_PLD + i] -= a10[t];
              Pan[(16 + r) * RSBA_PLD + 16 + i] -= a11[t];
            }
          }
          __syncthreads();
        }
      }
      __syncthreads();
      const int nrb = (R + 15) >> 4;
      const int nw1 = nwave > 1 ? nwave - 1 : 1;
      // a partial last panel (nb < 32) keeps its rows nb..R-1 (the rhs row) outside the strip: they are updated here
      const int rb0 = nb == RSBA_PB ? 2 : (nb >> 4);
      for (int rb = rb0 + (nwave > 1 ? wave - 1 : 0); rb < nrb && (wave > 0 || nwave == 1); rb += nw1) {
        const int prow = rb * 16 + i;  // panel-relative row of this lane's A operand
        const bool gl = prow < R;
        const double* arow = A + (size_t)(kb + (gl ? prow : 0)) * n + kk;
        d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        double an[8], an2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { an[u] = gl ? arow[4 * u] : 0.0; an2[u] = (gl && RSBA_PB < kb) ? arow[RSBA_PB + 4 * u] : 0.0; }
        for (int q0 = 0; q0 < kb; q0 += RSBA_PB) {
          double ac[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { ac[u] = an[u]; an[u] = an2[u]; }
          if (q0 + 2 * RSBA_PB < kb) {
#pragma unroll
            for (int u = 0; u < 8; ++u) an2[u] = gl ? arow[q0 + 2 * RSBA_PB + 4 * u] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const double b0 = Bst[(q0 + 4 * u + kk) * RSBA_PLD + i];
            const double b1 = Bst[(q0 + 4 * u + kk) * RSBA_PLD + 16 + i];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], b1, acc1, 0, 0, 0);
          }
        }
        // D layout: col = lane & 15, row = (lane >> 4) + 4 t
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = rb * 16 + kk + 4 * t;
          if (r < R && r >= nb) {
            Pan[r * RSBA_PLD + i] -= acc0[t];
            Pan[r * RSBA_PLD + 16 + i] -= acc1[t];
          }
        }
      }
    }
    // (no barrier here: wave 0 wrote the diagonal block's rows itself)
    // 3. diagonal block + its inverse, wave 0.  Rows/columns >= nb are padded with identity so that all 32
    //    steps run unconditionally; lanes 32..63 shadow lanes 0..31 (same values, same addresses), so the
    //    whole sequence is one branch-free basic block.  The empty asm statements pin every updated value at
    //    its step: without them LLVM sinks the updates towards their uses and spills ~1300 registers.
    if (wave == 0) {
      if (!DiagFactorInverse(Pan, nb, T, Lt, invd, lane) && lane == 0) s_ok = 0;
    }
    __syncthreads();
    RSBA_STAMP(2);
    // 4. rows below the diagonal block (and the rhs row): X = Rows * T' on the matrix cores.
    //    X[r][j] = sum_k Rows[r][k] T[j][k]:  A-op[i][k] = Pan[r0+i][k],  B-op[k][j] = T[j][k]
    {
      const int nrb4 = (R - nb + 15) >> 4;
      const int i = lane & 15, kk = lane >> 4;
      for (int rb = wave; rb < nrb4; rb += nwave) {
        const int prow = nb + rb * 16 + i;
        d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4) {
          const double a = prow < R ? Pan[prow * RSBA_PLD + qs + kk] : 0.0;
          const double b0 = T[i * RSBA_PLD + qs + kk];
          const double b1 = T[(16 + i) * RSBA_PLD + qs + kk];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();  // all of this block's rows are read before any is overwritten
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int r = nb + rb * 16 + kk + 4 * tt;
          if (r < R) {
            Pan[r * RSBA_PLD + i] = acc0[tt];
            Pan[r * RSBA_PLD + 16 + i] = acc1[tt];
          }
        }
      }
    }
    __syncthreads();
    RSBA_STAMP(3);
    // 5. write the panel back; strict upper triangle of the diagonal block <- T (transposed), row n+1 <- 1/pivots
#pragma unroll 4
    for (int e = tid; e < R * RSBA_PB; e += nt) {
      const int r = e >> 5, c = e & 31;
      if (c < nb) {
        double v = Pan[r * RSBA_PLD + c];
        if (r < nb && c > r) v = T[c * RSBA_PLD + r];  // A[kb+r][kb+c] = T[c][r], c > r
        A[(size_t)(kb + r) * n + kb + c] = v;
      }
    }
    if (tid < nb) A[(size_t)(n + 1) * n + kb + tid] = invd[tid];
    __threadfence_block();
    __syncthreads();
    RSBA_STAMP(4);
  }

  if (solve) {
    double* y = BackSubstituteBlocks(n, A, lds);
    for (int i = tid; i < n; i += nt) x_out[i] = y[i];
    __syncthreads();
  }
  RSBA_STAMP(5);
  if (tid == 0) *ok_out = s_ok;
  __syncthreads();
}

}  // namespace rsba
