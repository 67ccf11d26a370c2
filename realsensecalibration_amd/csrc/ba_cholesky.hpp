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

// LDS doubles: (n+2) x 33 shared by the panel and the B strip, 32 x 33 tiles for T and the padded L11, 32 inverse
// pivots, scratch, the column scale, four 32 x 33 tiles for the look-ahead products of the next diagonal block.
__host__ __device__ inline size_t CholeskyLdsDoubles(int n) {
  const size_t fact = (size_t)(n + 2) * RSBA_PLD + 2 * RSBA_PB * RSBA_PLD + RSBA_PB + 64 + (size_t)n + 4 * RSBA_PB * RSBA_PLD;
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
  int transposed;       // read entry (i, j), i >= j, as S[j][i]: the panel of camera group g then only touches the row slab
                        // of group g, which is what has been all-reduced when the multi-GPU pipeline opens the group's gate
};

// Columns of the matrix may still be in production when the factorisation starts (pipelined solve): gate.ready[1 + g]
// becomes gate.tag when the columns of camera group g (gate.cols wide) are complete; ready == nullptr: no gating.
// A factorisation launched AHEAD of its step (before the host knew the previous step's outcome, while that step's
// back-substitution was still running): state and radius are the device's decision (LmNext, ba_point_kernels.hpp).  The kernel
// waits for dec[3] == seq (the decision of exactly that step), then: dec[1] != 0 — accepted — the roles of the two camera
// buffers are swapped (x is the former candidate; the new candidate and its constants go where x was: camc_x), radius = dec[2].
// The host may still stop where the device went on (a tolerance, the time limit: tests the device does not run) — then x is the
// state the host returns, and an accepted-step factorisation would have written its candidate over it: before it writes anything
// the kernel saves x's cameras and constants (cam_backup: 6 C, camc_backup: C x CC_STRIDE doubles), and the host puts them back
// when it lets such a step run out unused (DrainAhead).
// Experimental paths (measured slower than, or worth nothing over, the defaults; HISTORY.md round 4) are compiled in only with
// -DRSBA_EXPERIMENTAL (tools/build_variant.sh exp -DRSBA_EXPERIMENTAL): the step launched ahead on the device's decision
// (RSBA_LAUNCH_AHEAD), the tiled factorisation gated beside the Schur kernel above 64 cameras (RSBA_PIPELINE_TILES: hung the
// suite once), the round-robin factorisation (RSBA_CHOL_DIAG=0).  RSBA_EXP(cond): `cond` there, a compile-time false in the
// product — the kernels do not carry the code.
#ifdef RSBA_EXPERIMENTAL
#define RSBA_EXP(cond) (cond)
#else
#define RSBA_EXP(cond) (false)
#endif
struct AheadSel {
  const double* dec = nullptr;
  double seq = 0.0;
  double* camc_x = nullptr;
  double* cam_backup = nullptr;
  double* camc_backup = nullptr;
};

struct StageGate {
  const int* ready;
  int tag, cols;
  int* done;          // = tag when the kernel has finished (or given up): the back-substitution, launched behind the Schur
                      // kernel on the main stream, waits for it inside the kernel instead of behind a cross-stream event
  long long* waited;  // += ticks (100 MHz) spent waiting, by thread 0: the kernel's duration minus this is its own work
  long long* trace;   // diagnostic (RSBA_TRACE=1): wall-clock stamps of the waits, nullptr otherwise
  long long budget;   // ticks a wait may last before the kernel gives up (0: RSBA_STALL_TICKS); the multi-GPU pipeline waits
                      // for other ranks' collectives and gets ten times as long
  // "every workgroup of this kernel is resident": the last one to start writes `tag` into the host's pinned word.  The host
  // waits for it before it launches the Schur kernel on the first step of a run (see PointsStep): the first launch on the
  // side stream was measured to start ~150 us late, and a factorisation that is not resident when the chip fills up does
  // not get a CU with enough LDS until the back-substitution — which waits for it — has left.
  int* started_cnt = nullptr;    // device
  int* started_host = nullptr;   // pinned host memory
  int started_need = 0;          // workgroups of the kernel
  // first iteration of a run, pipelined: = tag once every camera's diag U is written (the Schur kernel ran every self tile
  // first) — the Jacobi scale can be formed and the panels gated stage by stage like in every other iteration; nullptr: the
  // first iteration waits for all stages
  const int* all_diag = nullptr;
  // multi-GPU pipeline, diagonal-workgroup factorisation: entry (i, j), i >= j, is read as S[j][i] — when camera group g's panels
  // start only the group's ROW slab of S has been all-reduced (S is written symmetric to the bit, so the values are the same)
  int transposed = 0;
};

__device__ __forceinline__ void AnnounceResident(const StageGate& gate) {
  if (gate.started_host == nullptr || threadIdx.x != 0) return;
  if (__hip_atomic_fetch_add(gate.started_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gate.started_need - 1) {
    __hip_atomic_store(gate.started_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(gate.started_host, gate.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Spin (one lane, sleeping between polls) until *flag == tag; false when the producer does not show up in
// RSBA_STALL_TICKS of the 100 MHz wall clock — the caller gives up instead of hanging the queue.
#ifndef RSBA_STALL_TICKS
#define RSBA_STALL_TICKS 50000000LL
#endif
__device__ __forceinline__ bool WaitReady(const int* flag, int tag, long long* waited_ticks, long long budget = 0) {
  __shared__ int s_wait_ok;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int ok = 1;
    // relaxed polls: an acquire load invalidates this XCD's L2 every time (measured: 391 workgroups polling with acquire
    // doubled the duration of a latency-bound kernel next to them); the fence below acquires once
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > (budget > 0 ? budget : RSBA_STALL_TICKS)) { ok = 0; break; }
    }
    s_wait_ok = ok;
    if (waited_ticks) *waited_ticks += wall_clock64() - t0;
  }
  __syncthreads();
  __threadfence();  // every thread's later loads see what the producer published before the flag
  return s_wait_ok != 0;
}

// T = L11^-1 from the padded factor in Lt (32 x 33, upper part zero) and its inverse pivots: the second part of
// DiagFactorInverse, on its own for callers that hand L11 on first and invert behind the critical path.  One wavefront.
__device__ __forceinline__ void DiagInverse(double* __restrict__ T, const double* __restrict__ Lt, const double* __restrict__ invd, int lane) {
  const int lr = lane & 31;
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
          // L[hb + i][hb + q] from the tile just written: an LDS read with two distinct addresses per wave (one per
          // block) instead of two v_readlane pairs and a select; the reads of a step are independent of the chain
          const double lv = Lt[(hb + i) * RSBA_PLD + hb + q];
          if (q & 1) sacc2 -= lv * t[q]; else sacc -= lv * t[q];
        }
      }
      t[i] = (sacc + sacc2) * invd[hb + i];
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
}

// Factor the padded 32 x 32 diagonal block held in LDS (rows of `Pan`, stride RSBA_PLD) and invert it.  One wavefront.
// Out: Pan rows < nb = L11 (upper part zeroed; kWritePan), Lt = padded L11, T = L11^-1 (kInverse), invd = 1/diag.  Returns
// false on a non-positive pivot (no select sits on the pivot chain: the pivot's 1/sqrt and everything after it are NaN then,
// one ballot at the end finds out; the caller marks the step invalid).  Rows/columns >= nb are padded with identity so all
// 32 steps run unconditionally: one branch-free basic block.
//
// Lane = row (lanes 32..63 shadow lanes 0..31).  A wavefront issues IN ORDER, so what a step costs is decided by the
// program order (tools/factor_bench.hip, tools/lat_bench.hip: a dependent v_fma_f64 has 8 cycles of latency, v_rsq_f64 20,
// a v_readlane pair into an SGPR operand 20, LDS write -> read 76):
//   * the pivot chain of step j + 1 — v_readlane of the pivot, v_rsq_f64, two Newton steps of three dependent operations,
//     the scaling of the column, ~125 cycles — alternates, stage by stage, with the updates step j still owes (the empty asm
//     statements pin that order; without them LLVM also sinks the updates towards their uses and spills);
//   * column j + 1 (the next pivot) and j + 2 take their multiplier l_cj straight out of lane c's register (v_readlane ->
//     SGPR operand); for the columns >= j + 3 the scaled column goes through a 32-double LDS buffer (two, by step parity)
//     and comes back as broadcast reads, two columns per ds_read2_b64 instead of four v_readlane and their hazard nops:
//     read at the START of the next step, a whole step after the write, applied a few stages later.  No pivot ever
//     waits for LDS, and one step's multipliers are all the registers this takes (the callers sit at the 256-VGPR cap
//     and keep their state in the registers this routine leaves alone);
//   * two 16-column halves: the steps j < 16 only update the columns up to 15 (for all 32 rows: L11 of the top-left block
//     and L21 below it), the bottom-right block then takes its rank-16 update L21 L21' from the matrix cores in one go,
//     and the steps j >= 16 work on it alone.
// 3.6 us alone on a SIMD against 6.0 us for the version that scaled, broadcast and updated step by step (+ 2.1 us for the
// inverse either way).  Every sum has a fixed order: bitwise reproducible.
template <bool kInverse = true, bool kWritePan = true>
__device__ __forceinline__ bool DiagFactorInverse(double* __restrict__ Pan, int nb, double* __restrict__ T, double* __restrict__ Lt,
                                                  double* __restrict__ invd, int lane) {
#ifdef RSBA_PROFILE_PHASES
      long long _w0 = clock64();
#endif
      double row[RSBA_PB];
      const int lr = lane & 31;
      if (lr < nb) {
#pragma unroll
        for (int c = 0; c < RSBA_PB; ++c) row[c] = Pan[lr * RSBA_PLD + c];
      } else {
#pragma unroll
        for (int c = 0; c < RSBA_PB; ++c) row[c] = (c == lr) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) asm volatile("" : "+v"(row[c]));   // all 32 in registers now (else the padding constants of the late columns stay live beside the loaded values)
      double* colbuf = T + 20 * RSBA_PLD;   // rows 20, 21 of the T tile (two buffers, by step parity): free until the inverse, clear of the rank-16 product (rows < 16)
      double nv[16];                        // multipliers of step j - 1, read at the start of step j
      double ilv = 0.0, lij, il;            // lane j keeps 1 / L[j][j]
#define RSBA_PIN(x) asm volatile("" : "+v"(x))
      // pivot J from scratch (first step of a half): il = 1/sqrt(d) from the hardware estimate + two Newton steps (full fp64)
#define RSBA_FACTOR_CHAIN0(J)                                                                                           \
      {                                                                                                                 \
        const double d = ReadLaneD(row[J], J);                                                                          \
        double y = __builtin_amdgcn_rsq(d);                                                                             \
        double e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                               \
        y = __builtin_fma(y, e, y);                                                                                     \
        e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                                      \
        il = __builtin_fma(y, e, y);                                                                                    \
        lij = row[J] * il;                                                                                              \
      }
      // what step j does between two stages of the next pivot's chain: item 0 = column j + 2 by v_readlane, items 1.. =
      // the delayed updates of step j - 1 (columns j + 2 .. CEND - 1), nearest column first
#define RSBA_FACTOR_ITEMS(S)                                                                                            \
      _Pragma("unroll") for (int i = (S); i < nitems; i += 6) {                                                         \
        if (i == 0) {                                                                                                   \
          if (j + 2 < (CEND_)) { const double lc = ReadLaneD(lij, j + 2); row[j + 2] -= lij * lc; RSBA_PIN(row[j + 2]); } \
        } else if (j > (BASE_)) {                                                                                       \
          const int c = j + 1 + i;                                                                                      \
          if (c < (CEND_)) { row[c] -= row[j - 1] * nv[c - (BASE_)]; RSBA_PIN(row[c]); }                                \
        }                                                                                                               \
      }
#define RSBA_FACTOR_STEP                                                                                                \
      {                                                                                                                 \
        row[j] = lij;   /* l_ij; lane j: sqrt(d) */                                                                     \
        if (lr == j) ilv = il;                                                                                          \
        RSBA_PIN(ilv);   /* select now: else every step's il stays live to the end */                                   \
        const int nitems = (CEND_) - j - 1;                                                                             \
        double lij_n = 0.0, il_n = 0.0;                                                                                 \
        /* the multipliers step j - 1 left in its buffer a whole step ago, for the columns >= j + 2 */                  \
        if (j > (BASE_)) { _Pragma("unroll") for (int c = j + 2; c < (CEND_); ++c) nv[c - (BASE_)] = colbuf[((j - 1) & 1) * RSBA_PLD + c]; } \
        if (j + 3 < (CEND_)) { if (lane < 32) colbuf[(j & 1) * RSBA_PLD + lane] = lij; }                                \
        if (j + 1 < (CEND_)) {                                                                                          \
          { const double lc = ReadLaneD(lij, j + 1); row[j + 1] -= lij * lc; RSBA_PIN(row[j + 1]); }                    \
          const double d = ReadLaneD(row[j + 1], j + 1);                                                                \
          double y0 = __builtin_amdgcn_rsq(d); RSBA_PIN(y0);                                                            \
          double t = d * y0, h = 0.5 * y0; RSBA_PIN(t); RSBA_PIN(h);                                                    \
          RSBA_FACTOR_ITEMS(0)                                                                                          \
          double e = __builtin_fma(-t, h, 0.5); RSBA_PIN(e);                                                            \
          RSBA_FACTOR_ITEMS(1)                                                                                          \
          double y1 = __builtin_fma(y0, e, y0); RSBA_PIN(y1);                                                           \
          RSBA_FACTOR_ITEMS(2)                                                                                          \
          t = d * y1; h = 0.5 * y1; RSBA_PIN(t); RSBA_PIN(h);                                                           \
          RSBA_FACTOR_ITEMS(3)                                                                                          \
          e = __builtin_fma(-t, h, 0.5); RSBA_PIN(e);                                                                   \
          RSBA_FACTOR_ITEMS(4)                                                                                          \
          il_n = __builtin_fma(y1, e, y1); RSBA_PIN(il_n);                                                              \
          RSBA_FACTOR_ITEMS(5)                                                                                          \
          lij_n = row[j + 1] * il_n; RSBA_PIN(lij_n);                                                                   \
        }                                                                                                               \
        lij = lij_n; il = il_n;                                                                                         \
      }
      RSBA_FACTOR_CHAIN0(0)
#define CEND_ 16
#define BASE_ 0
#pragma unroll
      for (int j = 0; j < 16; ++j) RSBA_FACTOR_STEP
#undef CEND_
#undef BASE_
      {
        // columns 0..15 are final: into the Lt tile (upper part zero); A22 -= L21 L21' with L21 (rows 16..31) taken from
        // there in MFMA operand layout, the product back through the (still unused) T tile into the rows' registers
        if (lane < 32) {
#pragma unroll
          for (int c = 0; c < 16; ++c) Lt[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        const int mi = lane & 15, mk = lane >> 4;
        d4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
          const double a = Lt[(16 + mi) * RSBA_PLD + ks + mk];   // A[i][k] = L21[i][k]; B[k][j] = L21[j][k]: the same value
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) T[(mk + 4 * tt) * RSBA_PLD + mi] = acc[tt];   // D[row][col]
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const double upd = T[(lr & 15) * RSBA_PLD + c];
          row[16 + c] -= (lr >= 16) ? upd : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) RSBA_PIN(row[16 + c]);
      }
      RSBA_FACTOR_CHAIN0(16)
#define CEND_ 32
#define BASE_ 16
#pragma unroll
      for (int j = 16; j < RSBA_PB; ++j) RSBA_FACTOR_STEP
#undef CEND_
#undef BASE_
#undef RSBA_FACTOR_STEP
#undef RSBA_FACTOR_ITEMS
#undef RSBA_FACTOR_CHAIN0
#undef RSBA_PIN
      // columns 16..31 of the padded factor -> Lt (32 x 33); the real rows also back into the panel
      if (lane < 32) {
        invd[lane] = ilv;
#pragma unroll
        for (int c = 16; c < RSBA_PB; ++c) Lt[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
        if (kWritePan && lr < nb) {
#pragma unroll
          for (int c = 0; c < RSBA_PB; ++c) Pan[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool good = __builtin_amdgcn_ballot_w64(!(ilv > 0.0) || !(ilv <= DBL_MAX)) == 0;
#ifdef RSBA_PROFILE_PHASES
      if (lane == 0) { long long _w1 = clock64(); g_phase_cycles[10] += _w1 - _w0; _w0 = _w1; }
#endif
      if (!kInverse) return good;
      DiagInverse(T, Lt, invd, lane);
#ifdef RSBA_PROFILE_PHASES
      if (lane == 0) g_phase_cycles[11] += clock64() - _w0;
#endif
  return good;
}

// Out-of-line copy for the one-workgroup solver: that kernel sits at the 256-VGPR cap of a 512-thread workgroup, and with
// the factorisation inlined the register allocator spills into this loop (the critical path) whenever the code of the
// other waves grows.  As a separate function it gets its own allocation and the caller saves what it needs around one
// call per panel.  The LDS pointers keep their address space across the call (ds_read, not flat loads).
typedef __attribute__((address_space(3))) double lds_double;
static __device__ __noinline__ bool DiagFactorInverseCall(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) {
  return DiagFactorInverse<true, true>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane);
}
// ... the two parts separately (ba_cholesky_tiles.hpp: L11 is handed on before it is inverted)
static __device__ __noinline__ bool DiagFactorOnlyCall(lds_double* Pan, int nb, lds_double* Tscratch, lds_double* Lt, lds_double* invd, int lane) {
  return DiagFactorInverse<false, true>((double*)Pan, nb, (double*)Tscratch, (double*)Lt, (double*)invd, lane);
}
static __device__ __noinline__ void DiagInverseCall(lds_double* T, const lds_double* Lt, const lds_double* invd, int lane) {
  DiagInverse((double*)T, (const double*)Lt, (const double*)invd, lane);
}
// X = A L^-T by substitution: x_j = a_j / l_jj, then a_c -= x_j l_cj for c > j, 32 steps — with neither the inverse of L (2.1 us on
// the producer's critical path) nor the matrix cores.  Four lanes per row: lane q of a row keeps its columns c = 4 k + q in
// eight registers; the lane that owns column j forms x_j, a quad broadcast (DPP) hands it to the row's other lanes, and each
// applies it to its own later columns — at most eight multiply-adds per step and lane, the multipliers read from LDS a step
// ahead.  (A row per lane, one wavefront: 496 multiply-adds in a row, 4 us with its call; left to itself the compiler also
// read every multiplier right before its use, 9 us.)  src: the rows (nrows <= 64, a multiple of 16, `stride` apart) — thread t
// works on row t >> 2; LtT: L transposed, LtT[j * RSBA_PLD + c] = l_cj for c > j, 1 / l_jj for c == j, 0 for c < j; dst: X, rows
// RSBA_PLD apart.  The zero multipliers leave a row's finished columns alone, the owner's own column is set by a select.
template <int K> __device__ __forceinline__ double QuadBroadcast(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), K * 0x55, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), K * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void TrsmRowsQuad(const double* src, int stride, const double* LtT, double* dst, int tid, int nrows) {
  const int row = tid >> 2, q = tid & 3;
  if (row >= nrows) return;
  double a[8], l0[8], l1[8], l2[8], l3[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = src[row * stride + 4 * k + q];
#pragma unroll
  for (int k = 0; k < 8; ++k) { l0[k] = LtT[4 * k + q]; l1[k] = LtT[RSBA_PLD + 4 * k + q]; l2[k] = LtT[2 * RSBA_PLD + 4 * k + q]; }
  // a step is ~45 cycles of issue, an LDS read ~100: the multipliers of step j + 3 are read in step j.  (Measured 125 cycles per step.
  // Laying the step's chain out with its other multiply-adds between the links — the next pivot's column first, its product, two
  // multiply-adds, the broadcast, the rest — made the tiled factorisation 7 us slower, not faster.)
#define RSBA_TRSM_STEP(J, CUR, NXT)                                                                              \
  {                                                                                                              \
    constexpr int kj = (J) >> 2, qj = (J) & 3;                                                                   \
    if ((J) + 3 < RSBA_PB) {                                                                                     \
      _Pragma("unroll") for (int k = ((J) + 3) >> 2; k < 8; ++k) NXT[k] = LtT[((J) + 3) * RSBA_PLD + 4 * k + q]; \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    const double x = QuadBroadcast<qj>(a[kj] * CUR[kj]);                                                         \
    const double t = __builtin_fma(-x, CUR[kj], a[kj]);                                                          \
    a[kj] = (q == qj) ? x : t;                                                                                   \
    _Pragma("unroll") for (int k = kj + 1; k < 8; ++k) a[k] = __builtin_fma(-x, CUR[k], a[k]);                   \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }
#define RSBA_TRSM_STEP4(J) RSBA_TRSM_STEP(J, l0, l3) RSBA_TRSM_STEP((J) + 1, l1, l0) RSBA_TRSM_STEP((J) + 2, l2, l1) RSBA_TRSM_STEP((J) + 3, l3, l2)
  RSBA_TRSM_STEP4(0) RSBA_TRSM_STEP4(4) RSBA_TRSM_STEP4(8) RSBA_TRSM_STEP4(12)
  RSBA_TRSM_STEP4(16) RSBA_TRSM_STEP4(20) RSBA_TRSM_STEP4(24) RSBA_TRSM_STEP4(28)
#undef RSBA_TRSM_STEP4
#undef RSBA_TRSM_STEP
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[row * RSBA_PLD + 4 * k + q] = a[k];
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
// With a gate (pipelined solver: the kernel is launched before the matrix exists) the first panel of every camera
// group waits until that group's columns have been published by the Schur kernel, which is still eliminating points
// for the later groups on the rest of the chip; a stalled wait returns with *ok_out = -1.
__device__ void CholeskySolvePanelLDS(int n, double* __restrict__ A, double* __restrict__ x_out, int* ok_out, double* lds, PanelSource src,
                                      StageGate gate = StageGate{nullptr, 0, 0, nullptr, nullptr, nullptr, 0}) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
  double* T = lds + (size_t)(n + 2) * RSBA_PLD;        // 32 x 33: T = L11^-1
  double* Lt = T + RSBA_PB * RSBA_PLD;                 // 32 x 33: padded L11
  double* invd = Lt + RSBA_PB * RSBA_PLD;              // 32
  double* colb = invd + RSBA_PB;                       // 32 (+32 spare)
  double* scl = colb + 64;                             // n: LDS copy of the column scale (fused source only)
  double* ahead = scl + n;                             // 4 x (32 x 33): look-ahead products for the next diagonal block
  __shared__ int s_ok;
  __shared__ int s_ahead_read;   // the panel whose look-ahead tiles wave 0 has consumed (waves 1..4 may overwrite them)
  __shared__ int s_rb_next;      // next 16-row block of the update (2b): the waves take them as they get free
  if (tid == 0) { s_ok = 1; s_ahead_read = -1; s_rb_next = 0; }
  if (src.S != nullptr) for (int i = tid; i < n; i += nt) scl[i] = src.scale[i];
  __syncthreads();
  RSBA_STAMP_INIT;

  for (int kb = 0; kb < n; kb += RSBA_PB) {
    if (gate.ready != nullptr && kb % gate.cols == 0) {
      if (gate.trace && tid == 0) gate.trace[2 + 2 * (kb / gate.cols)] = wall_clock64();
      if (!WaitReady(gate.ready + 1 + kb / gate.cols, gate.tag, gate.waited, gate.budget)) { if (tid == 0) *ok_out = -1; __syncthreads(); return; }
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
    if (src.S != nullptr && src.transposed) {
      // column c of the panel = row kb + c of S, read along the row: lanes along r, stride-33 LDS stores
      const int nrow = n - kb;
      for (int r = tid & 127; r < nrow; r += 128) {
        const int gi = kb + r;
        const double si = scl[gi];
#pragma unroll
        for (int c = tid >> 7; c < RSBA_PB; c += 4) {   // nt = 512: four column classes
          const int gj = kb + c;
          double v = 0.0;
          if (c < nb) {
            v = src.S[(size_t)gj * n + gi] * (si * scl[gj]);
            if (gi == gj) v += fmin(fmax(si * si * src.diagU[gi], src.lo), src.hi) * src.inv_radius;
          }
          Pan[r * RSBA_PLD + c] = v;
        }
      }
      for (int c = tid; c < RSBA_PB; c += nt) Pan[(R - 1) * RSBA_PLD + c] = c < nb ? A[(size_t)n * n + kb + c] : 0.0;   // rhs row
    } else
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
    if (tid == 0) s_rb_next = 0;
    __syncthreads();
    RSBA_STAMP(0);
#ifdef RSBA_PROFILE_PHASES
    long long _x0 = 0;
    const long long _x00 = clock64();
#endif
    // 2. update with previous panels (MFMA), one wave per 16-row block, no barriers inside
    // 2a. the diagonal block's own 32 rows, by wave 0 alone: a rank-32 update with the previous panel from the LDS
    //     strip + the look-ahead products over all earlier columns that waves 1..4 computed during the previous panel
    //     (2c), so nothing but ~0.5 us of wave 0 stands between the panel load and the factorisation;
    // 2b. waves 1.. update the rows below while wave 0 factors the diagonal block (step 3): the serial factorisation
    //     hides behind the GEMM of the rest;
    // 2c. waves 1..4 also form, for the NEXT diagonal block, L[kb+32.., 0:kb] L[kb+32.., 0:kb]' from global memory
    //     (final entries of L, written back by the earlier panels), K-split four ways into `ahead`.
    if (kb > 0) {
      const int i = lane & 15, kk = lane >> 4;
      if (wave == 0) {
        // the previous panel's 32 columns: rank-32 update straight from the strip
        d4_t a00 = {0, 0, 0, 0}, a01 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double* bq = Bst + (size_t)(kb - RSBA_PB + 4 * u + kk) * RSBA_PLD;
          const double x0 = bq[i], x1 = bq[16 + i];   // row i / 16+i of the diagonal block, also columns i / 16+i
          a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, a00, 0, 0, 0);
          a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, a01, 0, 0, 0);
          a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, a10, 0, 0, 0);
          a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, a11, 0, 0, 0);
        }
#ifdef RSBA_PROFILE_PHASES
        if (lane == 0) g_phase_cycles[1] += clock64() - _x00;
#endif
        // all earlier columns: the four partial products waves 1..4 left in `ahead` during the previous panel, added
        // in a fixed order (bitwise reproducible: every rank of a multi-GPU run must end up with identical cameras)
        const bool have_ahead = kb > RSBA_PB;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = kk + 4 * t;
          const int e00 = r * RSBA_PLD + i, e01 = e00 + 16, e10 = (16 + r) * RSBA_PLD + i, e11 = e10 + 16;
          double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
          if (have_ahead) {
            const double *h0 = ahead, *h1 = ahead + RSBA_PB * RSBA_PLD, *h2 = h1 + RSBA_PB * RSBA_PLD, *h3 = h2 + RSBA_PB * RSBA_PLD;
            s00 = (h0[e00] + h1[e00]) + (h2[e00] + h3[e00]); s01 = (h0[e01] + h1[e01]) + (h2[e01] + h3[e01]);
            s10 = (h0[e10] + h1[e10]) + (h2[e10] + h3[e10]); s11 = (h0[e11] + h1[e11]) + (h2[e11] + h3[e11]);
          }
          // rows >= nb of a partial last panel are not the diagonal block's (the rhs row sits there): the waves of 2b
          // own them, and a read-modify-write of "minus zero" from here would race with their update
          if (r < nb) { Pan[e00] -= s00 + a00[t]; Pan[e01] -= s01 + a01[t]; }
          if (16 + r < nb) { Pan[e10] -= s10 + a10[t]; Pan[e11] -= s11 + a11[t]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_ahead_read, kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      // (no barrier: only wave 0 touches the diagonal block between the panel load and its factorisation)
#ifdef RSBA_PROFILE_PHASES
      _x0 = clock64();
#endif
      if (wave >= 1 && wave <= 4 && kb + RSBA_PB < n) {
        const int kbn = kb + RSBA_PB;
        const int r0g = kbn + i, r1g = kbn + 16 + i;
        const double* p0 = A + (size_t)(r0g < n ? r0g : 0) * n + 8 * kk;   // k order 8 kk + u, as in 2b
        const double* p1 = A + (size_t)(r1g < n ? r1g : 0) * n + 8 * kk;
        d4_t a00 = {0, 0, 0, 0}, a01 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
        {
          // this wave's column blocks qf, qf+128, qf+256 (kb <= 352: at most three): all loads first, one round trip
          const int qf = (wave - 1) * RSBA_PB;
          double x0[3][8], x1[3][8];
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const int q = qf + b * 4 * RSBA_PB;
#pragma unroll
            for (int u = 0; u < 8; ++u) { x0[b][u] = (r0g < n && q < kb) ? p0[q + u] : 0.0; x1[b][u] = (r1g < n && q < kb) ? p1[q + u] : 0.0; }
          }
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            if (qf + b * 4 * RSBA_PB < kb) {
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[b][u], x0[b][u], a00, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[b][u], x1[b][u], a01, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[b][u], x0[b][u], a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[b][u], x1[b][u], a11, 0, 0, 0);
              }
            }
          }
          for (int q0 = qf + 12 * RSBA_PB; q0 < kb; q0 += 4 * RSBA_PB) {   // n > 416 only (never with RSBA_CHOL_MAXN = 384)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const double z0 = r0g < n ? p0[q0 + u] : 0.0, z1 = r1g < n ? p1[q0 + u] : 0.0;
              a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, z0, a00, 0, 0, 0);
              a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, z1, a01, 0, 0, 0);
              a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, z0, a10, 0, 0, 0);
              a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, z1, a11, 0, 0, 0);
            }
          }
        }
        // wave 0 may still be reading the tiles of THIS panel (2a, a microsecond at the start of the phase)
        while (__hip_atomic_load(&s_ahead_read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != kb) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double* h = ahead + (size_t)(wave - 1) * RSBA_PB * RSBA_PLD;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = kk + 4 * t;
          h[r * RSBA_PLD + i] = a00[t]; h[r * RSBA_PLD + 16 + i] = a01[t];
          h[(16 + r) * RSBA_PLD + i] = a10[t]; h[(16 + r) * RSBA_PLD + 16 + i] = a11[t];
        }
      }
#ifdef RSBA_PROFILE_PHASES
      if (tid == 64) { long long _x1 = clock64(); g_phase_cycles[12] += _x1 - _x0; _x0 = _x1; }
#endif
      const int nrb = (R + 15) >> 4;
      // a partial last panel (nb < 32) keeps its rows nb..R-1 (the rhs row) outside the strip: they are updated here
      const int rb0 = nb == RSBA_PB ? 2 : (nb >> 4);
      // 16-row blocks handed out through an LDS counter (waves 1..4 arrive late from 2c and take fewer); the A operand
      // two steps ahead of the MFMAs
      for (;;) {
        if (!(wave > 0 || nwave == 1)) break;
        int blk = 0;
        if (lane == 0) blk = __hip_atomic_fetch_add(&s_rb_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        blk = __builtin_amdgcn_readfirstlane(blk);
        const int rb = rb0 + blk;
        if (rb >= nrb) break;
        const int prow = rb * 16 + i;  // panel-relative row of this lane's A operand
        const bool gl = prow < R;
        // k index of MFMA step u in lane group kk is 8 kk + u (for both operands): a lane's eight A values are then 64
        // contiguous bytes, four lanes cover 256 B of a row — every sector fetched is used (one CU only pulls a few
        // tens of GB/s; with the natural k = 4 u + kk order each load touched 16 rows x 32 B)
        const double* arow = A + (size_t)(kb + (gl ? prow : 0)) * n + 8 * kk;
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
#ifdef RSBA_PROFILE_PHASES
    if (kb > 0 && tid == 64) g_phase_cycles[13] += clock64() - _x0;
    if (kb > 0 && tid == 448) g_phase_cycles[14] += clock64() - _x0;
#endif
    // (no barrier here: wave 0 wrote the diagonal block's rows itself)
    // 3. diagonal block + its inverse, wave 0.  Rows/columns >= nb are padded with identity so that all 32
    //    steps run unconditionally; lanes 32..63 shadow lanes 0..31 (same values, same addresses), so the
    //    whole sequence is one branch-free basic block.  The empty asm statements pin every updated value at
    //    its step: without them LLVM sinks the updates towards their uses and spills ~1300 registers.
    if (wave == 0) {
#ifdef RSBA_PROFILE_PHASES
      if (lane == 0) g_phase_cycles[6] += clock64() - _x00;
      long long _c0 = clock64();
#endif
      if (!DiagFactorInverseCall((lds_double*)Pan, nb, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && lane == 0) s_ok = 0;
#ifdef RSBA_PROFILE_PHASES
      if (lane == 0) g_phase_cycles[7] += clock64() - _c0;
#endif
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

  double* y = BackSubstituteBlocks(n, A, lds);
  for (int i = tid; i < n; i += nt) x_out[i] = y[i];
  __syncthreads();
  RSBA_STAMP(5);
  if (tid == 0) *ok_out = s_ok;
  __syncthreads();
}

}  // namespace rsba
