// The reduced camera system on several workgroups with the DIAGONAL CHAIN IN ONE WORKGROUP (32 to 64 cameras).
//
// ba_cholesky_multi.hpp deals the 32-row blocks round-robin, so the chain factor(p) -> T(p) -> X(p+1, p) -> X X' ->
// factor(p+1) crosses from one workgroup to the next at EVERY panel: store, acknowledgement, flag, poll, load — ~2.6 us of
// the ~16 us a panel took, and the next owner's row update had to be finished by then as well.  Here
//
//   * workgroup 0 owns the diagonal: at panel p its wave 0 factors block p (DiagFactorInverse) while six other waves bring
//     the columns of panel p up to date in the rows of block p + 1 (three K slices per 16-row half, matrix cores); then
//     X(p+1, p) = Rows T', X X', and wave 0 goes on with block p + 1.  T never leaves the workgroup on the critical path.
//     Wave 4 shares wave 0's SIMD and stays out of everything between two factorisations: fp64 MFMAs run on the SIMD's fp64
//     vector lanes on this chip, and a wave issuing them back to back slows its SIMD partner down 14-fold
//     (tools/factor_bench.hip) — for the same reason nothing but MFMA work runs beside the update waves.
//   * workgroups 1 .. G-1 own the rows BELOW: block b belongs to workgroup 1 + b mod (G - 1) for the panels p <= b - 2.  Per
//     panel: the strip of block p (flag strip_ready, LDS-DMA) and the update of their blocks' columns of the panel in one
//     round trip (RowUpdateHalf); the block that leaves (b = p + 2) is handed over RIGHT THEN — unsolved, with its look-ahead
//     sum  sum_{q <= b-3} X(b, q) X(b, q)'  (three 16 x 16 tiles per block, kept in LDS), flag rows_ready[b], through one store
//     wave — and workgroup 0 forms X(b, b-2) with the T(b-2) it still has in LDS; for the others they wait for T(p) (flag
//     tdone), solve X = Rows T', store it and add X X' to the blocks' sums.  Workgroup 0 therefore never waits for an X that
//     needs its own T to come back first, and only adds the last two panels' X X' to a diagonal block itself — it used to
//     stream the block's rows a second time for the whole sum, 43 % of its matrix-core work.
//   * the right-hand-side row is block np (one row): a row workgroup's like any other until panel np - 2 (handed over solved:
//     rows_ready[np]), workgroup 0's at the last panel.
//   * stores are wave 4's business in workgroup 0 (L11 / T the moment the block is factored, the two newest X blocks, their
//     acknowledgements, the flags tdone and strip_ready); nobody on the chain waits for a store.
//
// STATUS (round 2): the default for 32 to 64 cameras, on six workgroups (RSBA_CHOL_DIAG=0 selects the round-robin kernel):
// 0.465 - 0.470 against 0.479 ms per LM iteration at 64 cameras, 184 against 221 us alone.  A panel of workgroup 0 takes 11 - 13 us when the next block is
// there (factorisation 6.3, update waves' data 4 - 5 us after they ask + 2 - 4 us of matrix cores, tail 2.2), but the row
// workgroups need 13 - 15 us per panel (every global round trip costs 2.3 - 8 us beside the Schur kernel), so it waits for
// rows_ready in about half of the panels.  DESIGN.md section 4, "round 2", item 5 has the measurements and what was dropped.
//
// Round 3: the stage of block p + 1's own columns of S is waited for where it is needed — in the panel's tail, for the next diagonal
// block's entries — not in front of panel p's update; T(p) is published as soon as its stores are acknowledged; a row workgroup
// acquires once behind a stage's flag.  0.4055 -> 0.3936 ms per LM iteration at 64 cameras ("Round 3").
//
// Same arithmetic per entry as the multi kernel's (products over fixed K slices, added in a fixed order): bitwise
// reproducible, identical on every rank.  All waits carry a budget: a stall gives up (RES_STALL), never hangs.
#pragma once
#include "ba_cholesky_multi.hpp"
#include "ba_cholesky_border.hpp"

namespace rsba {

struct DiagCholFlags {
  int* tdone;         // [16]  == tag when panel p's L11 / T are in global memory
  int* strip_ready;   // [16]  == tag when the rows of block p hold L for all columns < 32 p          (workgroup 0)
  int* rows_ready;    // [16]  == tag when the rows of block b hold L for all columns < 32 (b - 1)    (its row workgroup)
  int* error;         // [0] != 0: somebody gave up waiting; [4 + w]: (tag << 4) | panels workgroup w is through
  double* dg;         // [np + 1][32 * 32] look-ahead sums handed over with the blocks (tiles (0,0), (1,0), (1,1) of 256)
  double* ah;         // [np + 1][32 * 32] the blocks as handed over: updated through their second-last panel, unsolved
};

#ifndef RSBA_DC_UPD_NPF
#define RSBA_DC_UPD_NPF 2   // 32-column slabs in flight in load_update_half (panel 0 only: nothing to subtract there yet)
#endif

__host__ __device__ inline size_t DiagCholLdsDoubles(int nc) {
  const int n = MultiCholPadded(nc);
  return (size_t)(n + RSBA_PB) * RSBA_PLD + 5 * RSBA_PB * RSBA_PLD + 32 + n + 1024;   // 163.6 KB at 64 cameras: the static __shared__ words still fit below 160 KiB
}

// What one of the diagonal workgroup's six update waves (wk = 0 .. 5) does in panel p >= 1 while wave 0 factors block p.
//
// The rows of block p + 1 come from their row workgroup WITHOUT their newest 32 columns: that workgroup hands the block over
// after the update of its panel p - 1, before T(p - 1) has reached it — the updated, unsolved block Ahat(p+1, p-1) (ah) and
// the look-ahead sum through panel p - 2 — and X(p+1, p-1) = Ahat T(p-1)' is formed HERE, with the T(p-1) this workgroup kept
// in LDS (tprev).  (It used to wait for the row workgroup's X: T(p-1) out, X back, two store acknowledgements, two flag polls
// and two loads, 8 us after T(p-1) was published.)
//
// A compute unit moves only ~20 GB/s of freshly handed-over data (its own memory queue, ~2 us per round trip beside the Schur
// kernel), so NOTHING is loaded twice: the strip of block p — the B operand, in LDS as the image of the block's 32 rows, SLD
// doubles apart, read back with ds_read_b128 — is what this routine held in registers as the A operand one panel ago (the
// rows of block p + 1 become the strip of panel p + 1): after barrier [A], when the old strip is dead, every wave writes its
// slabs into the image.  The newest slab comes from xprev, the one after that (X(p+1, p)) from the panel's tail.
//
//   1. wave 0 polls the block's flag and the camera group of its own columns of S; the six meet at the LDS counter;
//   2. ONE round trip: for waves 0, 1 their 16 rows of Ahat and of S (slice 0's panel rows start as S'), everybody's share of the
//      next diagonal block (pre_n = S'(p+1, p+1) - look-ahead sum as handed over), then this wave's slabs of the rows of block
//      p + 1 (half h, K slice [sa, sb) of whole 32-column slabs, <= 4; lane (mi, kk): row mi, columns 8 kk ..);
//   3. waves 0, 1: X(p+1, p-1) -> xprev (LDS; the workgroup's wave 4 sends it to memory);
//   4. the unit: acc = A[rows, slabs] B', the older slabs as they arrive; the six meet again, then the newest slab (p - 1) from
//      xprev; waves 2, 3, 4 also take one tile each of X X' of the newest panel (what the look-ahead sum still lacks)
//      off pre_n; the three slices then subtract their products from the panel rows (out, stride 33) one after the other;
//   5. barrier [A] of the whole workgroup (s_barrier: the other two waves are at theirs), then the slabs into the strip image.
//
// Out of line and static: its own register allocation instead of pushing the kernel's state into scratch (the reloads would land
// in the tail of every panel), and with internal linkage nothing is saved for the caller as long as it stays within the
// caller-saved registers.  *ok_lds = 0 if a flag did not come.
typedef __attribute__((address_space(3))) int lds_int;
// The routine's constants, in LDS (set once per launch): as arguments they would not fit the argument registers, and the
// ones passed on the stack cost a scratch round trip at every entry.
struct DiagConst {
  const double* A; const double* S; const double* ah; const double* dg; const double* diag_u; const double* gc; const double* corr;
  const int* rows_ready; const int* error; const int* gate_ready;
  long long budget, gate_budget;
  double min_diag, max_diag, inv_radius;
  int n, nreal, ld, SLD, tag, gate_tag, gate_cols, gated;   // ld: columns of S in memory (> nreal when the last camera group is a border, ba_cholesky_border.hpp)
  int trs;  // 1: entry (i, j) of S is read as S[j][i] (multi-GPU pipeline: only camera group g's ROW slab is all-reduced when its panels start)
  lds_double* Bst; lds_double* t_tile[2]; lds_double* xprev; lds_double* scl;
  lds_int* s_wb; lds_int* ok_lds;
  lds_int* acq;   // [0]: the newest block some update wave has claimed to acquire for the workgroup, [1]: the newest one acquired,
                  // [2]: 1 when the stage of that block's own columns of S was published by then (DiagUpdateWave, step 1)
  lds_double* snext;   // 32 x 32: the next diagonal block's scaled, damped entries of S on their way into pre_n
  long long* tr;
};
typedef __attribute__((address_space(3))) const DiagConst lds_DiagConst;
static __device__ __noinline__ void DiagUpdateWave(lds_DiagConst* dc, int kb, int wk, int sa, int sb, int p, int do_unit, lds_double* out, lds_double* pre_n) {
  const double* __restrict__ A = dc->A;
  const double* __restrict__ S = dc->S;
  const int n = dc->n, nreal = dc->nreal, ld = dc->ld, SLD = dc->SLD, tag = dc->tag;
  lds_double* Bst = dc->Bst; const lds_double* tprev = dc->t_tile[(p - 1) & 1]; lds_double* xprev = dc->xprev; const lds_double* scl = dc->scl;   // (T(p-1): the factorisation alternates between two tiles)
  lds_int* s_wb = dc->s_wb; lds_int* ok_lds = dc->ok_lds;
  const int wb_target = 18 * (p - 1) + 6;
  long long* tr = dc->tr ? dc->tr + (size_t)p * 8 : nullptr;   // (workgroup 0's stamps start at mtrace)
  const int lane = threadIdx.x & 63, mi = lane & 15, kk = lane >> 4, h = wk & 1;
  const int nb0 = kb + RSBA_PB;   // block p + 1's first row
  typedef double d2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) d2_t lds_d2;
  auto meet = [&](int target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
      __hip_atomic_fetch_add(s_wb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(s_wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  // 1. the block's flag (at the last panel: the right-hand-side row's, which still comes solved), and the camera group of block
  // p + 1's own columns of S (a stage of the Schur kernel; the row workgroups wait for it too).  Every wave looks for itself:
  // normally both are up and nobody waits for anybody.
  // One wave acquires for the workgroup (caches are per CU and per XCD: WaitFlagWG, ba_cholesky_multi.hpp): the first to get
  // here claims block p + 1, looks for the flags and issues the one agent-scope acquire; the others wait for it in LDS.
  const bool gate_panel = do_unit && dc->gated && nb0 % dc->gate_cols == 0;
  if (lane == 0) {
    const int want = p + 1;
    if (__hip_atomic_fetch_max(dc->acq, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(dc->rows_ready + p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
        __builtin_amdgcn_s_sleep(2);
        if (__hip_atomic_load(dc->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > dc->budget) { __hip_atomic_store(ok_lds, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
      }
      // the stage of block p + 1's own columns of S is NOT waited for here: only the next diagonal block's entries come from
      // it, and they are added to the block last, in the panel's tail — this panel's update needs nothing of that stage.  (Waiting
      // here put the whole update of the panels 2, 5 and 8 behind the stage's flag: 6 - 10 us of the step whichever stage it waits
      // for.)
      int stage_up = 1;
      if (gate_panel) stage_up = __hip_atomic_load(dc->gate_ready + 1 + nb0 / dc->gate_cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == dc->gate_tag ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(dc->acq + 2, stage_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(dc->acq + 1, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      while (__hip_atomic_load(dc->acq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const bool s_now = __hip_atomic_load(dc->acq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;   // (uniform: written before acq[1])
  if (tr && wk == 0 && lane == 0) tr[2] = wall_clock64();
  if (!do_unit) { meet(wb_target); meet(wb_target + 6); meet(wb_target + 12); return; }   // (the last panel: the caller goes on to the right-hand-side row, barrier [A] is his)
  // 2. everything in one round trip
  double pf[4][8], ax[8], sv8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ns[3] = {0.0, 0.0, 0.0}, ndd[3] = {0.0, 0.0, 0.0}, ndu = 0.0;
  const int sr = lane >> 2, sc0 = (lane & 3) * 8, sgi = nb0 + h * 16 + sr;
  if (wk < 2) {
#pragma unroll
    for (int u = 0; u < 8; ++u) ax[u] = dc->ah[(size_t)(p + 1) * 1024 + (16 * wk + mi) * RSBA_PB + 4 * u + kk];   // MFMA A operand: row mi, k = 4 u + kk
  }
  // the next diagonal block: entries e = thread + 384 u of S(p+1, p+1), its damping diagonal, the look-ahead sum
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int e = wk * 64 + lane + u * 384, r = e >> 5, c = e & 31;
    if (e < RSBA_PB * RSBA_PB) {
      if (s_now) {
        if (nb0 + r < nreal && nb0 + c < nreal) ns[u] = __hip_atomic_load(&S[dc->trs ? (size_t)(nb0 + c) * ld + nb0 + r : (size_t)(nb0 + r) * ld + nb0 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r == c && nb0 + r < nreal) ndu = __hip_atomic_load(&dc->diag_u[nb0 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ndd[u] = dc->dg[(size_t)(p + 1) * 1024 + r * 32 + c];
    }
  }
  // slice 0 also brings the rows' own entries of S (off the diagonal: scaled, no damping term)
  if (wk < 2 && sgi < nreal) {
#pragma unroll
    for (int u = 0; u < 8; ++u) if (kb + sc0 + u < nreal) sv8[u] = S[dc->trs ? (size_t)(kb + sc0 + u) * ld + sgi : (size_t)sgi * ld + kb + sc0 + u];
  }
  // (the slabs last: the loads return in order, and X(p+1, p-1) is formed while they are still arriving)
  {
    const double* arow = A + (size_t)(nb0 + h * 16 + mi) * n + 8 * kk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (sa + i < sb && sa + i < p - 1) {
        const double2* pa = reinterpret_cast<const double2*>(arow + (sa + i) * RSBA_PB);
#pragma unroll
        for (int v2 = 0; v2 < 4; ++v2) { const double2 t = pa[v2]; pf[i][2 * v2] = t.x; pf[i][2 * v2 + 1] = t.y; }
      }
    }
  }
  if (wk < 2) {
    // slice 0's rows of the panel start as S'
    const double si = sgi < nreal ? scl[sgi] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) out[sr * RSBA_PLD + sc0 + u] = kb + sc0 + u < nreal ? sv8[u] * (si * scl[kb + sc0 + u]) : 0.0;
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int e = wk * 64 + lane + u * 384, r = e >> 5, c = e & 31, gi = nb0 + r, gj = nb0 + c;
    if (e < RSBA_PB * RSBA_PB) {
      // pre_n = -(look-ahead sum as handed over) now, - X X' of the newest panel (below), - X X' of this one and + the entry of S
      // (dc->snext, or fetched there if the stage is not up yet) in the panel's tail: the same terms in the same order whether the
      // stage was up or not
      pre_n[r * RSBA_PLD + c] = -ndd[u];
      if (s_now) {
        double v = gi == gj ? 1.0 : 0.0;   // padding
        if (gi < nreal && gj < nreal) {
          v = ns[u] * (scl[gi] * scl[gj]);
          if (gi == gj) v += fmin(fmax(scl[gi] * scl[gi] * ndu, dc->min_diag), dc->max_diag) * dc->inv_radius;
        }
        dc->snext[e] = v;
      }
    }
  }
  if (tr && wk == 0 && lane == 0) tr[5] = wall_clock64();
  // 3. X(p+1, p-1) = Ahat T(p-1)', 16 rows per wave
  if (wk < 2) {
    d4_t x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[u], tprev[mi * RSBA_PLD + 4 * u + kk], x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[u], tprev[(16 + mi) * RSBA_PLD + 4 * u + kk], x1, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int lr = 16 * wk + kk + 4 * t;
      xprev[lr * SLD + mi] = x0[t]; xprev[lr * SLD + 16 + mi] = x1[t];   // (wave 4 sends it to memory)
    }
  }
  // 4. the unit: the older slabs as they arrive (the strip is in LDS, nothing else is needed), ...
  d4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  auto slab_mfma = [&](const double (&av)[8], int q0) {
    // B[k][j] = L[kb + j][k]: lane (mi, kk) takes the eight k = q0 + 8 kk + u of strip rows mi and 16 + mi in four 16-byte reads each
    double bb[8];   // (one tile's operands at a time: the routine has to stay within the registers no caller has to save)
#pragma unroll
    for (int u = 0; u < 8; u += 2) { const d2_t t0 = *reinterpret_cast<const lds_d2*>(Bst + (size_t)mi * SLD + 8 * kk + q0 + u); bb[u] = t0[0]; bb[u + 1] = t0[1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bb[u], a0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 8; u += 2) { const d2_t t1 = *reinterpret_cast<const lds_d2*>(Bst + (size_t)(16 + mi) * SLD + 8 * kk + q0 + u); bb[u] = t1[0]; bb[u + 1] = t1[1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bb[u], a1, 0, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) if (sa + i < sb && sa + i < p - 1) slab_mfma(pf[i], (sa + i) * RSBA_PB);
  // ... then, when X(p+1, p-1) is there (the first meeting), the newest one
  meet(wb_target);
  if (tr && wk == 0 && lane == 0) tr[6] = wall_clock64();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (sa + i < sb && sa + i == p - 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) pf[i][u] = xprev[(h * 16 + mi) * SLD + 8 * kk + u];
      slab_mfma(pf[i], (sa + i) * RSBA_PB);
    }
  }
  // X X' of the newest panel — what the look-ahead sum as handed over lacks: tiles (0,0), (1,0), (1,1) by waves 2, 3, 4, taken off
  // pre_n (whose entries everybody wrote before the meeting); tile (1,0) also off its mirror
  if (wk >= 2 && wk <= 4) {
    const int t3 = wk - 2, ti = t3 == 0 ? 0 : 1, tj = t3 == 2 ? 1 : 0;
    d4_t xx = {0, 0, 0, 0};
#pragma unroll
    for (int qs = 0; qs < RSBA_PB; qs += 4)
      xx = __builtin_amdgcn_mfma_f64_16x16x4f64(xprev[(16 * ti + mi) * SLD + qs + kk], xprev[(16 * tj + mi) * SLD + qs + kk], xx, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = 16 * ti + kk + 4 * t, c = 16 * tj + mi;
      pre_n[r * RSBA_PLD + c] -= xx[t];
      if (t3 == 1) pre_n[c * RSBA_PLD + r] -= xx[t];
    }
  }
  // the three slices' products leave one after the other, straight into the panel rows (slice 0 wrote S' there): S' - P0 - P1 - P2
  // in this order, no partial tiles, nothing left to merge in the panel's tail
  const int ks = wk >> 1;
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    if (ks == s3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { out[(kk + 4 * t) * RSBA_PLD + mi] -= a0[t]; out[(kk + 4 * t) * RSBA_PLD + 16 + mi] -= a1[t]; }
    }
    if (s3 < 2) meet(wb_target + 6 * (s3 + 1));
  }
  if (tr && wk == 0 && lane == 0) tr[3] = wall_clock64();
  // 5. [A]; then this wave's slabs of the rows of block p + 1 into the strip image of the next panel
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (sa + i < sb) {
#pragma unroll
      for (int u = 0; u < 8; u += 2) { d2_t t; t[0] = pf[i][u]; t[1] = pf[i][u + 1]; *reinterpret_cast<lds_d2*>(Bst + (size_t)(h * 16 + mi) * SLD + (sa + i) * RSBA_PB + 8 * kk + u) = t; }
    }
  }
}

// A row workgroup's update unit: one 16-row half of block b (at the rows `rows` of the workgroup's panel block, stride 33): its
// columns of panel p (scaled; the right-hand-side row, b = np: s (gc + corr)), minus A[rows, K slice ks of nsplit] Bst' (Bst: the
// strip of block p, k-major, stride 33).  Slice 0 owns the rows, the others leave a 16 x 32 partial tile at pdst.  The whole
// slice (up to four 32-column slabs of 8 doubles per lane) is fetched in ONE round trip, together with the entries of S — two
// slabs at a time and refilled as they were consumed, the update took two to three round trips of 3.5 us beside the Schur
// kernel — and with this thread's share of the STRIP (every wave of the workgroup calls, with or without a unit; the barrier
// behind the strip is inside): as a phase of its own in front of the units the strip cost another round trip.  Out of line and
// static for its registers (see DiagUpdateWave).
static __device__ __noinline__ void RowUpdateHalf(lds_DiagConst* dc, int kb, int do_strip, int has_unit, int b, int half, int ks, int nsplit, lds_double* Bst, lds_double* rows, lds_double* pdst) {
  const double* __restrict__ A = dc->A;
  const double* __restrict__ S = dc->S;
  const int n = dc->n, nreal = dc->nreal, ld = dc->ld;
  const lds_double* scl = dc->scl;
  const int lane = threadIdx.x & 63, mi = lane & 15, kk = lane >> 4;
  const int sr = lane >> 2, sc0 = (lane & 3) * 8;
  const int sgi = b * RSBA_PB + half * 16 + sr;
  // the strip of block p straight into LDS (global_load_lds, no registers; as inline assembly, see DiagUpdateWave... the compiler
  // must not know): the image of its 32 rows, kb + 2 doubles apart — one wave instruction = 64 lanes x 16 bytes = 128 doubles of
  // ONE row; the two doubles of padding shift consecutive rows by 16 bytes, so the ds_read_b128 of the sixteen rows of a B tile
  // do not meet in a bank.  Rows wave, wave + 8, ...  First in the queue: the loads return in order.
  const int sld = kb + 2;
  if (do_strip) {
    for (int r = threadIdx.x >> 6; r < RSBA_PB; r += 8) {
      for (int c0 = 0; c0 < kb; c0 += 128) {
        if (c0 + 2 * lane < kb) {
          const double* gsrc = A + (size_t)(kb + r) * n + c0 + 2 * lane;
          const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(Bst + (size_t)r * sld + c0));
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        }
      }
    }
  }
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (has_unit && ks == 0) {
    if (sgi < nreal) {
#pragma unroll
      for (int u = 0; u < 8; ++u) if (kb + sc0 + u < nreal) v[u] = S[dc->trs ? (size_t)(kb + sc0 + u) * ld + sgi : (size_t)sgi * ld + kb + sc0 + u];
    } else if (sgi == n) {
#pragma unroll
      for (int u = 0; u < 8; ++u) if (kb + sc0 + u < nreal) { v[u] = dc->gc[kb + sc0 + u]; v2[u] = dc->corr[kb + sc0 + u]; }
    }
  }
  const int nq = kb / RSBA_PB, qper = (nq + nsplit - 1) / nsplit;
  const int qa = ks * qper * RSBA_PB, qb = min(kb, (ks + 1) * qper * RSBA_PB);
  const int grow = b * RSBA_PB + half * 16 + mi;
  const bool gl = grow <= n;
  const double* arow = A + (size_t)(gl ? grow : 0) * n + 8 * kk;
  d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  double buf[4][8];
  auto fetch = [&](double (&d)[8], int q) {
    const double2* pa = reinterpret_cast<const double2*>(arow + q);
#pragma unroll
    for (int v4 = 0; v4 < 4; ++v4) { const double2 t = pa[v4]; d[2 * v4] = gl ? t.x : 0.0; d[2 * v4 + 1] = gl ? t.y : 0.0; }
  };
  if (has_unit) {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (qa + i * RSBA_PB < qb) fetch(buf[i], qa + i * RSBA_PB);
  }
  if (do_strip) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the strip has landed (the compiler does not count its loads)
  if (do_strip) __syncthreads();
  if (!has_unit) return;
  if (ks == 0) {
    const double si = sgi < nreal ? scl[sgi] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gj = kb + sc0 + u;
      double val = 0.0;
      if (gj < nreal) val = sgi == n ? scl[gj] * (v[u] + v2[u]) : v[u] * (si * scl[gj]);
      rows[sr * RSBA_PLD + sc0 + u] = val;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (kb == 0) return;
  for (int qg = qa; qg < qb; qg += 4 * RSBA_PB) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q0 = qg + i * RSBA_PB;
      if (q0 < qb) {
        double ac[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ac[u] = buf[i][u];
        if (q0 + 4 * RSBA_PB < qb) fetch(buf[i], q0 + 4 * RSBA_PB);
        {
          // B[k][j] = L[kb + j][k]: lane (mi, kk) takes the eight k = q0 + 8 kk + u of strip rows mi and 16 + mi in four 16-byte reads each
          typedef double d2_t __attribute__((ext_vector_type(2)));
          typedef __attribute__((address_space(3))) const d2_t lds_cd2;
          double bb[8];
#pragma unroll
          for (int u = 0; u < 8; u += 2) { const d2_t t0 = *reinterpret_cast<lds_cd2*>(Bst + (size_t)mi * sld + 8 * kk + q0 + u); bb[u] = t0[0]; bb[u + 1] = t0[1]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], bb[u], acc0, 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 8; u += 2) { const d2_t t1 = *reinterpret_cast<lds_cd2*>(Bst + (size_t)(16 + mi) * sld + 8 * kk + q0 + u); bb[u] = t1[0]; bb[u + 1] = t1[1]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], bb[u], acc1, 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (ks == 0) {
      rows[(kk + 4 * t) * RSBA_PLD + mi] -= acc0[t];
      rows[(kk + 4 * t) * RSBA_PLD + 16 + mi] -= acc1[t];
    } else {
      pdst[(kk + 4 * t) * 32 + mi] = acc0[t];
      pdst[(kk + 4 * t) * 32 + 16 + mi] = acc1[t];
    }
  }
}

// kTr: StageGate::transposed as a compile-time constant (the single-GPU instance must not carry the other one's registers)
template <bool kTr>
__global__ void __launch_bounds__(512)
k_reduced_system_solve_diag(int C, double* __restrict__ red, RedLayout L, double* __restrict__ A, double* __restrict__ scale_c,
                            const double* __restrict__ cam_x, double* __restrict__ cam_c, const double* __restrict__ intr,
                            double* __restrict__ camc_c, double* __restrict__ dcam, const double* __restrict__ gmax_p,
                            double* __restrict__ res, IterParams ip, int* __restrict__ chol_ok, StageGate gate, DiagCholFlags f, int tag,
                            long long* __restrict__ mtrace /* diagnostic: [G][16][8] wall-clock stamps, or nullptr */, AheadSel ahead = AheadSel{},
                            int border_cols = 0 /* > 0: the leading system's columns (a multiple of 96); the last camera group is the border, formed by the launch's last workgroup (ba_cholesky_border.hpp) */) {
  extern __shared__ __attribute__((aligned(16))) double lds[];   // (16: the strip image is read and written 16 bytes at a time)
  const int ld = L.nc;                                          // columns of S in memory
  const int nreal = border_cols > 0 ? border_cols : ld, n = (nreal + RSBA_PB - 1) / RSBA_PB * RSBA_PB;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
  const int G = (int)gridDim.x - (border_cols > 0 ? 1 : 0), w = blockIdx.x;   // G: the diagonal workgroup and the row workgroups
  const bool border_wg = border_cols > 0 && w == G;
  const int np = n / RSBA_PB;              // column panels; blocks 0 .. np (block np: the rhs row alone, workgroup 0's)
  const long long budget = gate.budget > 0 ? gate.budget : RSBA_STALL_TICKS;
  __shared__ int s_ok, s_wb, s_w7ok, s_fdone, s_acq[3];
  int wb_gen = 0;
  // LDS: strip (32 p rows of 33) | this workgroup's blocks of the panel (32 x 33 each) | T | Lt | Xl | invd | scale | Pre | Pre2 | scratch
  const int max_rows = n + RSBA_PB;
  double* T = lds + (size_t)max_rows * RSBA_PLD;
  double* Lt = T + RSBA_PB * RSBA_PLD;
  double* Xl = Lt + RSBA_PB * RSBA_PLD;
  double* invd = Xl + RSBA_PB * RSBA_PLD;
  double* scl = invd + 32;
  double* PreA = scl + n;
  double* PreB = PreA + RSBA_PB * RSBA_PLD;
  double* scratch = PreB + RSBA_PB * RSBA_PLD;     // 1024 doubles
  double* Pre = PreA;        // the diagonal block being factored
  double* PreN = PreB;       // the next one, built during this panel
  if (tid == 0) { s_ok = 1; s_wb = 0; s_w7ok = 1; s_fdone = 0; s_acq[0] = -1; s_acq[1] = -1; s_acq[2] = 1; }
  if (gate.trace && tid == 0 && w == 0) gate.trace[0] = wall_clock64();
  AnnounceResident(gate);
  bool stalled = false;
  if (RSBA_EXP(ahead.dec != nullptr)) {
    // launched ahead: the previous step's decision (see AheadSel).  One lane polls, asleep in between; never hang.
    __shared__ int s_dec_ok;
    if (tid == 0) {
      const long long t0 = wall_clock64();
      int ok = 1;
      while (__hip_atomic_load(ahead.dec + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ahead.seq) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > budget) { ok = 0; break; }
      }
      s_dec_ok = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (s_dec_ok == 0) stalled = true;
    if (__hip_atomic_load(ahead.dec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) {
      const double* t = cam_x; cam_x = cam_c; cam_c = const_cast<double*>(t); camc_c = ahead.camc_x;
      if (w == 0) {   // (see AheadSel: what the candidate will overwrite at the end of this kernel)
        for (int i = tid; i < 6 * C; i += nt) ahead.cam_backup[i] = cam_c[i];
        for (int i = tid; i < C * CC_STRIDE; i += nt) ahead.camc_backup[i] = camc_c[i];
      }
    }
    ip.radius = __hip_atomic_load(ahead.dec + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const double* S = red + L.S();
  const double inv_radius = 1.0 / ip.radius;
  const int mi = lane & 15, kk = lane >> 4;
  if (border_wg) {
    // (its constants at the end of ITS carve of the dynamic LDS: the kernel's static words plus the diagonal workgroup's carve fill the 160 KiB)
    BorderCtx& s_bc = *reinterpret_cast<BorderCtx*>(lds + BorderLdsDoubles(ld) - RSBA_BORDER_CTX_DOUBLES);
    if (tid == 0) {
      s_bc.S = S; s_bc.diag_u = red + L.diagU(); s_bc.gc = red + L.gc(); s_bc.corr = red + L.corr(); s_bc.scal = red + L.scal();
      s_bc.A = A; s_bc.XB = A + (size_t)(n + 2) * n; s_bc.scale_c = scale_c;
      s_bc.gate_ready = gate.ready; s_bc.all_diag = gate.all_diag; s_bc.gate_tag = gate.tag; s_bc.gated = gate.ready != nullptr ? 1 : 0; s_bc.gate_budget = gate.budget;
      s_bc.tdone = f.tdone; s_bc.strip_ready = f.strip_ready; s_bc.rows_ready = f.rows_ready; s_bc.a_done = f.error + 12; s_bc.error = f.error; s_bc.tag = tag; s_bc.budget = budget;
      s_bc.nrow_wgs = G - 1; s_bc.ld = ld; s_bc.nA = nreal; s_bc.nB = ld - nreal; s_bc.B = nreal / RSBA_BW;
      s_bc.min_diag = ip.min_lm_diagonal; s_bc.max_diag = ip.max_lm_diagonal; s_bc.inv_radius = inv_radius; s_bc.first = ip.first; s_bc.jacobi = ip.jacobi_scaling;
      s_bc.C = C; s_bc.cam_x = cam_x; s_bc.cam_c = cam_c; s_bc.intr = intr; s_bc.camc_c = camc_c; s_bc.dcam = dcam; s_bc.gmax_p = gmax_p; s_bc.res = res; s_bc.cam_free = ip.cam_free;
      s_bc.chol_ok = chol_ok; s_bc.done = gate.done; s_bc.trace = gate.trace;
      s_bc.mtrace = mtrace ? mtrace + (size_t)w * 16 * 8 : nullptr;
    }
    __syncthreads();
    BorderWorkgroup((lds_BorderCtx*)&s_bc, (lds_double*)lds);
    return;
  }

  // gated stage by stage: every iteration but a run's first — and the first one too when the Schur kernel ran every self
  // tile ahead of the pair tiles and says so (gate.all_diag)
  const bool staged = gate.ready != nullptr && (!ip.first || gate.all_diag != nullptr);
  // warm the factorisation's code while there is nothing to do (see ba_cholesky_multi.hpp)
  if (staged && w == 0) {
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) { const int r = e >> 5, c = e & 31; Pre[r * RSBA_PLD + c] = r == c ? 1.0 : 0.0; }
    __syncthreads();
    if (wave == 0) (void)DiagFactorInverseCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane);
    __syncthreads();
  }
  // pipelined first iteration: the Jacobi scale needs the whole damping diagonal
  if (gate.ready != nullptr && ip.first && !staged) {
    for (int g = 0; g * gate.cols < nreal; ++g)
      if (!WaitReady(gate.ready + 1 + g, gate.tag, w == 0 ? gate.waited : nullptr, gate.budget)) { stalled = true; break; }
  } else if (staged) {
    if (ip.first && !WaitReady(gate.all_diag, gate.tag, w == 0 ? gate.waited : nullptr, gate.budget)) stalled = true;   // every workgroup: the scales
    if (!stalled && w == 0 && !WaitReady(gate.ready + 1, gate.tag, gate.waited, gate.budget)) stalled = true;   // workgroup 0 starts with S(0, 0)
  }
  if (!stalled) {
    for (int i = tid; i < n; i += nt) {
      double sc = 1.0;
      if (i < nreal) {
        sc = ip.first ? (ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(red[L.diagU() + i])) : 1.0) : scale_c[i];
        if (ip.first && w == 0) scale_c[i] = sc;
      }
      scl[i] = sc;
    }
    if (w == 0 && tid == 0) __hip_atomic_store(chol_ok, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // scaled, damped entry (gi, gj) of the padded system; the rhs row (gi == n) from gc + corr.  raw = S[gi][gj] (see Sat)
  auto sys = [&](int gi, int gj, double raw) {
    if (gi == n) return gj < nreal ? scl[gj] * (red[L.gc() + gj] + red[L.corr() + gj]) : 0.0;
    if (gi >= nreal || gj >= nreal) return gi == gj ? 1.0 : 0.0;   // padding
    double v = raw * (scl[gi] * scl[gj]);
    if (gi == gj) v += fmin(fmax(scl[gi] * scl[gi] * red[L.diagU() + gi], ip.min_lm_diagonal), ip.max_lm_diagonal) * inv_radius;
    return v;
  };
  auto sys_pre = [&](int gi, int gj, double raw, double du) {   // the same with the damping term's diag U prefetched
    if (gi >= nreal || gj >= nreal) return gi == gj ? 1.0 : 0.0;
    double v = raw * (scl[gi] * scl[gj]);
    if (gi == gj) v += fmin(fmax(scl[gi] * scl[gi] * du, ip.min_lm_diagonal), ip.max_lm_diagonal) * inv_radius;
    return v;
  };
  constexpr bool trs = kTr;
  auto Sat = [&](int gi, int gj) { return (gi < nreal && gj < nreal) ? S[trs ? (size_t)gj * ld + gi : (size_t)gi * ld + gj] : 0.0; };
  if (w == 0 && !stalled) {   // the first diagonal block
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) { const int r = e >> 5, c = e & 31; Pre[r * RSBA_PLD + c] = sys(r, c, Sat(r, c)); }
    __syncthreads();
  }
  // barrier of the diagonal workgroup's working waves 1, 2, 3, 5, 6, 7 while wave 0 is in the factorisation (wave 4 shares its
  // SIMD and idles; DiagUpdateWave counts on the same LDS word)
  const bool idle4 = w == 0 && wave == 4;
  auto bar6 = [&]() {
    ++wb_gen;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
      __hip_atomic_fetch_add(&s_wb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(&s_wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (nwave - 2) * wb_gen) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  __shared__ DiagConst s_dc;
  if (tid == 0) {
    s_dc.A = A; s_dc.S = S; s_dc.ah = f.ah; s_dc.dg = f.dg; s_dc.diag_u = red + L.diagU(); s_dc.gc = red + L.gc(); s_dc.corr = red + L.corr();
    s_dc.rows_ready = f.rows_ready; s_dc.error = f.error; s_dc.gate_ready = gate.ready;
    s_dc.budget = budget; s_dc.gate_budget = gate.budget;
    s_dc.min_diag = ip.min_lm_diagonal; s_dc.max_diag = ip.max_lm_diagonal; s_dc.inv_radius = inv_radius;
    s_dc.n = n; s_dc.nreal = nreal; s_dc.ld = ld; s_dc.SLD = n - 30; s_dc.tag = tag; s_dc.gate_tag = gate.tag; s_dc.gate_cols = gate.cols;
    s_dc.gated = staged ? 1 : 0; s_dc.trs = kTr ? 1 : 0;
    s_dc.Bst = (lds_double*)lds; s_dc.t_tile[0] = (lds_double*)T; s_dc.t_tile[1] = (lds_double*)(lds + (size_t)32 * (n - 30) + RSBA_PB * RSBA_PLD); s_dc.xprev = (lds_double*)(lds + (n - 30 - RSBA_PB)); s_dc.scl = (lds_double*)scl;
    s_dc.s_wb = (lds_int*)&s_wb; s_dc.ok_lds = (lds_int*)&s_w7ok; s_dc.acq = (lds_int*)&s_acq[0];
    s_dc.tr = mtrace; s_dc.snext = (lds_double*)scratch;
  }
  __syncthreads();
  // a row workgroup's look-ahead tiles: 4 block slots x 3 tiles x 256 doubles where workgroup 0 keeps Pre / PreN / scratch
  double* dtile = PreA;
  const int gmr = G - 1;

#define RSBA_DC_STAMP(k) do { if (mtrace && tid == 0) mtrace[((size_t)w * 16 + p) * 8 + (k)] = wall_clock64(); } while (0)
  for (int p = 0; p < np && !stalled; ++p) {
    const int kb = p * RSBA_PB;
    RSBA_DC_STAMP(0);
    // LDS below T.  A row workgroup: strip of block p as the image of its rows, 32 x (kb + 2) doubles (RowUpdateHalf), then its
    // blocks of the panel.  The diagonal workgroup: the same image, 32 x SLD doubles (see DiagUpdateWave; the two doubles of padding shift
    // consecutive rows by 16 bytes, so the ds_read_b128 of the sixteen rows of a B tile do not meet in a bank), then the panel
    // block of the next rows and the partial tiles of K slice 2:  32 (n - 30) + 1056 + 1024 <= 33 (n + 32).
    const int SLD = n - 30;
    double* Bst = lds;
    double* Pan = w == 0 ? lds + (size_t)32 * SLD : lds + (size_t)32 * (kb + 2);
    // One 16-row half of a block b in slot j: its columns of the panel (scaled, damped) into Pan, minus A[rows, 0:kb] Bst'
    // over K slice ks of nsplit (slice 0 owns the rows in Pan, the others leave 16 x 32 partial tiles at pdst).
    auto load_update_half = [&](int b, int j, int half, int ks, int nsplit, double* pdst) {
      const int prow = j * RSBA_PB + half * 16;
      const int sr = lane >> 2, sc0 = (lane & 3) * 8;
      const int sgi = b * RSBA_PB + half * 16 + sr;
      double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ks == 0 && sgi < nreal) {
        if (kb + sc0 + 8 <= nreal && !trs) {
          const double2* sp = reinterpret_cast<const double2*>(S + (size_t)sgi * ld + kb + sc0);
#pragma unroll
          for (int u = 0; u < 4; ++u) { const double2 t = sp[u]; v[2 * u] = t.x; v[2 * u + 1] = t.y; }
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = Sat(sgi, kb + sc0 + u);   // the padded last panel
        }
      }
      const int nq = kb / RSBA_PB, qper = (nq + nsplit - 1) / nsplit;
      const int qa = ks * qper * RSBA_PB, qb = min(kb, (ks + 1) * qper * RSBA_PB);
      const int grow = b * RSBA_PB + half * 16 + mi;
      const bool gl = grow <= n;
      const double* arow = A + (size_t)(gl ? grow : 0) * n + 8 * kk;
      d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      double buf[RSBA_DC_UPD_NPF][8];
      auto fetch = [&](double (&d)[8], int q) {
        const double2* pa = reinterpret_cast<const double2*>(arow + q);
#pragma unroll
        for (int v2 = 0; v2 < 4; ++v2) { const double2 t = pa[v2]; d[2 * v2] = gl ? t.x : 0.0; d[2 * v2 + 1] = gl ? t.y : 0.0; }
      };
#pragma unroll
      for (int i = 0; i < RSBA_DC_UPD_NPF; ++i) if (qa + i * RSBA_PB < qb) fetch(buf[i], qa + i * RSBA_PB);
      if (ks == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) Pan[(prow + sr) * RSBA_PLD + sc0 + u] = sgi <= n ? sys(sgi, kb + sc0 + u, v[u]) : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      if (p == 0) return;
      for (int qg = qa; qg < qb; qg += RSBA_DC_UPD_NPF * RSBA_PB) {
#pragma unroll
        for (int i = 0; i < RSBA_DC_UPD_NPF; ++i) {
          const int q0 = qg + i * RSBA_PB;
          if (q0 < qb) {
            double ac[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ac[u] = buf[i][u];
            if (q0 + RSBA_DC_UPD_NPF * RSBA_PB < qb) fetch(buf[i], q0 + RSBA_DC_UPD_NPF * RSBA_PB);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const double b0 = Bst[(q0 + 8 * kk + u) * RSBA_PLD + mi];
              const double b1 = Bst[(q0 + 8 * kk + u) * RSBA_PLD + 16 + mi];
              acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], b0, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[u], b1, acc1, 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (ks == 0) {
          const int r = prow + kk + 4 * t;
          Pan[r * RSBA_PLD + mi] -= acc0[t];
          Pan[r * RSBA_PLD + 16 + mi] -= acc1[t];
        } else {
          pdst[(kk + 4 * t) * 32 + mi] = acc0[t];
          pdst[(kk + 4 * t) * 32 + 16 + mi] = acc1[t];
        }
      }
    };
    // X = Rows T' for one 16-row half in slot j, stored as L (and kept in Xl for the next diagonal block)
    // (keep 1: also into Xl for the next diagonal block; keep 2: also back into the block's panel rows, for X X'; keep 3: into
    //  Xl ONLY — somebody else sends it to memory)
    auto solve_half = [&](int b, int j, int half, int keep) {
      const int prow = j * RSBA_PB + half * 16;
      d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int qs = 0; qs < RSBA_PB; qs += 4) {
        const double a = Pan[(prow + mi) * RSBA_PLD + qs + kk];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[mi * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(16 + mi) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int lr = half * 16 + kk + 4 * tt, grow = b * RSBA_PB + lr;
        if (grow <= n && keep != 3) {
          StoreShared(&A[(size_t)grow * n + kb + mi], acc0[tt]);
          StoreShared(&A[(size_t)grow * n + kb + 16 + mi], acc1[tt]);
        }
        if (keep == 1 || keep == 3) { Xl[lr * RSBA_PLD + mi] = acc0[tt]; Xl[lr * RSBA_PLD + 16 + mi] = acc1[tt]; }
        if (keep == 2) { Pan[(prow + kk + 4 * tt) * RSBA_PLD + mi] = acc0[tt]; Pan[(prow + kk + 4 * tt) * RSBA_PLD + 16 + mi] = acc1[tt]; }
      }
    };

    if (w == 0) {
      // ============================================================ the diagonal workgroup
      const bool has_next = p + 1 < np;
      const int nb0 = kb + RSBA_PB;                 // first row / column of block p + 1
      // X(p+1, p-1), formed during the update, lives in the last 32 columns of the strip image's rows (free while there is a next
      // block: the strip is at most n - 64 columns wide then)
      double* xprev = Bst + (SLD - RSBA_PB);
      const bool unit_wave = wave != 0 && !idle4;
      // T(p) goes into one of two tiles in turn (the second one behind the panel block): T(p-1) is still being used — X(p+1, p-1)
      // is formed with it — while block p is factored
      double* Tc = (p & 1) ? Pan + RSBA_PB * RSBA_PLD : T;
      if (wave == 0) {
        if (!DiagFactorInverseCall((lds_double*)Pre, RSBA_PB, (lds_double*)Tc, (lds_double*)Lt, (lds_double*)invd, lane) && lane == 0) s_ok = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_fdone, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // wave 4 sends T on its way at once
        RSBA_DC_STAMP(1);
      } else if (!idle4) {
        // ---- waves 1, 2, 3, 5, 6, 7: strip of block p, then the columns of panel p in the rows of block p + 1: half h of the
        // rows, K slice ks of three (whole 32-column slabs), see DiagUpdateWave
        const int wk = wave < 4 ? wave - 1 : wave - 2;   // 0 .. 5
        const int h = wk & 1, ks = wk >> 1;
        const int qper = (p + 2) / 3, sa = ks * qper, sb = min(p, (ks + 1) * qper);   // slabs [sa, sb), sb - sa <= 4
        const bool unit = has_next && p > 0;
        if (p > 0) {
          DiagUpdateWave((lds_DiagConst*)&s_dc, kb, wk, sa, sb, p, unit ? 1 : 0,
                         (lds_double*)(Pan + h * 16 * RSBA_PLD), (lds_double*)PreN);
          if (mtrace && tid == 64) mtrace[((size_t)w * 16 + p) * 8 + 3] = wall_clock64();   // wave 1's unit done
          if (!unit && wave == 1) {
            // the last panel: the right-hand-side row alone, rhs[kb + c] - sum_q L[n][q] L[kb + c][q]: one row, so no matrix
            // cores; lane (c, half) adds every second group of eight terms into eight running sums, the halves meet in a shuffle
            const int c = lane & 31, half = lane >> 5;
            const double* arow_n = A + (size_t)n * n;
            double sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int q0 = half * 8; q0 < kb; q0 += 16) {
#pragma unroll
              for (int u = 0; u < 8; ++u) sacc[u] += arow_n[q0 + u] * Bst[(size_t)c * SLD + q0 + u];
            }
            double sum = ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) + ((sacc[4] + sacc[5]) + (sacc[6] + sacc[7]));
            sum += __shfl_xor(sum, 32, 64);
            const double rc = sys(n, kb + c, 0.0);
            for (int e = lane; e < 16 * RSBA_PB; e += 64) Pan[(size_t)(e / RSBA_PB) * RSBA_PLD + (e % RSBA_PB)] = 0.0;   // rows 1 .. 15 feed the X product and are never stored: keep them finite
            __builtin_amdgcn_wave_barrier();
            if (half == 0) Pan[c] = rc - sum;
          }
        } else if (has_next) {
          if (wk < 2) load_update_half(p + 1, 0, wk, 0, 1, nullptr);   // panel 0: nothing to subtract yet
        } else if (wave == 1) {
          load_update_half(np, 0, 0, 0, 1, nullptr);
        }
      }
      if (idle4) {
        // wave 4, the STORE wave (it shares the factoring wave's SIMD and does nothing else): L11 / T as soon as the factorisation
        // is through — the row workgroups' next panel hangs on T(p), and it used to leave only in the panel's tail — their
        // acknowledgements awaited right behind [A]; nobody on the chain waits for a store (a round trip costs 3.5 us beside the
        // Schur kernel)
        if (lane == 0) { while (__hip_atomic_load(&s_fdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < p + 1) __builtin_amdgcn_s_sleep(4); }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 2
        for (int u = 0; u < 16; ++u) {
          const int e = lane + 64 * u, r = e >> 5, c = e & 31;
          StoreShared(&A[(size_t)(kb + r) * n + kb + c], c > r ? Tc[c * RSBA_PLD + r] : Pre[r * RSBA_PLD + c]);
        }
        if (lane < RSBA_PB) StoreShared(&A[(size_t)(n + 1) * n + kb + lane], invd[lane]);
        // ... and published as soon as they are acknowledged — not behind barrier [A], where the row workgroups' T(p) waited for
        // whatever the update waves were waiting for (a block handed over late, a stage), finished their panel late and handed the
        // next block over late in turn
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) __hip_atomic_store(f.tdone + p, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (idle4 && has_next && p > 0) {
        // ... and X(p+1, p-1) as soon as the update waves have formed it (their first meeting)
        if (lane == 0) { while (__hip_atomic_load(&s_wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 18 * (p - 1) + 6) __builtin_amdgcn_s_sleep(2); }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 2
        for (int u = 0; u < 16; ++u) { const int e = lane + 64 * u, r = e >> 5, c = e & 31; StoreShared(&A[(size_t)(nb0 + r) * n + kb - RSBA_PB + c], xprev[(size_t)r * SLD + c]); }
      }
      if (!(unit_wave && has_next && p > 0)) __syncthreads();   // [A] the factor (T, Lt, invd, Pre = L11) and the updates are done (the update waves: inside DiagUpdateWave)
      RSBA_DC_STAMP(4);
      if (!s_ok && tid == 0) __hip_atomic_store(chol_ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (s_w7ok == 0) { stalled = true; break; }
      // ---- the tail: two barriers.  The panel rows of block p + 1 are complete (DiagUpdateWave), PreN holds everything of the
      // next diagonal block but this panel's X X' (panel 0: built here).
      if (has_next && p == 0) {
        for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) { const int r = e >> 5, c = e & 31; PreN[r * RSBA_PLD + c] = sys(nb0 + r, nb0 + c, Sat(nb0 + r, nb0 + c)); }
        __syncthreads();
      }
      // X = Rows T': block p + 1 (two halves, kept in Xl), or the rhs row at the last panel; meanwhile wave 4 waits for the
      // acknowledgements of L11 / T (on their way since the factorisation ended) and publishes them
      {
        double* Tsave = T; T = Tc;   // (solve_half reads T)
        if (has_next) { if (wave == 1 || wave == 2) solve_half(p + 1, 0, wave - 1, 3); }
        else if (wave == 3) solve_half(np, 0, 0, 0);
        T = Tsave;
      }
      __syncthreads();   // [C] Xl
      if (has_next) {
        // X X' straight off PreN (tiles by waves 1, 2, 3, 5); X(p+1, p) into the next strip's image, behind the columns the update
        // waves left there (waves 0, 6, 7), and to memory (wave 4)
        if (wave == 0 || wave >= 6) {
          for (int e = (wave == 0 ? 0 : wave - 5) * 64 + lane; e < RSBA_PB * RSBA_PB; e += 192) { const int r = e >> 5, c = e & 31; Bst[(size_t)r * SLD + kb + c] = Xl[r * RSBA_PLD + c]; }
        } else if (idle4) {
#pragma unroll 2
          for (int u = 0; u < 16; ++u) { const int e = lane + 64 * u, r = e >> 5, c = e & 31; StoreShared(&A[(size_t)(nb0 + r) * n + kb + c], Xl[r * RSBA_PLD + c]); }
        } else {
          const int xt = wave < 4 ? wave - 1 : 3;
          const int ti = xt >> 1, tj = xt & 1;
          d4_t acc = {0, 0, 0, 0};
#pragma unroll
          for (int qs = 0; qs < RSBA_PB; qs += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xl[(16 * ti + mi) * RSBA_PLD + qs + kk], Xl[(16 * tj + mi) * RSBA_PLD + qs + kk], acc, 0, 0, 0);
          // ... and, last, the block's own (scaled, damped) entries of S: DiagUpdateWave left them in `scratch` if their stage of the
          // Schur kernel was published when the panel began; if not, THIS is where the factorisation waits for the stage — behind
          // everything of panel p, which needs nothing of it (agent-scope loads: no fence, nothing else is read behind this flag)
          double vv[4] = {0.0, 0.0, 0.0, 0.0};
          if (p > 0) {
            if (s_acq[2] != 0) {
#pragma unroll
              for (int t = 0; t < 4; ++t) vv[t] = scratch[(16 * ti + kk + 4 * t) * RSBA_PB + 16 * tj + mi];
            } else {
              if (lane == 0) {
                const long long t1 = wall_clock64();
                while (__hip_atomic_load(gate.ready + 1 + nb0 / gate.cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gate.tag) {
                  __builtin_amdgcn_s_sleep(4);
                  if (wall_clock64() - t1 > (gate.budget > 0 ? gate.budget : RSBA_STALL_TICKS)) { __hip_atomic_store(&s_w7ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                }
              }
              __builtin_amdgcn_wave_barrier();
              double raw[4], du[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int gi = nb0 + 16 * ti + kk + 4 * t, gj = nb0 + 16 * tj + mi;
                const bool in = gi < nreal && gj < nreal;
                raw[t] = in ? __hip_atomic_load(&S[trs ? (size_t)gj * ld + gi : (size_t)gi * ld + gj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                du[t] = in && gi == gj ? __hip_atomic_load(&red[L.diagU() + gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
              }
#pragma unroll
              for (int t = 0; t < 4; ++t) vv[t] = sys_pre(nb0 + 16 * ti + kk + 4 * t, nb0 + 16 * tj + mi, raw[t], du[t]);
            }
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            double* e = &PreN[(16 * ti + kk + 4 * t) * RSBA_PLD + 16 * tj + mi];
            *e = p > 0 ? (*e - acc[t]) + vv[t] : *e - acc[t];
          }
        }
        __syncthreads();   // [E] the next diagonal block is ready: wave 0 goes on
        if (s_w7ok == 0) { stalled = true; break; }   // (a stage that did not come)
      }
      // X(p+1, p) is on its way to memory: wave 4 waits for the acknowledgements and tells the row workgroups;
      // nobody else does (this workgroup's next strip does not come from those stores)
      if (idle4) {
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0 && has_next) __hip_atomic_store(f.strip_ready + p + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      { double* t = Pre; Pre = PreN; PreN = t; }
      RSBA_DC_STAMP(7);
    } else {
      // ============================================================ a row workgroup: blocks b >= p + 2, b mod (G - 1) == w - 1,
      // the right-hand-side row counting as block np (its first half only)
      const int gm = gmr;
      int first = p + 2;
      first += ((w - 1) - first % gm + gm) % gm;
      const int nfull = first >= np ? 0 : (np - 1 - first) / gm + 1;     // whole blocks
      const bool rhs_mine = np % gm == w - 1 && p + 2 <= np;             // block np: panels p <= np - 2
      const int nh = 2 * nfull + (rhs_mine ? 1 : 0);                     // 16-row halves to bring up to date and solve
      if (nh == 0) continue;
      auto blk = [&](int j) { return j < nfull ? first + j * gm : np; };
      if (p > 0) {
        if (!WaitFlagWG(f.strip_ready + p, tag, f.error, budget)) { stalled = true; break; }
        RSBA_DC_STAMP(2);
      } else {
        // the look-ahead sums start at zero
        for (int e = tid; e < 12 * 256; e += nt) dtile[e] = 0.0;
      }
      RSBA_DC_STAMP(3);
      if (staged && kb % gate.cols == 0) {
        // (the stage's flag, then ONE acquire by the wavefront that saw it — WaitFlagWG; WaitReady's __threadfence() is a write-back
        //  and an invalidate of the L2 in each of the eight wavefronts: the panels 3, 6 and 9 of the row workgroups took 4 - 7 us
        //  longer than the others, and the blocks they hand over were late)
        if (!WaitFlagWG(gate.ready + 1 + kb / gate.cols, gate.tag, f.error, gate.budget > 0 ? gate.budget : RSBA_STALL_TICKS)) { stalled = true; break; }
      }
      {
        const int nsplit = (p > 0 && nh <= 2) ? 4 : ((p > 0 && nh <= 4) ? 2 : 1);
        double* part = T;   // T | Lt | Xl are idle before T(p) arrives: up to six 16 x 32 partial tiles
        // (normally at most eight units, one per wave; every wave calls once with the strip's load and its barrier in there)
        for (int it = wave, first_call = 1; first_call || it < nh * nsplit; it += nwave, first_call = 0) {
          const int has_unit = it < nh * nsplit ? 1 : 0;
          const int hb = has_unit ? it / nsplit : 0, ks = has_unit ? it - hb * nsplit : 0;
          RowUpdateHalf((lds_DiagConst*)&s_dc, kb, first_call, has_unit, blk(hb >> 1), hb & 1, ks, nsplit, (lds_double*)Bst, (lds_double*)(Pan + ((hb >> 1) * RSBA_PB + (hb & 1) * 16) * RSBA_PLD),
                        (lds_double*)(part + (hb * (nsplit - 1) + ks - 1) * 512));
        }
        __syncthreads();
        if (nsplit > 1) {
          for (int e = tid; e < nh * 512; e += nt) {
            const int hb = e >> 9, r = (e >> 5) & 15, c = e & 31;
            double sum = part[(hb * (nsplit - 1)) * 512 + r * 32 + c];
            for (int k2 = 1; k2 < nsplit - 1; ++k2) sum += part[(hb * (nsplit - 1) + k2) * 512 + r * 32 + c];
            Pan[((hb >> 1) * RSBA_PB + (hb & 1) * 16 + r) * RSBA_PLD + c] -= sum;
          }
          __syncthreads();
        }
      }
      RSBA_DC_STAMP(4);
      // Block p + 2 leaves this workgroup NOW, before T(p) is here: updated through this panel, unsolved (slot 0 of the panel
      // rows), with its look-ahead sum through panel p - 1.  The diagonal workgroup solves it with its own T(p) (see
      // DiagUpdateWave); the rows' columns < 32 p have been in memory since the last panel.
      const bool lv = nfull > 0 && first == p + 2;
      if (lv && wave == nwave - 1) {
        // (one wave stores, waits for the acknowledgements and publishes; the others go on to T(p) — the wait for the stores would
        //  cost everybody a round trip, and the last wave never has a half to solve in a panel that hands a block over)
#pragma unroll 2
        for (int u = 0; u < 16; ++u) { const int e = lane + 64 * u; StoreShared(&f.ah[(size_t)(p + 2) * 1024 + e], Pan[(e >> 5) * RSBA_PLD + (e & 31)]); }
        const double* dt = dtile + (((p + 2) / gm) & 3) * 3 * 256;
#pragma unroll 2
        for (int u = 0; u < 12; ++u) {
          const int e = lane + 64 * u, t3 = e >> 8, r = (e >> 4) & 15, c = e & 15, ti = t3 == 0 ? 0 : 1, tj = t3 == 2 ? 1 : 0;
          StoreShared(&f.dg[(size_t)(p + 2) * 1024 + (16 * ti + r) * 32 + 16 * tj + c], dt[e]);
          if (t3 == 1) StoreShared(&f.dg[(size_t)(p + 2) * 1024 + c * 32 + 16 + r], dt[e]);   // the mirror of tile (1, 0): the reader takes whole rows
        }
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) __hip_atomic_store(f.rows_ready + p + 2, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (nh == (lv ? 2 : 0)) {   // nothing left to solve in this panel
        if (tid == 0) __hip_atomic_store(f.error + 4 + w, (tag << 4) | (p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      if (!WaitFlagWG(f.tdone + p, tag, f.error, budget)) { stalled = true; break; }
      {
        // (read along the rows of the stored T': whole cache lines; transposed on the way into LDS)
        double tv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + u * nt, i = e >> 5, j = e & 31;
          tv[u] = j > i ? A[(size_t)(kb + i) * n + kb + j] : (j == i ? A[(size_t)(n + 1) * n + kb + i] : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + u * nt, i = e >> 5, j = e & 31;
          if (j >= i) { T[j * RSBA_PLD + i] = tv[u]; if (j > i) T[i * RSBA_PLD + j] = 0.0; }
        }
      }
      __syncthreads();
      RSBA_DC_STAMP(5);
      for (int hb = wave + (lv ? 2 : 0); hb < nh; hb += nwave) solve_half(blk(hb >> 1), hb >> 1, hb & 1, (hb >> 1) < nfull ? 2 : 0);
      RSBA_DC_STAMP(6);
      // X(b, p) is on its way to memory and sits in the blocks' panel rows: the look-ahead sums take their X X'
      __syncthreads();
      for (int it = wave + (lv ? 3 : 0); it < 3 * nfull; it += nwave) {
        const int j = it / 3, t3 = it - 3 * j, ti = t3 == 0 ? 0 : 1, tj = t3 == 2 ? 1 : 0;
        double* dt = dtile + (((blk(j) / gm) & 3) * 3 + t3) * 256;
        d4_t acc;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = dt[(kk + 4 * t) * 16 + mi];
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pan[(j * RSBA_PB + 16 * ti + mi) * RSBA_PLD + qs + kk], Pan[(j * RSBA_PB + 16 * tj + mi) * RSBA_PLD + qs + kk], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) dt[(kk + 4 * t) * 16 + mi] = acc[t];
      }
      // the right-hand-side row (block np) leaves after panel np - 2, solved: rows_ready[np]
      if (rhs_mine && p + 2 == np) PublishFlagWG(f.rows_ready + np, tag);
      else { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
      // progress: every X(., p) of this workgroup's blocks is in memory
      if (tid == 0) __hip_atomic_store(f.error + 4 + w, (tag << 4) | (p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      RSBA_DC_STAMP(7);
    }
  }

  if (stalled) {
    // (with a border the last workgroup ends the solve: it sees the error flag in whatever it waits for next)
    if (tid == 0) { __hip_atomic_store(f.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (w == 0 && border_cols == 0) res[RES_STALL] = 1.0; }
    if (w == 0 && border_cols == 0) SolveDone(gate);
    return;
  }
  if (w != 0) return;
  if (border_cols > 0) {
    // the leading system is factored and y_A = L_A^-1 b_A lies in its right-hand-side row: over to the border's workgroup
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(f.error + 12, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // workgroup 0: every block's rows were handed over before its diagonal panel, so L is complete; L' x = y and the camera step
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (wave 0 has not acquired the row workgroups' stores yet)
  if (gate.trace && tid == 0) gate.trace[13] = wall_clock64();
  // What the camera step needs from memory besides the solution — the Jacobi scale, x, the camera gradient, the intrinsics, the
  // Schur kernel's scalars — is asked for HERE, ahead of the back-substitution (25 us), not behind it: CameraStepEpilogue's
  // arithmetic (ba_point_kernels.hpp) on y where it lies in LDS, with the candidate cameras handed to the lanes that form their
  // constants through LDS instead of a store and a load (the step's tail lost three dependent trips to memory: ~6 -> ~2.5 us).
  // nreal <= nt: one entry per thread.
  const bool e_on = tid < nreal;
  double pf_scale = 0.0, pf_x = 0.0, pf_g = 0.0, pf_free = 1.0, pf_in[4] = {0.0, 0.0, 0.0, 0.0}, pf_s[4] = {0.0, 0.0, 0.0, 0.0};
  if (e_on) {
    pf_scale = scale_c[tid]; pf_x = cam_x[tid]; pf_g = red[L.gc() + tid];
    if (ip.cam_free != nullptr) pf_free = ip.cam_free[tid / 6];
  }
  if (tid < C) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pf_in[q] = intr[4 * tid + q];
  }
  if (tid == 0) { pf_s[0] = red[L.scal() + 0]; pf_s[1] = red[L.scal() + 1]; pf_s[2] = red[L.scal() + 2]; pf_s[3] = *gmax_p; }
  double* y = BackSubstituteBlocksWaves(n, A, lds);
  int ok = 1;
  if (tid == 0) { res[RES_STALL] = 0.0; ok = __hip_atomic_load(chol_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (gate.trace && tid == 0) gate.trace[14] = wall_clock64();
  {
    double* scr = lds + 2048;          // 4 x nt (the back-substitution's tables end below 2048 doubles; y stays at lds[0 .. n))
    double* s_xc = lds + 2048 + 4 * 512;   // the candidate cameras
    double d2 = 0.0, x2 = 0.0, xc2 = 0.0, gm = 0.0;
    if (e_on) {
      const double d = -pf_scale * y[tid];
      dcam[tid] = d;
      const double xc = pf_x + d;
      cam_c[tid] = xc;
      s_xc[tid] = xc;
      if (pf_free != 0.0) { d2 += d * d; x2 += pf_x * pf_x; xc2 += xc * xc; }
      gm = fmax(gm, fabs(pf_g));
    }
    // the norms: lanes by butterfly, the wavefronts in order — one barrier (CameraStepEpilogue's tree over the workgroup: nine)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      d2 += __shfl_xor(d2, off, 64); x2 += __shfl_xor(x2, off, 64); xc2 += __shfl_xor(xc2, off, 64); gm = fmax(gm, __shfl_xor(gm, off, 64));
    }
    if (lane == 0) { scr[4 * wave + 0] = d2; scr[4 * wave + 1] = x2; scr[4 * wave + 2] = xc2; scr[4 * wave + 3] = gm; }
    __syncthreads();
    if (tid < C) {
      double cc[CC_STRIDE];
      CameraConstants(s_xc + 6 * tid, pf_in, cc);
      typedef double d2s_t __attribute__((ext_vector_type(2)));
      d2s_t* out2 = reinterpret_cast<d2s_t*>(camc_c + (size_t)tid * CC_STRIDE);
#pragma unroll
      for (int i = 0; i < CC_STRIDE / 2; ++i) { d2s_t v = {cc[2 * i], cc[2 * i + 1]}; out2[i] = v; }
    }
    if (tid == 0) {
      double t4[4] = {scr[0], scr[1], scr[2], scr[3]};
      for (int w8 = 1; w8 < nwave; ++w8) { t4[0] += scr[4 * w8]; t4[1] += scr[4 * w8 + 1]; t4[2] += scr[4 * w8 + 2]; t4[3] = fmax(t4[3], scr[4 * w8 + 3]); }
      res[RES_COST_X] = 0.5 * pf_s[0];
      res[RES_GMAX] = fmax(pf_s[3], t4[3]);
      res[RES_XNORM2] = pf_s[1] + t4[1];
      res[RES_POINT_FAIL] = pf_s[2];
      res[RES_CHOL_OK] = (ok && pf_s[2] == 0.0) ? 1.0 : 0.0;
      res[RES_STEP2] = t4[0];
      res[RES_XCNORM2] = t4[2];
    }
  }
  if (gate.trace && tid == 0) gate.trace[15] = wall_clock64();
  SolveDone(gate);
}

}  // namespace rsba
