// The reduced camera system on several workgroups with the DIAGONAL CHAIN IN ONE WORKGROUP (32 to 64 cameras).
//
// ba_cholesky_multi.hpp deals the 32-row blocks round-robin, so the chain factor(p) -> T(p) -> X(p+1, p) -> X X' ->
// factor(p+1) crosses from one workgroup to the next at EVERY panel: store, acknowledgement, flag, poll, load — ~2.6 us of
// the ~16 us a panel took, and the next owner's row update had to be finished by then as well.  Here
//
//   * workgroup 0 owns the diagonal: at panel p its wave 0 factors block p (DiagFactorInverse) while its other seven waves
//     bring block p + 1 up to date (the update of panel p's columns fused with the look-ahead product, as in the multi
//     kernel) and the right-hand-side row; then X(p+1, p) = Rows T', X X', and wave 0 goes on with block p + 1.  T never
//     leaves the workgroup on the critical path: a panel costs factor + X + X X' + four barriers.
//   * workgroups 1 .. G-1 own the rows BELOW: block b belongs to workgroup 1 + b mod (G - 1) for the panels p <= b - 2;
//     they receive the strip of block p and T(p) through global memory (flags strip_ready / tdone, as before) one hop behind
//     the diagonal, store their X, and hand block b over (flag rows_ready[b]) after panel b - 2; nothing waits for them but
//     the last slabs of workgroup 0's update one panel later.
//
// STATUS (round 2): the default for 32 to 64 cameras (RSBA_CHOL_DIAG=0 selects the round-robin kernel).  Measured at 64
// cameras beside the Schur kernel: 0.466 ms per LM iteration against 0.477 ms round-robin, 206 against 221 us alone
// (sequential schedule).  What it took after the first version (0.493 ms): T and L11 stored by ONE wave right after the
// factorisation and published before the X barrier (the row workgroups start 3 us earlier; publishing at the end of the
// panel instead gives the gain back: 0.475 ms), the right-hand-side row brought up to date with plain dot products from a
// prefetched row instead of a ninth MFMA tile, and the fused update's operand loads issued before its LDS stores.  A panel
// of the chain is now factor (8.5 - 10 us) + 3.2 us for X, X X' and four barriers; the seven other waves' strip, wait for
// rows_ready and fused update fit behind the factorisation for all but the first two panels.
//
// Same arithmetic per entry as the multi kernel's (products over fixed K slices, added in a fixed order): bitwise
// reproducible, identical on every rank.  All waits carry a budget: a stall gives up (RES_STALL), never hangs.
#pragma once
#include "ba_cholesky_multi.hpp"

namespace rsba {

struct DiagCholFlags {
  int* tdone;         // [16]  == tag when panel p's L11 / T are in global memory
  int* strip_ready;   // [16]  == tag when the rows of block p hold L for all columns < 32 p          (workgroup 0)
  int* rows_ready;    // [16]  == tag when the rows of block b hold L for all columns < 32 (b - 1)    (its row workgroup)
  int* error;         // != 0: somebody gave up waiting
};

#ifndef RSBA_DC_NPF
#define RSBA_DC_NPF 1   // 32-column slabs per operand stream in flight in the cross-tile waves (twice as many in the row waves)
#endif
#ifndef RSBA_DC_UPD_NPF
#define RSBA_DC_UPD_NPF 2   // ... in the plain row update (row workgroups, right-hand-side row)
#endif

__host__ __device__ inline size_t DiagCholLdsDoubles(int nc) {
  const int n = MultiCholPadded(nc);
  return (size_t)(n + RSBA_PB) * RSBA_PLD + 5 * RSBA_PB * RSBA_PLD + 32 + n + 1024;   // 163.6 KB at 64 cameras: the static __shared__ words still fit below 160 KiB
}

__global__ void __launch_bounds__(512)
k_reduced_system_solve_diag(int C, double* __restrict__ red, RedLayout L, double* __restrict__ A, double* __restrict__ scale_c,
                            const double* __restrict__ cam_x, double* __restrict__ cam_c, const double* __restrict__ intr,
                            double* __restrict__ camc_c, double* __restrict__ dcam, const double* __restrict__ gmax_p,
                            double* __restrict__ res, IterParams ip, int* __restrict__ chol_ok, StageGate gate, DiagCholFlags f, int tag,
                            long long* __restrict__ mtrace /* diagnostic: [G][16][8] wall-clock stamps, or nullptr */) {
  extern __shared__ double lds[];
  const int nreal = L.nc, n = (nreal + RSBA_PB - 1) / RSBA_PB * RSBA_PB;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
  const int G = gridDim.x, w = blockIdx.x;
  const int np = n / RSBA_PB;              // column panels; blocks 0 .. np (block np: the rhs row alone, workgroup 0's)
  const long long budget = gate.budget > 0 ? gate.budget : RSBA_STALL_TICKS;
  __shared__ int s_ok, s_wb, s_w7ok;
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
  if (tid == 0) { s_ok = 1; s_wb = 0; s_w7ok = 1; }
  if (gate.trace && tid == 0 && w == 0) gate.trace[0] = wall_clock64();
  bool stalled = false;
  const double* S = red + L.S();
  const double inv_radius = 1.0 / ip.radius;
  const int mi = lane & 15, kk = lane >> 4;

  // warm the factorisation's code while there is nothing to do (see ba_cholesky_multi.hpp)
  if (gate.ready != nullptr && !ip.first && w == 0) {
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) { const int r = e >> 5, c = e & 31; Pre[r * RSBA_PLD + c] = r == c ? 1.0 : 0.0; }
    __syncthreads();
    if (wave == 0) (void)DiagFactorInverseCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane);
    __syncthreads();
  }
  // pipelined first iteration: the Jacobi scale needs the whole damping diagonal
  if (gate.ready != nullptr && ip.first) {
    for (int g = 0; g * gate.cols < nreal; ++g)
      if (!WaitReady(gate.ready + 1 + g, gate.tag, w == 0 ? gate.waited : nullptr, gate.budget)) { stalled = true; break; }
  } else if (gate.ready != nullptr && w == 0) {
    if (!WaitReady(gate.ready + 1, gate.tag, gate.waited, gate.budget)) stalled = true;   // workgroup 0 starts with S(0, 0)
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
  auto Sat = [&](int gi, int gj) { return (gi < nreal && gj < nreal) ? S[(size_t)gi * nreal + gj] : 0.0; };
  if (w == 0 && !stalled) {   // the first diagonal block
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) { const int r = e >> 5, c = e & 31; Pre[r * RSBA_PLD + c] = sys(r, c, Sat(r, c)); }
    __syncthreads();
  }
  // barrier of waves 1 .. 7 (wave 0 is in the factorisation)
  auto bar7 = [&]() {
    ++wb_gen;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
      __hip_atomic_fetch_add(&s_wb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(&s_wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (nwave - 1) * wb_gen) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };

#define RSBA_DC_STAMP(k) do { if (mtrace && tid == 0) mtrace[((size_t)w * 16 + p) * 8 + (k)] = wall_clock64(); } while (0)
  for (int p = 0; p < np && !stalled; ++p) {
    const int kb = p * RSBA_PB;
    RSBA_DC_STAMP(0);
    double* Bst = lds;
    double* Pan = lds + (size_t)kb * RSBA_PLD;
    // One 16-row half of a block b in slot j: its columns of the panel (scaled, damped) into Pan, minus A[rows, 0:kb] Bst'
    // over K slice ks of nsplit (slice 0 owns the rows in Pan, the others leave 16 x 32 partial tiles at pdst).
    auto load_update_half = [&](int b, int j, int half, int ks, int nsplit, double* pdst) {
      const int prow = j * RSBA_PB + half * 16;
      const int sr = lane >> 2, sc0 = (lane & 3) * 8;
      const int sgi = b * RSBA_PB + half * 16 + sr;
      double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ks == 0 && sgi < nreal) {
        if (kb + sc0 + 8 <= nreal) {
          const double2* sp = reinterpret_cast<const double2*>(S + (size_t)sgi * nreal + kb + sc0);
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
    // The right-hand-side row's columns of the panel (one wave): rhs[kb + c] - sum_q L[n][q] L[kb + c][q], c < 32 — one row,
    // so no matrix cores: the row's kb entries come into LDS in one round trip (the unused rows of its panel block), lane
    // (c, half) adds every second term, the two halves meet in a shuffle.  Fixed order.  As an MFMA half streamed over the
    // whole K range by one wave it was the longest item of the diagonal workgroup's panel (13 us at kb = 288).
    double rhs_av[6] = {0, 0, 0, 0, 0, 0}, rhs_c = 0.0;   // one wave's share of the row and of the right-hand side, fetched early
    auto rhs_row_prefetch = [&]() {
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int q = lane + 64 * u; rhs_av[u] = q < kb ? A[(size_t)n * n + q] : 0.0; }
      rhs_c = sys(n, kb + (lane & 31), 0.0);
    };
    auto rhs_row_update = [&](int j) {
      double* arow_l = Pan + (size_t)(j * RSBA_PB + 1) * RSBA_PLD;   // rows 1.. of the block: 31 x 33 >= kb doubles
      const int c = lane & 31, half = lane >> 5;
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int q = lane + 64 * u; if (q < kb) arow_l[q] = rhs_av[u]; }
      __builtin_amdgcn_wave_barrier();
      // lane (c, half) adds the terms q = half, half + 2, ...: eight running sums (q mod 16), so that the LDS reads of
      // one round do not wait for the additions of the last (kb is a multiple of 32)
      double sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int q0 = half; q0 < kb; q0 += 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc[u] += arow_l[q0 + 2 * u] * Bst[(q0 + 2 * u) * RSBA_PLD + c];
      }
      double sum = ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) + ((sacc[4] + sacc[5]) + (sacc[6] + sacc[7]));
      sum += __shfl_xor(sum, 32, 64);
      __builtin_amdgcn_wave_barrier();
      if (half == 0) Pan[(size_t)(j * RSBA_PB) * RSBA_PLD + c] = rhs_c - sum;
      // rows 1 .. 15 of the half feed the X product row by row and are never stored: whatever they hold is harmless, but
      // keep them finite
      for (int e = lane; e < 15 * RSBA_PB; e += 64) Pan[(size_t)(j * RSBA_PB + 1 + e / RSBA_PB) * RSBA_PLD + (e % RSBA_PB)] = 0.0;
    };
    // X = Rows T' for one 16-row half in slot j, stored as L (and kept in Xl for the next diagonal block)
    auto solve_half = [&](int b, int j, int half, bool keep) {
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
        if (grow <= n) {
          StoreShared(&A[(size_t)grow * n + kb + mi], acc0[tt]);
          StoreShared(&A[(size_t)grow * n + kb + 16 + mi], acc1[tt]);
        }
        if (keep) { Xl[lr * RSBA_PLD + mi] = acc0[tt]; Xl[lr * RSBA_PLD + 16 + mi] = acc1[tt]; }
      }
    };

    if (w == 0) {
      // ============================================================ the diagonal workgroup
      const bool has_next = p + 1 < np;
      const int nb0 = kb + RSBA_PB;                 // first row / column of block p + 1
      const int jr = has_next ? 1 : 0;              // slot of the right-hand-side row (block np); slot 0: block p + 1
      // block p + 1's diagonal entries of S and their damping terms: fetched now when their camera group is known to be
      // published (same group as this panel, or the next one whose flag is already up), else after this panel's X
      bool have_s = false, s_pending = false;
      double sv[2] = {0.0, 0.0}, du[2] = {0.0, 0.0};
      if (has_next) {
        if (gate.ready == nullptr || ip.first || nb0 % gate.cols != 0) have_s = true;
        else {
          __shared__ int s_gate_open;
          if (tid == 0) s_gate_open = __hip_atomic_load(gate.ready + 1 + nb0 / gate.cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gate.tag;
          __syncthreads();
          have_s = s_gate_open != 0;
          if (have_s) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        s_pending = !have_s;
        if (have_s) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = tid + u * nt, r = e >> 5, c = e & 31;
            sv[u] = Sat(nb0 + r, nb0 + c);
            du[u] = (r == c && nb0 + r < nreal) ? red[L.diagU() + nb0 + r] : 0.0;
          }
        }
      }
      // partial tiles of the fused update (K slices 1): diagonal tiles (h, ks) behind the panel blocks (>= 1056 doubles free),
      // cross tiles (cs) in Xl, update partials (h) in the scratch area
      double* fp_diag = lds + (size_t)(kb + (jr + 1) * RSBA_PB) * RSBA_PLD;
      double* fp_cross = Xl;
      double* fp_upd = scratch;
      if (wave == 0) {
        if (!DiagFactorInverseCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && lane == 0) s_ok = 0;
      } else {
        // ---- waves 1 .. 7: strip of block p, then block p + 1 (fused update + look-ahead product) and the rhs row
        if (p > 0 && wave == (has_next ? 7 : 1)) rhs_row_prefetch();
        if (p > 0) {
          for (int e = tid - 64; e < (kb >> 2) * RSBA_PB; e += nt - 64) {
            const int c = e / (kb >> 2), q0 = (e - c * (kb >> 2)) * 4;
            const double* lrow = A + (size_t)(kb + c) * n + q0;
            const double2 a01 = *reinterpret_cast<const double2*>(lrow), a23 = *reinterpret_cast<const double2*>(lrow + 2);
            Bst[(q0 + 0) * RSBA_PLD + c] = a01.x; Bst[(q0 + 1) * RSBA_PLD + c] = a01.y;
            Bst[(q0 + 2) * RSBA_PLD + c] = a23.x; Bst[(q0 + 3) * RSBA_PLD + c] = a23.y;
          }
          // block p + 1's rows must hold L through column kb - 1: the last 32 columns come from its row workgroup's panel
          // p - 1 (one hop behind the diagonal)
          if (has_next && p >= 1 && wave == 1 && lane == 0) {
            const long long t0 = wall_clock64();
            int ok = 1;
            while (__hip_atomic_load(f.rows_ready + p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
              __builtin_amdgcn_s_sleep(2);
              if (__hip_atomic_load(f.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > budget) { ok = 0; break; }
            }
            if (!ok) __hip_atomic_store(&s_w7ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          bar7();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          if (mtrace && tid == 64) mtrace[((size_t)w * 16 + p) * 8 + 1] = wall_clock64();   // strip in LDS, rows_ready seen
        }
        if (has_next && p > 0) {
          // waves 1..4: row waves (half h, K slice ks of 2): update + diagonal tile (h, h); waves 5, 6: cross tile (1, 0), K
          // slice cs of 2; wave 7: the right-hand-side row
          const int rw = wave - 1;
          if (rw < 4) {
            const int h = rw >> 1, ks = rw & 1;
            const int sr = lane >> 2, sc0 = (lane & 3) * 8;
            const int sgi = nb0 + h * 16 + sr;
            double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ks == 0 && sgi < nreal) {
              if (kb + sc0 + 8 <= nreal) {
                const double2* sp = reinterpret_cast<const double2*>(S + (size_t)sgi * nreal + kb + sc0);
#pragma unroll
                for (int u = 0; u < 4; ++u) { const double2 t = sp[u]; v[2 * u] = t.x; v[2 * u + 1] = t.y; }
              } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = Sat(sgi, kb + sc0 + u);
              }
            }
            const int nq = kb / RSBA_PB, qper = (nq + 1) / 2;
            const int qa = ks * qper * RSBA_PB, qb = min(kb, (ks + 1) * qper * RSBA_PB);
            const double* ra = A + (size_t)(nb0 + 16 * h + mi) * n + 8 * kk;
            constexpr int D = 2 * RSBA_DC_NPF;
            double pf[D][8];
            auto fetch8 = [&](double (&d)[8], const double* src) {
              const double2* pa = reinterpret_cast<const double2*>(src);
#pragma unroll
              for (int v2 = 0; v2 < 4; ++v2) { const double2 t = pa[v2]; d[2 * v2] = t.x; d[2 * v2 + 1] = t.y; }
            };
#pragma unroll
            for (int i = 0; i < D; ++i) if (qa + i * RSBA_PB < qb) fetch8(pf[i], ra + qa + i * RSBA_PB);
            d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, dg = {0, 0, 0, 0};
            for (int qg = qa; qg < qb; qg += D * RSBA_PB) {
#pragma unroll
              for (int i = 0; i < D; ++i) {
                const int q0 = qg + i * RSBA_PB;
                if (q0 < qb) {
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    const double b0 = Bst[(q0 + 8 * kk + u) * RSBA_PLD + mi];
                    const double b1 = Bst[(q0 + 8 * kk + u) * RSBA_PLD + 16 + mi];
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pf[i][u], b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pf[i][u], b1, acc1, 0, 0, 0);
                    dg = __builtin_amdgcn_mfma_f64_16x16x4f64(pf[i][u], pf[i][u], dg, 0, 0, 0);
                  }
                  if (q0 + D * RSBA_PB < qb) fetch8(pf[i], ra + q0 + D * RSBA_PB);
                }
              }
            }
            if (ks == 0) {
#pragma unroll
              for (int u = 0; u < 8; ++u) Pan[(h * 16 + sr) * RSBA_PLD + sc0 + u] = sys(sgi, kb + sc0 + u, v[u]);
              __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (ks == 0) {
                const int r = h * 16 + kk + 4 * t;
                Pan[r * RSBA_PLD + mi] -= acc0[t];
                Pan[r * RSBA_PLD + 16 + mi] -= acc1[t];
              } else {
                fp_upd[h * 512 + (kk + 4 * t) * 32 + mi] = acc0[t];
                fp_upd[h * 512 + (kk + 4 * t) * 32 + 16 + mi] = acc1[t];
              }
              fp_diag[(h * 2 + ks) * 256 + (kk + 4 * t) * 16 + mi] = dg[t];
            }
            if (mtrace && tid == 128) mtrace[((size_t)w * 16 + p) * 8 + 2] = wall_clock64();   // row wave (h 0, slice 1) done
          } else if (rw < 6) {
            const int cs = rw - 4;
            const int nq = kb / RSBA_PB, qper = (nq + 1) / 2;
            const int qa = cs * qper * RSBA_PB, qb = min(kb, (cs + 1) * qper * RSBA_PB);
            const double* r1 = A + (size_t)(nb0 + 16 + mi) * n + 8 * kk;
            const double* r0 = A + (size_t)(nb0 + mi) * n + 8 * kk;
            constexpr int D = RSBA_DC_NPF;
            double p1[D][8], p0[D][8];
            auto fetch8 = [&](double (&d)[8], const double* src) {
              const double2* pa = reinterpret_cast<const double2*>(src);
#pragma unroll
              for (int v2 = 0; v2 < 4; ++v2) { const double2 t = pa[v2]; d[2 * v2] = t.x; d[2 * v2 + 1] = t.y; }
            };
#pragma unroll
            for (int i = 0; i < D; ++i) if (qa + i * RSBA_PB < qb) { fetch8(p1[i], r1 + qa + i * RSBA_PB); fetch8(p0[i], r0 + qa + i * RSBA_PB); }
            d4_t cr = {0, 0, 0, 0};
            for (int qg = qa; qg < qb; qg += D * RSBA_PB) {
#pragma unroll
              for (int i = 0; i < D; ++i) {
                const int q0 = qg + i * RSBA_PB;
                if (q0 < qb) {
#pragma unroll
                  for (int u = 0; u < 8; ++u) cr = __builtin_amdgcn_mfma_f64_16x16x4f64(p1[i][u], p0[i][u], cr, 0, 0, 0);
                  if (q0 + D * RSBA_PB < qb) { fetch8(p1[i], r1 + q0 + D * RSBA_PB); fetch8(p0[i], r0 + q0 + D * RSBA_PB); }
                }
              }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) fp_cross[cs * 256 + (kk + 4 * t) * 16 + mi] = cr[t];
          } else {
            rhs_row_update(jr);
            if (mtrace && lane == 0) mtrace[((size_t)w * 16 + p) * 8 + 3] = wall_clock64();   // rhs row done
          }
        } else {
          // panel 0 (nothing to subtract yet) or the last panel (no next block): plain loads of the panel's columns
          if (has_next) { for (int hb = wave - 1; hb < 2; hb += nwave - 1) load_update_half(p + 1, 0, hb, 0, 1, nullptr); }
          if (wave == (has_next ? 7 : 1)) { if (p > 0) rhs_row_update(jr); else load_update_half(np, jr, 0, 0, 1, nullptr); }
        }
      }
      __syncthreads();   // [A] the factor (T, Lt, invd, Pre = L11) and the updates are done
      RSBA_DC_STAMP(4);
      if (!s_ok && tid == 0) __hip_atomic_store(chol_ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (s_w7ok == 0) { stalled = true; break; }
      // slice 1 of the update in fixed order (slice 0 is in Pan); waves 1..7 also send L11 / T on their way
      if (has_next && p > 0) {
        for (int e = tid; e < 2 * 512; e += nt) {
          const int h = e >> 9, r = (e >> 5) & 15, c = e & 31;
          Pan[(h * 16 + r) * RSBA_PLD + c] -= fp_upd[h * 512 + r * 32 + c];
        }
      }
      if (wave == 7) {
        // L11 / T leave through ONE wave: it alone waits for their acknowledgements (below, after the next barrier) and
        // publishes them — the row workgroups get T(p) while this workgroup is still busy with X
        for (int e = lane; e < RSBA_PB * RSBA_PB; e += 64) {
          const int r = e >> 5, c = e & 31;
          StoreShared(&A[(size_t)(kb + r) * n + kb + c], c > r ? T[c * RSBA_PLD + r] : Pre[r * RSBA_PLD + c]);
        }
        if (lane < RSBA_PB) StoreShared(&A[(size_t)(n + 1) * n + kb + lane], invd[lane]);
      }
      // look-ahead part of the next diagonal block (everything but this panel's X X'), kept in registers until Xl is free
      if (has_next) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + u * nt, r = e >> 5, c = e & 31;
          double d = 0.0;
          if (p > 0) {
            if ((r >> 4) == (c >> 4)) {
              const int hh = r >> 4, o = (r & 15) * 16 + (c & 15);
              d = fp_diag[(hh * 2) * 256 + o] + fp_diag[(hh * 2 + 1) * 256 + o];
            } else {
              const int o = r >= 16 ? (r & 15) * 16 + (c & 15) : (c & 15) * 16 + (r & 15);   // tile (1,0), or its mirror
              d = fp_cross[o] + fp_cross[256 + o];
            }
          }
          sv[u] = (have_s ? sys_pre(nb0 + r, nb0 + c, sv[u], du[u]) : 0.0) - d;
        }
      }
      __syncthreads();   // [B] Pan complete, partial tiles consumed (Xl is free again)
      // X = Rows T': block p + 1 (two halves, kept in Xl) and the rhs row
      if (has_next) { if (wave == 1 || wave == 2) solve_half(p + 1, 0, wave - 1, true); }
      if (wave == 3) solve_half(np, jr, 0, false);
      if (wave == 7) {
        // (publishing T only at the end of the panel takes this wait off the chain but delays the row workgroups, whose
        //  last slab the next panel's update waits for: measured 0.475 against 0.466 ms per LM iteration)
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) __hip_atomic_store(f.tdone + p, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();   // [C] Xl
      RSBA_DC_STAMP(5);
      if (has_next) {
        // X X' (tiles by waves 1..4) -> scratch, then the next diagonal block
        if (wave >= 1 && wave <= 4) {
          const int ti = (wave - 1) >> 1, tj = (wave - 1) & 1;
          d4_t acc = {0, 0, 0, 0};
#pragma unroll
          for (int qs = 0; qs < RSBA_PB; qs += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xl[(16 * ti + mi) * RSBA_PLD + qs + kk], Xl[(16 * tj + mi) * RSBA_PLD + qs + kk], acc, 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) scratch[(16 * ti + kk + 4 * t) * 32 + 16 * tj + mi] = acc[t];
        }
        __syncthreads();   // [D]
        if (s_pending) {   // the next camera group had not been published when this panel began: its entries of S now
          if (!WaitReady(gate.ready + 1 + nb0 / gate.cols, gate.tag, nullptr, gate.budget)) { stalled = true; break; }
          // (-d) + S, then - X X': the same three roundings in the same order as (S - d) - X X' of the early path, so the
          // bits do not depend on whether the flag was up when the panel began
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = tid + u * nt, r = e >> 5, c = e & 31;
            sv[u] += sys(nb0 + r, nb0 + c, Sat(nb0 + r, nb0 + c));
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; PreN[r * RSBA_PLD + c] = sv[u] - scratch[r * 32 + c]; }
        __syncthreads();   // [E] the next diagonal block is ready: wave 0 goes on
      }
      RSBA_DC_STAMP(6);
      // X is on its way: waves 1..7 wait for the acknowledgements and publish the strip; wave 0 does not wait
      if (wave != 0) {
        __builtin_amdgcn_s_waitcnt(0);
        bar7();
        if (wave == 1 && lane == 0 && has_next) __hip_atomic_store(f.strip_ready + p + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      { double* t = Pre; Pre = PreN; PreN = t; }
      RSBA_DC_STAMP(7);
    } else {
      // ============================================================ a row workgroup: blocks b >= p + 2, b mod (G - 1) == w - 1
      const int gm = G - 1;
      int first = p + 2;
      first += ((w - 1) - first % gm + gm) % gm;
      const int nown = first >= np ? 0 : (np - 1 - first) / gm + 1;
      if (nown == 0) continue;
      auto blk = [&](int j) { return first + j * gm; };
      if (p > 0) {
        if (!WaitFlagWG(f.strip_ready + p, tag, f.error, budget)) { stalled = true; break; }
        RSBA_DC_STAMP(2);
        for (int e = tid; e < (kb >> 2) * RSBA_PB; e += nt) {
          const int c = e / (kb >> 2), q0 = (e - c * (kb >> 2)) * 4;
          const double* lrow = A + (size_t)(kb + c) * n + q0;
          const double2 a01 = *reinterpret_cast<const double2*>(lrow), a23 = *reinterpret_cast<const double2*>(lrow + 2);
          Bst[(q0 + 0) * RSBA_PLD + c] = a01.x; Bst[(q0 + 1) * RSBA_PLD + c] = a01.y;
          Bst[(q0 + 2) * RSBA_PLD + c] = a23.x; Bst[(q0 + 3) * RSBA_PLD + c] = a23.y;
        }
        __syncthreads();
      }
      RSBA_DC_STAMP(3);
      if (gate.ready != nullptr && kb % gate.cols == 0 && !ip.first) {
        if (!WaitReady(gate.ready + 1 + kb / gate.cols, gate.tag, nullptr, gate.budget)) { stalled = true; break; }
      }
      {
        const int nh = 2 * nown, nsplit = (p > 0 && nh <= 2) ? 4 : ((p > 0 && nh <= 4) ? 2 : 1);
        double* part = T;   // T | Lt | Xl are idle before T(p) arrives: up to six 16 x 32 partial tiles
        for (int it = wave; it < nh * nsplit; it += nwave) {
          const int hb = it / nsplit, ks = it - hb * nsplit;
          load_update_half(blk(hb >> 1), hb >> 1, hb & 1, ks, nsplit, part + (hb * (nsplit - 1) + ks - 1) * 512);
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
      if (!WaitFlagWG(f.tdone + p, tag, f.error, budget)) { stalled = true; break; }
      {
        double tv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + u * nt, r = e >> 5, c = e & 31;
          tv[u] = r > c ? A[(size_t)(kb + c) * n + kb + r] : (r == c ? A[(size_t)(n + 1) * n + kb + c] : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * nt; T[(e >> 5) * RSBA_PLD + (e & 31)] = tv[u]; }
      }
      __syncthreads();
      RSBA_DC_STAMP(5);
      for (int hb = wave; hb < 2 * nown; hb += nwave) solve_half(blk(hb >> 1), hb >> 1, hb & 1, false);
      RSBA_DC_STAMP(6);
      // block p + 2 leaves this workgroup after this panel: its rows are final through column 32 (p + 1) - 1
      if (first == p + 2) PublishFlagWG(f.rows_ready + p + 2, tag);
      else { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
      RSBA_DC_STAMP(7);
    }
  }

  if (stalled) {
    if (tid == 0) { __hip_atomic_store(f.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (w == 0) res[RES_STALL] = 1.0; }
    if (w == 0) SolveDone(gate);
    return;
  }
  if (w != 0) return;
  // workgroup 0: every block's rows were handed over before its diagonal panel, so L is complete; L' x = y and the camera step
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (wave 0 has not acquired the row workgroups' stores yet)
  if (gate.trace && tid == 0) gate.trace[13] = wall_clock64();
  double* ysol = A + (size_t)n * n;
  double* y = BackSubstituteBlocksPrefetch(n, A, lds);
  for (int i = tid; i < n; i += nt) ysol[i] = y[i];
  __threadfence_block();
  __syncthreads();
  int ok = 1;
  if (tid == 0) { res[RES_STALL] = 0.0; ok = __hip_atomic_load(chol_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (gate.trace && tid == 0) gate.trace[14] = wall_clock64();
  CameraStepEpilogue(C, red, L, scale_c, ysol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, ok, lds, ip.cam_free);
  if (gate.trace && tid == 0) gate.trace[15] = wall_clock64();
  SolveDone(gate);
}

}  // namespace rsba
