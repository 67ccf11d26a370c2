// The LAST camera group of the reduced camera system as a BORDER of the leading system (33 to 64 cameras, one GPU).
//
// The diagonal-chain factorisation (ba_cholesky_diag.hpp) walks the 32-column panels in order and needs, at panel p, the columns
// of p in ALL the rows below: the Schur kernel therefore has to finish every tile (g, g' >= g) of camera group g before the
// group's panels start, the last group's three panels — and with them the rest of the step — start behind the kernel's last
// tile, and each of them still pays the hand-overs between the diagonal workgroup and the row workgroups (~12.5 us a panel, two
// thirds of it round trips to memory).  With the system cut as
//
//        [ A    B' ]   A: the leading B camera groups (nA = 96 B columns)      L = [ L_A      ]
//    S = [ B    C  ]   C: the last group (up to 16 cameras, padded to 96)          [ X    L_C ]
//
// the diagonal-chain kernel factors A alone (its own panels need the tiles (g, g') with g' < B only: the Schur kernel runs them
// first and publishes them early), and ONE more workgroup — this file — forms the border:
//
//   * X_g = (B_g - sum_{k<g} X_k L_A[g,k]') L_A[g,g]^-T, group by group, as tile (g, B) of the Schur kernel arrives (the work
//     list runs these tiles behind the leading system's, the last group's own tile last); the product with the groups before
//     is formed BEFORE the tile is there, from the rows of L_A the diagonal-chain kernel has finished long ago;
//   * C' = C - sum_g X_g X_g' is kept in LDS as it grows; when the last tile is published, the 96 x 96 block is factored where
//     it lies — three 32 x 32 factorisations (DiagFactorInverse) with matrix-core products between them, no hand-over, no trip
//     to memory;
//   * y_C = L_C^-1 (b_C - X y_A), x_C = L_C^-T y_C, then x_A = L_A^-T (y_A - X' x_C) with the block back-substitution of the
//     other kernels, the camera step and the candidate cameras' constants.
//
// Same fixed summation orders everywhere: bitwise reproducible.  All waits carry a budget (RES_STALL, never a hang).
#pragma once
#include "ba_cholesky_multi.hpp"

namespace rsba {

#define RSBA_BW 96     // columns of the border: one camera group, three panels
// An LDS offset the compiler must not see through: one address register per operand stream and immediate offsets behind it — left
// alone it forms every address of every unrolled loop up front from the thread index, keeps them all, and spills them (a scratch
// load in front of every matrix-core instruction).
#define RSBA_OPQ(x) asm volatile("" : "+v"(x))
#define RSBA_BLD 97    // row stride of the 96 x 96 working block in LDS (odd: the sixteen rows of an MFMA operand meet in no bank)

struct BorderCtx {
  const double* S; const double* diag_u; const double* gc; const double* corr; const double* scal;   // the reduced system as the Schur kernel leaves it (leading dimension ld)
  double* A;             // the leading system's factor: (nA + 2) x nA — row nA: y_A', row nA + 1: 1 / diag
  double* XB;            // [96][nA]: the border's rows of L
  double* scale_c;       // Jacobi scale of the columns (formed on a run's first step)
  const int* gate_ready; const int* all_diag; int gate_tag, gated; long long gate_budget;   // the Schur kernel's stage flags (nullptr / 0: everything is there)
  const int* tdone; const int* strip_ready; const int* rows_ready; const int* a_done; int* error; int tag; long long budget;   // the diagonal-chain kernel's flags
  int nrow_wgs;          // row workgroups of the diagonal-chain kernel (block b of the leading system is workgroup 1 + b mod nrow_wgs's)
  int ld, nA, nB, B;     // columns of S, of the leading system (96 B), real columns of the border (<= 96), leading camera groups
  double min_diag, max_diag, inv_radius; int first, jacobi;
  // the camera step
  int C; const double* cam_x; double* cam_c; const double* intr; double* camc_c; double* dcam; const double* gmax_p; double* res; const double* cam_free;
  int* chol_ok; int* done; long long* trace;   // StageGate::done (= gate_tag when the solve is through, or has given up), StageGate::trace
  long long* mtrace;     // diagnostic: this workgroup's stamps (RSBA_MC_TRACE), or nullptr
};

#define RSBA_BORDER_CTX_DOUBLES 64   // the workgroup's BorderCtx and its flag word, at the end of its carve
static_assert(sizeof(BorderCtx) + 8 <= RSBA_BORDER_CTX_DOUBLES * sizeof(double), "BorderCtx");
__host__ __device__ inline size_t BorderLdsDoubles(int nc) {
  return (size_t)RSBA_BW * RSBA_BLD + 6 * RSBA_PB * RSBA_PLD + 3 * RSBA_PB * RSBA_PLD + RSBA_BW + (size_t)nc + RSBA_BW + 2 * RSBA_BW + (size_t)nc + RSBA_BORDER_CTX_DOUBLES;
}

// block (i, j), i >= j, of the border's 3 x 3 blocks in the packed lower block triangle
__device__ __forceinline__ int BorderBlk(int i, int j) { return i * (i + 1) / 2 + j; }

// acc += A[oa + 4 k] * B[ob + 4 k], k < NK (one 16 x 16 tile over 4 NK columns of both operands, LDS): ALL the operands are asked for
// before the first matrix-core instruction — read where they are used, every instruction waited for its own two LDS reads
// (~100 cycles in front of 64 of arithmetic, in order), and the products ran at half the matrix cores' rate.
template <int NK>
__device__ __forceinline__ d4_t BorderMfmaStrip(const double* A, int oa, const double* B, int ob, d4_t acc) {
  double av[NK], bv[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) { av[k] = A[oa + 4 * k]; bv[k] = B[ob + 4 * k]; }
#pragma unroll
  for (int k = 0; k < NK; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], bv[k], acc, 0, 0, 0);
  return acc;
}
// ... two tiles that share the A operand (B rows 16 apart)
template <int NK>
__device__ __forceinline__ void BorderMfmaStrip2(const double* A, int oa, const double* B, int ob, d4_t& a0, d4_t& a1) {
  double av[NK], b0[NK], b1[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) { av[k] = A[oa + 4 * k]; b0[k] = B[ob + 4 * k]; b1[k] = B[ob + 16 * RSBA_PLD + 4 * k]; }
#pragma unroll
  for (int k = 0; k < NK; ++k) { a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], b0[k], a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], b1[k], a1, 0, 0, 0); }
}

// INLINED into the kernel (it is the whole life of one workgroup, behind an early return): out of line — as the diagonal workgroup's
// routines are, for THEIR registers — it ran 8 us slower per step (0.3645 against 0.356 ms): a callee saves what it uses of the
// caller's registers in scratch, its own spills sat in the matrix-core loops (704 against 140 bytes of scratch a lane), and the
// back-substitution inside it took 18 instead of 13 us.  The kernel's other paths did not move (panel period, the sequential
// schedule, RSBA_BORDER=0, the multi-GPU instance: measured, same box).  The LDS pointers carry their address space in their types.
typedef __attribute__((address_space(3))) const BorderCtx lds_BorderCtx;
static __device__ __forceinline__ void BorderWorkgroup(lds_BorderCtx* bcp, lds_double* lds_in) {
  double* const lds = (double*)lds_in;
  const BorderCtx& bc = *(const BorderCtx*)bcp;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, mi = lane & 15, kk = lane >> 4;
  const int nA = bc.nA, nB = bc.nB, ld = bc.ld, B = bc.B, tag = bc.tag;
  // (global address space, said explicitly: the pointers come out of a structure in LDS as generic ones, and every access would be a flat one)
  // ... and UNIFORM, said explicitly too (read out of LDS they sit in vector registers: a buffer load with such a descriptor is
  // wrapped in a loop over the lanes' values — the back-substitution's 64 strip loads took twice their time)
  typedef __attribute__((address_space(1))) double gdouble;
  auto uniform_ptr = [](const void* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (double*)(((unsigned long long)hi << 32) | lo);
  };
  double* const A_generic = uniform_ptr(bc.A);
  gdouble* const A = (gdouble*)A_generic;
  gdouble* const XB = (gdouble*)uniform_ptr(bc.XB);
  const gdouble* const S = (const gdouble*)uniform_ptr(bc.S);
  double* R = lds;                                   // 96 x 97: the group's block B_g on its way to X_g; later scratch
  double* Cb = R + RSBA_BW * RSBA_BLD;               // six 32 x 33 blocks: sum X X', then C', then L_C
  double* Tt = Cb + 6 * RSBA_PB * RSBA_PLD;          // three 32 x 33 tiles: T / L blocks of the leading group, then T of the border's panels
  double* invd = Tt + 3 * RSBA_PB * RSBA_PLD;        // 96
  double* scl = invd + RSBA_BW;                      // nA + 96
  double* rB = scl + nA + RSBA_BW;                   // 96: right-hand side of the border, then y_C
  double* xB = rB + RSBA_BW;                         // 96
  double* yA2 = xB + RSBA_BW;                        // nA: y_A = L_A^-1 b_A, once the leading system is through
  int& sb_ok = *(int*)((__attribute__((address_space(3))) int*)(bcp + 1));   // (behind the constants: no static LDS)
  if (tid == 0) sb_ok = 1;
  const long long gbudget = bc.gate_budget > 0 ? bc.gate_budget : RSBA_STALL_TICKS;
  long long* tr = bc.mtrace;
#define RSBA_BORDER_STAMP(k) do { if (tr && tid == 0) tr[k] = wall_clock64(); } while (0)
  bool stalled = false;
  auto solve_done = [&]() {   // (SolveDone, ba_point_kernels.hpp)
    if (bc.done == nullptr) return;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(bc.done, bc.gate_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  };
  RSBA_BORDER_STAMP(0);
  // ---- the Jacobi scale of every column
  if (bc.first && bc.gated) {
    if (bc.all_diag != nullptr) { if (!WaitFlagWG(bc.all_diag, bc.gate_tag, bc.error, gbudget)) stalled = true; }
    else { for (int g = 0; g < 2 * B + 1 && !stalled; ++g) if (!WaitFlagWG(bc.gate_ready + 1 + g, bc.gate_tag, bc.error, gbudget)) stalled = true; }
  }
  if (!stalled) {
    for (int i = tid; i < nA + RSBA_BW; i += nt) {
      double sc = 1.0;
      if (i < nA + nB) {
        sc = bc.first ? (bc.jacobi ? 1.0 / (1.0 + sqrt(bc.diag_u[i])) : 1.0) : bc.scale_c[i];
        if (bc.first && i >= nA) bc.scale_c[i] = sc;
      }
      scl[i] = sc;
    }
    for (int e = tid; e < 6 * RSBA_PB * RSBA_PLD; e += nt) Cb[e] = 0.0;
  }
  __syncthreads();

  // ---- X_g, group by group; inside a group panel by panel, as the diagonal-chain kernel gets there
  auto store_T = [&](double* T, const double (&t2)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = tid + u * nt, i = e >> 5, j = e & 31; if (j >= i) { T[j * RSBA_PLD + i] = t2[u]; if (j > i) T[i * RSBA_PLD + j] = 0.0; } }
  };
  auto store_L = [&](double* Lb, const double (&l2)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = tid + u * nt; Lb[(e >> 5) * RSBA_PLD + (e & 31)] = l2[u]; }
  };
  // a wavefront owns sixteen rows of R for the whole group: R_p -= X_q L(p, q)' and X_p = R_p T(p)' need no barrier between them
  auto solve_T = [&](int p, const double* T) {
    if (wave < 6) {
      const int i0 = 16 * wave;
      int oa = (i0 + mi) * RSBA_BLD + RSBA_PB * p + kk, ob = mi * RSBA_PLD + kk, oc = (i0 + kk) * RSBA_BLD + RSBA_PB * p + mi;
      RSBA_OPQ(oa); RSBA_OPQ(ob); RSBA_OPQ(oc);
      d4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
      BorderMfmaStrip2<8>(R, oa, T, ob, a0, a1);
#pragma unroll
      for (int q = 0; q < 4; ++q) { R[oc + 4 * q * RSBA_BLD] = a0[q]; R[oc + 4 * q * RSBA_BLD + 16] = a1[q]; }
    }
  };
  auto update_L = [&](int q, int p, const double* Lb) {   // R_p -= X_q Lb', Lb = L(p, q)
    if (wave < 6) {
      const int i0 = 16 * wave;
      int oa = (i0 + mi) * RSBA_BLD + RSBA_PB * q + kk, ob = mi * RSBA_PLD + kk, oc = (i0 + kk) * RSBA_BLD + RSBA_PB * p + mi;
      RSBA_OPQ(oa); RSBA_OPQ(ob); RSBA_OPQ(oc);
      d4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
      BorderMfmaStrip2<8>(R, oa, Lb, ob, a0, a1);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) { R[oc + 4 * t4 * RSBA_BLD] -= a0[t4]; R[oc + 4 * t4 * RSBA_BLD + 16] -= a1[t4]; }
    }
  };
  // One lane polls a row workgroup's progress word (error[4 + w] = (tag << 4) | panels it is through), then the workgroup acquires.
  int& sb_w = *((int*)((__attribute__((address_space(3))) int*)(bcp + 1)) + 1);
  auto wait_progress = [&](int w, int need) -> bool {
    if (tid == 0) {
      const long long t0 = wall_clock64();
      int ok = 1;
      for (;;) {
        const int v = __hip_atomic_load(bc.error + 4 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 4) == tag && (v & 15) >= need) break;
        __builtin_amdgcn_s_sleep(2);
        if (__hip_atomic_load(bc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > bc.budget) { ok = 0; break; }
      }
      sb_w = ok;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return sb_w != 0;
  };
  // P += X_g L_A[group h, columns of group g]' (36 tiles, tile t = wave + 8 u): X_g where it lies in R, the rows of the leading factor
  // 32 columns at a time through the three tiles' space, the next chunk in flight while this one is multiplied
  auto product = [&](d4_t (&P)[5], int h, int c0) {
    double* AB = Tt;   // 96 x 33
    double pa[6];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; pa[u] = A[(size_t)(RSBA_BW * h + r) * nA + c0 + k0 + c]; }
    };
    fetch(0);
    for (int k0 = 0; k0 < RSBA_BW; k0 += RSBA_PB) {
      __syncthreads();   // the previous chunk has been consumed
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; AB[r * RSBA_PLD + c] = pa[u]; }
      if (k0 + RSBA_PB < RSBA_BW) fetch(k0 + RSBA_PB);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        int t = wave + 8 * u;
        RSBA_OPQ(t);   // (the tile's offsets formed HERE, every chunk: hoisted out of the chunk loop they are spilled, and the reload's wait also
                       //  waits for the next chunk's loads from memory — nothing overlapped)
        if (t < 36) {
          const int ti = t / 6, tj = t - 6 * ti;
          int oa = (16 * ti + mi) * RSBA_BLD + k0 + kk, ob = (16 * tj + mi) * RSBA_PLD + kk;
          RSBA_OPQ(oa); RSBA_OPQ(ob);
          P[u] = BorderMfmaStrip<8>(R, oa, AB, ob, P[u]);
        }
      }
    }
    __syncthreads();
  };
  // sum X X' += X_g[:, k0:k1) X_g[:, k0:k1)': 24 tiles (the diagonal blocks whole), tile q = wave + 8 u
  auto syrk = [&](int k0, int k1) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int q = wave + 8 * u, blk = q >> 2, ti = (q >> 1) & 1, tj = q & 1;
      const int bi = blk == 0 ? 0 : (blk < 3 ? 1 : 2), bj = blk - bi * (bi + 1) / 2;
      int oa = (RSBA_PB * bi + 16 * ti + mi) * RSBA_BLD + kk + k0, ob = (RSBA_PB * bj + 16 * tj + mi) * RSBA_BLD + kk + k0, oc = blk * RSBA_PB * RSBA_PLD + (16 * ti + kk) * RSBA_PLD + 16 * tj + mi;
      RSBA_OPQ(oa); RSBA_OPQ(ob); RSBA_OPQ(oc);
      d4_t xx = {0, 0, 0, 0};
      for (int ks = 0; ks < k1 - k0; ks += RSBA_PB) xx = BorderMfmaStrip<8>(R, oa + ks, R, ob + ks, xx);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) Cb[oc + 4 * t4 * RSBA_PLD] += xx[t4];
    }
  };
  d4_t P1[5], P2[5];   // X[:, groups before] L_A[group, groups before]' of the next group and of the one after it (at most four groups: B <= 3)
#pragma unroll
  for (int u = 0; u < 5; ++u) { P1[u] = d4_t{0, 0, 0, 0}; P2[u] = d4_t{0, 0, 0, 0}; }
  for (int e = tid; e < RSBA_BW; e += nt) rB[e] = 0.0;
  const int gm = bc.nrow_wgs, npA = 3 * B;
  for (int g = 0; g < B && !stalled; ++g) {
    const int c0 = RSBA_BW * g;
    const bool last_g = g == B - 1;
    d4_t acc[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) { acc[u] = P1[u]; P1[u] = P2[u]; P2[u] = d4_t{0, 0, 0, 0}; }
    RSBA_BORDER_STAMP(1 + 4 * g);
    // tile (g, B) of the Schur kernel: R = B_g (scaled) - X[:, groups before] L_A[group g, groups before]'
    if (bc.gated && !WaitFlagWG(bc.gate_ready + 1 + B + g, bc.gate_tag, bc.error, gbudget)) { stalled = true; break; }
    RSBA_BORDER_STAMP(2 + 4 * g);
    {
      double sv[5][4];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int t = min(wave + 8 * u, 35), ti = t / 6, tj = t - 6 * ti;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 16 * ti + kk + 4 * q, j = 16 * tj + mi;
          sv[u][q] = S[(size_t)(nA + min(i, nB - 1)) * ld + c0 + j];   // (every lane a valid address: no branch per load)
        }
      }
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int t = wave + 8 * u, ti = t / 6, tj = t - 6 * ti;
        if (t < 36) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 16 * ti + kk + 4 * q, j = 16 * tj + mi;
            R[i * RSBA_BLD + j] = (i < nB ? sv[u][q] * (scl[nA + i] * scl[c0 + j]) : 0.0) - acc[u][q];
          }
        }
      }
    }
    // The group's three panels.  Everything but T of the last one in ONE round trip, once the group's blocks of the leading factor are
    // complete and its first two panels factored (the flags go up in this order); T of the last panel behind its own flag — the
    // border's workgroup is usually waiting for exactly that one (five dependent round trips, a flag and a load each, cost 14 us).
    if (!WaitFlagWG(bc.strip_ready + 3 * g + 2, tag, bc.error, bc.budget) || !WaitFlagWG(bc.tdone + 3 * g + 1, tag, bc.error, bc.budget)) { stalled = true; break; }
    {
      auto load_T = [&](int p, double (&t2)[2]) {
        const int kb = c0 + RSBA_PB * p;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + u * nt, i = e >> 5, j = e & 31;
          const double tvv = A[j > i ? (size_t)(kb + i) * nA + kb + j : (size_t)(nA + 1) * nA + kb + i];   // T' above the diagonal, 1 / diag in row nA + 1
          t2[u] = j >= i ? tvv : 0.0;
        }
      };
      auto load_L = [&](int pp, int q, double (&l2)[2]) {   // L(3 g + pp, 3 g + q)
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * nt; l2[u] = A[(size_t)(c0 + RSBA_PB * pp + (e >> 5)) * nA + c0 + RSBA_PB * q + (e & 31)]; }
      };
      double t0[2], t1[2], t2[2] = {0.0, 0.0}, l10[2], l20[2], l21[2];
      // (is the last panel factored already?  Then its T rides with the others: the wait above ends in a barrier, the peek is one more)
      if (tid == 0) sb_w = __hip_atomic_load(bc.tdone + 3 * g + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag ? 1 : 0;
      __syncthreads();
      const bool all_up = sb_w != 0;
      if (all_up) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      load_T(0, t0); load_L(1, 0, l10); load_L(2, 0, l20); load_T(1, t1); load_L(2, 1, l21);
      if (all_up) load_T(2, t2);
      store_T(Tt, t0); store_L(Tt + RSBA_PB * RSBA_PLD, l10); store_L(Tt + 2 * RSBA_PB * RSBA_PLD, l20);
      __syncthreads();
      if (g == B - 1) RSBA_BORDER_STAMP(20);
      solve_T(0, Tt); update_L(0, 1, Tt + RSBA_PB * RSBA_PLD); update_L(0, 2, Tt + 2 * RSBA_PB * RSBA_PLD);
      __syncthreads();
      store_T(Tt, t1); store_L(Tt + RSBA_PB * RSBA_PLD, l21);
      if (all_up) store_T(Tt + 2 * RSBA_PB * RSBA_PLD, t2);
      __syncthreads();
      solve_T(1, Tt); update_L(1, 2, Tt + RSBA_PB * RSBA_PLD);
      // (the first two panels' columns of X_g are final: their part of the sum while the last panel's T is on its way — the wait ends in
      //  the barrier the sum needs in front of it)
      if (!all_up) {
        if (!WaitFlagWG(bc.tdone + 3 * g + 2, tag, bc.error, bc.budget)) { stalled = true; break; }
        load_T(2, t2);
      } else __syncthreads();
      if (g == B - 1) RSBA_BORDER_STAMP(21);
      if (!last_g) syrk(0, 2 * RSBA_PB);   // (the last group's X X': block (0, 0) behind the last tile's loads, the rest beside the first factorisation)
      if (g == B - 1) RSBA_BORDER_STAMP(22);
      if (!all_up) {
        store_T(Tt + 2 * RSBA_PB * RSBA_PLD, t2);
        __syncthreads();
      }
      solve_T(2, Tt + 2 * RSBA_PB * RSBA_PLD);
    }
    if (stalled) break;
    __syncthreads();
    RSBA_BORDER_STAMP(3 + 4 * g);
    // the group's entries of y_A = L_A^-1 b_A (for r -= X_g y_g below): final once the right-hand-side row is through the group's panels — it is
    // a row workgroup's (block np of the leading system) until the last two panels, the diagonal workgroup's then; asked for ahead of the sum
    // (the last group's own: behind the last tile, below — its entries of the last panel come with the end of the leading system)
    const bool last = g == B - 1;
    double yg_v = 0.0;
    if (!last) {
      if (!wait_progress(1 + npA % gm, 3 * g + 3)) { stalled = true; break; }
      if (tid < RSBA_BW) yg_v = A[(size_t)nA * nA + c0 + tid];
    }
    if (!last) syrk(2 * RSBA_PB, RSBA_BW);
    if (g == B - 1) RSBA_BORDER_STAMP(23);
    // the border's right-hand side: r -= X_g y_g
    if (!last) {
      double* yg = Tt;   // 96 (the tiles have been read: the barrier in front of the stamp above)
      if (tid < RSBA_BW) yg[tid] = yg_v;
      __syncthreads();
      if (tid < 4 * RSBA_BW) {
        const int i = tid >> 2, part = tid & 3;
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < 24; ++j) sum = fma(R[i * RSBA_BLD + 24 * part + j], yg[24 * part + j], sum);
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if (part == 0) rB[i] -= sum;
      }
    }
    // X_g to memory (read back for X' x_C; nobody waits for the stores here; the last group's leave behind the last tile's loads, below)
    if (!last) { for (int e = tid; e < RSBA_BW * RSBA_BW; e += nt) { const int i = e / RSBA_BW, j = e - RSBA_BW * i; XB[(size_t)i * nA + c0 + j] = R[i * RSBA_BLD + j]; } }
    RSBA_BORDER_STAMP(4 + 4 * g);
    // X_g's part of the later groups' products, as soon as those groups' rows of the leading factor hold the columns of group g:
    // the group after the next first (its rows have them since their row workgroups went through panel 3 g + 2), then the next
    // (its blocks become strips / are handed over a panel later)
    if (g + 2 < B) {
      for (int b3 = 0; b3 < 3 && !stalled; ++b3) if (!wait_progress(1 + (3 * (g + 2) + b3) % gm, 3 * g + 3)) stalled = true;
      if (stalled) break;
      product(P2, g + 2, c0);
    }
    if (g + 1 < B) {
      if (!WaitFlagWG(bc.strip_ready + 3 * (g + 1) + 1, tag, bc.error, bc.budget) || !WaitFlagWG(bc.rows_ready + 3 * (g + 1) + 2, tag, bc.error, bc.budget)) { stalled = true; break; }
      if (g == B - 2) RSBA_BORDER_STAMP(24);
      product(P1, g + 1, c0);
      if (g == B - 2) RSBA_BORDER_STAMP(25);
    }
  }

  // ---- y_A into LDS, once the leading system is through; the border's own gradient entries come with the last tile
  double* yA = R;   // nA, at lds[0 ..): the back-substitution's right-hand side (R holds the last group's X until r has taken its part)
  if (!stalled && !WaitFlagWG(bc.a_done, tag, bc.error, bc.budget)) stalled = true;
  if (!stalled) {
    RSBA_BORDER_STAMP(14);
    // ---- the last tile: C' = C (scaled, damped; identity where the border is padded) - sum X X'
    if (bc.gated && !WaitFlagWG(bc.gate_ready + 1 + 2 * B, bc.gate_tag, bc.error, gbudget)) stalled = true;
  }
  if (stalled) {
    if (tid == 0) { __hip_atomic_store(bc.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bc.res[RES_STALL] = 1.0; }
    solve_done();
    return;
  }
  RSBA_BORDER_STAMP(15);
  {
    double sv[12], du[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int e = tid + u * nt, blk = e >> 10, r = (e >> 5) & 31, c = e & 31;
      const int bi = blk == 0 ? 0 : (blk < 3 ? 1 : 2), bj = blk - bi * (bi + 1) / 2;
      const int gi = RSBA_PB * bi + r, gj = RSBA_PB * bj + c;
      sv[u] = S[(size_t)(nA + min(gi, nB - 1)) * ld + nA + min(gj, nB - 1)];   // (every lane a valid address: no branch per load)
      du[u] = gi == gj ? bc.diag_u[nA + min(gi, nB - 1)] : 0.0;
    }
    // (behind the tile's loads, while they are on their way: y_A)
    const double ya_v = tid < nA ? A[(size_t)nA * nA + tid] : 0.0;
    if (tid < nA) yA2[tid] = ya_v;
    // the last group's X X' (X_g in R): block (0, 0) — what the first factorisation needs — while the loads above are on their way; the other
    // five blocks are subtracted beside that factorisation
    if (wave < 4) {
      const int ti = (wave >> 1) & 1, tj = wave & 1;
      int oa = (16 * ti + mi) * RSBA_BLD + kk, ob = (16 * tj + mi) * RSBA_BLD + kk, oc = (16 * ti + kk) * RSBA_PLD + 16 * tj + mi;
      RSBA_OPQ(oa); RSBA_OPQ(ob); RSBA_OPQ(oc);
      d4_t xx = {0, 0, 0, 0};
      for (int ks = 0; ks < RSBA_BW; ks += RSBA_PB) xx = BorderMfmaStrip<8>(R, oa + ks, R, ob + ks, xx);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) Cb[oc + 4 * t4 * RSBA_PLD] += xx[t4];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int e = tid + u * nt, blk = e >> 10, r = (e >> 5) & 31, c = e & 31;
      const int bi = blk == 0 ? 0 : (blk < 3 ? 1 : 2), bj = blk - bi * (bi + 1) / 2;
      const int gi = RSBA_PB * bi + r, gj = RSBA_PB * bj + c;
      double v = gi == gj ? 1.0 : 0.0;
      if (gi < nB && gj < nB) {
        v = sv[u] * (scl[nA + gi] * scl[nA + gj]);
        if (gi == gj) v += fmin(fmax(scl[nA + gi] * scl[nA + gi] * du[u], bc.min_diag), bc.max_diag) * bc.inv_radius;
      }
      double* cb = Cb + blk * RSBA_PB * RSBA_PLD + r * RSBA_PLD + c;
      *cb = v - *cb;
    }
    if (tid < nB) rB[tid] += scl[nA + tid] * (bc.gc[nA + tid] + bc.corr[nA + tid]);
  }
  __syncthreads();
  RSBA_BORDER_STAMP(26);
  // ---- L_C: three panels where they lie, then y_C = L_C^-1 r and x_C = L_C^-T y_C — wavefront 0.  The others form the products between
  // the panels and, before anything else, ask for all of X (96 x nA, 221 KB through this one compute unit: 8 us if waited for) — rows
  // w - 1, w + 6, ... of wavefront w, a row's columns over the lanes — so that X' x_C is a few multiply-adds once x_C is there.
  // (the factorisation's second tile: the third T's place while the first panel is factored — R still holds the last group's X then —, in R after)
  double* vpart = R + 2304;   // [7][nA] the wavefronts' partial sums of X' x_C
#define RSBA_BORDER_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
  // One block of the forward substitution y_C = L_C^-1 r by one wavefront (lane (r, h) takes half of a row's terms, the halves meet by a
  // lane exchange): t = r_p - sum_{q < p} X(p, q) y_q, then y_p = T_p t into r's place — or, while T_p does not exist yet, t itself.
  // Blocks 0 and 1 and the last block's t are formed by a wavefront that has nothing to do BESIDE the factorisations (y_0 needs T_0,
  // y_1 needs T_1 and X(1, 0), t_2 needs X(2, 0), X(2, 1)): behind the third factorisation stand y_2 = T_2 t_2 and the three blocks back.
  auto forward_block = [&](int p, bool apply_T, double* tm) {
    const int r = lane & 31, h = lane >> 5;
    double t = 0.0;
    for (int q = 0; q < p; ++q) {
      const double* Xb = Cb + BorderBlk(p, q) * RSBA_PB * RSBA_PLD;
#pragma unroll
      for (int k = 0; k < 16; ++k) t = fma(Xb[r * RSBA_PLD + 16 * h + k], rB[RSBA_PB * q + 16 * h + k], t);
    }
    t += __shfl_xor(t, 32, 64);
    const double tv = rB[RSBA_PB * p + r] - t;
    RSBA_BORDER_WSYNC();
    if (!apply_T) { if (lane < 32) rB[RSBA_PB * p + r] = tv; RSBA_BORDER_WSYNC(); return; }
    if (lane < 32) tm[r] = tv;
    RSBA_BORDER_WSYNC();
    const double* T = Tt + p * RSBA_PB * RSBA_PLD;
    double y = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) y = fma(T[r * RSBA_PLD + 16 * h + k], tm[16 * h + k], y);
    y += __shfl_xor(y, 32, 64);
    RSBA_BORDER_WSYNC();
    if (lane < 32) rB[RSBA_PB * p + r] = y;
    RSBA_BORDER_WSYNC();
  };
  if (wave == 0) {
    for (int p = 0; p < 3; ++p) {
      double* Lt = p == 0 ? Tt + 2 * RSBA_PB * RSBA_PLD : R + 1024;
      if (!DiagFactorInverseCall((lds_double*)(Cb + BorderBlk(p, p) * RSBA_PB * RSBA_PLD), RSBA_PB, (lds_double*)(Tt + p * RSBA_PB * RSBA_PLD), (lds_double*)Lt, (lds_double*)(invd + RSBA_PB * p), lane) && lane == 0) sb_ok = 0;
      RSBA_BORDER_STAMP(27 + p);
      if (p == 2) break;
      __syncthreads();   // T(p)
      __syncthreads();   // X(., p)
      __syncthreads();   // the blocks behind panel p are up to date
    }
    if (bc.trace && tid == 0) bc.trace[13] = wall_clock64();
    RSBA_BORDER_STAMP(16);
    // lane (r, h) takes half of a row's terms, the halves meet by a lane exchange
    double* tmp = R + 2112;
    const int r = lane & 31, h = lane >> 5;
    RSBA_BORDER_WSYNC();
    {
      // y_2 = T_2 t_2 (t_2 stands in r's last block: the seventh wavefront put it there)
      if (lane < 32) tmp[r] = rB[2 * RSBA_PB + r];
      RSBA_BORDER_WSYNC();
      const double* T = Tt + 2 * RSBA_PB * RSBA_PLD;
      double y = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) y = fma(T[r * RSBA_PLD + 16 * h + k], tmp[16 * h + k], y);
      y += __shfl_xor(y, 32, 64);
      RSBA_BORDER_WSYNC();
      if (lane < 32) rB[2 * RSBA_PB + r] = y;
      RSBA_BORDER_WSYNC();
    }
    for (int p = 2; p >= 0; --p) {
      double t = 0.0;
      for (int q = p + 1; q < 3; ++q) {
        const double* Xb = Cb + BorderBlk(q, p) * RSBA_PB * RSBA_PLD;
#pragma unroll
        for (int k = 0; k < 16; ++k) t = fma(Xb[(16 * h + k) * RSBA_PLD + r], xB[RSBA_PB * q + 16 * h + k], t);
      }
      t += __shfl_xor(t, 32, 64);
      if (lane < 32) tmp[r] = rB[RSBA_PB * p + r] - t;
      RSBA_BORDER_WSYNC();
      const double* T = Tt + p * RSBA_PB * RSBA_PLD;
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) x = fma(T[(16 * h + k) * RSBA_PLD + r], tmp[16 * h + k], x);
      x += __shfl_xor(x, 32, 64);
      RSBA_BORDER_WSYNC();
      if (lane < 32) xB[RSBA_PB * p + r] = x;
      RSBA_BORDER_WSYNC();
    }
    RSBA_BORDER_STAMP(17);
    __syncthreads();   // x_C
  } else {
    const int w7 = wave - 1;
    // beside the first factorisation: the last group's X to memory (by the wavefronts that will read it back: the factoring wavefront waits
    // for nobody's stores), r -= X_g y_g of the last group, X_g where it still lies in R
    { const int c0 = RSBA_BW * (B - 1); for (int e = tid - 64; e < RSBA_BW * RSBA_BW; e += nt - 64) { const int i = e / RSBA_BW, j = e - RSBA_BW * i; XB[(size_t)i * nA + c0 + j] = R[i * RSBA_BLD + j]; } }
    if (tid - 64 < 4 * RSBA_BW) {
      const int i = (tid - 64) >> 2, part = tid & 3;
      const double* yg = yA2 + RSBA_BW * (B - 1);
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < 24; ++j) sum = fma(R[i * RSBA_BLD + 24 * part + j], yg[24 * part + j], sum);
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      if (part == 0) rB[i] -= sum;
    }
    if (wave != 4) {
      const int k6 = wave < 4 ? wave - 1 : wave - 2;
      for (int q = 4 + k6; q < 24; q += 6) {
        const int blk = q >> 2, ti = (q >> 1) & 1, tj = q & 1;
        const int bi = blk < 3 ? 1 : 2, bj = blk - bi * (bi + 1) / 2;
        int oa = (RSBA_PB * bi + 16 * ti + mi) * RSBA_BLD + kk, ob = (RSBA_PB * bj + 16 * tj + mi) * RSBA_BLD + kk, oc = blk * RSBA_PB * RSBA_PLD + (16 * ti + kk) * RSBA_PLD + 16 * tj + mi;
        RSBA_OPQ(oa); RSBA_OPQ(ob); RSBA_OPQ(oc);
        d4_t xx = {0, 0, 0, 0};
        for (int ks = 0; ks < RSBA_BW; ks += RSBA_PB) xx = BorderMfmaStrip<8>(R, oa + ks, R, ob + ks, xx);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) Cb[oc + 4 * t4 * RSBA_PLD] -= xx[t4];
      }
    }
    __builtin_amdgcn_s_waitcnt(0);   // (this thread's stores of X are performed; everybody's behind the barrier below)
    double xv[14][5];
    for (int p = 0; p < 2; ++p) {
      const double* T = Tt + p * RSBA_PB * RSBA_PLD;
      if (p == 1) {
        // all of X, asked for while the second panel is factored (nothing for these wavefronts to do; behind the first barrier
        // everybody's stores of X are performed): used behind x_C
#pragma unroll
        for (int rr = 0; rr < 14; ++rr) {
#pragma unroll
          for (int c = 0; c < 5; ++c) xv[rr][c] = XB[(size_t)min(w7 + 7 * rr, RSBA_BW - 1) * nA + min(lane + 64 * c, nA - 1)];
        }
      }
      __syncthreads();   // T(p)
      if (w7 == 6) forward_block(p, true, R + 2144);   // y_p (X(p, q < p) have been there since the last panel)
      // X(p', p) = C'(p', p) T(p)', in place: a 16-row half per wavefront
      if (w7 < 2 * (2 - p)) {
        const int pp = p + 1 + (w7 >> 1), i0 = 16 * (w7 & 1);
        double* X = Cb + BorderBlk(pp, p) * RSBA_PB * RSBA_PLD;
        int oa = (i0 + mi) * RSBA_PLD + kk, ob = mi * RSBA_PLD + kk, oc = (i0 + kk) * RSBA_PLD + mi;
        RSBA_OPQ(oa); RSBA_OPQ(ob); RSBA_OPQ(oc);
        d4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        BorderMfmaStrip2<8>(X, oa, T, ob, a0, a1);
#pragma unroll
        for (int q = 0; q < 4; ++q) { X[oc + 4 * q * RSBA_PLD] = a0[q]; X[oc + 4 * q * RSBA_PLD + 16] = a1[q]; }
      }
      __syncthreads();   // X(., p)
      if (w7 == 6 && p == 1) forward_block(2, false, R + 2144);   // t_2 = r_2 - X(2, 0) y_0 - X(2, 1) y_1
      // C'(i, j) -= X(i, p) X(j, p)' for p < j <= i: twelve tiles behind panel 0, four behind panel 1
      const int ntile = p == 0 ? 12 : 4;
      for (int q = w7; q < ntile; q += 7) {
        const int which = q >> 2, ti = (q >> 1) & 1, tj = q & 1;
        const int bi = p == 0 ? (which == 0 ? 1 : 2) : 2, bj = p == 0 ? (which == 2 ? 2 : 1) : 2;   // (1,1), (2,1), (2,2) | (2,2)
        const double* Xi = Cb + BorderBlk(bi, p) * RSBA_PB * RSBA_PLD;
        const double* Xj = Cb + BorderBlk(bj, p) * RSBA_PB * RSBA_PLD;
        int oa = (16 * ti + mi) * RSBA_PLD + kk, ob = (16 * tj + mi) * RSBA_PLD + kk, oc = BorderBlk(bi, bj) * RSBA_PB * RSBA_PLD + (16 * ti + kk) * RSBA_PLD + 16 * tj + mi;
        RSBA_OPQ(oa); RSBA_OPQ(ob); RSBA_OPQ(oc);
        d4_t xx = {0, 0, 0, 0};
        xx = BorderMfmaStrip<8>(Xi, oa, Xj, ob, xx);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) Cb[oc + 4 * t4 * RSBA_PLD] -= xx[t4];
      }
      __syncthreads();   // the blocks behind panel p are up to date
    }
    __syncthreads();   // x_C
    // this wavefront's rows of X' x_C
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      double sum = 0.0;
#pragma unroll
      for (int rr = 0; rr < 14; ++rr) { const int i = w7 + 7 * rr; sum = fma(xv[rr][c], i < RSBA_BW ? xB[min(i, RSBA_BW - 1)] : 0.0, sum); }
      if (lane + 64 * c < nA) vpart[w7 * nA + lane + 64 * c] = sum;
    }
  }
#undef RSBA_BORDER_WSYNC
  __syncthreads();
  // ---- x_A = L_A^-T (y_A - X' x_C)
  if (tid < nA) {
    double sum = vpart[tid];
#pragma unroll
    for (int k = 1; k < 7; ++k) sum += vpart[k * nA + tid];
    yA[tid] = yA2[tid] - sum;
  }
  // what the camera step needs from memory, asked for ahead of the back-substitution (see k_reduced_system_solve_diag)
  const int nreal = nA + nB;
  const bool e_on = tid < nreal;
  double pf_x = 0.0, pf_g = 0.0, pf_free = 1.0, pf_in[4] = {0.0, 0.0, 0.0, 0.0}, pf_s[4] = {0.0, 0.0, 0.0, 0.0};
  if (e_on) {
    pf_x = bc.cam_x[tid]; pf_g = bc.gc[tid];
    if (bc.cam_free != nullptr) pf_free = bc.cam_free[tid / 6];
  }
  if (tid < bc.C) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pf_in[q] = bc.intr[4 * tid + q];
  }
  if (tid == 0) { pf_s[0] = bc.scal[0]; pf_s[1] = bc.scal[1]; pf_s[2] = bc.scal[2]; pf_s[3] = *bc.gmax_p; }
  const double my_scale = e_on ? scl[tid] : 0.0;          // (scl, xB: taken before the back-substitution reuses the head of the LDS)
  const double my_xb = (tid >= nA && e_on) ? xB[tid - nA] : 0.0;
  __syncthreads();
  RSBA_BORDER_STAMP(13);
  double* xs = BackSubstituteBlocksWaves(nA, A_generic, lds, /*y_in_place=*/true);
  RSBA_BORDER_STAMP(19);
  int ok = 1;
  if (tid == 0) { bc.res[RES_STALL] = 0.0; ok = (__hip_atomic_load(bc.chol_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 && sb_ok != 0) ? 1 : 0; }
  if (bc.trace && tid == 0) bc.trace[14] = wall_clock64();
  {
    const double xsol = e_on ? (tid < nA ? xs[tid] : my_xb) : 0.0;
    __syncthreads();   // (xs has been read: the scratch below may overlap the back-substitution's arrays)
    double* scr = lds + 2048;
    double* s_xc = lds + 2048 + 4 * 512;
    double d2 = 0.0, x2 = 0.0, xc2 = 0.0, gm = 0.0;
    if (e_on) {
      const double d = -my_scale * xsol;
      bc.dcam[tid] = d;
      const double xc = pf_x + d;
      bc.cam_c[tid] = xc;
      s_xc[tid] = xc;
      if (pf_free != 0.0) { d2 += d * d; x2 += pf_x * pf_x; xc2 += xc * xc; }
      gm = fmax(gm, fabs(pf_g));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      d2 += __shfl_xor(d2, off, 64); x2 += __shfl_xor(x2, off, 64); xc2 += __shfl_xor(xc2, off, 64); gm = fmax(gm, __shfl_xor(gm, off, 64));
    }
    if (lane == 0) { scr[4 * wave + 0] = d2; scr[4 * wave + 1] = x2; scr[4 * wave + 2] = xc2; scr[4 * wave + 3] = gm; }
    __syncthreads();
    if (tid < bc.C) {
      double cc[CC_STRIDE];
      CameraConstants(s_xc + 6 * tid, pf_in, cc);
      typedef double d2s_t __attribute__((ext_vector_type(2)));
      d2s_t* out2 = reinterpret_cast<d2s_t*>(bc.camc_c + (size_t)tid * CC_STRIDE);
#pragma unroll
      for (int i = 0; i < CC_STRIDE / 2; ++i) { d2s_t v = {cc[2 * i], cc[2 * i + 1]}; out2[i] = v; }
    }
    if (tid == 0) {
      double t4[4] = {scr[0], scr[1], scr[2], scr[3]};
      for (int w8 = 1; w8 < (nt >> 6); ++w8) { t4[0] += scr[4 * w8]; t4[1] += scr[4 * w8 + 1]; t4[2] += scr[4 * w8 + 2]; t4[3] = fmax(t4[3], scr[4 * w8 + 3]); }
      bc.res[RES_COST_X] = 0.5 * pf_s[0];
      bc.res[RES_GMAX] = fmax(pf_s[3], t4[3]);
      bc.res[RES_XNORM2] = pf_s[1] + t4[1];
      bc.res[RES_POINT_FAIL] = pf_s[2];
      bc.res[RES_CHOL_OK] = (ok && pf_s[2] == 0.0) ? 1.0 : 0.0;
      bc.res[RES_STEP2] = t4[0];
      bc.res[RES_XCNORM2] = t4[2];
    }
  }
  if (bc.trace && tid == 0) bc.trace[15] = wall_clock64();
  RSBA_BORDER_STAMP(18);
  solve_done();
#undef RSBA_BORDER_STAMP
}

}  // namespace rsba
