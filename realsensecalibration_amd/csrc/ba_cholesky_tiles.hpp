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
#include <algorithm>
#include <vector>
#include "ba_cholesky_large.hpp"
#include "ba_cholesky_multi.hpp"
#include "ba_cholesky_follow.hpp"

namespace rsba {

#define RSBA_TL 65   // leading dimension of the resident tile (64 x 64, odd stride)

struct TileCholFlags {
  int* tdone;    // [np]            == tag: L11 / T of panel p are in F
  int* xdone;    // [np][nrt]       == tag: X of panel p for tile row I is in F
  int* error;    // != 0: somebody gave up
  int nrt;
  // The chain from one tile column's last factorisation to the next one's first runs through PRIVATE hand-over buffers, one
  // producer and one consumer each, where the data is its own flag (TakeHandOver): the producer just stores, the consumer's
  // threads poll their own elements until none is the sentinel.  No wait for the stores' acknowledgements, no flag, no
  // second trip to memory — a hop of ~1.3 us instead of ~3.  Per tile row / column J (TileHand*):
  //   T2   the factor of tile column J's second diagonal block (DenseLtT) diagonal tile J      -> diagonal tile J + 1
  //   S1   ... of its first | the diagonal tile's rows 32..63 of X       diagonal tile J      -> sub-diagonal tile (J + 1, J)
  //        (the FACTOR, not its inverse: the consumer solves its rows by substitution, TrsmRowsQuad, as fast as the product with
  //        the inverse on the matrix cores — and the inversion, 2.1 us, leaves the chain: it is done behind the hand-over)
  //   XH   X of tile row J for the first panel of column J - 1         tile (J, J - 1)      -> diagonal tile J
  //   AH   tile row J's columns of column J - 1's SECOND panel, final but unsolved (after the first panel's update): the next
  //        diagonal tile solves X = Ahat L22^-T itself the moment L22 arrives (the lesson of ba_cholesky_diag.hpp)
  //                                                                    tile (J, J - 1)      -> diagonal tile J
  // Two sets, used by launch parity: a consumer resets the OTHER set's slots at the start of every launch, a whole launch
  // before they are written again — also behind a launch that gave up half-way.  Everybody else goes through tdone / xdone.
  double* hand = nullptr;    // [2][nrt][kTileHandDoubles]
  int parity = 0;
  int test_stall = 0;              // RSBA_TEST_STALL=3: the diagonal tiles look for their hand-over in the set nobody writes
  const int* tile_map = nullptr;   // [tiles] I << 8 | J of the workgroup with that index (TileOrder); nullptr: row by row
  long long* trace = nullptr;   // RSBA_MC_TRACE=1: [nrt][24] stamps of the diagonal tiles' chain (wall clock, 10 ns)
};

constexpr int kTileHandT2 = 0, kTileHandAH = 1024, kTileHandS1 = 3072, kTileHandXH = 5120, kTileHandDoubles = 7168;

__device__ __forceinline__ double* TileHandSlot(const TileCholFlags& f, int parity, int idx) { return f.hand + ((size_t)parity * f.nrt + idx) * kTileHandDoubles; }
__device__ __forceinline__ bool HandThere(double v) { return __double_as_longlong(v) != -1LL; }   // the sentinel: all bits set (hipMemset 0xff)

// v[u] = buf[tid + 256 u] once the producer's stores have arrived: every thread polls its own elements (agent-scope loads, the
// wavefront leaves together).  false: the budget ran out or somebody gave up — the caller's barrier collects it (TakeFailed).
template <int NV>
__device__ __forceinline__ bool TakeHandOver(const double* __restrict__ buf, double (&v)[NV], const int* error, long long budget) {
  const long long t0 = wall_clock64();
  for (int round = 0;; ++round) {
    bool all = true;
#pragma unroll
    for (int u = 0; u < NV; ++u) { v[u] = __hip_atomic_load(&buf[threadIdx.x + 256 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); all = all && HandThere(v[u]); }
    if (__builtin_amdgcn_ballot_w64(!all) == 0) return true;
    if ((round & 15) == 15 && (wall_clock64() - t0 > budget || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// WaitFlagWG without its acquire fences: this kernel reads what other workgroups wrote with agent-scope loads (TileLoadF, TileLoadT,
// TakeHandOver), element by element past the caches that could hold an old copy.  The fence (buffer_inv sc1) throws away
// the XCD's whole L2 — 300 workgroups at two waits per panel kept every L2 of the chip empty, also for the workgroups on the
// chain (their instruction fetches, their saved registers).
__device__ __forceinline__ bool WaitFlagPlainWG(const int* flag, int tag, const int* error, long long budget) {
  __shared__ int s_ok3;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      __builtin_amdgcn_s_sleep(2);
      if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > budget) { ok = 0; break; }
    }
    s_ok3 = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (the produced data is read with agent-scope loads behind this: ordered after the flag for the compiler and the wavefront)
  return s_ok3 != 0;
}

// A stage's flag, waited for by up to 325 resident tiles at once: polled every ~3 us (the hand-overs above are polled every ~50 ns
// by one or two workgroups each; 325 workgroups polling ONE cache line that often keep its memory channel busy for everybody
// else).  One lane polls, then one acquire for the workgroup (as WaitFlagWG).
__device__ __forceinline__ bool WaitStageCoarseWG(const int* flag, int tag, const int* error, long long budget) {
  __shared__ int s_ok4;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      __builtin_amdgcn_s_sleep(127);
      if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > budget) { ok = 0; break; }
    }
    s_ok4 = ok;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return s_ok4 != 0;
}

// Dense copy of a factored diagonal block for TrsmRowsQuad: e = 32 j + c -> l_cj (c > j) | 1 / l_jj (c == j) | 0
__device__ __forceinline__ double DenseLtT(const double* Lt, const double* invd, int e) {
  const int j = e >> 5, c = e & 31;
  return c > j ? Lt[c * RSBA_PLD + j] : (c == j ? invd[j] : 0.0);
}

// What the roles of the kernel share.  The diagonal tiles' and the sub-diagonal tiles' steps on the chain are OUT OF LINE
// (DiagTileColumn, SubDiagFirstHalf): inlined, their registers (the substitution, the hand-overs' values) pushed the kernel to
// 82 spilled registers, and the trailing update — every tile's bread and butter — reloaded a spilled pointer in front of
// each of its loads, one trip to memory after the other: 7 us per update instead of 2.5.
struct TileCtx {
  int n, m, np, I, J, tag;
  double* F;
  int* ok_flag;
  TileCholFlags f;
  long long budget;
};
struct TileLds {
  double *Tl, *XI, *T, *Pan, *Lt, *XJ, *invd;
  int* flags;   // [0] the factorisation's pivots were positive, [1] a hand-over did not come, [2] wavefronts whose late stores of X are acknowledged
};
__device__ __forceinline__ TileLds TileLdsOf(double* lds) {
  TileLds L;
  L.Tl = lds;                              // the tile, 64 x 65
  L.XI = L.Tl + 64 * RSBA_TL;              // 64 x 33: X of this tile's rows for the current panel
  L.T = L.XI + 64 * RSBA_PLD;              // 32 x 33
  L.Pan = L.T + RSBA_PB * RSBA_PLD;        // 32 x 33 (diagonal tiles)   |  together: XJ, 64 x 33 (the others)
  L.Lt = L.Pan + RSBA_PB * RSBA_PLD;       // 32 x 33 (diagonal tiles)   |
  L.XJ = L.Pan;
  L.invd = L.Lt + RSBA_PB * RSBA_PLD;      // 64
  L.flags = reinterpret_cast<int*>(L.invd + 64);
  return L;
}
__host__ __device__ inline size_t TileCholLdsDoubles() { return (size_t)64 * RSBA_TL + 64 * RSBA_PLD + 3 * RSBA_PB * RSBA_PLD + 64 + 2; }   // (+ 2: four ints of flags)

// Where a tile finds its entries of the system.  fused = 0: in W, as k_sys_build left it (scaled, damped, mirrored, rhs in row n).
// fused = 1: k_sys_build's arithmetic on the way into LDS, straight from the Schur kernel's payload `red` — the (n + 1) x n
// intermediate is neither written nor read back and a launch is gone (10 us of a 1 ms step at 256 cameras).  On the first
// iteration the Jacobi scales are defined here (from diag U): the diagonal tiles store them for the rest of the solve.
// Pipelined above 64 cameras (round 4): the kernel is launched BEFORE the Schur kernel and every tile waits, before it builds its
// entries of the system, for the stage(s) of the Schur elimination that complete them — entry (i, j) of S, i >= j, is final once
// the camera group of column j is published (the finishers write every block and its mirror), the damping diagonal, the gradient
// and the right-hand-side correction of a column come with its group's self tile, and on the first step of a run the Jacobi scale
// of a ROW needs that row's diag U: every self tile (all_diag).  The tiles of the first tile columns start at once, the
// others sleep; a tile column's two panels take ~18 us, a stage arrives every ~25 us: the chain keeps up and ends a few tile
// columns behind the last stage.  ready == nullptr: the sequential schedule (no waits).
struct TileGate {
  const int* ready = nullptr;      // TiledSchur::ready: [1 + g] == tag once camera group g's columns are in S
  int tag = 0;
  int cols = 96;                   // reduced columns per camera group
  const int* all_diag = nullptr;   // first step of a run: == tag once every camera's diag U is written
  int* started_cnt = nullptr;      // "every tile is resident" (see StageGate): device counter, host word, tiles
  int* started_host = nullptr;
  int started_need = 0;
  long long* waited = nullptr;     // += ticks (100 MHz) the LAST diagonal tile slept on its stage: the kernel's span minus this is its own work
  long long budget = 0;
};

struct TileSysSource {
  int fused = 0;
  const double* red = nullptr;
  RedLayout L;
  double* scale_c = nullptr;
  IterParams ip;
  int sym_full = 0;
};

// Which tile the workgroup with index t works on.  With more tiles than CUs, the workgroups t and t + (number of CUs) end up
// on the same CU (measured: RSBA_MC_TRACE=1 lists the pairs), and a diagonal tile that shares its CU with a tile busy
// updating takes ~6 us longer per tile column: row by row, the first eleven diagonal tiles of 256 cameras were paired with
// tiles of the last three tile rows, busy from the first panel to the last.  So: the first `extra` indices go to tiles that
// retire early and are off the chain (I >= J + 2 of the first tile columns), their partners at the end of the launch are the
// tiles of the LAST tile columns (idle until late), everybody else — every early diagonal and sub-diagonal tile — has a CU alone.
inline std::vector<int> TileOrder(int nrt, int num_cus) {
  const int ntiles = nrt * (nrt + 1) / 2, extra = std::max(0, std::min(ntiles - num_cus, ntiles / 2));
  std::vector<int> order;
  order.reserve(ntiles);
  if (extra == 0 || nrt > 255) {
    for (int I = 0; I < nrt; ++I) for (int J = 0; J <= I; ++J) order.push_back(I << 8 | J);
    return order;
  }
  std::vector<char> taken((size_t)nrt * nrt, 0);
  std::vector<int> first, last;
  for (int J = 0; J < nrt && (int)first.size() < extra; ++J)
    for (int I = J + 2; I < nrt && (int)first.size() < extra; ++I) { first.push_back(I << 8 | J); taken[(size_t)I * nrt + J] = 1; }
  for (int J = nrt - 1; J >= 0 && (int)last.size() < (int)first.size(); --J)
    for (int I = nrt - 1; I >= J && (int)last.size() < (int)first.size(); --I)
      if (!taken[(size_t)I * nrt + J]) { last.push_back(I << 8 | J); taken[(size_t)I * nrt + J] = 1; }
  order = first;
  for (int I = 0; I < nrt; ++I) for (int J = 0; J <= I; ++J) if (!taken[(size_t)I * nrt + J]) order.push_back(I << 8 | J);
  order.insert(order.end(), last.begin(), last.end());
  return order;
}

__device__ __forceinline__ void TileStamp(const TileCholFlags& f, int tile, int k) { if (f.trace != nullptr && threadIdx.x == 0) f.trace[tile * 24 + k] = wall_clock64(); }
// a value of L / y into F (real entries only; the rhs row m lands in row n)
__device__ __forceinline__ void TileStoreF(double* __restrict__ F, int n, int m, int gi, int gj, double v) {
  if (gj >= n) return;
  if (gi < n) StoreShared(&F[(size_t)gi * n + gj], v);
  else if (gi == m) StoreShared(&F[(size_t)n * n + gj], v);
}
// L entry (gi, gj) of the padded factor (padded columns: identity, nothing below the diagonal).  No branch around the load, and
// none the compiler could make: a clamped address, the value multiplied by 0 or 1 (a select lets it sink the load into a
// branch, and then the loads of a strip wait for each other — F[0] is finite, the product exact).
__device__ __forceinline__ double TileLoadF(const double* __restrict__ F, int n, int m, int gi, int gj) {
  const bool real = gj < n && (gi < n || gi == m);
  const size_t idx = real ? (size_t)(gi < n ? gi : n) * n + gj : 0;
  return __hip_atomic_load(&F[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (real ? 1.0 : 0.0);
}
// T[r][c] (r >= c) of the panel at kb as the diagonal tile left it in F: transposed above the diagonal of its block, the
// diagonal in row n + 1; identity on the padding
__device__ __forceinline__ double TileLoadT(const double* __restrict__ F, int n, int kb, int r, int c) {
  const bool real = kb + r < n && kb + c < n, low = r > c, dia = r == c;
  const bool use = real && (low || dia);
  const size_t idx = use ? (low ? (size_t)(kb + c) * n + kb + r : (size_t)(n + 1) * n + kb + c) : 0;
  return __builtin_fma(__hip_atomic_load(&F[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), use ? 1.0 : 0.0, (!real && dia) ? 1.0 : 0.0);
}
// tile -= XI XB' (64 x 64, K = 32) and tile[:, 32..63] -= XI XB[32..63]' on the matrix cores; wave w takes rows 16 w .. 16 w + 15
__device__ __forceinline__ void TileUpdateFull(double* Tl, const double* XI, const double* XB) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mi = lane & 15, kk = lane >> 4;
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
}
__device__ __forceinline__ void TileUpdateHalf(double* Tl, const double* XI, const double* XB) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mi = lane & 15, kk = lane >> 4;
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
}
// X = Rows T' for the rows >= rlo of the tile's columns [lc, lc + 32): into XI (zero above rlo), the tile and F
__device__ __forceinline__ void TileFormX(double* __restrict__ F, int n, int m, int I, const TileLds& L, int kb, int lc, int rlo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mi = lane & 15, kk = lane >> 4;
  d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const int row = 16 * wave + mi;
#pragma unroll
  for (int qs = 0; qs < RSBA_PB; qs += 4) {
    const double a = L.Tl[row * RSBA_TL + lc + qs + kk];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, L.T[mi * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, L.T[(16 + mi) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int r = 16 * wave + kk + 4 * tt;
    const bool below = r >= rlo;
    const double x0 = below ? acc0[tt] : 0.0, x1 = below ? acc1[tt] : 0.0;
    L.XI[r * RSBA_PLD + mi] = x0; L.XI[r * RSBA_PLD + 16 + mi] = x1;
    if (below) {
      L.Tl[r * RSBA_TL + lc + mi] = x0; L.Tl[r * RSBA_TL + lc + 16 + mi] = x1;
      TileStoreF(F, n, m, 64 * I + r, kb + mi, x0); TileStoreF(F, n, m, 64 * I + r, kb + 16 + mi, x1);
    }
  }
}
// X (in XI, rows >= rlo) into the tile's columns [lc, lc + 32), into F, and straight to its consumer on the chain
__device__ __forceinline__ void TileSpreadX(double* __restrict__ F, int n, int m, int I, const TileLds& L, int kb, int lc, int rlo, double* hx) {
  for (int e = threadIdx.x + 32 * rlo; e < 64 * 32; e += 256) {
    const int r = e >> 5, cc = e & 31;
    const double x = L.XI[r * RSBA_PLD + cc];
    L.Tl[r * RSBA_TL + lc + cc] = x;
    StoreShared(&hx[e], x);
    TileStoreF(F, n, m, 64 * I + r, kb + cc, x);
  }
}

// ---- The diagonal tile (J, J) from its last-but-one update to the end: the chain.
//   panel 2J - 2   tile -= X X' with X of its rows straight from the sub-diagonal tile (XH)
//   panel 2J - 1   the sub-diagonal tile's rows as handed over (AH) and the factor of the panel's diagonal block (T2 slot) ->
//                  X = Ahat L22^-T by substitution -> tile -= X X'
//   panel 2J       factor the first block, hand L11 on (S1), solve the own rows 32..63 by substitution, hand them on, update
//   panel 2J + 1   factor the second block — wavefront 1 inverts the FIRST one meanwhile, stores T and raises the first half's
//                  flags for everybody off the chain — hand L22 on (T2), invert, store, publish
// Measured and dropped (256 cameras, same box, 460 us as it stands): the last update split around the first factorisation (only the
// first diagonal block before it, the rows 32..63 next to it on the wavefronts 1 to 3): 495; the first block's inverse next to the
// substitution (wavefront 3) instead of next to the second factorisation: the substitution's barrier waits 1.5 us for it; the
// inverse inlined instead of called: 656; the factorisation inlined: 546; the rows 32..63 eliminated WITH the first block (64 lanes,
// no substitution): the factorisation 1.1 us longer, 535.
// Buffers (32 x 33 each): first half Pan | T (scratch, then L11 transposed for the substitution) | Lt; second half Pan | XI rows
// 0..31 (scratch, then T) | XI rows 32..63 (Lt), while T = inverse of the first block's Lt.  false: a hand-over did not come.
static __device__ __noinline__ bool DiagTileColumn(const TileCtx& c, lds_double* lds_base) {
  // (the LDS base comes as an argument: `extern __shared__` in a function that is not a kernel is a look-up in a table in
  //  memory, redone wherever the compiler finds it convenient — seen in the middle of the substitution, with a wait for every
  //  store and load in flight in front of it)
  const TileLds L = TileLdsOf((double*)lds_base);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = c.n, m = c.m, np = c.np, J = c.J, tag = c.tag;
  const TileCholFlags f = c.f;   // (copies: the context itself lives in private memory, and a load from there in the middle
  double* F = c.F;               //  of a run of loads or stores waits for all of them)
  int* ok_flag = c.ok_flag;
  const long long budget = c.budget;
  if (J > 0) {
    {
      double v[8];
      if (!TakeHandOver<8>(TileHandSlot(f, f.parity ^ f.test_stall, J) + kTileHandXH, v, f.error, budget)) L.flags[1] = 1;
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = tid + u * 256; L.XI[(e >> 5) * RSBA_PLD + (e & 31)] = v[u]; }
    }
    __syncthreads();
    if (L.flags[1]) return false;
    TileUpdateFull(L.Tl, L.XI, L.XI);
    __syncthreads();
    TileStamp(f, J, 0);
    {
      double av[8], tv[4];
      if (!TakeHandOver<8>(TileHandSlot(f, f.parity, J) + kTileHandAH, av, f.error, budget)) L.flags[1] = 1;
      TileStamp(f, J, 10);
      if (!TakeHandOver<4>(TileHandSlot(f, f.parity, J - 1) + kTileHandT2, tv, f.error, budget)) L.flags[1] = 1;
      TileStamp(f, J, 11);
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = tid + u * 256; L.XJ[(e >> 5) * RSBA_PLD + (e & 31)] = av[u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; L.T[(e >> 5) * RSBA_PLD + (e & 31)] = tv[u]; }
    }
    __syncthreads();
    if (L.flags[1]) return false;
    TileStamp(f, J, 12);
    TrsmRowsQuad(L.XJ, RSBA_PLD, L.T, L.XI, tid, 64);
    __syncthreads();
    TileStamp(f, J, 1);
    TileUpdateFull(L.Tl, L.XI, L.XI);
    __syncthreads();
  }
  const bool two = 2 * J + 1 < np;
  if (!two) {
    // the last, one-panel column: factor and invert in one go, publish; X of the rows below (the right-hand side's) as everybody's
    const int p = 2 * J, kb = p * RSBA_PB;
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) { const int r = e >> 5, cc = e & 31; L.Pan[r * RSBA_PLD + cc] = L.Tl[r * RSBA_TL + cc]; }
    __syncthreads();
    if (wave == 0) {
      const bool good = DiagFactorInverseCall((lds_double*)L.Pan, RSBA_PB, (lds_double*)L.T, (lds_double*)L.Lt, (lds_double*)L.invd, lane);
      if (lane == 0) L.flags[0] = good ? 1 : 0;
    }
    __syncthreads();
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) {
      const int r = e >> 5, cc = e & 31;
      if (kb + r < n && kb + cc < n) StoreShared(&F[(size_t)(kb + r) * n + kb + cc], cc > r ? L.T[cc * RSBA_PLD + r] : L.Pan[r * RSBA_PLD + cc]);
    }
    if (tid < RSBA_PB && kb + tid < n) StoreShared(&F[(size_t)(n + 1) * n + kb + tid], L.invd[tid]);
    if (tid == 0 && !L.flags[0]) __hip_atomic_store(ok_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PublishFlagWG(f.tdone + p, tag);
    TileFormX(F, n, m, J, L, kb, 0, RSBA_PB);
    PublishFlagWG(f.xdone + (size_t)p * f.nrt + J, tag);
    return true;
  }
  double xr[8];   // (wavefronts 2 and 3: X of the first half, on its way out while wavefront 0 factors the second block)
  for (int hf = 0; hf < 2; ++hf) {
    const int p = 2 * J + hf, kb = p * RSBA_PB, lc = 32 * hf;
    double* Tscr = hf ? L.XI : L.T;
    double* Ltx = hf ? L.XI + 32 * RSBA_PLD : L.Lt;
    double* ivx = L.invd + 32 * hf;
    TileStamp(f, J, hf ? 6 : 2);
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) { const int r = e >> 5, cc = e & 31; L.Pan[r * RSBA_PLD + cc] = L.Tl[(lc + r) * RSBA_TL + lc + cc]; }
    // (first half: the tile's rows 32..63 of these columns FOLLOW the factorisation — lanes 32..63 of the factoring wavefront, DiagFactorFollow —
    //  and are X when it returns: no substitution behind it, 2.8 us of every tile column's 18.5)
    if (hf == 0) { for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) { const int r = e >> 5, cc = e & 31; L.XI[(32 + r) * RSBA_PLD + cc] = L.Tl[(32 + r) * RSBA_TL + cc]; } }
    __syncthreads();
    if (wave == 0) {
      const bool good = hf == 0 ? DiagFactorFollowACall((lds_double*)L.Pan, (lds_double*)Tscr, (lds_double*)Ltx, (lds_double*)ivx, (lds_double*)(L.XI + 32 * RSBA_PLD), lane)
                                : DiagFactorOnlyCall((lds_double*)L.Pan, RSBA_PB, (lds_double*)Tscr, (lds_double*)Ltx, (lds_double*)ivx, lane);   // (inlined here: slower, 546 vs 520 us)
      if (lane == 0) L.flags[0] = good ? 1 : 0;
    } else if (wave == 1 && hf == 1) {
      // the first block's inverse, its place in F, the first half's flags (every wavefront's stores of L11 and X were
      // acknowledged before the barrier above)
      DiagInverseCall((lds_double*)L.T, (const lds_double*)L.Lt, (const lds_double*)L.invd, lane);
      __builtin_amdgcn_wave_barrier();
      const int kb1 = kb - RSBA_PB;
      for (int e = lane; e < RSBA_PB * RSBA_PB; e += 64) {
        const int r = e >> 5, cc = e & 31;
        if (cc > r && kb1 + cc < n) StoreShared(&F[(size_t)(kb1 + r) * n + kb1 + cc], L.T[cc * RSBA_PLD + r]);
      }
      if (lane < RSBA_PB && kb1 + lane < n) StoreShared(&F[(size_t)(n + 1) * n + kb1 + lane], L.invd[lane]);
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) {
        // (X of the rows 32..63 went out behind the barrier, from wavefronts 2 and 3: the inversion above takes longer than their
        // stores, this wait is a formality)
        while (__hip_atomic_load(&L.flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 2) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(f.tdone + p - 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(f.xdone + (size_t)(p - 1) * f.nrt + J, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (wave >= 2 && hf == 1) {
      // X of the first half: to the sub-diagonal tile (behind L11's 1024 in S1) and into F
      double* hx = TileHandSlot(f, f.parity, J) + kTileHandS1;
      const int kb1 = kb - RSBA_PB;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = 1024 + (tid - 128) + 128 * u, r = e >> 5, cc = e & 31;
        StoreShared(&hx[e], xr[u]);
        TileStoreF(F, n, m, 64 * J + r, kb1 + cc, xr[u]);
      }
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) __hip_atomic_fetch_add(&L.flags[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    TileStamp(f, J, hf ? 7 : 3);
    {
      // the factor first, to whoever waits on the chain: the next diagonal tile (second half), the sub-diagonal tile (first)
      double* hs = TileHandSlot(f, f.parity, J) + (hf ? kTileHandT2 : kTileHandS1);
      const bool wanted = hf == 1 ? 64 * (J + 1) < m : true;
      for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) {
        const double v = DenseLtT(Ltx, ivx, e);
        if (wanted) StoreShared(&hs[e], v);
      }
    }
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) {
      const int r = e >> 5, cc = e & 31;
      if (cc <= r && kb + r < n) StoreShared(&F[(size_t)(kb + r) * n + kb + cc], L.Pan[r * RSBA_PLD + cc]);
    }
    if (tid == 0 && !L.flags[0]) __hip_atomic_store(ok_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hf == 1) {
      // nothing of this tile lies below the block: no X, nobody waits for this tile's xdone; what is left is T for the others
      if (wave == 0) DiagInverseCall((lds_double*)Tscr, (const lds_double*)Ltx, (const lds_double*)ivx, lane);
      __syncthreads();
      for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) {
        const int r = e >> 5, cc = e & 31;
        if (cc > r && kb + cc < n) StoreShared(&F[(size_t)(kb + r) * n + kb + cc], Tscr[cc * RSBA_PLD + r]);
      }
      if (tid < RSBA_PB && kb + tid < n) StoreShared(&F[(size_t)(n + 1) * n + kb + tid], ivx[tid]);
      PublishFlagWG(f.tdone + p, tag);
      TileStamp(f, J, 8);
      break;
    }
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += 256) L.XI[(e >> 5) * RSBA_PLD + (e & 31)] = 0.0;   // rows above the block: no X (rows 32..63: the followers')
    if (tid == 0) L.flags[2] = 0;
    __syncthreads();
    // X (XI rows 32..63) leaves from registers of the wavefronts 2 and 3 in the second half: nothing waits for its stores here
    if (wave >= 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = 1024 + (tid - 128) + 128 * u; xr[u] = L.XI[(e >> 5) * RSBA_PLD + (e & 31)]; }
    }
    TileStamp(f, J, 4);
    TileUpdateHalf(L.Tl, L.XI, L.XI);
    // every wavefront's stores so far are acknowledged behind this barrier: wavefront 1 raises the first half's flags in the second
    __builtin_amdgcn_s_waitcnt(0);
    TileStamp(f, J, 5);
    __syncthreads();
  }
  return true;
}

// ---- The sub-diagonal tile (J + 1, J) in the first half of its column, on the chain: L11 (transposed, with the inverse pivots)
// straight from the diagonal tile, its 64 rows solved by substitution (L11 comes ~3 us before the diagonal tile's rows of X:
// the substitution does not wait for them), X on to the next diagonal tile (XH), the update of its columns 32..63, those
// columns — final but unsolved — on to the next diagonal tile as well (AH); xdone for everybody else at the end.
static __device__ __noinline__ bool SubDiagFirstHalf(const TileCtx& c, lds_double* lds_base) {
  const TileLds L = TileLdsOf((double*)lds_base);
  const int tid = threadIdx.x, n = c.n, m = c.m, I = c.I, J = c.J, p = 2 * c.J, kb = p * RSBA_PB, tag = c.tag;
  const TileCholFlags f = c.f;
  double* F = c.F;
  const long long budget = c.budget;
  TileStamp(f, I, 16);
  {
    double v[4];
    if (!TakeHandOver<4>(TileHandSlot(f, f.parity, J) + kTileHandS1, v, f.error, budget)) L.flags[1] = 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; L.T[(e >> 5) * RSBA_PLD + (e & 31)] = v[u]; }
  }
  __syncthreads();
  if (L.flags[1]) return false;
  TileStamp(f, I, 13);
  TrsmRowsQuad(L.Tl, RSBA_TL, L.T, L.XI, tid, 64);
  TileStamp(f, I, 14);
  {
    double v[4];
    if (!TakeHandOver<4>(TileHandSlot(f, f.parity, J) + kTileHandS1 + 1024, v, f.error, budget)) L.flags[1] = 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; L.XJ[(32 + (e >> 5)) * RSBA_PLD + (e & 31)] = v[u]; }
  }
  __syncthreads();
  if (L.flags[1]) return false;
  TileStamp(f, I, 15);
  TileSpreadX(F, n, m, I, L, kb, 0, 0, TileHandSlot(f, f.parity, I) + kTileHandXH);
  TileStamp(f, I, 17);
  TileUpdateHalf(L.Tl, L.XI, L.XJ);
  TileStamp(f, I, 18);
  __syncthreads();
  double* ah = TileHandSlot(f, f.parity, I) + kTileHandAH;
  for (int e = tid; e < 64 * 32; e += 256) StoreShared(&ah[e], L.Tl[(e >> 5) * RSBA_TL + 32 + (e & 31)]);
  TileStamp(f, I, 9);
  PublishFlagWG(f.xdone + (size_t)p * f.nrt + I, tag);
  return true;
}

__global__ void __launch_bounds__(256, 2)
k_chol_tiles_persistent(int n, const double* __restrict__ W /* (n + 1) x n: scaled, damped system + rhs row (k_sys_build) */,
                        double* __restrict__ F /* (n + 2) x n */, int* __restrict__ ok_flag, TileCholFlags f, int tag,
                        double* __restrict__ res, TileSysSource src, TileGate gate) {
  extern __shared__ double lds[];
  const TileLds L = TileLdsOf(lds);
  const int tid = threadIdx.x;
  const int m = (n + RSBA_PB - 1) / RSBA_PB * RSBA_PB, np = m / RSBA_PB;
  // tile (I, J), I >= J, from the linear index
  const int t = blockIdx.x;
  int I, J;
  if (f.tile_map != nullptr) {
    const int code = f.tile_map[t];
    I = code >> 8; J = code & 255;
  } else {
    I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (I * (I + 1) / 2 > t) --I;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    J = t - I * (I + 1) / 2;
  }
  const int r0 = 64 * I, c0 = 64 * J;
  // "every tile is resident" (pipelined schedule, first step of a run): counted by EVERY workgroup of the launch, also the one
  // that has nothing to do
  if (RSBA_EXP(gate.ready != nullptr) && tid == 0 && gate.started_host != nullptr &&
      __hip_atomic_fetch_add(gate.started_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gate.started_need - 1) {
    __hip_atomic_store(gate.started_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(gate.started_host, gate.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (c0 >= m) return;   // (the tile row of the rhs row reaches one column past the matrix when m is a multiple of 64)
  if (tid == 0) L.flags[1] = 0;
  if (f.trace != nullptr && tid == 0) {
    // (trace: where this tile runs — XCC | SE | CU — behind the diagonal tiles' stamps)
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    f.trace[(size_t)(f.nrt + 1) * 24 + t] = ((long long)(xcc & 0xf) << 32) | hw;
  }
  {
    // the hand-over slots this tile consumes, in the set the NEXT launch uses: back to the sentinel
    const double sent = __longlong_as_double(-1LL);
    if (I == J) {
      double* o = TileHandSlot(f, f.parity ^ 1, I);
      for (int e = tid; e < 2048; e += 256) { StoreShared(&o[kTileHandAH + e], sent); StoreShared(&o[kTileHandXH + e], sent); }
      if (J > 0) { double* q = TileHandSlot(f, f.parity ^ 1, J - 1); for (int e = tid; e < 1024; e += 256) StoreShared(&q[kTileHandT2 + e], sent); }
    } else if (I == J + 1) {
      double* q = TileHandSlot(f, f.parity ^ 1, J);
      for (int e = tid; e < 2048; e += 256) StoreShared(&q[kTileHandS1 + e], sent);
    }
  }
  // (gated = beside the Schur kernel, whose hit loops run at priority 0: the factorisation is a latency chain of a few instructions
  //  between trips to memory — served first)
  if (RSBA_EXP(gate.ready != nullptr)) __builtin_amdgcn_s_setprio(3);
  bool gate_stalled = false;
  int stall_code = 1;   // what RES_STALL carries: 1 a hand-over inside the factorisation, 2 the wait for every camera's diag U, 10 + g the wait for stage g
  if (RSBA_EXP(gate.ready != nullptr)) {
    const long long gb = gate.budget > 0 ? gate.budget : RSBA_STALL_TICKS;
    const long long tw0 = wall_clock64();
    if (gate.all_diag != nullptr && !WaitStageCoarseWG(gate.all_diag, gate.tag, f.error, gb)) { gate_stalled = true; stall_code = 2; }
    // (columns at or beyond n are identity padding; the right-hand-side row's entries come with their column's group)
    const int g_lo = min(c0, n - 1) / gate.cols, g_hi = min(c0 + 63, n - 1) / gate.cols;
    for (int g = g_lo; g <= g_hi && !gate_stalled; ++g)
      if (!WaitStageCoarseWG(gate.ready + 1 + g, gate.tag, f.error, gb)) { gate_stalled = true; stall_code = 10 + g; }
    if (gate.waited != nullptr && tid == 0 && I == J && 64 * (J + 1) >= m) *gate.waited += wall_clock64() - tw0;
  }
  // entry (gi, gj) of the padded system; row m is the right-hand side
  auto sysv = [&](int gi, int gj) {
    if (gi > m || gj >= m) return 0.0;
    if (gi == m) return gj < n ? W[(size_t)n * n + gj] : 0.0;
    if (gi >= n || gj >= n) return gi == gj ? 1.0 : 0.0;
    return W[(size_t)gi * n + gj];
  };
  if (src.fused) {
    // k_sys_build's arithmetic (ba_cholesky_large.hpp), entry by entry: the scales of the tile's rows and columns first
    const double* __restrict__ red = src.red;
    const RedLayout RL = src.L;
    double* sc_r = L.XI;        // 64 + 64 scales (XI is free until the first panel)
    double* sc_c = L.XI + 64;
    if (tid < 128) {
      const int g = (tid < 64 ? r0 : c0 - 64) + tid;
      double v = 0.0;
      if (g < n) v = src.ip.first ? (src.ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(red[RL.diagU() + g])) : 1.0) : src.scale_c[g];
      L.XI[tid] = v;
      if (src.ip.first && I == J && tid < 64 && g < n) src.scale_c[g] = v;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63, gi = r0 + r, gj = c0 + c;
      double v;
      if (gi > m || gj >= m) v = 0.0;
      else if (gi == m) v = gj < n ? sc_c[c] * (red[RL.gc() + gj] + red[RL.corr() + gj]) : 0.0;
      else if (gi >= n || gj >= n) v = gi == gj ? 1.0 : 0.0;
      else {
        const int bi = gi / 6, bj = gj / 6;
        const bool upper = src.sym_full || (bi < bj) || (bi == bj && gi <= gj);
        const double raw = upper ? red[RL.S() + (size_t)gi * n + gj] : red[RL.S() + (size_t)gj * n + gi];
        v = raw * (sc_r[r] * sc_c[c]);
        if (gi == gj) v += fmin(fmax(sc_r[r] * sc_r[r] * red[RL.diagU() + gi], src.ip.min_lm_diagonal), src.ip.max_lm_diagonal) / src.ip.radius;
      }
      L.Tl[r * RSBA_TL + c] = v;
    }
  } else {
    for (int e = tid; e < 64 * 64; e += 256) { const int r = e >> 6, c = e & 63; L.Tl[r * RSBA_TL + c] = sysv(r0 + r, c0 + c); }
  }
  if (I == 0 && J == 0 && tid == 0) { __hip_atomic_store(ok_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); res[RES_STALL] = 0.0; }   // (a stall, half a second later, sets it)
  __syncthreads();
  // (gated: a tile waits for tiles that wait for stages — the budget of a hand-over covers a stage's wait as well)
  const long long budget = RSBA_EXP(gate.ready != nullptr) ? 3 * RSBA_STALL_TICKS : RSBA_STALL_TICKS;
  const TileCtx ctx{n, m, np, I, J, tag, F, ok_flag, f, budget};
  bool stalled = gate_stalled;
  const int plast = min(2 * J + 1, np - 1);
  for (int p = 0; p <= plast && !stalled; ++p) {
    const int Jp = p >> 1, hf = p & 1, kb = p * RSBA_PB, lc = 32 * hf;   // lc: the panel's first column inside a tile of column Jp
    const bool second_follows = hf == 0 && 2 * J + 1 < np;   // this tile column has a second panel and this is its first
    // (the thread's index, opaque per iteration: otherwise the loads' addresses are computed once in front of the loop, do not fit
    //  into the registers that survive the calls below, and come back from private memory in the middle of every run of loads)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (I == J && p == max(0, 2 * J - 2)) {
      // the diagonal tile's last steps, out of line
      stalled = !DiagTileColumn(ctx, (lds_double*)lds);
      break;
    }
    if (J == Jp) {
      if (I == J + 1 && second_follows) {
        if (!SubDiagFirstHalf(ctx, (lds_double*)lds)) stalled = true;
        continue;
      }
      // ---- column tile: T from the diagonal tile through F, X = Rows T' on the matrix cores
      if (!WaitFlagPlainWG(f.tdone + p, tag, f.error, budget)) { stalled = true; break; }
      // first half: the diagonal tile published its rows 32..63 of X together with T (one flag covers both), and this tile's
      // columns 32..63 need them right after its own X — fetched in the same trip to memory.  (Two straight-line variants: a
      // uniform branch around some of the loads makes each of them wait for the ones before.)
      if (second_follows) {
        double tv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = tid + u * 256, r = e >> 5, c = e & 31;
          tv[u] = TileLoadT(F, n, kb, c, r);   // (T[c][r]: lanes along the row of F it is stored in, transposed on the way into LDS)
          xv[u] = TileLoadF(F, n, m, c0 + 32 + r, kb + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = tid + u * 256;
          L.T[(e & 31) * RSBA_PLD + (e >> 5)] = tv[u];
          L.XJ[(32 + (e >> 5)) * RSBA_PLD + (e & 31)] = xv[u];
        }
      } else {
        double tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; tv[u] = TileLoadT(F, n, kb, e & 31, e >> 5); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = tid + u * 256; L.T[(e & 31) * RSBA_PLD + (e >> 5)] = tv[u]; }
      }
      __syncthreads();
      TileFormX(F, n, m, I, L, kb, lc, 0);
      PublishFlagWG(f.xdone + (size_t)p * f.nrt + I, tag);
      // ---- first half done: this tile's columns 32..63 take -X_I X_Jp[rows 32..63]'  (the barrier in PublishFlagWG: XI complete)
      if (second_follows) {
        TileUpdateHalf(L.Tl, L.XI, L.XJ);
        __syncthreads();
      }
    } else {
      // ---- trailing tile: tile -= X_I X_J'
      const bool sdt = I == J + 1 && Jp == J - 1;   // (trace: the sub-diagonal tile's last two updates)
      if (sdt) TileStamp(f, I, hf ? 21 : 19);
      if (!WaitFlagPlainWG(f.xdone + (size_t)p * f.nrt + I, tag, f.error, budget)) { stalled = true; break; }
      if (I != J && !WaitFlagPlainWG(f.xdone + (size_t)p * f.nrt + J, tag, f.error, budget)) { stalled = true; break; }
      if (sdt) TileStamp(f, I, hf ? 23 : 20);
      if (I != J) {
        double xi[8], xj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + u * 256, r = e >> 5, c = e & 31;
          xi[u] = TileLoadF(F, n, m, r0 + r, kb + c);
          xj[u] = TileLoadF(F, n, m, c0 + r, kb + c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + u * 256, r = e >> 5, c = e & 31;
          L.XI[r * RSBA_PLD + c] = xi[u];
          L.XJ[r * RSBA_PLD + c] = xj[u];
        }
      } else {
        double xi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = tid + u * 256; xi[u] = TileLoadF(F, n, m, r0 + (e >> 5), kb + (e & 31)); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = tid + u * 256; L.XI[(e >> 5) * RSBA_PLD + (e & 31)] = xi[u]; }
      }
      __syncthreads();
      TileUpdateFull(L.Tl, L.XI, (I == J) ? L.XI : L.XJ);
      __syncthreads();
    }
  }
  if (stalled && tid == 0) {
    // (the first to give up names its wait; the others follow the error flag)
    const bool first_out = __hip_atomic_exchange(f.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    __hip_atomic_store(ok_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (first_out || res[RES_STALL] == 0.0) res[RES_STALL] = (double)stall_code;   // the host repeats the step (sequential schedule / multi-launch factorisation)
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
  if (stalled) { if (tid == 0) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); res[RES_STALL] = 5.0; } return; }
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
// order, and never waits for a store: per block it subtracts the nearest strips' blocks itself (L(i, b) x_i for b < i <= top
// of its column range + RSBA_BSC_LAG: at most LAG + 2 products of 32 x 32, their entries of L prefetched one block ahead, x of
// those blocks in registers) and x_b = T_b' y_b; one wavefront sends x_b out.  Everything FARTHER above is the helpers'
// business: helper h owns 96 columns of y and applies the strip of every block i more than RSBA_BSC_LAG above its range as soon
// as x_i arrives (strips prefetched three ahead, as in k_backsub_multi), then hands its slice of y over.  x and the slices travel
// with the data as its own flag (below).  Same sums per entry in a fixed order: bitwise reproducible.
// What bounds it (84 us at 256 cameras, 48 blocks): the chain's own loads — one CU pulls ~70 GB/s through its L1, whether the
// lines sit in its XCD's L2 or not (workgroups that touched the lines ahead of the chain changed nothing) — 0.4 us per block,
// the two products with their two barriers 0.6, the hand-over of a slice every three blocks.  Measured: with LAG = 3 the chain
// read 40 KB per block and took 117 us, LAG = 2: 110, LAG = 1: 106 (all with flags); with the sentinel hand-overs LAG = 1: 94,
// LAG = 0: 96.5 (LAG = 2 in the end: the loop 68.6 against 70.5 us, the helpers' slices there when asked for; 3: 75); T_b fetched along the rows of F it is stored in and transposed through LDS (fetch_plain): 94 -> 84; four barriers per block instead of two: no difference (the row groups' sums met in LDS behind a barrier each);
// a column's eight row groups in neighbouring lanes (the loads then ask for every line twice): 164 us.
// ------------------------------------------------------------------------------------------------
#define RSBA_BSC_LAG 2
__global__ void __launch_bounds__(256)
k_backsub_chain(int C, const double* __restrict__ red, RedLayout L, const double* __restrict__ F, double* __restrict__ xs2 /* [2][m] */,
                const double* __restrict__ scale_c, const double* __restrict__ cam_x, double* __restrict__ cam_c,
                const double* __restrict__ intr, double* __restrict__ camc_c, double* __restrict__ dcam,
                const double* __restrict__ gmax_p, double* __restrict__ res, const int* __restrict__ ok_flag,
                const double* __restrict__ cam_free, double* __restrict__ ys2 /* [2][96 helpers] */, int* __restrict__ error, int parity) {
  const int n = L.nc, tid = threadIdx.x, w = blockIdx.x;
  const int m = (n + RSBA_PB - 1) / RSBA_PB * RSBA_PB, nblk = m / RSBA_PB, btop = nblk - 1;
  const int nhelp = (nblk + 2) / 3;
  // x and the helpers' slices of y travel as in the tiled factorisation (TakeHandOver): the data is its own flag — the sentinel
  // until the producer's store arrives — in two sets by launch parity, the chain resetting the other set at the start.  With
  // flags the way from a solved x_b to the slice that needs it was six microseconds (the store's acknowledgement, the flag, the
  // helper's poll and fetch, its store, acknowledgement and flag, the chain's poll and fetch): three blocks of the chain.
  double* __restrict__ xsol = xs2 + (size_t)parity * m;
  double* __restrict__ ys = ys2 + (size_t)parity * nhelp * 96;
  const long long budget = RSBA_STALL_TICKS;
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
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
      if (tid < RSBA_PB) {
        double v = 0.0;
        if (32 * b + tid < n) {
          const long long t0 = wall_clock64();
          for (int round = 0;; ++round) {
            v = __hip_atomic_load(&xsol[32 * b + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (HandThere(v)) break;
            if ((round & 15) == 15 && (wall_clock64() - t0 > budget || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { s_fail = 1 + b; v = 0.0; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        xb[tid] = v;
      }
      __syncthreads();
      if (s_fail) { stalled = true; break; }
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
    if (stalled) { if (tid == 0) { if (__hip_atomic_exchange(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) res[RES_STALL] = 1000.0 + 100.0 * h + s_fail; } return; }
    if (tid < 96) StoreShared(&ys[96 * h + tid], yown[tid]);
    return;
  }
  {
    // the other set back to the sentinel (nobody touches it during this launch)
    const double sent = __longlong_as_double(-1LL);
    double* xo = xs2 + (size_t)(parity ^ 1) * m;
    double* yo = ys2 + (size_t)(parity ^ 1) * nhelp * 96;
    for (int i = tid; i < m; i += 256) StoreShared(&xo[i], sent);
    for (int i = tid; i < nhelp * 96; i += 256) StoreShared(&yo[i], sent);
  }
  // ---- the chain
  __shared__ double yl[96];                            // the current helper range's slice of y
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
      // T[i][cc] for i = this thread's c, cc = 4 p8 + r (see fetch_plain): at F[kb + cc][kb + i] for i > cc, the diagonal in row n + 1;
      // identity on the padding
      const int i = c, cc = 4 * p8 + r;
      tn[r] = b < 0 ? 0.0 : ((kb + i < n && kb + cc < n) ? (i > cc ? F[(size_t)(kb + cc) * n + kb + i] : (i == cc ? F[(size_t)(n + 1) * n + kb + cc] : 0.0))
                                                          : (i == cc ? 1.0 : 0.0));
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
    // T: the block is stored TRANSPOSED (T[i][cc] at row kb + cc, column kb + i of F), and the product below wants T[4 p8 + r][c]
    // per thread — fetched that way every lane of a load sits in a row of F of its own (64 lines per instruction, 0.3 us per
    // block).  So the roles are swapped for the fetch: this thread brings T[i = c][cc = 4 p8 + r] (32 consecutive columns per half
    // wavefront), and the block goes through LDS once, transposed, beside the first product.  Raw here: row kb + cc holds T[i][cc]
    // for i > cc, the diagonal comes from row n + 1 (this thread's: column c), the rest of the row belongs to other blocks — picked
    // when the values are stored to LDS (picking here made the compiler wait for the loads here).
    const unsigned o_t = (kb + 4u * p8) * rowb + 8u * (kb + c);
    dgn = *reinterpret_cast<const double*>(Fb + (size_t)((unsigned)(n + 1) * rowb + 8u * (kb + c)));
#pragma unroll
    for (int r = 0; r < 4; ++r) tn[r] = *reinterpret_cast<const double*>(Fb + (size_t)(o_t + (unsigned)r * rowb));
  };
  auto fetch = [&](int b, double (&ln)[kNear][4], double (&tn)[4], double& dgn) {
    const int b1 = min(btop, 3 * (b / 3) + 2);
    if (b >= 0 && 32 * (min(btop, b1 + RSBA_BSC_LAG) + 1) <= n) { fetch_plain(b, ln, tn, dgn); return true; }
    fetch_checked(b, ln, tn);
    return false;
  };
  double lcur[kNear][4], tcur[4], lnext[kNear][4], tnext[4], dgcur = 0.0, dgnext = 0.0;
  double xq[kNear][4];                                 // this thread's four rows of x of the last kNear blocks (nearest first)
#pragma unroll
  for (int u = 0; u < kNear; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) xq[u][r] = 0.0;
  __shared__ double p4[4][RSBA_PB], q4[4][RSBA_PB];   // the wavefronts' partial sums of the two products of a block
  __shared__ double Tt[RSBA_PB * RSBA_PLD];           // T_b: Tt[cc * RSBA_PLD + i] = T[i][cc]
  bool raw_cur = fetch(btop, lcur, tcur, dgcur), raw_next = false;
  bool stalled = false;
  double ypre = __longlong_as_double(-1LL);            // the next range's slice as last seen (the sentinel: not asked for yet / not there yet)
  for (int b = btop; b >= 0; --b) {
    const int h = b / 3, b1 = min(btop, 3 * h + 2), ihi = min(btop, b1 + RSBA_BSC_LAG);
    raw_next = fetch(b - 1, lnext, tnext, dgnext);
    if (b == b1) {
      // a new range of columns: its helper has applied every strip above b1 + LAG (or there are none) and stored its slice —
      // asked for one block ago (ypre), asked again here until it is there
      if (tid < 96) {
        double v = ypre;
        const long long t0 = wall_clock64();
        for (int round = 0; !HandThere(v); ++round) {
          if ((round & 15) == 15 && (wall_clock64() - t0 > budget || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { s_fail = 1; v = 0.0; break; }
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(&ys[96 * h + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        yl[tid] = v;
      }
      __syncthreads();
      if (s_fail) { stalled = true; break; }
    }
    // the next range's slice: asked for while this range's last block is solved
    if (b == 3 * h && h > 0 && tid < 96) ypre = __hip_atomic_load(&ys[96 * (h - 1) + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Two barriers per block.  The near blocks: sum_i L(i, b)' x_i over b < i <= ihi, with x of the
    // last kNear blocks in registers (this thread's four rows of each); the wavefront's two row groups meet in a lane exchange,
    // the four wavefronts' sums in LDS — and every thread adds up the four partial sums of the rows IT needs next (sixteen
    // LDS reads) instead of waiting for one wavefront to do it behind another barrier.  Every sum in a fixed order.
    {
      double sacc = 0.0;
#pragma unroll
      for (int u = 0; u < kNear; ++u) {
        if (b + 1 + u <= ihi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc += lcur[u][r] * xq[u][r];
        }
      }
      sacc += __shfl_xor(sacc, 32, 64);
      if ((tid & 63) < 32) p4[tid >> 6][c] = sacc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = 4 * p8 + r;   // (this thread fetched T[c][cc])
        Tt[cc * RSBA_PLD + c] = raw_cur ? (c > cc ? tcur[r] : (c == cc ? dgcur : 0.0)) : tcur[r];
      }
    }
    __syncthreads();
    // x_b = T_b' y_b
    {
      double sacc = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * p8 + r;
        const double yv = yl[32 * (b - 3 * h) + i] - (((p4[0][i] + p4[1][i]) + p4[2][i]) + p4[3][i]);
        sacc += Tt[c * RSBA_PLD + i] * yv;
      }
      sacc += __shfl_xor(sacc, 32, 64);
      if ((tid & 63) < 32) q4[tid >> 6][c] = sacc;
    }
    __syncthreads();
#pragma unroll
    for (int u = kNear - 1; u > 0; --u)
#pragma unroll
      for (int r = 0; r < 4; ++r) xq[u][r] = xq[u - 1][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int i = 4 * p8 + r; xq[0][r] = ((q4[0][i] + q4[1][i]) + q4[2][i]) + q4[3][i]; }
    // x_b goes out (no flag behind it: the helpers poll the values themselves)
    if (tid >= 192 && tid < 224) {
      const int t = tid - 192;
      if (32 * b + t < n) StoreShared(&xsol[32 * b + t], ((q4[0][t] + q4[1][t]) + q4[2][t]) + q4[3][t]);
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
  if (stalled) { if (tid == 0) { if (__hip_atomic_exchange(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) res[RES_STALL] = 6.0; } return; }
  // every x is in xsol (this workgroup stored them) — visible to its own plain loads behind the acknowledgements and one acquire
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  int ok = 1;
  if (tid == 0) ok = __hip_atomic_load(ok_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  CameraStepEpilogue(C, red, L, scale_c, xsol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, ok, epi, cam_free);
}

}  // namespace rsba
