// The reduced camera system on SEVERAL workgroups (32 to 64 cameras; the system is padded with identity rows and columns
// to whole 32-wide panels).
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
#ifndef RSBA_MC_NPF
#define RSBA_MC_NPF 3         // 32-column slabs of a 16-row half in flight in the row update (4 spills registers)
#endif
#ifndef RSBA_MC_NPF_FUSED
#define RSBA_MC_NPF_FUSED 1   // slabs per operand stream fetched at the top of the panel by the cross-tile waves; the single-stream row waves keep twice as many (2 -> 25 spilled registers, and a spill costs more than the round trip it hides)
#endif

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
    // ONE acquire for the workgroup, by the wavefront that saw the flag: the invalidate it issues (buffer_inv sc1) empties
    // this CU's L1 and the XCD's L2 of what other workgroups have rewritten — caches all wavefronts of the workgroup share —
    // and the barrier orders everybody's loads behind it.  Every wavefront issuing its own (eight per wait, two waits per
    // panel and workgroup) cost the factorisation ~0.25 us per wait: an acquire fence alone is 90 ns on an idle chip
    // (tools/lat_bench.hip), and they queue.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return s_ok2 != 0;
}

// After the workgroup's agent-scope stores: all of them performed, then the flag.
__device__ __forceinline__ void PublishFlagWG(int* flag, int tag) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void StoreShared(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// BackSubstituteBlocks (ba_cholesky.hpp) with the strip of the NEXT block row in flight while this one is applied: the
// twelve strips of L come back from memory one dependent round trip each (~2.5 us), which was all this phase cost.
// 512 threads, n a multiple of 32 (the padded dimension of the kernel below); same arithmetic, same order.
__device__ __forceinline__ double* BackSubstituteBlocksPrefetch(int n, double* __restrict__ A, double* lds) {
  const int tid = threadIdx.x, nt = blockDim.x;
  double* y = lds;                                 // n
  double* Tb = lds + ((n + 63) & ~63);             // 32 x 33: Tb[i][c] = T[i][c]
  double* xb = Tb + RSBA_PB * RSBA_PLD;            // 32
  double* xw = xb + RSBA_PB;                       // 16 x 32: the wavefronts' partial sums of x_b
  const int kb_last = n - RSBA_PB;
  auto fetch_T = [&](int kb, int slot) {
    const int e = tid + slot * nt, i = e & 31, c = e >> 5;   // (lanes along i: T[i][c] lies in row kb + c of A — whole lines, not a line per lane)
    // T[i][c] for i > c sits at A[kb+c][kb+i]; the diagonal in row n+1
    return (i > c) ? A[(size_t)(kb + c) * n + kb + i] : (i == c ? A[(size_t)(n + 1) * n + kb + c] : 0.0);
  };
  const int q = tid;  // kb <= 352 < nt
  // (two strips ahead instead of one — 96 doubles per thread — measured slower: 29.7 vs 25 us for the phase, twice, before
  //  and after the product above was spread over the workgroup; a strip's loads already keep this CU's address path busy
  //  for ~0.9 us)
  // (round 4, RSBA_TRACE=1, factored -> back-substituted at 64 cameras: ONE barrier per block row instead of four — every
  //  wavefront forming x_b for itself, its entries taken out of the lanes by v_readlane — 24 us; that with two block rows ahead,
  //  three register sets copied round 27, taking turns without copies 28; one ahead, two sets taking turns 26.  Neither the
  //  barriers nor the round trip: the phase moves all of L — 590 KB written by workgroups on other XCDs, every line from memory
  //  — through ONE compute unit, which sustains ~24 GB/s of such loads)
  double tpre[2], lv[RSBA_PB], ln[RSBA_PB];
  // (the first T and strip are asked for BEFORE the right-hand side is waited for and stored: one trip to memory at the head of
  //  the phase instead of two behind each other)
  for (int sl = 0; sl < 2; ++sl) tpre[sl] = fetch_T(kb_last, sl);
#pragma unroll
  for (int c = 0; c < RSBA_PB; ++c) lv[c] = q < kb_last ? A[(size_t)(kb_last + c) * n + q] : 0.0;
  for (int i = tid; i < n; i += nt) y[i] = A[(size_t)n * n + i];
  __syncthreads();
  for (int kb = kb_last; kb >= 0; kb -= RSBA_PB) {
    for (int sl = 0; sl < 2; ++sl) { const int e = tid + sl * nt; Tb[(e & 31) * RSBA_PLD + (e >> 5)] = tpre[sl]; }
    if (kb >= RSBA_PB) {
      const int kn = kb - RSBA_PB;
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) ln[c] = q < kn ? A[(size_t)(kn + c) * n + q] : 0.0;
      for (int sl = 0; sl < 2; ++sl) tpre[sl] = fetch_T(kn, sl);
    }
    __syncthreads();
    // x_b = T_b' y_b on the whole workgroup: thread (c, g) = (tid & 31, tid >> 5) takes rows 2 g, 2 g + 1 of column c, lanes
    // l and l + 32 add up, eight wavefronts' sums meet in LDS — a fixed order.  (32 threads with a 32-long chain of dependent
    // LDS reads and FMAs each took ~1 us per block: half of what this phase cost.)
    {
      const int c = tid & 31, g = tid >> 5;
      double part = 0.0;
      if (g < RSBA_PB / 2) part = Tb[(2 * g) * RSBA_PLD + c] * y[kb + 2 * g] + Tb[(2 * g + 1) * RSBA_PLD + c] * y[kb + 2 * g + 1];
      part += __shfl_xor(part, 32, 64);
      if ((tid & 63) < 32) xw[(tid >> 6) * RSBA_PB + c] = part;
    }
    __syncthreads();
    if (tid < RSBA_PB) {
      double sacc = xw[tid];
      for (int wv = 1; wv < (nt >> 6); ++wv) sacc += xw[wv * RSBA_PB + tid];
      xb[tid] = sacc;
      y[kb + tid] = sacc;
    }
    __syncthreads();
    if (q < kb) {
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) sacc += lv[c] * xb[c];
      y[q] -= sacc;
    }
#pragma unroll
    for (int c = 0; c < RSBA_PB; ++c) lv[c] = ln[c];
    __syncthreads();
  }
  return y;
}

// BackSubstituteBlocksPrefetch rewritten for INSTRUCTION COUNT (round 4).  The phase — 25 us of the 64-camera step's tail, 2.1 us per
// block row — was taken for memory-bound (one compute unit pulling all of L): it is not.  Neither two block rows in flight, nor
// helper workgroups on the same XCD pulling the strips into the shared L2 ahead of it (placement verified with XCC_ID), nor one
// barrier instead of four moved it; the ISA did: ~450 instructions per block row and wavefront — every one of the 32 loads of a
// strip under its own compare / exec-mask save / branch / restore with a 64-bit vector address computation, 64 v_readlane with
// their hazard nops — on two wavefronts per SIMD.  Here: one branch around the strip's loads, their addresses a scalar row base
// plus one lane offset, x_b = T_b' y_b formed by every wavefront for itself (lane (c, h): half of column c's 32 terms, the halves
// meet by a lane exchange — the same operations in every wavefront, the same bits) and handed to its lanes through a
// wavefront-private LDS row read back 16 bytes at a time (broadcast), one barrier per block row, T double-buffered.
// x is returned in its own array.  512 threads, n a multiple of 32, n <= 512.
// y_in_place: the right-hand side already lies at lds[0 .. n) (ba_cholesky_border.hpp); otherwise it is row n of A.
__device__ __forceinline__ double* BackSubstituteBlocksWaves(int n, double* __restrict__ A, double* lds, bool y_in_place = false) {
  typedef double d2b_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int n64 = (n + 63) & ~63;
  double* y = lds;                                 // n
  double* xs = lds + n64;                          // n
  double* Tb0 = xs + n64;                          // 2 x (32 x 33): Tb[i][c] = T[i][c]
  double* xwv = Tb0 + 2 * RSBA_PB * RSBA_PLD + 32 * wave;   // [8][32] a wavefront's copy of x_b (16-byte aligned: offsets are even)
  const int kb_last = n - RSBA_PB;
  auto fetch_T = [&](int kb, int slot) {
    const int e = tid + slot * nt, i = e & 31, c = e >> 5;
    return (i > c) ? A[(size_t)(kb + c) * n + kb + i] : (i == c ? A[(size_t)(n + 1) * n + kb + c] : 0.0);
  };
  const unsigned q = (unsigned)tid, qoff = 8u * (unsigned)tid;  // kb <= 352 < nt
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);   // (gfx9 raw buffer, no swizzle: dword 3 as composable_kernel sets it for gfx90a / gfx94x)
  double ra[RSBA_PB], rb[RSBA_PB], ta[2], tb[2];
  auto load_strip = [&](int kb, double (&r)[RSBA_PB], double (&t)[2]) {
    for (int sl = 0; sl < 2; ++sl) t[sl] = kb >= 0 ? fetch_T(kb, sl) : 0.0;
    if (kb >= 0 && (int)q < kb) {
      // (buffer loads: the matrix as a raw buffer, the row a SCALAR offset that moves on by one row per load, the column one 32-bit
      //  lane offset — one instruction per load; as plain pointers every load came with a 64-bit vector add and two scalar ones.
      //  ONE branch around the 32 loads, not one per load.  Measured and dropped: the loads without any branch — every lane, every
      //  block row — so that the compiler's wait for T leaves the strip in flight: 36 us, three times the bytes)
      int soff = kb * n * (int)sizeof(double);
      const int row_bytes = n * (int)sizeof(double);
#pragma unroll
      for (int cc = 0; cc < RSBA_PB; ++cc) {
        const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, qoff, soff, 0);
        r[cc] = __hiloint2double(v.y, v.x);
        soff += row_bytes;
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < RSBA_PB; ++cc) r[cc] = 0.0;
    }
  };
  auto store_T = [&](int b, const double* t) {
    double* Tn = Tb0 + b * (RSBA_PB * RSBA_PLD);
    for (int sl = 0; sl < 2; ++sl) { const int e = tid + sl * nt; Tn[(e & 31) * RSBA_PLD + (e >> 5)] = t[sl]; }
  };
  const int c = lane & 31, h = lane >> 5;
  int buf = 0;
  auto block_row = [&](int kb, const double (&cur)[RSBA_PB], double (&nxt)[RSBA_PB], double (&tnxt)[2]) {
    const double* Tb = Tb0 + buf * (RSBA_PB * RSBA_PLD);
    load_strip(kb - RSBA_PB, nxt, tnxt);
    // (a wavefront all of whose rows lie at or beyond this block row has nothing to apply x_b to: wavefront 0 alone then forms it)
    const bool wave_on = 64 * wave < kb || wave == 0;
    double part = 0.0;
    if (wave_on) {
#pragma unroll
      for (int i = 0; i < RSBA_PB / 2; ++i) part = fma(Tb[(16 * h + i) * RSBA_PLD + c], y[kb + 16 * h + i], part);
      part += __shfl_xor(part, 32, 64);   // (a + b in one half, b + a in the other: the same bits)
      if (lane < RSBA_PB) xwv[lane] = part;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if ((int)q < kb) {
      double s4[4] = {0.0, 0.0, 0.0, 0.0};   // four chains of eight, added in a fixed order
      const d2b_t* xv = reinterpret_cast<const d2b_t*>(xwv);
#pragma unroll
      for (int k2 = 0; k2 < RSBA_PB / 2; ++k2) {
        const d2b_t v = xv[k2];
        s4[(2 * k2) & 3] = fma(cur[2 * k2], v.x, s4[(2 * k2) & 3]);
        s4[(2 * k2 + 1) & 3] = fma(cur[2 * k2 + 1], v.y, s4[(2 * k2 + 1) & 3]);
      }
      y[q] -= (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    if (tid < RSBA_PB) xs[kb + tid] = part;
    buf ^= 1;
    if (kb >= RSBA_PB) store_T(buf, tnxt);
    __syncthreads();
  };
  load_strip(kb_last, ra, ta);
  if (!y_in_place) { for (int i = tid; i < n; i += nt) y[i] = A[(size_t)n * n + i]; }
  store_T(0, ta);
  __syncthreads();
  for (int kb = kb_last; kb >= 0; kb -= 2 * RSBA_PB) {
    block_row(kb, ra, rb, tb);
    if (kb - RSBA_PB >= 0) block_row(kb - RSBA_PB, rb, ra, ta);
  }
  return xs;
}

#ifdef RSBA_EXPERIMENTAL   // the round-robin factorisation (RSBA_CHOL_DIAG=0): superseded by ba_cholesky_diag.hpp, which keeps this file's helpers
__global__ void __launch_bounds__(512)
k_reduced_system_solve_multi(int C, double* __restrict__ red, RedLayout L, double* __restrict__ A, double* __restrict__ scale_c,
                             const double* __restrict__ cam_x, double* __restrict__ cam_c, const double* __restrict__ intr,
                             double* __restrict__ camc_c, double* __restrict__ dcam, const double* __restrict__ gmax_p,
                             double* __restrict__ res, IterParams ip, int* __restrict__ chol_ok, StageGate gate, MultiCholFlags f, int tag,
                             long long* __restrict__ mtrace /* diagnostic: [G][16][8] wall-clock stamps, or nullptr */) {
  extern __shared__ double lds[];
  // nreal unknowns, padded to n = whole panels: rows / columns nreal .. n-1 are identity (their solution is 0, their part of
  // L the identity), the right-hand side row sits at row n of A (leading dimension n)
  const int nreal = L.nc, n = (nreal + RSBA_PB - 1) / RSBA_PB * RSBA_PB;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6;
  const int G = gridDim.x, w = blockIdx.x;
  const int np = n / RSBA_PB;              // column panels; blocks 0 .. np (block np: the rhs row alone)
  const long long budget = gate.budget > 0 ? gate.budget : RSBA_STALL_TICKS;
  __shared__ int s_ok;
  __shared__ int s_wb;     // arrivals at the working waves' own barrier
  int wb_gen = 0;
  // LDS: strip (32 p rows of 33) | this workgroup's blocks of the panel (32 x 33 each) | T | Lt | invd | scale | Pre | Xl
  const int max_rows = n + RSBA_PB;       // strip + panel blocks never exceed (p + ceil((np + 1 - p) / G)) * 32 <= n + 32 rows
  double* T = lds + (size_t)max_rows * RSBA_PLD;
  double* Lt = T + RSBA_PB * RSBA_PLD;
  double* Xl = Lt + RSBA_PB * RSBA_PLD;    // X of the next diagonal block in the current panel (rank-32 update, last rows of its strip)
  double* invd = Xl + RSBA_PB * RSBA_PLD;
  double* scl = invd + 64;
  double* Pre = scl + n;                   // the NEXT diagonal block of this workgroup, updated ahead of its panel
  double* part = T;                        // T | Lt | Xl are idle before T(p) arrives: partial tiles of the K-split update
                                           // (up to six 16 x 32) and of the look-ahead product (eight 16 x 16)
  if (tid == 0) { s_ok = 1; s_wb = 0; }
  if (gate.trace && tid == 0 && w == 0) gate.trace[0] = wall_clock64();
  AnnounceResident(gate);
  bool stalled = false;
  const double* S = red + L.S();
  const double inv_radius = 1.0 / ip.radius;
  const int mi = lane & 15, kk = lane >> 4;

  // Pipelined: the kernel is resident long before its first columns exist.  The diagonal factorisation is ~15 KB of
  // straight-line code whose first execution on a CU ran 2 - 5 us longer than the later ones (instruction cache): wave 0
  // of every workgroup runs it once on an identity block while there is nothing to do anyway.
  if (gate.ready != nullptr && !ip.first) {
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
  auto Sat = [&](int gi, int gj) { return (gi < nreal && gj < nreal) ? S[(size_t)gi * nreal + gj] : 0.0; };
  if (w == 0 && !stalled) {   // the first diagonal block
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) { const int r = e >> 5, c = e & 31; Pre[r * RSBA_PLD + c] = sys(r, c, Sat(r, c)); }
    __syncthreads();
  }

#define RSBA_MC_STAMP(k) do { if (mtrace && tid == 0) mtrace[((size_t)w * 16 + p) * 8 + (k)] = wall_clock64(); } while (0)
  for (int p = 0; p < np && !stalled; ++p) {
    const int kb = p * RSBA_PB;
    RSBA_MC_STAMP(0);
    bool s_pending = false;   // the next diagonal block still lacks its entries of S (added after this panel's X)
    // this workgroup's blocks b >= p: b = first, first + G, ...
    const int first = p + ((w - p % G) + G) % G;
    const int nown = first > np ? 0 : (np - first) / G + 1;
    const bool owner = first == p;
    const bool next_owner = p + 1 < np && (p + 1) % G == w;    // then first == p + 1: slot 0 is the next diagonal block
    if (nown == 0) continue;
    double* Bst = lds;
    double* Pan = lds + (size_t)kb * RSBA_PLD;    // slot j: rows of block first + j G
    // One 16-row half of an owned block: its columns of the panel (scaled, damped) into Pan, then minus A[rows, 0:kb] Bst'.
    // A wave loads what it updates itself: no workgroup barrier in between.
    // (ks, nsplit): this wave's slice of the K range; slice 0 owns the rows in Pan, the others leave their products in
    // `part` (slot pslot) for slice 0 to add in a fixed order after the barrier.
    auto load_update_half = [&](int hb, int ks, int nsplit, double* pdst /* 16 x 32 partial tile of a slice ks > 0 */) {
      const int j = hb >> 1, b = first + j * G;
      const int prow = j * RSBA_PB + (hb & 1) * 16;
      // all global loads of the item are issued before anything waits for one of them (every dependent round trip costs
      // ~2 us beside the Schur kernel): the panel's columns of S first, then the first slabs of the rows of L
      const int sr = lane >> 2, sc0 = (lane & 3) * 8;
      const int sgi = b * RSBA_PB + (hb & 1) * 16 + sr;
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
      const int qa = ks * qper * RSBA_PB, qb = min(kb, (ks + 1) * qper * RSBA_PB);   // this slice's columns
      const int grow = b * RSBA_PB + (hb & 1) * 16 + mi;
      const bool gl = grow <= n;
      const double* arow = A + (size_t)(gl ? grow : 0) * n + 8 * kk;
      d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      // RSBA_MC_NPF slabs of the rows in flight: the rows were written by other CUs a moment ago and come back at memory
      // latency, so the length of this loop is (slabs / slabs in flight) round trips
      double buf[RSBA_MC_NPF][8];
      auto fetch = [&](double (&d)[8], int q) {
        const double2* pa = reinterpret_cast<const double2*>(arow + q);
#pragma unroll
        for (int v2 = 0; v2 < 4; ++v2) { const double2 t = pa[v2]; d[2 * v2] = gl ? t.x : 0.0; d[2 * v2 + 1] = gl ? t.y : 0.0; }
      };
#pragma unroll
      for (int i = 0; i < RSBA_MC_NPF; ++i) if (qa + i * RSBA_PB < qb) fetch(buf[i], qa + i * RSBA_PB);
      if (ks == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) Pan[(prow + sr) * RSBA_PLD + sc0 + u] = sgi <= n ? sys(sgi, kb + sc0 + u, v[u]) : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      if (p == 0) return;
      for (int qg = qa; qg < qb; qg += RSBA_MC_NPF * RSBA_PB) {
#pragma unroll
        for (int i = 0; i < RSBA_MC_NPF; ++i) {
          const int q0 = qg + i * RSBA_PB;
          if (q0 < qb) {
            double ac[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ac[u] = buf[i][u];
            if (q0 + RSBA_MC_NPF * RSBA_PB < qb) fetch(buf[i], q0 + RSBA_MC_NPF * RSBA_PB);
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

    // The NEXT owner's diagonal block (block p + 1, slot 0), half h, K slice ks of nf: ONE pass over its rows of L gives
    // both the update of this panel's columns (rows x strip of block p, as load_update_half) and the look-ahead product
    // L[block p+1, 0:kb] L[block p+1, 0:kb]' that the next factorisation needs (tile (h, h); the waves of half 1 also load
    // half 0's rows for the cross tile (1, 0)).  The two used to be separate passes over the same rows — the look-ahead
    // before the strip arrived, the update after it — and one CU streams such rows at memory latency: together they were
    // the longest item of a panel (14.5 us against 9 us of factorisation on the owner), i.e. what the chain of panels
    // waited for.  Two phases: fused_prefetch issues the first slabs' loads at the top of the panel (the rows are this
    // workgroup's own and final; nothing of block p is needed for them), fused_consume runs once the strip is in LDS.
    // Slices ks > 0 leave their products in the partial area `fp` (slice 0 keeps the rows in Pan); fixed order of addition
    // below.  fp: diag tile (h, ks) at (h nf + ks) 256 | cross tile (ks) at (2 nf + ks) 256 | update (h, ks >= 1) at
    // 3 nf 256 + (h (nf - 1) + ks - 1) 512.
    // Roles of the eight waves (next owner, p > 0): "row" waves (half h, K slice ks of nf) stream ONE operand — their
    // half's rows — for the update and the diagonal tile (h, h); two "cross" waves stream both halves' rows, each over half
    // of K, for tile (1, 0); with other owned blocks to update the last two waves take those (load_update_half).
    // RSBA_MC_NPF_FUSED slabs per stream are fetched at the top of the panel and kept in flight.
    double pf[2 * RSBA_MC_NPF_FUSED][8];   // row waves: [0, NPF) their rows; cross waves: [0, NPF) half 1's rows, [NPF, 2 NPF) half 0's
    auto rows_ptr = [&](int h) { return A + (size_t)(kb + RSBA_PB + 16 * h + mi) * n + 8 * kk; };   // row nb0 + 16 h + mi <= n - 1
    auto fetch8 = [&](double (&d)[8], const double* src) {
      const double2* pa = reinterpret_cast<const double2*>(src);
#pragma unroll
      for (int v2 = 0; v2 < 4; ++v2) { const double2 t = pa[v2]; d[2 * v2] = t.x; d[2 * v2 + 1] = t.y; }
    };
    auto fused_range = [&](int ks, int nf, int& qa, int& qb) {
      const int nq = kb / RSBA_PB, qper = (nq + nf - 1) / nf;
      qa = ks * qper * RSBA_PB; qb = min(kb, (ks + 1) * qper * RSBA_PB);
    };
    auto row_prefetch = [&](int h, int ks, int nf) {
      int qa, qb;
      fused_range(ks, nf, qa, qb);
#pragma unroll
      for (int i = 0; i < 2 * RSBA_MC_NPF_FUSED; ++i) if (qa + i * RSBA_PB < qb) fetch8(pf[i], rows_ptr(h) + qa + i * RSBA_PB);
    };
    auto cross_prefetch = [&](int cs) {
      int qa, qb;
      fused_range(cs, 2, qa, qb);
#pragma unroll
      for (int i = 0; i < RSBA_MC_NPF_FUSED; ++i)
        if (qa + i * RSBA_PB < qb) { fetch8(pf[i], rows_ptr(1) + qa + i * RSBA_PB); fetch8(pf[RSBA_MC_NPF_FUSED + i], rows_ptr(0) + qa + i * RSBA_PB); }
    };
    // fp: diag tile (h, ks) at (h nf + ks) 256 | cross tile (cs) at (2 nf + cs) 256 | update (h, ks >= 1) at
    // (2 nf + 2) 256 + (h (nf - 1) + ks - 1) 512
    auto row_consume = [&](int h, int ks, int nf, double* fp) {
      const int nb0 = kb + RSBA_PB;
      const int prow = h * 16;                       // slot 0
      // the panel's columns of S for these rows (slice 0): issued now, used after the products
      const int sr = lane >> 2, sc0 = (lane & 3) * 8;
      const int sgi = nb0 + h * 16 + sr;             // < n: block p + 1 is a block of the matrix
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
      int qa, qb;
      fused_range(ks, nf, qa, qb);
      constexpr int D = 2 * RSBA_MC_NPF_FUSED;
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
            if (q0 + D * RSBA_PB < qb) fetch8(pf[i], rows_ptr(h) + q0 + D * RSBA_PB);   // refilled behind its own products: D - 1 slabs of products ahead of its use
          }
        }
      }
      if (ks == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) Pan[(prow + sr) * RSBA_PLD + sc0 + u] = sys(sgi, kb + sc0 + u, v[u]);
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (ks == 0) {
          const int r = prow + kk + 4 * t;
          Pan[r * RSBA_PLD + mi] -= acc0[t];
          Pan[r * RSBA_PLD + 16 + mi] -= acc1[t];
        } else {
          double* up = fp + (2 * nf + 2) * 256 + (h * (nf - 1) + ks - 1) * 512;
          up[(kk + 4 * t) * 32 + mi] = acc0[t];
          up[(kk + 4 * t) * 32 + 16 + mi] = acc1[t];
        }
        fp[(h * nf + ks) * 256 + (kk + 4 * t) * 16 + mi] = dg[t];
      }
    };
    auto cross_consume = [&](int cs, int nf, double* fp) {
      int qa, qb;
      fused_range(cs, 2, qa, qb);
      constexpr int D = RSBA_MC_NPF_FUSED;
      d4_t cr = {0, 0, 0, 0};
      for (int qg = qa; qg < qb; qg += D * RSBA_PB) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const int q0 = qg + i * RSBA_PB;
          if (q0 < qb) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cr = __builtin_amdgcn_mfma_f64_16x16x4f64(pf[i][u], pf[D + i][u], cr, 0, 0, 0);
            if (q0 + D * RSBA_PB < qb) { fetch8(pf[i], rows_ptr(1) + q0 + D * RSBA_PB); fetch8(pf[D + i], rows_ptr(0) + q0 + D * RSBA_PB); }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) fp[(2 * nf + cs) * 256 + (kk + 4 * t) * 16 + mi] = cr[t];
    };
    const int nhp_no = 2 * nown - 2;                // 16-row halves of the other owned blocks
    double* const fp_area = lds + (size_t)(kb + nown * RSBA_PB) * RSBA_PLD;   // free rows behind the panel blocks, contiguous with T | Lt | Xl:
                                                                               // (max_rows - kb - 32 nown) 33 + 3168 >= 5280 doubles
    const int nf_no = nhp_no > 0 ? 2 : 3;           // 1536 nf + 512 (nf - 1) ... = 2560 / 4096 doubles of partials
    const bool no_panel = next_owner && p > 0;
    const bool row_wave = no_panel && wave < 2 * nf_no;
    const bool cross_wave = no_panel && !row_wave && wave < 2 * nf_no + 2;
    if (row_wave) row_prefetch(wave / nf_no, wave % nf_no, nf_no);
    if (cross_wave) cross_prefetch(wave - 2 * nf_no);

    if (owner) {
      // Pre = this panel's diagonal block, fully updated during the previous panel.
      // Wave 0 factors at once; the other waves bring the remaining blocks up to date meanwhile.
      if (wave == 0) {
        if (!DiagFactorInverseCall((lds_double*)Pre, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && lane == 0) s_ok = 0;
      } else if (nown > 1) {
        if (p > 0) {
          // the strip of block p: this workgroup's own rows (the last 32 columns stored a moment ago), by the seven working
          // waves alone, with a barrier of their own (wave 0 is in the factorisation)
          for (int e = tid - 64; e < (kb >> 2) * RSBA_PB; e += nt - 64) {
            const int c = e / (kb >> 2), q0 = (e - c * (kb >> 2)) * 4;
            const double* lrow = A + (size_t)(kb + c) * n + q0;
            const double2 a01 = *reinterpret_cast<const double2*>(lrow), a23 = *reinterpret_cast<const double2*>(lrow + 2);
            Bst[(q0 + 0) * RSBA_PLD + c] = a01.x; Bst[(q0 + 1) * RSBA_PLD + c] = a01.y;
            Bst[(q0 + 2) * RSBA_PLD + c] = a23.x; Bst[(q0 + 3) * RSBA_PLD + c] = a23.y;
          }
          ++wb_gen;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) {
            __hip_atomic_fetch_add(&s_wb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&s_wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (nwave - 1) * wb_gen) __builtin_amdgcn_s_sleep(1);
          }
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        for (int hb = 2 + (wave - 1); hb < 2 * nown; hb += nwave - 1) load_update_half(hb, 0, 1, nullptr);
      }
      __syncthreads();
      RSBA_MC_STAMP(4);
      // L11 in the lower triangle, T transposed into the strict upper one, inverse pivots into row n + 1 (the layout
      // BackSubstituteBlocks reads)
      for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) {
        const int r = e >> 5, c = e & 31;
        StoreShared(&A[(size_t)(kb + r) * n + kb + c], c > r ? T[c * RSBA_PLD + r] : Pre[r * RSBA_PLD + c]);
      }
      if (tid < RSBA_PB) StoreShared(&A[(size_t)(n + 1) * n + kb + tid], invd[tid]);
      if (tid == 0 && !s_ok) __hip_atomic_store(chol_ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      PublishFlagWG(f.tdone + p, tag);
      RSBA_MC_STAMP(5);
    } else {
      // The next owner's look-ahead: Pre = the NEXT diagonal block's entries of the system minus
      // L[block p+1, 0:kb] L[block p+1, 0:kb]'.  The product is formed in the same pass over block p + 1's rows as this
      // panel's update (fused_half below); here only the entries of S are fetched, early, so that their latency is hidden.
      const int nb0_la = kb + RSBA_PB;   // first row / column of block p + 1
      // block p + 1 opens a new camera group whose columns may not be published yet: its entries of S are added after
      // this panel's X instead (the product only needs L) — waiting here would hold up this panel for everybody
      const bool s_late = next_owner && gate.ready != nullptr && !ip.first && (nb0_la % gate.cols == 0 || kb % gate.cols == 0);   // (this panel's own gate is passed further down)
      double sv[2] = {0.0, 0.0};
      if (next_owner && !s_late) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; sv[u] = Sat(nb0_la + r, nb0_la + c); }
      }
      // (the diagonal entries' damping term, diag U, is fetched here too: inside sys() it would be one more dependent round
      // trip in the assembly of Pre; it belongs to the payload of the block's camera group like the entries of S)
      auto sys_pre = [&](int gi, int gj, double raw, double du) {
        if (gi >= nreal || gj >= nreal) return gi == gj ? 1.0 : 0.0;
        double v = raw * (scl[gi] * scl[gj]);
        if (gi == gj) v += fmin(fmax(scl[gi] * scl[gi] * du, ip.min_lm_diagonal), ip.max_lm_diagonal) * inv_radius;
        return v;
      };
      auto load_du = [&](int u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; return (r == c && nb0_la + r < nreal) ? red[L.diagU() + nb0_la + r] : 0.0; };
      double du[2] = {0.0, 0.0};
      if (next_owner && !s_late) { du[0] = load_du(0); du[1] = load_du(1); }
      if (next_owner && p == 0) {
        s_pending = s_late;
        // nothing to subtract yet
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; Pre[r * RSBA_PLD + c] = s_late ? 0.0 : sys(nb0_la + r, nb0_la + c, sv[u]); }
        __syncthreads();
      }
      RSBA_MC_STAMP(1);
      // strip: rows of block p, columns 0 .. kb, transposed into Bst[q][c]
      if (p > 0) {
        if (!WaitFlagWG(f.strip_ready + p, tag, f.error, budget)) { stalled = true; break; }
        RSBA_MC_STAMP(2);
        for (int e = tid; e < (kb >> 2) * RSBA_PB; e += nt) {
          const int c = e / (kb >> 2), q0 = (e - c * (kb >> 2)) * 4;
          const double* lrow = A + (size_t)(kb + c) * n + q0;
          const double2 a01 = *reinterpret_cast<const double2*>(lrow), a23 = *reinterpret_cast<const double2*>(lrow + 2);
          Bst[(q0 + 0) * RSBA_PLD + c] = a01.x; Bst[(q0 + 1) * RSBA_PLD + c] = a01.y;
          Bst[(q0 + 2) * RSBA_PLD + c] = a23.x; Bst[(q0 + 3) * RSBA_PLD + c] = a23.y;
        }
        __syncthreads();
      }
      RSBA_MC_STAMP(3);
      // the panel's own columns of S are first needed here: everything above (strip) ran while the
      // Schur kernel was still producing this camera group
      if (gate.ready != nullptr && kb % gate.cols == 0 && !ip.first) {
        if (!WaitReady(gate.ready + 1 + kb / gate.cols, gate.tag, nullptr, gate.budget)) { stalled = true; break; }
      }
      // The next diagonal block's entries of S, if they were held back above (s_late): block p + 1 belongs to this panel's
      // camera group (gate just passed), or opens the next one — whose stage the Schur kernel has usually published by now
      // (the factorisation runs behind it): one look at the flag, no waiting.  Only if it is not there yet do the entries
      // come after this panel's X (the tail below), where they cost a wait and a round trip on the chain of panels.
      bool have_s = !s_late;
      if (s_late && p > 0) {
        if (nb0_la % gate.cols != 0) have_s = true;   // same group as this panel
        else {
          __shared__ int s_gate_open;
          if (tid == 0) s_gate_open = __hip_atomic_load(gate.ready + 1 + nb0_la / gate.cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gate.tag;
          __syncthreads();
          have_s = s_gate_open != 0;
          if (have_s) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (have_s) {
#pragma unroll
          for (int u = 0; u < 2; ++u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; sv[u] = Sat(nb0_la + r, nb0_la + c); }
          du[0] = load_du(0); du[1] = load_du(1);
        }
        s_pending = !have_s;
      } else if (s_late && p == 0 && nb0_la % gate.cols != 0) {
        // panel 0: block 1 belongs to group 0, whose gate has just been passed
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * nt, r = e >> 5, c = e & 31; Pre[r * RSBA_PLD + c] = sys(nb0_la + r, nb0_la + c, Sat(nb0_la + r, nb0_la + c)); }
        s_pending = false;
        __syncthreads();
      }
      if (next_owner && p > 0) {
        const int nf = nf_no, nhp = nhp_no;
        double* fp = fp_area;
        if (row_wave) row_consume(wave / nf, wave % nf, nf, fp);
        else if (cross_wave) cross_consume(wave - 2 * nf, nf, fp);
        else for (int hb = 2 + (wave - 2 * nf - 2); hb < 2 + nhp; hb += nwave - 2 * nf - 2) load_update_half(hb, 0, 1, nullptr);
        __syncthreads();
        // slices >= 1 of the update in slice order (slice 0 is already in Pan), then the next diagonal block ahead of its panel
        for (int e = tid; e < 2 * 512; e += nt) {
          const int h = e >> 9, r = (e >> 5) & 15, c = e & 31;
          double sum = 0.0;
          for (int k2 = 0; k2 < nf - 1; ++k2) sum += fp[(2 * nf + 2) * 256 + (h * (nf - 1) + k2) * 512 + r * 32 + c];
          Pan[(h * 16 + r) * RSBA_PLD + c] -= sum;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + u * nt, r = e >> 5, c = e & 31;
          double d = 0.0;
          if ((r >> 4) == (c >> 4)) {
            const int hh = r >> 4, o = (r & 15) * 16 + (c & 15);
            for (int k2 = 0; k2 < nf; ++k2) d += fp[(hh * nf + k2) * 256 + o];
          } else {
            const int o = r >= 16 ? (r & 15) * 16 + (c & 15) : (c & 15) * 16 + (r & 15);   // tile (1,0), or its mirror
            d = fp[(2 * nf) * 256 + o] + fp[(2 * nf + 1) * 256 + o];
          }
          sv[u] = (have_s ? sys_pre(nb0_la + r, nb0_la + c, sv[u], du[u]) : 0.0) - d;
        }
        __syncthreads();   // T | Lt | Xl may hold partials: everybody has read them before Pre and, later, T are written
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * nt; Pre[(e >> 5) * RSBA_PLD + (e & 31)] = sv[u]; }
        __syncthreads();
      } else {
        // few 16-row halves (late panels): several waves share one, each a slice of the K range — the update is a chain
        // of dependent load latencies per wave, so its length is what counts
        const int nh = 2 * nown, nsplit = (p > 0 && nh <= 2) ? 4 : ((p > 0 && nh <= 4) ? 2 : 1);
        for (int it = wave; it < nh * nsplit; it += nwave) {
          const int hb = it / nsplit, ks = it - hb * nsplit;
          load_update_half(hb, ks, nsplit, part + (hb * (nsplit - 1) + ks - 1) * 512);
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
      RSBA_MC_STAMP(4);
      if (!WaitFlagWG(f.tdone + p, tag, f.error, budget)) { stalled = true; break; }
      {
        // T[r][c], r >= c: stored at A[kb + c][kb + r] for r > c, the diagonal in row n + 1.  Both of a thread's entries
        // are loaded before either is stored: this is on the chain from one factorisation to the next
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
      RSBA_MC_STAMP(5);
    }
    // X = Rows T' for the blocks below the diagonal, stored as L; the next diagonal block's X also stays in LDS
    {
      const int hb0 = owner ? 2 : 0;
      for (int hb = hb0 + wave; hb < 2 * nown; hb += nwave) {
        const int j = hb >> 1, b = first + j * G;
        const int prow = j * RSBA_PB + (hb & 1) * 16;
        d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4) {
          const double a = Pan[(prow + mi) * RSBA_PLD + qs + kk];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[mi * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(16 + mi) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int lr = (hb & 1) * 16 + kk + 4 * tt, grow = b * RSBA_PB + lr;
          if (grow <= n) {
            StoreShared(&A[(size_t)grow * n + kb + mi], acc0[tt]);
            StoreShared(&A[(size_t)grow * n + kb + 16 + mi], acc1[tt]);
          }
          if (next_owner && j == 0) { Xl[lr * RSBA_PLD + mi] = acc0[tt]; Xl[lr * RSBA_PLD + 16 + mi] = acc1[tt]; }
        }
      }
    }
    RSBA_MC_STAMP(6);
    if (next_owner) {
      // block p + 1's rows in global memory are complete through this panel: the others may read them as the next strip
      // (before anything that may wait for the Schur kernel)
      PublishFlagWG(f.strip_ready + p + 1, tag);
      const int nb0 = kb + RSBA_PB;
      if (s_pending) {   // the deferred entries of S (see the look-ahead)
        if (!WaitReady(gate.ready + 1 + nb0 / gate.cols, gate.tag, nullptr, gate.budget)) { stalled = true; break; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + u * nt, r = e >> 5, c = e & 31;
          Pre[r * RSBA_PLD + c] += sys(nb0 + r, nb0 + c, Sat(nb0 + r, nb0 + c));
        }
        __syncthreads();
      }
      // Pre -= X X' (one 16 x 16 tile per wave 0..3)
      if (wave < 4) {
        const int ti = wave >> 1, tj = wave & 1;
        d4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xl[(16 * ti + mi) * RSBA_PLD + qs + kk], Xl[(16 * tj + mi) * RSBA_PLD + qs + kk], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) Pre[(16 * ti + kk + 4 * t) * RSBA_PLD + 16 * tj + mi] -= acc[t];
      }
      __syncthreads();
    } else { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
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
  if (gate.trace && tid == 0) gate.trace[13] = wall_clock64();   // factorisation complete on every workgroup
  double* ysol = A + (size_t)n * n;
  double* y = BackSubstituteBlocksPrefetch(n, A, lds);
  for (int i = tid; i < n; i += nt) ysol[i] = y[i];   // (the epilogue reads the first nreal entries)
  __threadfence_block();
  __syncthreads();
  int ok = 1;
  if (tid == 0) { res[RES_STALL] = 0.0; ok = __hip_atomic_load(chol_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (gate.trace && tid == 0) gate.trace[14] = wall_clock64();   // back-substitution done
  CameraStepEpilogue(C, red, L, scale_c, ysol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, ok, lds, ip.cam_free);
  if (gate.trace && tid == 0) gate.trace[15] = wall_clock64();
  SolveDone(gate);
}
#endif   // RSBA_EXPERIMENTAL

// n = the padded dimension
__host__ __device__ inline int MultiCholPadded(int nc) { return (nc + RSBA_PB - 1) / RSBA_PB * RSBA_PB; }
__host__ __device__ inline size_t MultiCholLdsDoubles(int nc) { const int n = MultiCholPadded(nc); return (size_t)(n + RSBA_PB) * RSBA_PLD + 4 * RSBA_PB * RSBA_PLD + 64 + n; }

}  // namespace rsba
