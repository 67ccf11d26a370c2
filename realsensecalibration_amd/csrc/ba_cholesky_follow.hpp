// DiagFactorInverse's factorisation (ba_cholesky.hpp) with FOLLOWER rows — the rows of the blocks below a 32 x 32 diagonal block take the
// factoring wavefront's own steps (scale column j, subtract l_ij l_cj) and end as X = A L11^-T, without the inverse of L11 and without a
// substitution behind the factorisation.  Set A: 32 rows in the lanes 32..63 (which DiagFactorInverse leaves to shadow the lanes 0..31);
// set B: 32 more rows in a second set of registers of the lanes 0..31.  Alone on a CU (tools/lab/follow_bench.hip, profiles/r06_follow_bench.txt):
// factor alone 4.81 us, + 32 follower rows 5.32, + 64: 6.83; factor then TrsmRowsQuad of 32 rows by two wavefronts 6.31.  Used by the
// tiled factorisation's diagonal tiles for their own rows 32..63 (ba_cholesky_tiles.hpp, round 6); in the border's workgroup of the
// 64-camera kernel it was slower (the substitution it replaces ran beside the next pivot chain there: HISTORY.md).
#pragma once
#include "ba_cholesky.hpp"
namespace rsba {
template <bool kSetB>
__device__ __forceinline__ bool DiagFactorFollow(double* __restrict__ Pan, double* __restrict__ T, double* __restrict__ Lt, double* __restrict__ invd,
                                                 double* __restrict__ FolA, double* __restrict__ FolB, int lane) {
      double row[RSBA_PB], rowb[kSetB ? RSBA_PB : 1];
      const int lr = lane & 31;
      const bool piv = lane < 32;
      {
        const double* src = piv ? Pan : FolA;
#pragma unroll
        for (int c = 0; c < RSBA_PB; ++c) row[c] = src[lr * RSBA_PLD + c];
      }
      if constexpr (kSetB) {
#pragma unroll
        for (int c = 0; c < RSBA_PB; ++c) rowb[c] = piv ? FolB[lr * RSBA_PLD + c] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) asm volatile("" : "+v"(row[c]));
      if constexpr (kSetB) {
#pragma unroll
        for (int c = 0; c < RSBA_PB; ++c) asm volatile("" : "+v"(rowb[c]));
      }
      double* colbuf = T + 20 * RSBA_PLD;   // two 32-double buffers by step parity (rows 20, 21 of the T tile: scratch here)
      double nv[16];
      double ilv = 0.0, lij, il, lijb = 0.0;
#define RSBA_PIN(x) asm volatile("" : "+v"(x))
#define RSBA_FACTOR_CHAIN0(J)                                                                                           \
      {                                                                                                                 \
        const double d = ReadLaneD(row[J], J);                                                                          \
        double y = __builtin_amdgcn_rsq(d);                                                                             \
        double e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                               \
        y = __builtin_fma(y, e, y);                                                                                     \
        e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                                      \
        il = __builtin_fma(y, e, y);                                                                                    \
        lij = row[J] * il;                                                                                              \
        if constexpr (kSetB) lijb = rowb[J] * il;                                                                                 \
      }
#define RSBA_FACTOR_ITEMS(S)                                                                                            \
      _Pragma("unroll") for (int i = (S); i < nitems; i += 6) {                                                         \
        if (i == 0) {                                                                                                   \
          if (j + 2 < (CEND_)) {                                                                                        \
            const double lc = ReadLaneD(lij, j + 2);                                                                    \
            row[j + 2] -= lij * lc; RSBA_PIN(row[j + 2]);                                                               \
            if constexpr (kSetB) { rowb[j + 2] -= lijb * lc; RSBA_PIN(rowb[j + 2]); }                                             \
          }                                                                                                             \
        } else if (j > (BASE_)) {                                                                                       \
          const int c = j + 1 + i;                                                                                      \
          if (c < (CEND_)) {                                                                                            \
            row[c] -= row[j - 1] * nv[c - (BASE_)]; RSBA_PIN(row[c]);                                                   \
            if constexpr (kSetB) { rowb[c] -= rowb[j - 1] * nv[c - (BASE_)]; RSBA_PIN(rowb[c]); }                                 \
          }                                                                                                             \
        }                                                                                                               \
      }
#define RSBA_FACTOR_STEP                                                                                                \
      {                                                                                                                 \
        row[j] = lij;   /* the block's rows: l_ij (lane j: sqrt(d)); a follower's: x_ij */                              \
        if constexpr (kSetB) rowb[j] = lijb;                                                                                      \
        if (lr == j) ilv = il;                                                                                          \
        RSBA_PIN(ilv);                                                                                                  \
        const int nitems = (CEND_) - j - 1;                                                                             \
        double lij_n = 0.0, il_n = 0.0, lijb_n = 0.0;                                                                   \
        if (j > (BASE_)) { _Pragma("unroll") for (int c = j + 2; c < (CEND_); ++c) nv[c - (BASE_)] = colbuf[((j - 1) & 1) * RSBA_PLD + c]; } \
        if (j + 3 < (CEND_)) { if (lane < 32) colbuf[(j & 1) * RSBA_PLD + lane] = lij; }                                \
        if (j + 1 < (CEND_)) {                                                                                          \
          { const double lc = ReadLaneD(lij, j + 1); row[j + 1] -= lij * lc; RSBA_PIN(row[j + 1]);                      \
            if constexpr (kSetB) { rowb[j + 1] -= lijb * lc; RSBA_PIN(rowb[j + 1]); } }                                           \
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
          if constexpr (kSetB) { lijb_n = rowb[j + 1] * il_n; RSBA_PIN(lijb_n); }                                                 \
        }                                                                                                               \
        lij = lij_n; il = il_n; lijb = lijb_n;                                                                          \
      }
      RSBA_FACTOR_CHAIN0(0)
#define CEND_ 16
#define BASE_ 0
#pragma unroll
      for (int j = 0; j < 16; ++j) RSBA_FACTOR_STEP
#undef CEND_
#undef BASE_
      {
        // columns 0..15 are final.  The block's into the Lt tile (upper part zero), the followers' — their X — back into their own
        // tiles; then the rank-16 update of the columns 16..31: the block's bottom-right quadrant as in DiagFactorInverse (through
        // the T tile), the followers' 32 x 16 each through columns 16..31 of their own tiles (dead: those values are in registers)
        if (piv) {
#pragma unroll
          for (int c = 0; c < 16; ++c) Lt[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
        } else {
#pragma unroll
          for (int c = 0; c < 16; ++c) FolA[lr * RSBA_PLD + c] = row[c];
        }
        if constexpr (kSetB) if (piv) {
#pragma unroll
          for (int c = 0; c < 16; ++c) FolB[lr * RSBA_PLD + c] = rowb[c];
        }
        __builtin_amdgcn_wave_barrier();
        const int mi = lane & 15, mk = lane >> 4;
        d4_t acc = {0, 0, 0, 0}, fa0 = {0, 0, 0, 0}, fa1 = {0, 0, 0, 0}, fb0 = {0, 0, 0, 0}, fb1 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
          const double a = Lt[(16 + mi) * RSBA_PLD + ks + mk];   // B[k][j] = L21[j][k]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
          fa0 = __builtin_amdgcn_mfma_f64_16x16x4f64(FolA[mi * RSBA_PLD + ks + mk], a, fa0, 0, 0, 0);
          fa1 = __builtin_amdgcn_mfma_f64_16x16x4f64(FolA[(16 + mi) * RSBA_PLD + ks + mk], a, fa1, 0, 0, 0);
          if constexpr (kSetB) {
            fb0 = __builtin_amdgcn_mfma_f64_16x16x4f64(FolB[mi * RSBA_PLD + ks + mk], a, fb0, 0, 0, 0);
            fb1 = __builtin_amdgcn_mfma_f64_16x16x4f64(FolB[(16 + mi) * RSBA_PLD + ks + mk], a, fb1, 0, 0, 0);
          }
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          T[(mk + 4 * tt) * RSBA_PLD + mi] = acc[tt];   // D[row][col]
          FolA[(mk + 4 * tt) * RSBA_PLD + 16 + mi] = fa0[tt];
          FolA[(16 + mk + 4 * tt) * RSBA_PLD + 16 + mi] = fa1[tt];
          if constexpr (kSetB) { FolB[(mk + 4 * tt) * RSBA_PLD + 16 + mi] = fb0[tt]; FolB[(16 + mk + 4 * tt) * RSBA_PLD + 16 + mi] = fb1[tt]; }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const double upd = piv ? T[(lr & 15) * RSBA_PLD + c] : FolA[lr * RSBA_PLD + 16 + c];
          row[16 + c] -= (piv && lr < 16) ? 0.0 : upd;
          if constexpr (kSetB) rowb[16 + c] -= piv ? FolB[lr * RSBA_PLD + 16 + c] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) { RSBA_PIN(row[16 + c]); if constexpr (kSetB) RSBA_PIN(rowb[16 + c]); }
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
      if (piv) {
        invd[lane] = ilv;
#pragma unroll
        for (int c = 16; c < RSBA_PB; ++c) Lt[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
#pragma unroll
        for (int c = 0; c < RSBA_PB; ++c) Pan[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
      } else {
#pragma unroll
        for (int c = 16; c < RSBA_PB; ++c) FolA[lr * RSBA_PLD + c] = row[c];
      }
      if constexpr (kSetB) if (piv) {
#pragma unroll
        for (int c = 16; c < RSBA_PB; ++c) FolB[lr * RSBA_PLD + c] = rowb[c];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
      return __builtin_amdgcn_ballot_w64(piv && (!(ilv > 0.0) || !(ilv <= DBL_MAX))) == 0;
}

static __device__ __noinline__ bool DiagFactorFollowACall(lds_double* Pan, lds_double* T, lds_double* Lt, lds_double* invd, lds_double* FolA, int lane) {
  return DiagFactorFollow<false>((double*)Pan, (double*)T, (double*)Lt, (double*)invd, (double*)FolA, nullptr, lane);
}
static __device__ __noinline__ bool DiagFactorFollowABCall(lds_double* Pan, lds_double* T, lds_double* Lt, lds_double* invd, lds_double* FolA, lds_double* FolB, int lane) {
  return DiagFactorFollow<true>((double*)Pan, (double*)T, (double*)Lt, (double*)invd, (double*)FolA, (double*)FolB, lane);
}
}  // namespace rsba
