// Experimental variants of DiagFactorInverse for tools/factor_bench.hip (not part of the product).
#pragma once
namespace rsba {

// T = L11^-1 from the padded factor in Lt and invd = 1 / diag (the tail of DiagFactorInverse, unchanged)
__device__ __forceinline__ void InverseFromLt(double* __restrict__ T, double* __restrict__ Lt, double* __restrict__ invd, int lane) {
  const int lr = lane & 31;
  const int hb = lr & 16, lc = lr & 15;
  double t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double sacc = (i == lc) ? 1.0 : 0.0, sacc2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < i) {
        const double lv = Lt[(hb + i) * RSBA_PLD + hb + q];
        if (q & 1) sacc2 -= lv * t[q]; else sacc -= lv * t[q];
      }
    }
    t[i] = (sacc + sacc2) * invd[hb + i];
    asm volatile("" : "+v"(t[i]));
  }
  if (lane < RSBA_PB) {
#pragma unroll
    for (int i = 0; i < 16; ++i) T[(hb + i) * RSBA_PLD + hb + lc] = t[i];
  }
  __builtin_amdgcn_wave_barrier();
  const int mi = lane & 15, mk = lane >> 4;
  d4_t m1 = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 16; ks += 4) m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[(16 + mi) * RSBA_PLD + ks + mk], T[(ks + mk) * RSBA_PLD + mi], m1, 0, 0, 0);
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) T[(mk + 4 * tt) * RSBA_PLD + 16 + mi] = m1[tt];
  __builtin_amdgcn_wave_barrier();
  d4_t t21 = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 16; ks += 4) t21 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(16 + mi) * RSBA_PLD + 16 + ks + mk], T[(ks + mk) * RSBA_PLD + 16 + mi], t21, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) { T[(16 + mk + 4 * tt) * RSBA_PLD + mi] = -t21[tt]; T[(mk + 4 * tt) * RSBA_PLD + 16 + mi] = 0.0; }
}

// Variant 1: the column is NOT scaled on the critical path.  Step j: d = a_jj, r = 1/d (v_rcp_f64 + two Newton steps),
// m_i = a_ij r, a_ic -= m_i a_cj with the RAW a_cj broadcast (v_readlane does not wait for the reciprocal); the factor's
// column is a_ij / sqrt(d_j), formed for all columns at once at the end (each lane the rsqrt of its own pivot).
template <bool kInverse>
__device__ __forceinline__ bool FactorV1(double* __restrict__ Pan, int nb, double* __restrict__ T, double* __restrict__ Lt,
                                         double* __restrict__ invd, int lane) {
  double row[RSBA_PB];
  const int lr = lane & 31;
#pragma unroll
  for (int c = 0; c < RSBA_PB; ++c) row[c] = (lr < nb) ? Pan[lr * RSBA_PLD + c] : (c == lr ? 1.0 : 0.0);
  bool good = true;
  double dmine = 1.0;
#define RSBA_V1_STEP(CEND)                                                                                              \
  {                                                                                                                     \
    const double d = ReadLaneD(row[j], j);                                                                              \
    if (!(d > 0.0) || !(d <= DBL_MAX)) good = false;                                                                    \
    const double dd = good ? d : 1.0;                                                                                   \
    double r = __builtin_amdgcn_rcp(dd);                                                                                \
    r = __builtin_fma(__builtin_fma(-dd, r, 1.0), r, r);                                                                \
    r = __builtin_fma(__builtin_fma(-dd, r, 1.0), r, r);                                                                \
    const double aij = row[j];                                                                                          \
    const double m = aij * r;                                                                                           \
    if (lr == j) dmine = dd;                                                                                            \
    _Pragma("unroll") for (int c0 = j + 1; c0 < (CEND); c0 += 4) {                                                      \
      double lc[4];                                                                                                     \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) lc[u] = (c0 + u < (CEND)) ? ReadLaneD(aij, c0 + u) : 0.0;           \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) if (c0 + u < (CEND)) row[c0 + u] -= m * lc[u];                      \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) if (c0 + u < (CEND)) asm volatile("" : "+v"(row[c0 + u]));          \
    }                                                                                                                   \
  }
  auto rsqrt_full = [](double dd) {
    double il = __builtin_amdgcn_rsq(dd);
    il = il * (1.5 - 0.5 * dd * il * il);
    il = il * (1.5 - 0.5 * dd * il * il);
    return il;
  };
#pragma unroll
  for (int j = 0; j < 16; ++j) RSBA_V1_STEP(16)
  {
    // columns 0..15 of the factor: scale by 1/sqrt(d_c) (lane c's), then A22 -= L21 L21' on the matrix cores
    const double il = rsqrt_full(dmine);
    if (lane < 16) invd[lane] = il;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) { row[c] = (c <= lr) ? row[c] * invd[c] : 0.0; Lt[lr * RSBA_PLD + c] = row[c]; }
    __builtin_amdgcn_wave_barrier();
    const int mi = lane & 15, mk = lane >> 4;
    d4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      const double a = Lt[(16 + mi) * RSBA_PLD + ks + mk];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) T[(mk + 4 * tt) * RSBA_PLD + mi] = acc[tt];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double upd = T[(lr & 15) * RSBA_PLD + c];
      if (lr >= 16) row[16 + c] -= upd;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[16 + c]));
  }
#pragma unroll
  for (int j = 16; j < RSBA_PB; ++j) RSBA_V1_STEP(RSBA_PB)
#undef RSBA_V1_STEP
  {
    const double il = rsqrt_full(dmine);
    if (lane >= 16 && lane < 32) invd[lane] = il;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 16; c < RSBA_PB; ++c) { row[c] = (c <= lr) ? row[c] * invd[c] : 0.0; Lt[lr * RSBA_PLD + c] = row[c]; }
#pragma unroll
    for (int c = 0; c < RSBA_PB; ++c) if (lr < nb) Pan[lr * RSBA_PLD + c] = row[c];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (kInverse) InverseFromLt(T, Lt, invd, lane);
  return good;
}

__device__ __noinline__ bool FactorV1Call(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) {
  return FactorV1<true>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane);
}
__device__ __noinline__ bool FactorV2Call(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) {
  return FactorV1<false>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane);
}


// The current algorithm's factor loop with knobs, for timing only (kNewton < 2 or !kBulk give wrong numbers):
// what do the reciprocal square root chain and the broadcast + update cost?
template <int kNewton, bool kBulk, bool kSelect>
__device__ __forceinline__ bool FactorKnobs(double* __restrict__ Pan, int nb, double* __restrict__ T, double* __restrict__ Lt,
                                            double* __restrict__ invd, int lane) {
  double row[RSBA_PB];
  const int lr = lane & 31;
#pragma unroll
  for (int c = 0; c < RSBA_PB; ++c) row[c] = (lr < nb) ? Pan[lr * RSBA_PLD + c] : (c == lr ? 1.0 : 0.0);
  bool good = true;
#define RSBA_K_STEP(CEND)                                                                                               \
  {                                                                                                                     \
    const double d = ReadLaneD(row[j], j);                                                                              \
    if (!(d > 0.0) || !(d <= DBL_MAX)) good = false;                                                                    \
    const double dd = kSelect ? (good ? d : 1.0) : d;                                                                   \
    double il = __builtin_amdgcn_rsq(dd);                                                                               \
    if (kNewton >= 1) il = il * (1.5 - 0.5 * dd * il * il);                                                             \
    if (kNewton >= 2) il = il * (1.5 - 0.5 * dd * il * il);                                                             \
    const double lij = kSelect ? ((lr == j) ? dd * il : row[j] * il) : row[j] * il;                                     \
    row[j] = lij;                                                                                                       \
    invd[j] = il;                                                                                                       \
    if (kBulk) {                                                                                                        \
    _Pragma("unroll") for (int c0 = j + 1; c0 < (CEND); c0 += 4) {                                                      \
      double lc[4];                                                                                                     \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) lc[u] = (c0 + u < (CEND)) ? ReadLaneD(lij, c0 + u) : 0.0;           \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) if (c0 + u < (CEND)) row[c0 + u] -= lij * lc[u];                    \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) if (c0 + u < (CEND)) asm volatile("" : "+v"(row[c0 + u]));          \
    }                                                                                                                   \
    } else if (j + 1 < (CEND)) { row[j + 1] -= lij * ReadLaneD(lij, j + 1); asm volatile("" : "+v"(row[j + 1])); }      \
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) RSBA_K_STEP(16)
  {
#pragma unroll
    for (int c = 0; c < 16; ++c) Lt[lr * RSBA_PLD + c] = row[c];
    __builtin_amdgcn_wave_barrier();
    const int mi = lane & 15, mk = lane >> 4;
    d4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      const double a = Lt[(16 + mi) * RSBA_PLD + ks + mk];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) T[(mk + 4 * tt) * RSBA_PLD + mi] = acc[tt];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double upd = T[(lr & 15) * RSBA_PLD + c];
      if (lr >= 16) row[16 + c] -= upd;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[16 + c]));
  }
#pragma unroll
  for (int j = 16; j < RSBA_PB; ++j) RSBA_K_STEP(RSBA_PB)
#undef RSBA_K_STEP
#pragma unroll
  for (int c = 0; c < RSBA_PB; ++c) {
    const double v = (c <= lr) ? row[c] : 0.0;
    Lt[lr * RSBA_PLD + c] = v;
    if (lr < nb) Pan[lr * RSBA_PLD + c] = v;
  }
  __builtin_amdgcn_wave_barrier();
  return good;
}
#define RSBA_KNOB_CALL(NAME, N, B, S) \
__device__ __noinline__ bool NAME(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) { \
  return FactorKnobs<N, B, S>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane); }
RSBA_KNOB_CALL(FactorK_2_1_1, 2, true, true)
RSBA_KNOB_CALL(FactorK_2_1_0, 2, true, false)
RSBA_KNOB_CALL(FactorK_1_1_0, 1, true, false)
RSBA_KNOB_CALL(FactorK_0_1_0, 0, true, false)
RSBA_KNOB_CALL(FactorK_2_0_0, 2, false, false)
RSBA_KNOB_CALL(FactorK_0_0_0, 0, false, false)


// Variant 3: software-pipelined.  A wavefront issues in order, so the reciprocal-square-root chain of step j + 1 (readlane,
// v_rsq_f64, two Newton steps of three dependent operations, the scaling: ~125 cycles of latency) only overlaps with step
// j's broadcasts and updates if the two are interleaved in PROGRAM order: column j + 1 is updated first, then one stage of
// the chain alternates with a share of the remaining columns (the empty asm statements pin that order).  No selects on the
// chain: a non-positive pivot turns its 1/sqrt and everything after it into NaN, and the check is one ballot at the end.
__device__ long long g_v3_phase[8];
#define RSBA_V3_PH(k) do {} while (0)
template <bool kInverse>
__device__ __forceinline__ bool FactorV3(double* __restrict__ Pan, int nb, double* __restrict__ T, double* __restrict__ Lt,
                                         double* __restrict__ invd, int lane) {
  double row[RSBA_PB];
  const int lr = lane & 31;
#pragma unroll
  for (int c = 0; c < RSBA_PB; ++c) row[c] = (lr < nb) ? Pan[lr * RSBA_PLD + c] : (c == lr ? 1.0 : 0.0);
#define RSBA_PIN(x) asm volatile("" : "+v"(x))
#define RSBA_V3_BULK(S)                                                                                                 \
  _Pragma("unroll") for (int i = (S); i < nbk; i += 7) {                                                                \
    const int c = j + 2 + i;                                                                                            \
    const double lc = ReadLaneD(lij, c);                                                                                \
    row[c] -= lij * lc;                                                                                                 \
    RSBA_PIN(row[c]);                                                                                                   \
  }
  double lij, il;
#define RSBA_V3_CHAIN0(J)                                                                                               \
  {                                                                                                                     \
    const double d = ReadLaneD(row[J], J);                                                                              \
    double y = __builtin_amdgcn_rsq(d);                                                                                 \
    double e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                                   \
    y = __builtin_fma(y, e, y);                                                                                         \
    e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                                          \
    il = __builtin_fma(y, e, y);                                                                                        \
    lij = row[J] * il;                                                                                                  \
  }
#define RSBA_V3_STEP(CEND)                                                                                              \
  {                                                                                                                     \
    row[j] = lij;                                                                                                       \
    invd[j] = il;                                                                                                       \
    const int nbk = (CEND) - j - 2;                                                                                     \
    double lij_n = 0.0, il_n = 0.0;                                                                                     \
    if (j + 1 < (CEND)) {                                                                                               \
      { const double lc = ReadLaneD(lij, j + 1); row[j + 1] -= lij * lc; RSBA_PIN(row[j + 1]); }                        \
      const double d = ReadLaneD(row[j + 1], j + 1);                                                                    \
      double y0 = __builtin_amdgcn_rsq(d); RSBA_PIN(y0);                                                                \
      RSBA_V3_BULK(0)                                                                                                   \
      double t = d * y0, h = 0.5 * y0; RSBA_PIN(t); RSBA_PIN(h);                                                        \
      RSBA_V3_BULK(1)                                                                                                   \
      double e = __builtin_fma(-t, h, 0.5); RSBA_PIN(e);                                                                \
      RSBA_V3_BULK(2)                                                                                                   \
      double y1 = __builtin_fma(y0, e, y0); RSBA_PIN(y1);                                                               \
      RSBA_V3_BULK(3)                                                                                                   \
      t = d * y1; h = 0.5 * y1; RSBA_PIN(t); RSBA_PIN(h);                                                               \
      RSBA_V3_BULK(4)                                                                                                   \
      e = __builtin_fma(-t, h, 0.5); RSBA_PIN(e);                                                                       \
      RSBA_V3_BULK(5)                                                                                                   \
      il_n = __builtin_fma(y1, e, y1); RSBA_PIN(il_n);                                                                  \
      RSBA_V3_BULK(6)                                                                                                   \
      lij_n = row[j + 1] * il_n; RSBA_PIN(lij_n);                                                                       \
    }                                                                                                                   \
    lij = lij_n; il = il_n;                                                                                             \
  }
  RSBA_V3_PH(0);
  RSBA_V3_CHAIN0(0)
#pragma unroll
  for (int j = 0; j < 16; ++j) RSBA_V3_STEP(16)
  RSBA_V3_PH(1);
  {
#pragma unroll
    for (int c = 0; c < 16; ++c) Lt[lr * RSBA_PLD + c] = row[c];
    __builtin_amdgcn_wave_barrier();
    const int mi = lane & 15, mk = lane >> 4;
    d4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      const double a = Lt[(16 + mi) * RSBA_PLD + ks + mk];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) T[(mk + 4 * tt) * RSBA_PLD + mi] = acc[tt];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double upd = T[(lr & 15) * RSBA_PLD + c];
      if (lr >= 16) row[16 + c] -= upd;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) RSBA_PIN(row[16 + c]);
  }
  RSBA_V3_PH(2);
  RSBA_V3_CHAIN0(16)
#pragma unroll
  for (int j = 16; j < RSBA_PB; ++j) RSBA_V3_STEP(RSBA_PB)
  RSBA_V3_PH(3);
#undef RSBA_V3_STEP
#undef RSBA_V3_BULK
#undef RSBA_V3_CHAIN0
#pragma unroll
  for (int c = 0; c < RSBA_PB; ++c) {
    const double v = (c <= lr) ? row[c] : 0.0;
    Lt[lr * RSBA_PLD + c] = v;
    if (lr < nb) Pan[lr * RSBA_PLD + c] = v;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
  const double mine = invd[lr];
  const bool good = __builtin_amdgcn_ballot_w64(!(mine > 0.0) || !(mine <= DBL_MAX)) == 0;
  RSBA_V3_PH(4);
  if (kInverse) InverseFromLt(T, Lt, invd, lane);
  RSBA_V3_PH(5);
  return good;
}
__device__ __noinline__ bool FactorV3Call(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) {
  return FactorV3<true>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane);
}
__device__ __noinline__ bool FactorV4Call(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) {
  return FactorV3<false>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane);
}


// Variant 5: variant 3 + the far columns' multipliers through LDS.  Step j scales its column (l_ij), writes it to a
// 32-double buffer and reads the multipliers of the columns >= j + 3 back as broadcasts (ds_read2_b64: one LDS instruction
// for two columns instead of four v_readlane + hazard nops); those updates are applied ONE STEP LATER, between the stages
// of the next pivot's chain, when the values have long arrived.  Columns j + 1 (the next pivot) and j + 2 stay on the
// v_readlane path, so no pivot ever waits for LDS.  The multipliers' order per column is fixed: bitwise reproducible.
template <bool kInverse, bool kWritePan>
__device__ __forceinline__ bool FactorV5(double* __restrict__ Pan, int nb, double* __restrict__ T, double* __restrict__ Lt,
                                         double* __restrict__ invd, int lane) {
  double row[RSBA_PB];
  const int lr = lane & 31;
#pragma unroll
  for (int c = 0; c < RSBA_PB; ++c) row[c] = (lr < nb) ? Pan[lr * RSBA_PLD + c] : (c == lr ? 1.0 : 0.0);
  double* colbuf = T + 20 * RSBA_PLD;   // rows 16.. of the T tile are free until the inverse
  double nv[2][16];
  double ilv = 0.0, lij, il;
#define RSBA_PIN(x) asm volatile("" : "+v"(x))
#define RSBA_V5_CHAIN0(J)                                                                                               \
  {                                                                                                                     \
    const double d = ReadLaneD(row[J], J);                                                                              \
    double y = __builtin_amdgcn_rsq(d);                                                                                 \
    double e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                                   \
    y = __builtin_fma(y, e, y);                                                                                         \
    e = __builtin_fma(-(d * y), 0.5 * y, 0.5);                                                                          \
    il = __builtin_fma(y, e, y);                                                                                        \
    lij = row[J] * il;                                                                                                  \
  }
  // work items of step j between the chain stages: item 0 = column j + 2 by v_readlane, items 1.. = the delayed updates of
  // step j - 1 (columns j + 2 .. CEND - 1, multipliers read one step ago)
#define RSBA_V5_ITEMS(S)                                                                                                \
  _Pragma("unroll") for (int i = (S); i < nitems; i += 7) {                                                             \
    if (i == 0) {                                                                                                       \
      if (j + 2 < (CEND_)) { const double lc = ReadLaneD(lij, j + 2); row[j + 2] -= lij * lc; RSBA_PIN(row[j + 2]); }   \
    } else if (j > (BASE_)) {                                                                                           \
      const int c = j + 1 + i;                                                                                          \
      if (c < (CEND_)) { row[c] -= row[j - 1] * nv[(j - 1) & 1][c - (BASE_)]; RSBA_PIN(row[c]); }                       \
    }                                                                                                                   \
  }
#define RSBA_V5_STEP                                                                                                    \
  {                                                                                                                     \
    row[j] = lij;                                                                                                       \
    if (lr == j) ilv = il;                                                                                              \
    const int nitems = (CEND_) - j - 1;                                                                                 \
    double lij_n = 0.0, il_n = 0.0;                                                                                     \
    if (j + 3 < (CEND_)) { if (lane < 32) colbuf[lane] = lij; __builtin_amdgcn_wave_barrier(); }                        \
    if (j + 1 < (CEND_)) {                                                                                              \
      { const double lc = ReadLaneD(lij, j + 1); row[j + 1] -= lij * lc; RSBA_PIN(row[j + 1]); }                        \
      const double d = ReadLaneD(row[j + 1], j + 1);                                                                    \
      double y0 = __builtin_amdgcn_rsq(d); RSBA_PIN(y0);                                                                \
      _Pragma("unroll") for (int c = j + 3; c < (CEND_); ++c) nv[j & 1][c - (BASE_)] = colbuf[c];                       \
      RSBA_V5_ITEMS(0)                                                                                                  \
      double t = d * y0, h = 0.5 * y0; RSBA_PIN(t); RSBA_PIN(h);                                                        \
      RSBA_V5_ITEMS(1)                                                                                                  \
      double e = __builtin_fma(-t, h, 0.5); RSBA_PIN(e);                                                                \
      RSBA_V5_ITEMS(2)                                                                                                  \
      double y1 = __builtin_fma(y0, e, y0); RSBA_PIN(y1);                                                               \
      RSBA_V5_ITEMS(3)                                                                                                  \
      t = d * y1; h = 0.5 * y1; RSBA_PIN(t); RSBA_PIN(h);                                                               \
      RSBA_V5_ITEMS(4)                                                                                                  \
      e = __builtin_fma(-t, h, 0.5); RSBA_PIN(e);                                                                       \
      RSBA_V5_ITEMS(5)                                                                                                  \
      il_n = __builtin_fma(y1, e, y1); RSBA_PIN(il_n);                                                                  \
      RSBA_V5_ITEMS(6)                                                                                                  \
      lij_n = row[j + 1] * il_n; RSBA_PIN(lij_n);                                                                       \
    }                                                                                                                   \
    lij = lij_n; il = il_n;                                                                                             \
  }
  RSBA_V5_CHAIN0(0)
#define CEND_ 16
#define BASE_ 0
#pragma unroll
  for (int j = 0; j < 16; ++j) RSBA_V5_STEP
#undef CEND_
#undef BASE_
  {
    // columns 0..15 are final: into the Lt tile (upper part zero), from where the matrix cores take L21 for A22 -= L21 L21'
    if (lane < 32) {
#pragma unroll
      for (int c = 0; c < 16; ++c) Lt[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    const int mi = lane & 15, mk = lane >> 4;
    d4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      const double a = Lt[(16 + mi) * RSBA_PLD + ks + mk];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) T[(mk + 4 * tt) * RSBA_PLD + mi] = acc[tt];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double upd = T[(lr & 15) * RSBA_PLD + c];
      row[16 + c] -= (lr >= 16) ? upd : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) RSBA_PIN(row[16 + c]);
  }
  RSBA_V5_CHAIN0(16)
#define CEND_ 32
#define BASE_ 16
#pragma unroll
  for (int j = 16; j < RSBA_PB; ++j) RSBA_V5_STEP
#undef CEND_
#undef BASE_
#undef RSBA_V5_STEP
#undef RSBA_V5_ITEMS
#undef RSBA_V5_CHAIN0
  if (lane < 32) {
    invd[lane] = ilv;
#pragma unroll
    for (int c = 16; c < RSBA_PB; ++c) Lt[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
    if (kWritePan) {
#pragma unroll
      for (int c = 0; c < RSBA_PB; ++c) if (lr < nb) Pan[lr * RSBA_PLD + c] = (c <= lr) ? row[c] : 0.0;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
  const bool good = __builtin_amdgcn_ballot_w64(!(ilv > 0.0) || !(ilv <= DBL_MAX)) == 0;
  if (kInverse) InverseFromLt(T, Lt, invd, lane);
  return good;
}
__device__ __noinline__ bool FactorV5Call(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) {
  return FactorV5<true, true>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane);
}
__device__ __noinline__ bool FactorV6Call(lds_double* Pan, int nb, lds_double* T, lds_double* Lt, lds_double* invd, int lane) {
  return FactorV5<false, false>((double*)Pan, nb, (double*)T, (double*)Lt, (double*)invd, lane);
}

}  // namespace rsba
