// Marker-chain model at scale: many times x markers x cameras.
//
// The residual blocks of /root/reference/Main_Calibration/bundle_adjustment.h:56-343 touch up to three 6-dof blocks:
// a camera transform, a marker transform and the pose of the base marker at one TIME.  Every residual block has
// exactly one time block, and no residual couples two times: given the cameras and markers the time blocks are
// mutually independent, the same structure the 3-D points have in the point model (and the reason Ceres' DENSE_SCHUR
// ordering would eliminate them first).  So the time blocks are eliminated,
//
//     S = U - sum_t W_t' (V_t + D_t)^-1 W_t        rhs = g_r - sum_t W_t' (V_t + D_t)^-1 g_t
//
// leaving a dense system over the (cameras + markers) only, which goes through the same Cholesky kernels as the
// reduced camera system of the point model, and the time steps come from back-substitution.  The one-workgroup dense
// solver of ba_marker_kernels.hpp stays for the committed 114-parameter data; this path takes over when the normal
// equations no longer fit it (rsba_options.schur_impl: 0 dense, 1 automatic, 2 always eliminate).
//
//   k_marker_eval        thread per residual block: r (8), J (8 x 18) by forward-mode duals      [ba_marker_kernels.hpp]
//   k_time_eliminate     workgroup per chunk of consecutive times, one time after the other: V, g, W and the reduced
//                        gradient from the staged Jacobians, E = (V + D)^-1, Y = E W, then every entry of the time's
//                        local block of U - W'Y is added to the workgroup's PRIVATE partial system (fixed order:
//                        bitwise reproducible, no atomics)
//   k_marker_reduce      thread per entry: the partial systems summed in chunk order, mirrored to full symmetric
//   k_marker_reduced_solve / k_sys_build + k_chol_step + k_marker_chol_finish     (n_r <= 384 / larger)
//   k_time_backsub_terms wavefront per time: delta_t = -E (g_t + W_t delta_r), candidate time pose, then the model cost
//                        change and the candidate's residuals of the time's residual blocks
//   k_marker_schur_finish   ordered final sums -> the 16-double result block
//
// Scaling and damping follow the point model's formulation: the elimination runs in unscaled coordinates with the
// effective damping D_t = clamp(s^2 V_ii) / (radius s^2), the Jacobi scale and the LM diagonal of the reduced blocks
// are applied by the Cholesky's panel loads (PanelSource).
#pragma once
#include "ba_cholesky_large.hpp"
#include "ba_cholesky_tiles.hpp"
#include "ba_marker_kernels.hpp"

namespace rsba {

#define RSBA_MT_TILE 32        // residual blocks staged per tile (32 x 152 doubles of LDS)
#define RSBA_MT_MAXD 1020      // widest local system of one time: 170 camera + marker blocks

#define RSBA_MT_THREADS 1024
// (ablation hooks of the profiling builds: tools/ablate_marker.sh)
#ifndef RSBA_ABL_Q
#define RSBA_ABL_Q 8
#endif
#ifndef RSBA_ABL_SUMS
#define RSBA_ABL_SUMS 1
#endif
#ifndef RSBA_ABL_STAGE
#define RSBA_ABL_STAGE 1
#endif
#define RSBA_MT_JLD 145        // LDS stride of a staged 8 x 18 Jacobian
#define RSBA_MT_PW (RSBA_MT_JLD + 8)   // doubles of a residual block's products (they take the place of its rows and residuals)

// A workgroup's partial system: S as packed lower triangle (row i, column j <= i at i (i + 1) / 2 + j), then the vectors.
struct PartLayout {
  int nr;
  __host__ __device__ size_t packed() const { return (size_t)nr * (nr + 1) / 2; }
  __host__ __device__ size_t S() const { return 0; }
  __host__ __device__ size_t gc() const { return packed(); }
  __host__ __device__ size_t corr() const { return packed() + nr; }
  __host__ __device__ size_t diagU() const { return packed() + 2 * (size_t)nr; }
  __host__ __device__ size_t scal() const { return packed() + 3 * (size_t)nr; }
  __host__ __device__ size_t size() const { return packed() + 3 * (size_t)nr + 8; }
};

struct TimeSlots {  // per residual block, in time order: where its camera / marker block sits
  int slot_cam, slot_marker;   // index in its time's slot list (-1: block not part of the residual)
  int col_cam, col_marker;     // first column in the reduced system (-1)
  int camera, pad;             // camera index of the detection (its intrinsics), whether or not the camera pose is a parameter
};

// LDS written by some lanes of a wavefront, read by others of the same wavefront.
#define RSBA_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// 6 x 6 SPD inverse through Cholesky on the first 36 lanes of a wavefront (all 64 call it): A (LDS, row-major, the lower
// triangle is read) -> E (LDS, both halves); Mx: 36 doubles of LDS scratch.  Every lane factors A (the same 21 entries of
// L), lane j < 6 inverts column j of L, lane e < 36 forms entry e of E = M'M: the operations of one thread doing all of
// it (sums padded with exact zeros), without the 200 registers that takes.  False (on every lane) when a pivot is not
// positive and finite.
__device__ inline bool InvertSpd6Lanes(int lane, const double* A, double* Mx, double* E) {
  double L[6][6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
    if (!(s > 0.0) || !(s <= DBL_MAX)) { ok = false; s = 1.0; }
    const double l = sqrt(s), inv = 1.0 / l;
    L[j][j] = l;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
      L[i][j] = t * inv;
    }
  }
  {
    // column j = lane of M = L^-1: zeros above the diagonal
    const int j = lane;
    double m[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) t -= L[i][k] * m[k];
      m[i] = i < j ? 0.0 : (i == j ? 1.0 / L[i][i] : t / L[i][i]);
    }
    if (j < 6) {
#pragma unroll
      for (int i = 0; i < 6; ++i) Mx[i * 6 + j] = m[i];
    }
  }
  RSBA_WAVE_LDS_SYNC();
  if (lane < 36) {
    const int r = lane / 6, c = lane - 6 * r, hi = r > c ? r : c, lo = r > c ? c : r;
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) t += Mx[k * 6 + hi] * Mx[k * 6 + lo];
    E[lane] = t;
  }
  RSBA_WAVE_LDS_SYNC();
  return ok;
}

// Constants of every pose of the parameter array at the current linearisation (rotation matrix, left Jacobian of SO(3),
// translation, AngleAxisRotatePoint's branch): once per pose instead of once per corner of every residual block.
__global__ void __launch_bounds__(256) k_pose_constants(int nposes, const double* __restrict__ params, double* __restrict__ posec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nposes) return;
  const double zero4[4] = {0.0, 0.0, 0.0, 0.0};
  double cc[CC_STRIDE];
  CameraConstants(params + 6 * (size_t)i, zero4, cc);
#pragma unroll
  for (int q = 0; q < CC_STRIDE; ++q) posec[(size_t)i * CC_STRIDE + q] = cc[q];
}

struct ElimArgs {
  int nr, dmax, nblocks;
  const int* __restrict__ chunk_ptr;   // [G + 1] times of each workgroup
  const int* __restrict__ time_ptr;    // [T + 1] residual blocks of each time
  const int* __restrict__ slot_ptr;    // [T + 1] slots (distinct camera / marker blocks) of each time
  const int* __restrict__ slot_col;    // first reduced column of each slot, ascending inside a time
  const int* __restrict__ time_full;   // offset of the time block in the full parameter array
  const int* __restrict__ col_full;    // reduced column -> index in the full parameter array
  const TimeSlots* __restrict__ ts;
  // the residual blocks' Jacobians are NOT stored: every tile of 32 blocks is evaluated where it is used, analytically
  // (MarkerCornerResidualJacobian, ba_math.hpp) from the poses' constants of this linearisation
  const MarkerObs* __restrict__ mo;    // [N]
  const double* __restrict__ obs8;     // [N][8]
  const double* __restrict__ intr;     // [C][4]
  const double* __restrict__ posec;    // [poses][CC_STRIDE]: R, left Jacobian, t, branch flag of every pose at x (k_pose_constants)
  double half_side;
  const double* __restrict__ params_x;
  double* __restrict__ scale_t;        // [6 T]
  double* __restrict__ tdata;          // [T][48]: E (36), g_t (6)
  double* __restrict__ part;           // [G][RedLayout(nr).size()]; S lower triangle only
  IterParams ip;
};

#ifdef RSBA_PROFILE_PHASES
__device__ long long g_mt_cycles[16];
#define RSBA_MT_STAMP(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { long long _t = clock64(); g_mt_cycles[k] += _t - _t0; _t0 = _t; } } while (0)
#else
#define RSBA_MT_STAMP(k) do {} while (0)
#endif

// Workgroup barrier that orders LDS only: the staging lanes' loads from HBM for the next tile stay in flight across it.
#define RSBA_MT_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

__device__ __forceinline__ int LocalColumn(int slot, int q, int sc, int sm) { return slot == sc ? q : (slot == sm ? 12 + q : -1); }

template <bool kLdsS>
__global__ void __launch_bounds__(RSBA_MT_THREADS)
k_time_eliminate(ElimArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nr = a.nr, dmax = a.dmax;
  const PartLayout RL{nr};
  double* W = lds;                        // [6][d]
  double* Y = W + 6 * dmax;               // [6][d]
  double* Gr = Y + 6 * dmax;              // [d]
  double* Vs = Gr + dmax;                 // V 36 | g 6 | (pad 6) | E 36 | Eg 6
  double* Jt = Vs + 96;                   // [TILE][145]: odd stride, the lanes of pass 2 read different residual blocks
  double* rt = Jt + RSBA_MT_TILE * RSBA_MT_JLD;   // [TILE][8]
  double* sqv = rt + RSBA_MT_TILE * 8;    // [TILE] squared residual norms
  double* PA = Jt;                        // [TILE][153] a residual block's products, in the place of its rows and residuals once those are read
  int* sl = (int*)(sqv + RSBA_MT_TILE);   // [TILE][2] slots, then per slot of the time [dmax / 6 + 1] each:
  int* scol = sl + 2 * RSBA_MT_TILE;      //   first reduced column
  int* mk = scol + dmax / 6 + 1;          //   bit i: staged residual block i has the slot
  int* kb = mk + dmax / 6 + 1;            //   where the slot's block sits in a residual's Jacobian: column 0 (camera) or 12 (marker)
  // the chunk's sum of S (packed lower triangle) and of the three vectors: in LDS when they fit, else straight in the workgroup's partial system
  // pose constants of the current time: the time pose, then one per slot (camera / marker block of the time)
  double* pcl = (double*)(sl + ((2 * RSBA_MT_TILE + 3 * (dmax / 6 + 1) + 1) & ~1));   // [dmax / 6 + 2][CC_STRIDE]
  double* Sl = pcl + (size_t)(dmax / 6 + 2) * CC_STRIDE;
  double* P = a.part + (size_t)blockIdx.x * RL.size();
  double* Sacc = kLdsS ? Sl : P + RL.S();
  const size_t nacc = RL.packed() + 3 * (size_t)nr;   // S, then the three vectors behind it (PartLayout): all of them in LDS or none
  for (size_t e = tid; e < nacc; e += RSBA_MT_THREADS) Sacc[e] = 0.0;
  for (size_t e = nacc + tid; e < RL.size(); e += RSBA_MT_THREADS) P[e] = 0.0;
  double cost = 0.0, xn2 = 0.0, gmax = 0.0, fail = 0.0;   // thread 0's running sums over the chunk's times
#ifdef RSBA_PROFILE_PHASES
  long long _t0 = clock64();
#endif
  __threadfence_block();
  __syncthreads();

  // staging lanes: part (camera, time, marker block of the rows) x residual block of the tile x corner
  const int sp = tid >> 7, sb_i = (tid & 127) >> 2, sk = tid & 3;
  int nsc = -1, nsm = -1, ncam = 0;
  double nu = 0.0, nv = 0.0;
  auto fetch = [&](int first) {   // what the lane needs of the tile that starts at residual block first
    const int i = min(first + sb_i, a.nblocks - 1);
    nsc = a.ts[i].slot_cam; nsm = a.ts[i].slot_marker; ncam = a.ts[i].camera;
    nu = a.obs8[8 * (size_t)i + 2 * sk]; nv = a.obs8[8 * (size_t)i + 2 * sk + 1];
  };
  if (tid < 3 * 4 * RSBA_MT_TILE) fetch(a.time_ptr[a.chunk_ptr[blockIdx.x]]);
  RSBA_MT_STAMP(0);
  for (int t = a.chunk_ptr[blockIdx.x]; t < a.chunk_ptr[blockIdx.x + 1]; ++t) {
    const int o0 = a.time_ptr[t], nobs = a.time_ptr[t + 1] - o0;
    const int s0 = a.slot_ptr[t], nslot = a.slot_ptr[t + 1] - s0, d = 6 * nslot;
    // the time's Jacobi scale and parameters, wanted after pass 1
    double pre_scale = 1.0, pre_x = 0.0;
    if (tid < 6) { if (!a.ip.first) pre_scale = a.scale_t[6 * t + tid]; pre_x = a.params_x[a.time_full[t] + tid]; }
    for (int e = tid; e < 6 * d; e += RSBA_MT_THREADS) W[e] = 0.0;
    for (int e = tid; e < d; e += RSBA_MT_THREADS) Gr[e] = 0.0;
    if (tid < 43) Vs[tid] = 0.0;
    for (int e = tid; e < nslot; e += RSBA_MT_THREADS) scol[e] = a.slot_col[s0 + e];
    for (int e = tid; e < (nslot + 1) * CC_STRIDE; e += RSBA_MT_THREADS) {
      const int ps = e / CC_STRIDE, q = e - ps * CC_STRIDE;
      const int pose = (ps == 0 ? a.time_full[t] : a.col_full[a.slot_col[s0 + ps - 1]]) / 6;
      pcl[e] = a.posec[(size_t)pose * CC_STRIDE + q];
    }
    const int ntile = (nobs + RSBA_MT_TILE - 1) / RSBA_MT_TILE;
    auto stage = [&](int tile) {
      const int b0 = o0 + tile * RSBA_MT_TILE, nb = min(RSBA_MT_TILE, nobs - tile * RSBA_MT_TILE);
      RSBA_MT_BARRIER();   // the previous tile has been consumed
      if (tid < 3 * 4 * RSBA_MT_TILE) {
        // one corner of one residual block per three lanes: the residuals (2) and one 2 x 6 block of the Jacobian rows each
        // (the rows go straight into the staged tile; the poses' constants come from LDS, staged once per time).  What the
        // lane read from HBM was fetched a tile ago; the next tile's (of this time or the first of the next: the residual
        // blocks are in time order) is on its way while this one is consumed.
        const int slot_cam = nsc, slot_marker = nsm;
        const double* in = a.intr + 4 * ncam;   // a handful of cameras per chunk: from the L1
        const double u = nu, v = nv, fx = in[0], fy = in[1], ppx = in[2], ppy = in[3];
        fetch(b0 + nb);
        if (sb_i < RSBA_ABL_STAGE * nb) {
          const double hs = a.half_side;
          const double cx = (sk == 0 || sk == 3) ? -hs : hs, cy = sk < 2 ? hs : -hs;
          MarkerCornerJacobianPart(sp, slot_cam >= 0 ? pcl + (size_t)(1 + slot_cam) * CC_STRIDE : nullptr, pcl,
                                   slot_marker >= 0 ? pcl + (size_t)(1 + slot_marker) * CC_STRIDE : nullptr,
                                   fx, fy, ppx, ppy, cx, cy, u, v, rt + sb_i * 8 + 2 * sk, Jt + sb_i * RSBA_MT_JLD + 36 * sk);
          if (sp == 0 && sk == 0) { sl[2 * sb_i] = slot_cam; sl[2 * sb_i + 1] = slot_marker; }
        }
      }
      for (int e = tid; e < nslot; e += RSBA_MT_THREADS) mk[e] = 0;
      RSBA_MT_BARRIER();
      // the last wavefront (the products below keep the first fifteen busy), no barrier: wanted by the sums, two barriers on
      const int mt = tid - (RSBA_MT_THREADS - 64);
      if (mt >= 0 && mt < nb) {   // squared residual norm of a block, corner by corner (before the products take the residuals' place)
        double ss = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) ss += rt[mt * 8 + 2 * k] * rt[mt * 8 + 2 * k] + rt[mt * 8 + 2 * k + 1] * rt[mt * 8 + 2 * k + 1];
        sqv[mt] = ss;
        // which staged residual blocks belong to a slot (bit order = block order: the sums below run in a fixed order)
        const int sc = sl[2 * mt], sm = sl[2 * mt + 1];
        if (sc >= 0) { atomicOr(&mk[sc], 1 << mt); kb[sc] = 0; }
        if (sm >= 0) { atomicOr(&mk[sm], 1 << mt); kb[sm] = 12; }
      }
      return nb;
    };
    RSBA_MT_STAMP(1);
    // ---- pass 1 (the only one over the Jacobians): V, g_t, W = J_t' J_r, reduced gradient, U
    for (int tile = 0; tile < ntile; ++tile) {
      const int nb = stage(tile);
      RSBA_MT_STAMP(8);
      // Products of a residual block's rows J_i (8 x 18: camera, time, marker block), a column per lane: with the time block (J_t' J_side = a block of W, or
      // V), with the residuals (reduced gradient, g_t), with the lane's own block (U's diagonal blocks), marker with camera
      // (U's cross block: every (time, camera, marker) occurs once, so it goes straight into the sum).  The rows are read
      // once per residual block here, not once per entry of the sums; the products then take the rows' place in LDS.
      // Five roles of 192 lanes (three wavefronts each, so a role's branch is uniform): 0 camera columns x time block + gradient,
      // 1 time columns (V, g_t), 2 marker columns x time block + gradient + cross block, 3 camera block x itself, 4 marker block x itself.
      double pt[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, pg = 0.0;   // (roles 3, 4: pt holds the block's own products)
      const int p_role = tid / (6 * RSBA_MT_TILE), p_in = tid - p_role * 6 * RSBA_MT_TILE, p_i = p_in / 6, p_c = p_in - 6 * p_i;
      const int p_sd = p_role == 1 ? 1 : ((p_role == 0 || p_role == 3) ? 0 : 2);
      const bool p_on = p_role < 5 && p_i < nb && (p_sd == 1 || sl[2 * p_i + (p_sd >> 1)] >= 0);
      if (p_on) {
        // row by row (q): the accumulators are what stays live, not a hundred loads
        double px[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        const double* Jq = Jt + p_i * RSBA_MT_JLD;
        const double* ri = rt + p_i * 8;
        const int other = p_role >= 3 ? 6 * p_sd : 6;   // the block the lane's column is multiplied with
#pragma unroll 2
        for (int q = 0; q < RSBA_ABL_Q; ++q, Jq += 18) {
          const double own = Jq[6 * p_sd + p_c];
#pragma unroll
          for (int x = 0; x < 6; ++x) pt[x] = fma(Jq[other + x], own, pt[x]);
          if (p_role < 3) pg = fma(own, ri[q], pg);
          if (p_role == 2) {
#pragma unroll
            for (int cq = 0; cq < 6; ++cq) px[cq] = fma(own, Jq[cq], px[cq]);
          }
        }
        if (p_role == 2 && sl[2 * p_i] >= 0) {
          const int gr = scol[sl[2 * p_i + 1]] + p_c, gc = scol[sl[2 * p_i]];
#pragma unroll
          for (int cq = 0; cq < 6; ++cq) Sacc[(size_t)gr * (gr + 1) / 2 + gc + cq] += px[cq];
        }
      }
      RSBA_MT_BARRIER();   // the rows have been read
      if (p_on) {
        // per residual block (stride RSBA_MT_PW): camera W, g (42) | marker W, g (42) | camera U (21) | marker U (21) | V (21) | g_t (6);
        // the symmetric blocks as lower triangles (row r >= column c at r (r + 1) / 2 + c)
        double* out = PA + p_i * RSBA_MT_PW;
        if (p_role == 0 || p_role == 2) {
          double* o1 = out + 42 * (p_sd >> 1) + p_c;
#pragma unroll
          for (int x = 0; x < 6; ++x) o1[6 * x] = pt[x];
          o1[36] = pg;
        } else if (p_role >= 3) {
          double* o2 = out + 84 + 21 * (p_sd >> 1);
#pragma unroll
          for (int r = 0; r < 6; ++r) if (r >= p_c) o2[r * (r + 1) / 2 + p_c] = pt[r];
        } else {
#pragma unroll
          for (int x = 0; x < 6; ++x) if (x >= p_c) out[126 + x * (x + 1) / 2 + p_c] = pt[x];
          out[147 + p_c] = pg;
        }
      }
      RSBA_MT_BARRIER();
      RSBA_MT_STAMP(9);
      // W = J_t' J_r (row x of the time block, reduced column col) and the reduced gradient (x = 6): the products of the
      // slot's residual blocks in block order
      for (int e = tid; e < RSBA_ABL_SUMS * 7 * d; e += RSBA_MT_THREADS) {
        const int x = e / d, col = e - x * d, s = col / 6;
        const double* pw = PA + (kb[s] ? 42 : 0) + 6 * x + (col - 6 * s);
        double acc = 0.0;
        for (unsigned m = (unsigned)mk[s]; m; m &= m - 1) acc += pw[(__ffs(m) - 1) * RSBA_MT_PW];
        if (x < 6) W[e] += acc; else Gr[col] += acc;
      }
      // U: a diagonal block per slot, the sum over the slot's residual blocks in block order (cameras have the lower
      // columns; two different cameras (or markers) never meet in a residual)
      for (int e = tid; e < RSBA_ABL_SUMS * nslot * 21; e += RSBA_MT_THREADS) {
        const int sidx = e / 21, tri = e - 21 * sidx;
        int rq = (int)((sqrtf(8.0f * (float)tri + 1.0f) - 1.0f) * 0.5f);
        while (rq * (rq + 1) / 2 > tri) --rq;
        while ((rq + 1) * (rq + 2) / 2 <= tri) ++rq;
        const int cq = tri - rq * (rq + 1) / 2;
        const double* pw = PA + 84 + (kb[sidx] ? 21 : 0) + tri;
        double u = 0.0;
        for (unsigned m = (unsigned)mk[sidx]; m; m &= m - 1) u += pw[(__ffs(m) - 1) * RSBA_MT_PW];
        const int gr = scol[sidx] + rq, gc = scol[sidx] + cq;
        if (rq == cq) Sacc[RL.diagU() + gr] += u;
        Sacc[(size_t)gr * (gr + 1) / 2 + gc] += u;
      }
      // V (21, mirrored), g_t (6), cost (1): every staged residual block contributes; eight lanes per entry take every
      // eighth block, a fixed shuffle tree adds them
      {
        const int t2 = tid - (RSBA_MT_THREADS - 28 * 8);   // the last waves: the first ones have the longest sums above
        if (t2 >= 0) {
          const int ent = t2 >> 3, k = t2 & 7;
          double acc = 0.0;
          if (ent < 27) { for (int i = k; i < nb; i += 8) acc += PA[i * RSBA_MT_PW + 126 + ent]; }
          else { for (int i = k; i < nb; i += 8) acc += sqv[i]; }
          acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
          if (k == 0) {
            if (ent < 21) {
              int x = (int)((sqrtf(8.0f * (float)ent + 1.0f) - 1.0f) * 0.5f);
              while (x * (x + 1) / 2 > ent) --x;
              while ((x + 1) * (x + 2) / 2 <= ent) ++x;
              const int y = ent - x * (x + 1) / 2;
              Vs[6 * x + y] += acc;
              if (x != y) Vs[6 * y + x] += acc;
            } else if (ent < 27) Vs[36 + ent - 21] += acc;
            else Vs[42] += acc;
          }
        }
      }
      RSBA_MT_STAMP(10);
    }
    __syncthreads();
    RSBA_MT_STAMP(2);
    // ---- E = (V + D)^-1, E g_t: the first wavefront (the residual blocks' products are dead: the damped block and the inverse's scratch)
    if (tid < 64) {
      double* Vd = PA + 36;
      double sc = 1.0;
      if (tid < 6) {
        if (a.ip.first) { sc = a.ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(Vs[7 * tid])) : 1.0; a.scale_t[6 * t + tid] = sc; }
        else sc = pre_scale;
      }
      const double s2 = sc * sc;
      const double dd = tid < 6 ? fmin(fmax(s2 * Vs[7 * tid], a.ip.min_lm_diagonal), a.ip.max_lm_diagonal) / (a.ip.radius * s2) : 0.0;
      {
        const int x = tid < 36 ? tid / 6 : 0, y = tid - 6 * x;
        const double dx = __shfl(dd, x, 64);
        if (tid < 36) Vd[tid] = x == y ? Vs[tid] + dx : Vs[tid];
      }
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        const double xv = __shfl(pre_x, x, 64);
        gmax = fmax(gmax, fabs(Vs[36 + x]));
        xn2 += xv * xv;
      }
      RSBA_WAVE_LDS_SYNC();
      if (!InvertSpd6Lanes(tid, Vd, PA, Vs + 48)) fail += 1.0;
      if (tid < 36) a.tdata[(size_t)t * 48 + tid] = Vs[48 + tid];
      if (tid < 6) {
        double e = 0.0;
#pragma unroll
        for (int y = 0; y < 6; ++y) e += Vs[48 + 6 * tid + y] * Vs[36 + y];
        Vs[84 + tid] = e;
        a.tdata[(size_t)t * 48 + 36 + tid] = Vs[36 + tid];
      }
      cost += Vs[42];
    }
    __syncthreads();
    RSBA_MT_STAMP(3);
    for (int e = tid; e < 6 * d; e += RSBA_MT_THREADS) {
      const int x = e / d, col = e - x * d;
      double s = 0.0;
#pragma unroll
      for (int y = 0; y < 6; ++y) s += Vs[48 + 6 * x + y] * W[y * d + col];
      Y[e] = s;
    }
    __syncthreads();
    RSBA_MT_STAMP(4);
    // ---- pass 2: minus W'Y, the dense part of the time's block.  One item = a row of a 6 x 6 block (rs, cs <= rs):
    // lower triangle only (slots ascend, so does the reduced column), six outputs per six + 36 loads
    {
      const int nbp = nslot * (nslot + 1) / 2;
      for (int it = tid; it < 6 * nbp; it += RSBA_MT_THREADS) {
        const int bp = it / 6, rq = it - 6 * bp;
        int rs = (int)((sqrtf(8.0f * (float)bp + 1.0f) - 1.0f) * 0.5f);
        while (rs * (rs + 1) / 2 > bp) --rs;
        while ((rs + 1) * (rs + 2) / 2 <= bp) ++rs;
        const int cs = bp - rs * (rs + 1) / 2;
        double wr[6];
#pragma unroll
        for (int x = 0; x < 6; ++x) wr[x] = W[x * d + 6 * rs + rq];
        const int gr = scol[rs] + rq;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          if (rs == cs && c > rq) continue;
          double wy = 0.0;
#pragma unroll
          for (int x = 0; x < 6; ++x) wy += wr[x] * Y[x * d + 6 * cs + c];
          Sacc[(size_t)gr * (gr + 1) / 2 + scol[cs] + c] -= wy;
        }
      }
    }
    RSBA_MT_STAMP(5);
    for (int e = tid; e < d; e += RSBA_MT_THREADS) {
      const int s = e / 6, gcol = scol[s] + (e - 6 * s);
      double c = 0.0;
#pragma unroll
      for (int x = 0; x < 6; ++x) c += W[x * d + e] * Vs[84 + x];
      Sacc[RL.gc() + gcol] += Gr[e];
      Sacc[RL.corr() + gcol] -= c;
    }
    __threadfence_block();
    __syncthreads();
    RSBA_MT_STAMP(6);
  }
  if (kLdsS) for (size_t e = tid; e < nacc; e += RSBA_MT_THREADS) P[e] = Sl[e];
  if (tid == 0) { P[RL.scal() + 0] = cost; P[RL.scal() + 1] = xn2; P[RL.scal() + 2] = fail; P[RL.scal() + 3] = gmax; }
}

// Partial systems -> red (full symmetric S | gc | corr | diagU | scal), chunk order.
// A workgroup takes 64 consecutive entries of the PACKED partial systems, wavefront k the chunks g = k (mod 8) in ascending
// order, eight loads in flight: every load of a wavefront is 512 contiguous bytes of one partial system (the round-3 form read
// the packed triangle twice — once per half of the full square it writes — with eight partial systems per load instruction:
// 29 us for 20 MB).  The eight wavefronts' sums meet in LDS in the order the eight lanes' did: ((0+1)+(2+3))+((4+5)+(6+7)) — the
// same bits.  An entry of the triangle is written to both halves of S.
__global__ void __launch_bounds__(512)
k_marker_reduce(int nr, int G, const double* __restrict__ part, double* __restrict__ red) {
  const RedLayout RL{nr};
  const PartLayout PL{nr};
  __shared__ double s_w[8][64];
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  const size_t src = (size_t)blockIdx.x * 64 + lane;
  const bool live = src < PL.size();
  const bool is_max = src == PL.scal() + 3;
  double s = 0.0;
  if (live) {
    int g = k;
    for (; g + 56 < G; g += 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(g + 8 * u) * PL.size() + src];
#pragma unroll
      for (int u = 0; u < 8; ++u) s = is_max ? fmax(s, v[u]) : s + v[u];
    }
    for (; g < G; g += 8) { const double v = part[(size_t)g * PL.size() + src]; s = is_max ? fmax(s, v) : s + v; }
  }
  s_w[k][lane] = s;
  __syncthreads();
  if (k != 0 || !live) return;
  auto add = [&](double x, double y) { return is_max ? fmax(x, y) : x + y; };
  const double t = add(add(add(s_w[0][lane], s_w[1][lane]), add(s_w[2][lane], s_w[3][lane])), add(add(s_w[4][lane], s_w[5][lane]), add(s_w[6][lane], s_w[7][lane])));
  if (src < PL.packed()) {
    size_t i = (size_t)((sqrt(8.0 * (double)src + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > src) --i;
    while ((i + 1) * (i + 2) / 2 <= src) ++i;
    const size_t j = src - i * (i + 1) / 2;
    red[i * nr + j] = t;
    red[j * nr + i] = t;
  } else {
    red[(size_t)nr * nr + (src - PL.packed())] = t;
  }
}

// Step of the reduced blocks from the solution y of the scaled system; candidate parameters; norms.
//   out[0..4] = |delta_r|^2, |x_r|^2, |x_r + delta_r|^2, max |g_r|, factorisation ok
__device__ __forceinline__ void ReducedStepEpilogue(int nr, const double* __restrict__ red, RedLayout RL, const double* __restrict__ scale_r,
                                                    const double* __restrict__ ysol, const int* __restrict__ col_full,
                                                    const double* __restrict__ params_x, double* __restrict__ params_c,
                                                    double* __restrict__ delta_r, double* __restrict__ out, int ok, double* scr) {
  const int tid = threadIdx.x, nt = blockDim.x;
  double d2 = 0, x2 = 0, xc2 = 0, gm = 0;
  for (int i = tid; i < nr; i += nt) {
    const double dd = -scale_r[i] * ysol[i];
    delta_r[i] = dd;
    const double x = params_x[col_full[i]], xc = x + dd;
    params_c[col_full[i]] = xc;
    d2 += dd * dd; x2 += x * x; xc2 += xc * xc; gm = fmax(gm, fabs(red[RL.gc() + i]));
  }
  __syncthreads();
  scr[tid] = d2; scr[nt + tid] = x2; scr[2 * nt + tid] = xc2; scr[3 * nt + tid] = gm;
  __syncthreads();
  for (int off = nt / 2; off > 0; off >>= 1) {
    if (tid < off) {
      scr[tid] += scr[tid + off]; scr[nt + tid] += scr[nt + tid + off]; scr[2 * nt + tid] += scr[2 * nt + tid + off];
      scr[3 * nt + tid] = fmax(scr[3 * nt + tid], scr[3 * nt + tid + off]);
    }
    __syncthreads();
  }
  if (tid == 0) { out[0] = scr[0]; out[1] = scr[nt]; out[2] = scr[2 * nt]; out[3] = scr[3 * nt]; out[4] = ok ? 1.0 : 0.0; }
}

__global__ void __launch_bounds__(512)
k_marker_reduced_solve(int nr, const double* __restrict__ red, double* __restrict__ A, double* __restrict__ scale_r,
                       const int* __restrict__ col_full, const double* __restrict__ params_x, double* __restrict__ params_c,
                       double* __restrict__ delta_r, double* __restrict__ out, IterParams ip) {
  extern __shared__ double lds[];
  const RedLayout RL{nr};
  const int tid = threadIdx.x, nt = blockDim.x;
  __shared__ int s_ok;
  for (int i = tid; i < nr; i += nt)
    if (ip.first) scale_r[i] = ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(red[RL.diagU() + i])) : 1.0;
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < nr; i += nt) A[(size_t)nr * nr + i] = scale_r[i] * (red[RL.gc() + i] + red[RL.corr() + i]);
  __threadfence_block();
  __syncthreads();
  double* ysol = A + (size_t)nr * nr;
  CholeskySolvePanelLDS(nr, A, ysol, &s_ok, lds,
                        PanelSource{red + RL.S(), scale_r, red + RL.diagU(), ip.min_lm_diagonal, ip.max_lm_diagonal, 1.0 / ip.radius, nullptr, nullptr, 0},
                        StageGate{nullptr, 0, 0, nullptr, nullptr, nullptr, 0});
  __syncthreads();
  ReducedStepEpilogue(nr, red, RL, scale_r, ysol, col_full, params_x, params_c, delta_r, out, s_ok > 0, lds);
}

// k_marker_reduced_solve for systems of up to 160 columns (26 camera + marker blocks: the 8 x 5000 x 16 benchmark has 132) with the WHOLE
// lower triangle in LDS.  CholeskySolvePanelLDS keeps a panel there and the matrix in memory: per 32-column panel the panel comes
// in, the strip of L to its left comes in, the panel goes back out — three dependent trips to memory around 7 us of factorisation,
// 12 us a panel, 77 us for the five of 132 columns (a quarter of the iteration once the elimination was split, ba_marker_split.hpp).
// Here the lower triangle lies in LDS as five column panels of 33-double rows (the right-hand side rides along as every panel's last
// row; padding columns are identity), loaded once; right-looking: per panel DiagFactorInverse (one wavefront), X = rows T' and the
// trailing panels' updates on the matrix cores, T kept where L_pp was; then L' x = y from LDS.  Nothing goes to memory but x.
// Same scaling and damping as PanelSource's (ba_cholesky.hpp); fixed orders: bitwise reproducible.
#define RSBA_MRS_MAXP 5
__host__ __device__ inline size_t MarkerSolveLdsDoubles(int nr) {
  const int np = (nr + RSBA_PB - 1) / RSBA_PB, npad = RSBA_PB * np;
  size_t rows = 0;
  for (int p = 0; p < np; ++p) rows += (size_t)(npad + 1 - RSBA_PB * p);
  return rows * RSBA_PLD + 2 * RSBA_PB * RSBA_PLD + RSBA_PB + 2 * (size_t)npad;   // panels (>= 2048 doubles: the epilogue's scratch, once they are done with) | T | Lt | invd | scl | y, then x
}
__global__ void __launch_bounds__(512)
k_marker_reduced_solve_lds(int nr, const double* __restrict__ red, double* __restrict__ A, double* __restrict__ scale_r,
                           const int* __restrict__ col_full, const double* __restrict__ params_x, double* __restrict__ params_c,
                           double* __restrict__ delta_r, double* __restrict__ out, IterParams ip) {
  extern __shared__ double lds[];
  const RedLayout RL{nr};
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nt >> 6, mi = lane & 15, kk = lane >> 4;
  const int np = (nr + RSBA_PB - 1) / RSBA_PB, npad = RSBA_PB * np;
  __shared__ int s_ok;
  __shared__ double s_red[8][4];
  int poff[RSBA_MRS_MAXP + 1];   // panel p: rows 32 p .. npad (the right-hand side) of its 32 columns
  poff[0] = 0;
#pragma unroll
  for (int p = 0; p < RSBA_MRS_MAXP; ++p) poff[p + 1] = poff[p] + (npad + 1 - RSBA_PB * p) * RSBA_PLD;
  double* T = lds + poff[np];
  double* Lt = T + RSBA_PB * RSBA_PLD;
  double* invd = Lt + RSBA_PB * RSBA_PLD;
  double* scl = invd + RSBA_PB;        // npad
  double* yv = scl + npad;             // npad: the damping diagonal while the panels are filled, then y, then x
  double* tmp32 = invd;                // (the back-substitution's 32 doubles: the inverse pivots are done with by then)
  if (tid == 0) s_ok = 1;
  // ---- everything the kernel reads from memory, asked for in ONE go: the thread's entry of the vectors (nr <= 160 < 512: one each),
  // then its entries of S — element e = tid + 512 u of panel p's (rows x 32) block, at most eleven a panel.  (Entry by entry under
  // their branches the loads were 30 dependent trips to memory a thread: 15 us of the kernel's 78.)
  const bool mine = tid < nr;
  const double my_du = mine ? red[RL.diagU() + tid] : 0.0, my_gc = mine ? red[RL.gc() + tid] : 0.0, my_corr = mine ? red[RL.corr() + tid] : 0.0;
  const double my_sc_old = (mine && !ip.first) ? scale_r[tid] : 1.0;
  const int my_cf = mine ? col_full[tid] : 0;
  constexpr int kMaxE = (RSBA_PB * RSBA_MRS_MAXP + 1 + 15) / 16;   // 11
  double sv[RSBA_MRS_MAXP][kMaxE];
#pragma unroll
  for (int p = 0; p < RSBA_MRS_MAXP; ++p) {
#pragma unroll
    for (int u = 0; u < kMaxE; ++u) {
      sv[p][u] = 0.0;
      if (u < kMaxE - 2 * p) {   // (panel p has 161 - 32 p rows at most: 11, 9, 7, 5, 3 elements a thread)
        const int kb = RSBA_PB * p, e = tid + u * nt, gi = kb + (e >> 5), gj = kb + (e & 31);
        if (p < np && gi < nr && gj < nr) sv[p][u] = red[RL.S() + (size_t)gi * nr + gj];
      }
    }
  }
  const double my_x = mine ? params_x[my_cf] : 0.0;
  if (tid < npad) {
    double sc = 1.0;
    if (mine) { sc = ip.first ? (ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(my_du)) : 1.0) : my_sc_old; if (ip.first) scale_r[tid] = sc; }
    scl[tid] = sc;
    yv[tid] = my_du;
  }
  __syncthreads();
  const double inv_radius = 1.0 / ip.radius;
#pragma unroll
  for (int p = 0; p < RSBA_MRS_MAXP; ++p) {
    if (p < np) {
      const int kb = RSBA_PB * p, R = npad + 1 - kb;
      double* Pan = lds + poff[p];
#pragma unroll
      for (int u = 0; u < kMaxE; ++u) {
        const int e = tid + u * nt;
        if (u < kMaxE - 2 * p && e < R * RSBA_PB) {
          const int r = e >> 5, c = e & 31, gi = kb + r, gj = kb + c;
          double v;
          if (gi == npad) v = 0.0;   // (the right-hand side: below)
          else if (gi >= nr || gj >= nr) v = gi == gj ? 1.0 : 0.0;
          else {
            v = sv[p][u] * (scl[gi] * scl[gj]);
            if (gi == gj) v += fmin(fmax(scl[gi] * scl[gi] * yv[gi], ip.min_lm_diagonal), ip.max_lm_diagonal) * inv_radius;
          }
          Pan[r * RSBA_PLD + c] = v;
        }
      }
    }
  }
  __syncthreads();   // (the damping diagonal in yv has been read)
  if (mine) { const int p = tid >> 5; (lds + poff[p])[(npad - RSBA_PB * p) * RSBA_PLD + (tid & 31)] = scl[tid] * (my_gc + my_corr); }   // s_j (g_j + corr_j): the panel's last row
  __syncthreads();
#pragma unroll
  for (int p = 0; p < RSBA_MRS_MAXP; ++p) {
    if (p >= np) break;
    const int R = npad + 1 - RSBA_PB * p;   // rows of the panel, the right-hand side's included
    double* Pan = lds + poff[p];
    if (wave == 0 && !DiagFactorInverseCall((lds_double*)Pan, RSBA_PB, (lds_double*)T, (lds_double*)Lt, (lds_double*)invd, lane) && lane == 0) s_ok = 0;
    __syncthreads();
    // X = rows T' for the rows below the diagonal block (CholeskySolvePanelLDS, step 4)
    {
      const int nrb = (R - RSBA_PB + 15) >> 4;
      for (int rb = wave; rb < nrb; rb += nwave) {
        const int prow = RSBA_PB + rb * 16 + mi;
        d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int qs = 0; qs < RSBA_PB; qs += 4) {
          const double a = prow < R ? Pan[prow * RSBA_PLD + qs + kk] : 0.0;
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[mi * RSBA_PLD + qs + kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(16 + mi) * RSBA_PLD + qs + kk], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();   // all of this block's rows are read before any is overwritten
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int r = RSBA_PB + rb * 16 + kk + 4 * tt;
          if (r < R) { Pan[r * RSBA_PLD + mi] = acc0[tt]; Pan[r * RSBA_PLD + 16 + mi] = acc1[tt]; }
        }
      }
    }
    __syncthreads();
    // T_p takes L_pp's place (the back-substitution wants T_p and the blocks below; L_pp itself is done with)
    for (int e = tid; e < RSBA_PB * RSBA_PB; e += nt) Pan[(e >> 5) * RSBA_PLD + (e & 31)] = T[(e >> 5) * RSBA_PLD + (e & 31)];
    // the trailing panels: (i, j) -= sum_k L(i, k) L(j, k) over this panel's columns k — 16 x 16 tiles, both operands from this panel
    {
      int tile0 = 0;
#pragma unroll
      for (int q = 1; q < RSBA_MRS_MAXP; ++q) {
        if (q <= p || q >= np) continue;
        const int Rq = npad + 1 - RSBA_PB * q, nrt = (Rq + 15) >> 4;     // rows of panel q, its 16-row tiles (two column tiles each)
        double* Pq = lds + poff[q];
        const int roff = RSBA_PB * (q - p);                                 // panel q's row 0 / column block inside panel p
        const int m = tile0 & (nwave - 1);
        for (int t = wave - m + (wave < m ? nwave : 0); t < 2 * nrt; t += nwave) {   // (tile tile0 + t is wavefront (tile0 + t) mod 8's)
          const int rt = t >> 1, ct = t & 1;
          const int prow = roff + 16 * rt + mi;
          d4_t acc = {0, 0, 0, 0};
#pragma unroll
          for (int qs = 0; qs < RSBA_PB; qs += 4) {
            const double a = prow < R ? Pan[prow * RSBA_PLD + qs + kk] : 0.0;
            const double b = Pan[(roff + 16 * ct + mi) * RSBA_PLD + qs + kk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
          }
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            const int r = 16 * rt + kk + 4 * tt;
            if (r < Rq) Pq[r * RSBA_PLD + 16 * ct + mi] -= acc[tt];
          }
        }
        tile0 += 2 * nrt;
      }
    }
    __syncthreads();
  }
  // y (the panels' last rows) and L' x = y, block by block from the last: one wavefront, lane (c, h) half of a column's terms
  if (wave == 0) {
    const int c = lane & 31, h = lane >> 5;
    for (int p = 0; p < np; ++p) if (lane < RSBA_PB) yv[RSBA_PB * p + lane] = (lds + poff[p])[(npad - RSBA_PB * p) * RSBA_PLD + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int p = np - 1; p >= 0; --p) {
      const double* Pan = lds + poff[p];
      // t = y_p - sum over the rows below of L(row, 32 p + c) x(row)
      double t = 0.0;
      const int nbelow = npad - RSBA_PB * (p + 1);
      for (int r = h; r < nbelow; r += 2) t = fma(Pan[(RSBA_PB + r) * RSBA_PLD + c], yv[RSBA_PB * (p + 1) + r], t);
      t += __shfl_xor(t, 32, 64);
      const double tv = yv[RSBA_PB * p + c] - t;
      __builtin_amdgcn_wave_barrier();
      if (lane < RSBA_PB) tmp32[lane] = tv;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // x_p = T_p' t  (T_p lies where L_pp was: T[i][c], i >= c)
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) x = fma(Pan[(16 * h + k) * RSBA_PLD + c], tmp32[16 * h + k], x);
      x += __shfl_xor(x, 32, 64);
      __builtin_amdgcn_wave_barrier();
      if (lane < RSBA_PB) yv[RSBA_PB * p + lane] = x;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  // the step of the reduced blocks and its norms (ReducedStepEpilogue's arithmetic on what this thread already holds; the lanes by
  // butterfly, the wavefronts in order: one barrier instead of a tree of eleven)
  double d2 = 0.0, x2 = 0.0, xc2 = 0.0, gm = 0.0;
  if (mine) {
    const double dd = -scl[tid] * yv[tid], xc = my_x + dd;
    delta_r[tid] = dd;
    params_c[my_cf] = xc;
    d2 = dd * dd; x2 = my_x * my_x; xc2 = xc * xc; gm = fabs(my_gc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    d2 += __shfl_xor(d2, off, 64); x2 += __shfl_xor(x2, off, 64); xc2 += __shfl_xor(xc2, off, 64); gm = fmax(gm, __shfl_xor(gm, off, 64));
  }
  if (lane == 0) { s_red[wave][0] = d2; s_red[wave][1] = x2; s_red[wave][2] = xc2; s_red[wave][3] = gm; }
  __syncthreads();
  if (tid == 0) {
    double t4[4] = {s_red[0][0], s_red[0][1], s_red[0][2], s_red[0][3]};
    for (int w8 = 1; w8 < nwave; ++w8) { t4[0] += s_red[w8][0]; t4[1] += s_red[w8][1]; t4[2] += s_red[w8][2]; t4[3] = fmax(t4[3], s_red[w8][3]); }
    out[0] = t4[0]; out[1] = t4[1]; out[2] = t4[2]; out[3] = t4[3]; out[4] = s_ok > 0 ? 1.0 : 0.0;
  }
  (void)A;
}

__global__ void __launch_bounds__(1024)
k_marker_chol_finish(int nr, const double* __restrict__ red, double* __restrict__ F, const double* __restrict__ scale_r,
                     const int* __restrict__ col_full, const double* __restrict__ params_x, double* __restrict__ params_c,
                     double* __restrict__ delta_r, double* __restrict__ out, const int* __restrict__ ok_flag) {
  extern __shared__ double lds[];
  const RedLayout RL{nr};
  const int tid = threadIdx.x, nt = blockDim.x;
  double* y = BackSubstituteBlocks(nr, F, lds);
  double* ysol = F + (size_t)nr * nr;
  for (int i = tid; i < nr; i += nt) ysol[i] = y[i];
  __threadfence_block();
  __syncthreads();
  ReducedStepEpilogue(nr, red, RL, scale_r, ysol, col_full, params_x, params_c, delta_r, out, *ok_flag, lds);
}

// Back-substitution of one time and everything that follows from its step, one wavefront per time:
//   pass 1  lanes over the time's residual blocks: delta_t = -E (g_t + sum J_t' (J_c delta_c + J_m delta_m)) (fixed
//           shuffle tree), candidate time pose;
//   pass 2  the same residual blocks again (their Jacobians are still in the L2): the model cost change
//           -(J d).(r + J d / 2) and the squared residuals of the candidate (four corners through the three candidate poses).
// Per workgroup (four times): |delta_t|^2, |x_t + delta_t|^2, model cost change, candidate sum of squares.
__global__ void __launch_bounds__(256)
k_time_backsub_terms(int T, const int* __restrict__ time_ptr, const int* __restrict__ time_full, const TimeSlots* __restrict__ ts,
                     const MarkerObs* __restrict__ mo, const double* __restrict__ obs8, const double* __restrict__ intr, double half_side,
                     const double* __restrict__ posec /* pose constants at x */, const double* __restrict__ tdata,
                     const double* __restrict__ delta_r, const double* __restrict__ params_x, double* __restrict__ params_c,
                     double* __restrict__ delta_t, double* __restrict__ bpart /* gridDim.x x 4 */) {
  __shared__ double s_part[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wave;
  double d2 = 0, xc2 = 0, mcc = 0, cc = 0;
  if (t < T) {
    const int o0 = time_ptr[t], o1 = time_ptr[t + 1], tf = time_full[t];
    double h[6] = {0, 0, 0, 0, 0, 0};
    for (int i = o0 + lane; i < o1; i += 64) {
      const TimeSlots s = ts[i];
      double dc[6], dm[6];
#pragma unroll
      for (int x = 0; x < 6; ++x) { dc[x] = s.col_cam >= 0 ? delta_r[s.col_cam + x] : 0.0; dm[x] = s.col_marker >= 0 ? delta_r[s.col_marker + x] : 0.0; }
      // the block's Jacobian rows, corner by corner, recomputed (never stored: see ElimArgs)
      const MarkerObs o = mo[i];
      const double* pcc = o.full_cam >= 0 ? posec + (size_t)(o.full_cam / 6) * CC_STRIDE : nullptr;
      const double* pct = posec + (size_t)(o.full_time / 6) * CC_STRIDE;
      const double* pcm = o.full_marker >= 0 ? posec + (size_t)(o.full_marker / 6) * CC_STRIDE : nullptr;
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        double rr[2], Jc[36];
        MarkerCornerResidualJacobian(pcc, pct, pcm, intr + 4 * o.camera, (k == 0 || k == 3) ? -half_side : half_side, k < 2 ? half_side : -half_side,
                                     obs8[8 * (size_t)i + 2 * k], obs8[8 * (size_t)i + 2 * k + 1], rr, Jc);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          double m = 0.0;
#pragma unroll
          for (int x = 0; x < 6; ++x) m += Jc[q * 18 + x] * dc[x] + Jc[q * 18 + 12 + x] * dm[x];
#pragma unroll
          for (int x = 0; x < 6; ++x) h[x] += Jc[q * 18 + 6 + x] * m;
        }
      }
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) {
      for (int off = 32; off > 0; off >>= 1) h[x] += __shfl_down(h[x], off, 64);
      h[x] = __shfl(h[x], 0, 64) + tdata[(size_t)t * 48 + 36 + x];
    }
    double dt[6], tc[6];   // the time's step and candidate pose, in every lane
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double sum = 0.0;
#pragma unroll
      for (int y = 0; y < 6; ++y) sum += tdata[(size_t)t * 48 + 6 * a + y] * h[y];
      dt[a] = -sum;
      tc[a] = params_x[tf + a] + dt[a];
      d2 += dt[a] * dt[a]; xc2 += tc[a] * tc[a];   // identical in every lane; lane 0's copy is used
    }
    if (lane < 6) {
      double dsel = dt[0], csel = tc[0];
#pragma unroll
      for (int a = 1; a < 6; ++a) if (lane == a) { dsel = dt[a]; csel = tc[a]; }
      delta_t[6 * t + lane] = dsel;
      params_c[tf + lane] = csel;
    }
    // pass 2
    const double cx[4] = {-half_side, half_side, half_side, -half_side};
    const double cy[4] = {half_side, half_side, -half_side, -half_side};
    for (int i = o0 + lane; i < o1; i += 64) {
      const TimeSlots s = ts[i];
      const MarkerObs o = mo[i];
      double dl[18];
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        dl[x] = s.col_cam >= 0 ? delta_r[s.col_cam + x] : 0.0;
        dl[6 + x] = dt[x];
        dl[12 + x] = s.col_marker >= 0 ? delta_r[s.col_marker + x] : 0.0;
      }
      {
        const double* pcc = o.full_cam >= 0 ? posec + (size_t)(o.full_cam / 6) * CC_STRIDE : nullptr;
        const double* pct = posec + (size_t)(o.full_time / 6) * CC_STRIDE;
        const double* pcm = o.full_marker >= 0 ? posec + (size_t)(o.full_marker / 6) * CC_STRIDE : nullptr;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
          double rr[2], Jc[36];
          MarkerCornerResidualJacobian(pcc, pct, pcm, intr + 4 * o.camera, (k == 0 || k == 3) ? -half_side : half_side, k < 2 ? half_side : -half_side,
                                       obs8[8 * (size_t)i + 2 * k], obs8[8 * (size_t)i + 2 * k + 1], rr, Jc);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            double m = 0.0;
#pragma unroll
            for (int x = 0; x < 18; ++x) m += Jc[q * 18 + x] * dl[x];
            mcc -= m * (rr[q] + 0.5 * m);
          }
        }
      }
      // candidate residuals: cameras and markers of the candidate are in params_c already (reduced solve), the time here
      const double fx = intr[4 * o.camera], fy = intr[4 * o.camera + 1], ppx = intr[4 * o.camera + 2], ppy = intr[4 * o.camera + 3];
      for (int k = 0; k < 4; ++k) {
        double pt[3] = {cx[k], cy[k], 0.0};
        if (o.full_marker >= 0) { const double* m = params_c + o.full_marker; RotateD(m, pt); pt[0] += m[3]; pt[1] += m[4]; pt[2] += m[5]; }
        RotateD(tc, pt); pt[0] += tc[3]; pt[1] += tc[4]; pt[2] += tc[5];
        if (o.full_cam >= 0) { const double* c = params_c + o.full_cam; RotateD(c, pt); pt[0] += c[3]; pt[1] += c[4]; pt[2] += c[5]; }
        const double r0 = fx * pt[0] / pt[2] + ppx - obs8[8 * (size_t)i + 2 * k];
        const double r1 = fy * pt[1] / pt[2] + ppy - obs8[8 * (size_t)i + 2 * k + 1];
        cc += r0 * r0 + r1 * r1;
      }
    }
  }
  // the lanes' sums in a fixed tree, then the four times of the workgroup in wave order
  for (int off = 32; off > 0; off >>= 1) { mcc += __shfl_down(mcc, off, 64); cc += __shfl_down(cc, off, 64); }
  if (lane == 0) { s_part[wave][0] = d2; s_part[wave][1] = xc2; s_part[wave][2] = mcc; s_part[wave][3] = cc; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    bpart[4 * blockIdx.x + k] = ((s_part[0][k] + s_part[1][k]) + s_part[2][k]) + s_part[3][k];
  }
}

// Pose constants of the poses behind the reduced columns (cameras, markers) of a parameter array: the candidate's, once the
// reduced solve has written them — k_time_backsub_wg rotates the candidate's corners with matrices instead of three
// sin / cos per corner.
__global__ void __launch_bounds__(64) k_pose_constants_reduced(int nred_poses, const int* __restrict__ col_full, const double* __restrict__ params,
                                                               double* __restrict__ posec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nred_poses) return;
  const int pose = col_full[6 * k] / 6;
  const double zero4[4] = {0.0, 0.0, 0.0, 0.0};
  double cc[CC_STRIDE];
  CameraConstants(params + 6 * (size_t)pose, zero4, cc);
#pragma unroll
  for (int q = 0; q < CC_STRIDE; ++q) posec[(size_t)pose * CC_STRIDE + q] = cc[q];
}

// k_time_backsub_terms with a WORKGROUP per time and a corner of a residual block per lane (round 4).  The wavefront-per-time
// form evaluates every corner's Jacobian rows twice — once for W_t delta_r, once more for the model cost change, which needs
// delta_t — on two residual blocks x four corners per lane one after the other, and rotates every candidate corner through three
// angle-axis poses (a sin / cos pair each).  Here a lane keeps what the second pass needs of its (at most kPer) corners in
// registers — J_r delta_r, the residual and the time block's two rows: 16 doubles a corner — so the rows are formed once; the
// candidate's cameras and markers come as rotation matrices (k_pose_constants_reduced), the time's own is formed once per lane.
// Sums in a fixed order: lanes by butterfly, the four wavefronts in order.  bpart: one entry (4 doubles) per time.
template <int kPer, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_time_backsub_wg(int T, const int* __restrict__ time_ptr, const int* __restrict__ time_full, const TimeSlots* __restrict__ ts,
                  const MarkerObs* __restrict__ mo, const double* __restrict__ obs8, const double* __restrict__ intr, double half_side,
                  const double* __restrict__ posec /* pose constants at x */, const double* __restrict__ posec_c /* candidate: cameras, markers */,
                  const double* __restrict__ tdata, const double* __restrict__ delta_r, const double* __restrict__ params_x,
                  double* __restrict__ params_c, double* __restrict__ delta_t, double* __restrict__ bpart /* T x 4 */,
                  const int* __restrict__ slot_ptr, const int* __restrict__ slot_col, const int* __restrict__ col_full) {
  constexpr int kWaves = kThreads / 64;
  __shared__ double s_h[kWaves][8];
  extern __shared__ double s_tab[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x;
  const int o0 = time_ptr[t], o1 = time_ptr[t + 1], tf = time_full[t], ncorner = 4 * (o1 - o0);
  // the shot's tables, once per workgroup instead of three dependent trips to memory per lane (slot -> column -> pose -> constants):
  // pose constants at x of the time (row 0) and of its slots' cameras / markers, the candidate's rotation and translation of
  // the slots, the slots' part of delta_r
  const int s0 = slot_ptr[t], nslot = slot_ptr[t + 1] - s0;
  double* s_pc = s_tab;                                  // [(nslot + 1)][CC_STRIDE]
  double* s_cd = s_pc + (size_t)(nslot + 1) * CC_STRIDE;   // [nslot][12]
  double* s_dr = s_cd + (size_t)nslot * 12;               // [nslot][6]
  for (int e = tid; e < (nslot + 1) * CC_STRIDE; e += kThreads) {
    const int ps = e / CC_STRIDE, q = e - ps * CC_STRIDE;
    const int pose = (ps == 0 ? tf : col_full[slot_col[s0 + ps - 1]]) / 6;
    s_pc[e] = posec[(size_t)pose * CC_STRIDE + q];
  }
  for (int e = tid; e < nslot * 12; e += kThreads) {
    const int ps = e / 12, q = e - 12 * ps;
    const int pose = col_full[slot_col[s0 + ps]] / 6;
    s_cd[e] = posec_c[(size_t)pose * CC_STRIDE + (q < 9 ? CC_R + q : CC_T + q - 9)];
  }
  for (int e = tid; e < nslot * 6; e += kThreads) { const int ps = e / 6; s_dr[e] = delta_r[slot_col[s0 + ps] + (e - 6 * ps)]; }
  // (lanes 0..5: their row of E, g_t and the time's parameters, wanted between the passes)
  double e_row[6] = {0, 0, 0, 0, 0, 0}, g_all[6] = {0, 0, 0, 0, 0, 0}, x_mine = 0.0;
  if (tid < 6) {
#pragma unroll
    for (int y = 0; y < 6; ++y) { e_row[y] = tdata[(size_t)t * 48 + 6 * tid + y]; g_all[y] = tdata[(size_t)t * 48 + 36 + y]; }
    x_mine = params_x[tf + tid];
  }
  __syncthreads();
  double m1[kPer][2], rk[kPer][2], jt[kPer][2][6];
  double h[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int e = tid + kThreads * u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      m1[u][q] = 0.0; rk[u][q] = 0.0;
#pragma unroll
      for (int x = 0; x < 6; ++x) jt[u][q][x] = 0.0;
    }
    if (e < ncorner) {
      const int i = o0 + (e >> 2), k = e & 3;
      const TimeSlots sl = ts[i];
      double dc[6], dm[6];
#pragma unroll
      for (int x = 0; x < 6; ++x) { dc[x] = sl.slot_cam >= 0 ? s_dr[6 * sl.slot_cam + x] : 0.0; dm[x] = sl.slot_marker >= 0 ? s_dr[6 * sl.slot_marker + x] : 0.0; }
      const double* pcc = sl.slot_cam >= 0 ? s_pc + (size_t)(1 + sl.slot_cam) * CC_STRIDE : nullptr;
      const double* pct = s_pc;
      const double* pcm = sl.slot_marker >= 0 ? s_pc + (size_t)(1 + sl.slot_marker) * CC_STRIDE : nullptr;
      double rr[2], Jc[36];
      MarkerCornerResidualJacobian(pcc, pct, pcm, intr + 4 * sl.camera, (k == 0 || k == 3) ? -half_side : half_side, k < 2 ? half_side : -half_side,
                                   obs8[8 * (size_t)i + 2 * k], obs8[8 * (size_t)i + 2 * k + 1], rr, Jc);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        double m = 0.0;
#pragma unroll
        for (int x = 0; x < 6; ++x) m += Jc[q * 18 + x] * dc[x] + Jc[q * 18 + 12 + x] * dm[x];
        m1[u][q] = m; rk[u][q] = rr[q];
#pragma unroll
        for (int x = 0; x < 6; ++x) { jt[u][q][x] = Jc[q * 18 + 6 + x]; h[x] += Jc[q * 18 + 6 + x] * m; }
      }
    }
  }
  // W_t delta_r: the lanes by butterfly (the same value in every lane), the wavefronts in order
#pragma unroll
  for (int x = 0; x < 6; ++x) {
    for (int off = 32; off > 0; off >>= 1) h[x] += __shfl_xor(h[x], off, 64);
    if (lane == 0) s_h[wave][x] = h[x];
  }
  __syncthreads();
  // the time's step and candidate pose: entry a on lane a of the first wavefront, the candidate's rotation matrix on one lane —
  // once per time, not once per lane (E h is 36 loads and as many FMAs, the matrix a sin / cos pair)
  __shared__ double s_dt[6], s_tc[6], s_rt[9], s_n2[2];
  if (tid < 6) {
    double hh[6];
#pragma unroll
    for (int x = 0; x < 6; ++x) {
      double sum = s_h[0][x];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) sum += s_h[w][x];
      hh[x] = sum + g_all[x];
    }
    double sum = 0.0;
#pragma unroll
    for (int y = 0; y < 6; ++y) sum += e_row[y] * hh[y];
    const double d = -sum, c = x_mine + d;
    s_dt[tid] = d; s_tc[tid] = c;
    delta_t[6 * t + tid] = d;
    params_c[tf + tid] = c;
  }
  __syncthreads();
  double dt[6], tc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) { dt[a] = s_dt[a]; tc[a] = s_tc[a]; }
  if (tid == 0) {
    const double zero4[4] = {0.0, 0.0, 0.0, 0.0};
    double cct[CC_STRIDE];
    CameraConstants(tc, zero4, cct);
#pragma unroll
    for (int q = 0; q < 9; ++q) s_rt[q] = cct[CC_R + q];
    double d2 = 0.0, xc2 = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) { d2 += dt[a] * dt[a]; xc2 += tc[a] * tc[a]; }
    s_n2[0] = d2; s_n2[1] = xc2;
  }
  // pass 2: the model cost change from what the lane kept (while one lane forms the candidate's rotation matrix), then the
  // candidate's residuals with the candidate's rotation matrices
  double mcc = 0.0, cc = 0.0;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int e = tid + kThreads * u;
    if (e < ncorner) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        double m = m1[u][q];
#pragma unroll
        for (int x = 0; x < 6; ++x) m += jt[u][q][x] * dt[x];
        mcc -= m * (rk[u][q] + 0.5 * m);
      }
    }
  }
  __syncthreads();
  double Rt[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) Rt[q] = s_rt[q];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int e = tid + kThreads * u;
    if (e < ncorner) {
      const int i = o0 + (e >> 2), k = e & 3;
      const TimeSlots sl = ts[i];
      const double fx = intr[4 * sl.camera], fy = intr[4 * sl.camera + 1], ppx = intr[4 * sl.camera + 2], ppy = intr[4 * sl.camera + 3];
      double pt[3] = {(k == 0 || k == 3) ? -half_side : half_side, k < 2 ? half_side : -half_side, 0.0};
      auto apply = [&](const double* R, const double* tr) {
        const double a0 = R[0] * pt[0] + R[1] * pt[1] + R[2] * pt[2], a1 = R[3] * pt[0] + R[4] * pt[1] + R[5] * pt[2], a2 = R[6] * pt[0] + R[7] * pt[1] + R[8] * pt[2];
        pt[0] = a0 + tr[0]; pt[1] = a1 + tr[1]; pt[2] = a2 + tr[2];
      };
      if (sl.slot_marker >= 0) { const double* pc = s_cd + 12 * sl.slot_marker; apply(pc, pc + 9); }
      apply(Rt, tc + 3);
      if (sl.slot_cam >= 0) { const double* pc = s_cd + 12 * sl.slot_cam; apply(pc, pc + 9); }
      const double r0 = fx * pt[0] / pt[2] + ppx - obs8[8 * (size_t)i + 2 * k];
      const double r1 = fy * pt[1] / pt[2] + ppy - obs8[8 * (size_t)i + 2 * k + 1];
      cc += r0 * r0 + r1 * r1;
    }
  }
  for (int off = 32; off > 0; off >>= 1) { mcc += __shfl_down(mcc, off, 64); cc += __shfl_down(cc, off, 64); }
  __syncthreads();   // (s_h has been read)
  if (lane == 0) { s_h[wave][0] = mcc; s_h[wave][1] = cc; }
  __syncthreads();
  if (tid == 0) {
    bpart[4 * t + 0] = s_n2[0]; bpart[4 * t + 1] = s_n2[1];
    double sm = s_h[0][0], sc = s_h[0][1];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) { sm += s_h[w][0]; sc += s_h[w][1]; }
    bpart[4 * t + 2] = sm; bpart[4 * t + 3] = sc;
  }
}

// (1024 threads, an entry = 32 bytes in two 16-byte loads, four entries in flight a thread, the lanes by butterfly and the sixteen
//  wavefronts in order: 256 threads walking 27 entries each, one dependent trip to memory after the other, and a tree of eight
//  barriers were 11 us for 200 KB)
__global__ void __launch_bounds__(1024)
k_marker_schur_finish(int nb_time, const double* __restrict__ bp_time, const double* __restrict__ red_scal,
                      const double* __restrict__ solve_out, double* __restrict__ res) {
  __shared__ double s[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double v[4] = {0, 0, 0, 0};
  const double2* __restrict__ bp2 = reinterpret_cast<const double2*>(bp_time);
  for (int b0 = tid; b0 < nb_time; b0 += 4 * 1024) {
    double2 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int b = min(b0 + 1024 * u, nb_time - 1); lo[u] = bp2[2 * b]; hi[u] = bp2[2 * b + 1]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (b0 + 1024 * u < nb_time) { v[0] += lo[u].x; v[1] += lo[u].y; v[2] += hi[u].x; v[3] += hi[u].y; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += __shfl_xor(v[k], off, 64);
  if (lane == 0) for (int k = 0; k < 4; ++k) s[k][wave] = v[k];
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 4; ++k) { double t = s[k][0]; for (int w = 1; w < 16; ++w) t += s[k][w]; s[k][0] = t; }
    res[RES_COST_X] = 0.5 * red_scal[0];
    res[RES_GMAX] = fmax(red_scal[3], solve_out[3]);
    res[RES_XNORM2] = red_scal[1] + solve_out[1];
    res[RES_POINT_FAIL] = red_scal[2];
    res[RES_CHOL_OK] = (solve_out[4] != 0.0 && red_scal[2] == 0.0) ? 1.0 : 0.0;
    res[RES_STEP2] = solve_out[0] + s[0][0];
    res[RES_XCNORM2] = solve_out[2] + s[1][0];
    res[RES_MCC] = s[2][0];
    double c = 0.5 * s[3][0];
    if (!(c == c) || !(fabs(c) <= DBL_MAX)) c = DBL_MAX;
    res[RES_COST_C] = c;
    res[RES_SUMSQ_C] = s[3][0];
    res[RES_STALL] = 0.0; res[RES_WAIT_TIMEOUT] = 0.0;
  }
}

}  // namespace rsba
#include "ba_marker_split.hpp"
namespace rsba {

struct MarkerSchurDevice {
  int N = 0, T = 0, nr = 0, nfull = 0, G = 0, dmax = 0, nb_time = 0;
  // the split elimination (ba_marker_split.hpp; RSBA_MT_SPLIT=0: k_time_eliminate)
  bool split = true;
  int nslots = 0, nx = 0, ncam_cols = 0, nx_threads = 0;
  int* x_order = nullptr;
  int4 *sb_blk = nullptr, *xi_blk = nullptr;
  int *slot_order = nullptr, *slot_time = nullptr, *sb_ptr = nullptr, *xi_ptr = nullptr, *xi_cc = nullptr, *xi_cm = nullptr, *xc_ptr = nullptr;
  double *sp = nullptr, *xout = nullptr, *tscal = nullptr;
  size_t lds_acc = 0;
  bool solve_lds = false;       // k_marker_reduced_solve_lds (the whole triangle in LDS: up to 160 reduced columns)
  bool split_backsub = false;   // k_mc_time_step + k_mc_candidate instead of k_time_backsub_wg / _terms
  int* blk_time = nullptr;
  int ncand_wg = 0;
  int acc_tiles = 0;   // > 0: k_mc_accumulate_mfma with that many tiles a wavefront
  int acc_tb = 2;      // ... and times per step (three tiles a wavefront; eight: one)
  hipStream_t fork_s[2] = {nullptr, nullptr};   // the three product kernels side by side
  hipEvent_t fork_ev[3] = {nullptr, nullptr, nullptr};
  bool backsub_wg = false;   // k_time_backsub_wg instead of k_time_backsub_terms
  double* posec_c = nullptr; // pose constants of the candidate's cameras and markers
  double half_side = 0;
  MarkerObs* mo = nullptr;
  TimeSlots* ts = nullptr;
  int *chunk_ptr = nullptr, *time_ptr = nullptr, *slot_ptr = nullptr, *slot_col = nullptr, *time_full = nullptr, *col_full = nullptr,
      *ok_flag = nullptr;
  double *obs8 = nullptr, *intr = nullptr, *params[2] = {nullptr, nullptr}, *params0 = nullptr;
  double *posec = nullptr, *ss_x = nullptr, *scale_t = nullptr, *scale_r = nullptr, *tdata = nullptr,
         *part = nullptr, *red = nullptr, *A = nullptr, *Wm = nullptr, *delta_r = nullptr, *delta_t = nullptr, *bp_time = nullptr,
         *solve_out = nullptr, *res = nullptr;
  int cur = 0;
  int* tc_flags = nullptr;   // persistent tiled factorisation of a large reduced system (ba_cholesky_tiles.hpp)
  double* tc_hand = nullptr; // ... its diagonal chain's hand-over buffers
  int* tc_map = nullptr;     // ... which tile each workgroup takes (TileOrder)
  int tc_np = 0, tc_nrt = 0, tc_tiles = 0, tc_tag = 0;
  size_t lds_elim = 0;
  bool lds_s = false;   // the chunk sums of S live in LDS

  void Free() {
    void* ptrs[] = {mo, ts, chunk_ptr, time_ptr, slot_ptr, slot_col, time_full, col_full, ok_flag, obs8, intr, params[0], params[1],
                    params0, posec, posec_c, ss_x, scale_t, scale_r, tdata, part, red, A, Wm, delta_r, delta_t, bp_time, solve_out, res, tc_flags, tc_hand, tc_map,
                    slot_order, slot_time, sb_ptr, sb_blk, xi_ptr, xi_blk, xi_cc, xi_cm, xc_ptr, sp, xout, tscal, blk_time, x_order};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    mo = nullptr; ts = nullptr;
    for (auto& q : fork_s) if (q) { (void)hipStreamDestroy(q); q = nullptr; }
    for (auto& q : fork_ev) if (q) { (void)hipEventDestroy(q); q = nullptr; }
  }

  // true when the problem should take this path (schur_impl: 0 never, 1 when the dense system outgrows one workgroup, 2 always)
  static bool Wanted(const rsba_problem& p, int schur_impl) {
    if (schur_impl == 0) return false;
    if (schur_impl >= 2) return true;
    const int nblocks = p.num_cameras + p.num_times + p.num_markers;
    return 6 * nblocks > RSBA_CHOL_MAXN;
  }

  int Upload(const rsba_problem& p) {
    N = (int)p.num_observations; nfull = (int)p.parameters.size(); half_side = p.marker_side / 2;
    if (N <= 0) return RSBA_ERR_ARG;
    const int C = p.num_cameras, Tn = p.num_times, M = p.num_markers;
    // reduced blocks: the cameras and markers some residual uses, cameras first
    std::vector<int> red_col(C + M, -1);
    std::vector<char> used(C + M, 0), tused(Tn, 0);
    for (int i = 0; i < N; ++i) {
      if (p.uses_camera(i)) used[p.camera_index[i]] = 1;
      if (p.uses_marker(i)) used[C + p.marker_index[i]] = 1;
      tused[p.time_index[i]] = 1;
    }
    std::vector<int> cf;
    int K = 0;
    for (int b = 0; b < C + M; ++b) if (used[b]) {
      red_col[b] = 6 * K++;
      const int full = 6 * (b < C ? b : Tn + b);   // [C | T | M] x 6
      for (int q = 0; q < 6; ++q) cf.push_back(full + q);
    }
    nr = 6 * K;
    if (nr == 0) return RSBA_ERR_UNSUPPORTED;   // nothing but time blocks: the dense path handles it
    std::vector<int> tid_of(Tn, -1), tfull;
    T = 0;
    for (int t = 0; t < Tn; ++t) if (tused[t]) { tid_of[t] = T++; tfull.push_back(6 * (C + t)); }
    // residual blocks in time order (stable)
    std::vector<int> order(N);
    for (int i = 0; i < N; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return p.time_index[x] < p.time_index[y]; });
    std::vector<int> tptr(T + 1, 0), sptr(T + 1, 0), scol;
    std::vector<MarkerObs> hmo(N);
    std::vector<TimeSlots> hts(N);
    std::vector<double> hobs(8 * (size_t)N);
    for (int k = 0; k < N; ++k) tptr[tid_of[p.time_index[order[k]]] + 1]++;
    for (int t = 0; t < T; ++t) tptr[t + 1] += tptr[t];
    dmax = 0;
    std::vector<double> work(T);
    for (int t = 0; t < T; ++t) {
      std::vector<int> cols;
      for (int k = tptr[t]; k < tptr[t + 1]; ++k) {
        const int i = order[k];
        if (p.uses_camera(i)) cols.push_back(red_col[p.camera_index[i]]);
        if (p.uses_marker(i)) cols.push_back(red_col[C + p.marker_index[i]]);
      }
      std::sort(cols.begin(), cols.end());
      cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
      {
        // the kernel relies on one residual block per (time, camera, marker); the same detection listed twice goes to
        // the dense path
        std::vector<std::pair<int, int>> cm;
        for (int k = tptr[t]; k < tptr[t + 1]; ++k) cm.emplace_back(p.camera_index[order[k]], p.marker_index[order[k]]);
        std::sort(cm.begin(), cm.end());
        if (std::adjacent_find(cm.begin(), cm.end()) != cm.end()) return RSBA_ERR_UNSUPPORTED;
      }
      for (int k = tptr[t]; k < tptr[t + 1]; ++k) {
        const int i = order[k];
        MarkerObs& o = hmo[k];
        o.full_cam = p.uses_camera(i) ? 6 * p.camera_block(i) : -1; o.full_time = 6 * p.time_block(i);
        o.full_marker = p.uses_marker(i) ? 6 * p.marker_block(i) : -1;
        o.act_cam = o.act_time = o.act_marker = -1; o.camera = p.camera_index[i]; o.pad = 0;
        TimeSlots& s = hts[k];
        s.col_cam = p.uses_camera(i) ? red_col[p.camera_index[i]] : -1;
        s.col_marker = p.uses_marker(i) ? red_col[C + p.marker_index[i]] : -1;
        s.camera = p.camera_index[i]; s.pad = 0;
        s.slot_cam = s.col_cam >= 0 ? (int)(std::lower_bound(cols.begin(), cols.end(), s.col_cam) - cols.begin()) : -1;
        s.slot_marker = s.col_marker >= 0 ? (int)(std::lower_bound(cols.begin(), cols.end(), s.col_marker) - cols.begin()) : -1;
        memcpy(&hobs[8 * (size_t)k], &p.observations[8 * (size_t)i], 8 * sizeof(double));
      }
      sptr[t + 1] = sptr[t] + (int)cols.size();
      scol.insert(scol.end(), cols.begin(), cols.end());
      const int d = 6 * (int)cols.size();
      dmax = std::max(dmax, d);
      work[t] = (double)d * d * (1.0 + 0.25 * (tptr[t + 1] - tptr[t])) + 2000.0;
    }
    if (dmax > RSBA_MT_MAXD) return RSBA_ERR_UNSUPPORTED;
    // chunks of consecutive times, balanced by work; the partial systems together stay below 2 GB
    const RedLayout RL{nr};
    const PartLayout PL{nr};
    lds_elim = (size_t)(13 * dmax + 96 + RSBA_MT_TILE * (RSBA_MT_JLD + 9) + (dmax / 6 + 2) * CC_STRIDE) * sizeof(double) + (size_t)(2 * RSBA_MT_TILE + 3 * (dmax / 6 + 1) + 2) * sizeof(int);
    lds_s = lds_elim + (PL.packed() + 3 * (size_t)nr) * sizeof(double) <= 156 * 1024;
    if (lds_s) lds_elim += (PL.packed() + 3 * (size_t)nr) * sizeof(double);
    const bool lds_s_elim = lds_s;   // (k_time_eliminate's own choice, should the split kernels not take the problem)
    split = !(getenv("RSBA_MT_SPLIT") && atoi(getenv("RSBA_MT_SPLIT")) == 0);
    if (split) {
      const int per_wave = (AccMfmaTiles(nr) + 15) / 16;
      acc_tiles = (per_wave <= 8 && !(getenv("RSBA_MT_ACC_MFMA") && atoi(getenv("RSBA_MT_ACC_MFMA")) == 0)) ? (per_wave <= 3 ? 3 : 8) : 0;
      if (acc_tiles > 0) {
        lds_s = true;   // (the chunk's sums never touch memory before the end: one chunk per CU, as with the sums in LDS)
        if (acc_tiles == 8) acc_tb = 1;
        lds_acc = AccMfmaLdsBytes(nr, acc_tb);
      } else {
        lds_s = AccLdsBytes(dmax, PL.packed() + 3 * (size_t)nr) <= 156 * 1024;
        lds_acc = AccLdsBytes(dmax, lds_s ? PL.packed() + 3 * (size_t)nr : 0);
      }
      if (lds_acc > 156 * 1024) {
        // times that touch more blocks than the accumulation's double-buffered records hold in LDS (~110 of the 170 the model allows):
        // round 4's kernel, which stages 32 residual blocks at a time whatever the width
        split = false; acc_tiles = 0; lds_s = lds_s_elim;
      }
    }
    // Without the LDS accumulators every entry is a read-modify-write in the partial system: few enough workgroups that
    // their partial systems stay in the L2s (8 x 4 MB) then, as many as there are times otherwise (at most 1024).
    // With the sums in LDS a workgroup fills a CU: one chunk per CU (more chunks only add partial systems to write, to
    // zero and to reduce: 1024 chunks cost 11 % of the iteration on 5000 times)
    size_t cap_lds = 256;
    { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cap_lds = (size_t)prop.multiProcessorCount; }
    if (getenv("RSBA_MT_CHUNKS")) cap_lds = (size_t)std::max(1, atoi(getenv("RSBA_MT_CHUNKS")));
    const size_t cap = lds_s ? cap_lds : std::min<size_t>(1024, std::max<size_t>(64, ((size_t)24 << 20) / (PL.size() * sizeof(double))));
    G = (int)std::min<size_t>((size_t)T, cap);
    std::vector<int> cptr(G + 1, 0);
    {
      double total = 0; for (double w : work) total += w;
      double acc = 0; int g = 0;
      for (int t = 0; t < T; ++t) {
        acc += work[t];
        // close chunk g when it has its share and enough times remain for the others
        while (g < G - 1 && acc >= total * (g + 1) / G && T - (t + 1) >= G - 1 - g) cptr[++g] = t + 1;
      }
      while (g < G) cptr[++g] = T;
      // chunks may not be empty in the middle: compact
      std::vector<int> c2; c2.push_back(0);
      for (int k = 1; k <= G; ++k) if (cptr[k] > c2.back()) c2.push_back(cptr[k]);
      G = (int)c2.size() - 1; cptr = c2;
    }
    // the split elimination's tables: a (time, slot)'s residual blocks, the thread order, the camera-marker pairs of every chunk
    std::vector<int> h_slot_time, h_sb_ptr, h_order, h_xi_ptr, h_xi_cc, h_xi_cm, h_xc_ptr, h_xorder;
    std::vector<int4> h_sb_blk, h_xi_blk;
    auto pose_of_col = [&](int col) { return col >= 0 ? cf[col] / 6 : -1; };   // reduced column -> pose in the parameter array
    const int kShareFrom = 10;   // an item of k_mc_cross with this many residual blocks or more runs on two neighbouring lanes
    if (split) {
      nslots = (int)scol.size();
      ncam_cols = 0;
      for (int b = 0; b < C; ++b) if (used[b]) ncam_cols += 6;
      h_slot_time.resize(nslots);
      h_sb_ptr.assign(nslots + 1, 0);
      for (int t = 0; t < T; ++t) {
        for (int S = sptr[t]; S < sptr[t + 1]; ++S) h_slot_time[S] = t;
        for (int k = tptr[t]; k < tptr[t + 1]; ++k) {
          if (hts[k].slot_cam >= 0) h_sb_ptr[sptr[t] + hts[k].slot_cam + 1]++;
          if (hts[k].slot_marker >= 0) h_sb_ptr[sptr[t] + hts[k].slot_marker + 1]++;
        }
      }
      for (int S = 0; S < nslots; ++S) h_sb_ptr[S + 1] += h_sb_ptr[S];
      h_sb_blk.resize(h_sb_ptr[nslots]);
      {
        std::vector<int> fill(h_sb_ptr.begin(), h_sb_ptr.end() - 1);
        for (int t = 0; t < T; ++t)
          for (int k = tptr[t]; k < tptr[t + 1]; ++k) {
            if (hts[k].slot_cam >= 0) h_sb_blk[fill[sptr[t] + hts[k].slot_cam]++] = int4{k, pose_of_col(hts[k].col_marker), hts[k].camera, 0};
            if (hts[k].slot_marker >= 0) h_sb_blk[fill[sptr[t] + hts[k].slot_marker]++] = int4{k, pose_of_col(hts[k].col_cam), hts[k].camera, 0};
          }
      }
      h_order.resize(nslots);
      for (int S = 0; S < nslots; ++S) h_order[S] = S;
      std::stable_sort(h_order.begin(), h_order.end(), [&](int x, int y) {
        const bool cx = scol[x] < ncam_cols, cy = scol[y] < ncam_cols;
        if (cx != cy) return cx;
        return h_sb_ptr[x + 1] - h_sb_ptr[x] > h_sb_ptr[y + 1] - h_sb_ptr[y];
      });
      h_xc_ptr.assign(G + 1, 0);
      h_xi_ptr.push_back(0);
      for (int g = 0; g < G; ++g) {
        std::vector<std::pair<std::pair<int, int>, int>> items;   // ((camera column, marker column), residual block)
        for (int t = cptr[g]; t < cptr[g + 1]; ++t)
          for (int k = tptr[t]; k < tptr[t + 1]; ++k)
            if (hts[k].col_cam >= 0 && hts[k].col_marker >= 0) items.push_back({{hts[k].col_cam, hts[k].col_marker}, k});
        std::stable_sort(items.begin(), items.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (size_t i = 0; i < items.size(); ++i) {
          if (i == 0 || items[i].first != items[i - 1].first) {
            if (i != 0) h_xi_ptr.push_back((int)h_xi_blk.size());
            h_xi_cc.push_back(items[i].first.first); h_xi_cm.push_back(items[i].first.second);
          }
          h_xi_blk.push_back(int4{items[i].second, hmo[items[i].second].full_time / 6, hts[items[i].second].camera, 0});
        }
        if (!items.empty()) h_xi_ptr.push_back((int)h_xi_blk.size());
        h_xc_ptr[g + 1] = (int)h_xi_cc.size();
      }
      nx = (int)h_xi_cc.size();
      {
        std::vector<int> ord(nx);
        for (int it = 0; it < nx; ++it) ord[it] = it;
        auto len = [&](int it) { return h_xi_ptr[it + 1] - h_xi_ptr[it]; };
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return len(x) > len(y); });
        for (int it : ord) if (len(it) >= kShareFrom) { h_xorder.push_back(4 * it + 2); h_xorder.push_back(4 * it + 3); }
        for (int it : ord) if (len(it) < kShareFrom) h_xorder.push_back(4 * it);
        nx_threads = (int)h_xorder.size();
      }
    }
    // (k_time_backsub_wg: a corner of a residual block per lane, two per lane at most; wider times take the wavefront-per-time kernel)
    { int widest = 0; for (int t = 0; t < T; ++t) widest = std::max(widest, tptr[t + 1] - tptr[t]); backsub_wg = widest <= 128 /* 4 x 128 corners = 512 lanes */ && !(getenv("RSBA_MT_BACKSUB_WG") && atoi(getenv("RSBA_MT_BACKSUB_WG")) == 0); }
    nb_time = backsub_wg ? T : (T + 3) / 4;
    split_backsub = split && !(getenv("RSBA_MT_SPLIT_BACKSUB") && atoi(getenv("RSBA_MT_SPLIT_BACKSUB")) == 0);
    if (split_backsub) { ncand_wg = (N + 255) / 256; nb_time = T + ncand_wg; }
    auto al = [](void** q, size_t bytes) { return hipMalloc(q, std::max<size_t>(bytes, 8)) == hipSuccess; };
    const size_t nA = (size_t)(nr + 2) * nr;
    if (!al((void**)&mo, N * sizeof(MarkerObs)) || !al((void**)&ts, N * sizeof(TimeSlots)) || !al((void**)&chunk_ptr, (G + 1) * 4) ||
        !al((void**)&time_ptr, (T + 1) * 4) || !al((void**)&slot_ptr, (T + 1) * 4) || !al((void**)&slot_col, scol.size() * 4) ||
        !al((void**)&time_full, T * 4) || !al((void**)&col_full, nr * 4) || !al((void**)&ok_flag, 4) ||
        !al((void**)&obs8, 8 * (size_t)N * 8) || !al((void**)&intr, p.intrinsics.size() * 8) || !al((void**)&params[0], nfull * 8) ||
        !al((void**)&params[1], nfull * 8) || !al((void**)&params0, nfull * 8) || !al((void**)&posec, (size_t)(nfull / 6) * CC_STRIDE * 8) ||
        !al((void**)&posec_c, (size_t)(nfull / 6) * CC_STRIDE * 8) ||
        !al((void**)&ss_x, N * 8) || !al((void**)&scale_t, 6 * (size_t)T * 8) ||
        !al((void**)&scale_r, nr * 8) || !al((void**)&tdata, 48 * (size_t)T * 8) || !al((void**)&part, (size_t)G * PL.size() * 8) ||
        !al((void**)&red, RL.size() * 8) || !al((void**)&A, nA * 8) || (nr > RSBA_CHOL_MAXN && !al((void**)&Wm, nA * 8)) ||
        !al((void**)&delta_r, nr * 8) || !al((void**)&delta_t, 6 * (size_t)T * 8) || !al((void**)&bp_time, 4 * (size_t)nb_time * 8) ||
        !al((void**)&solve_out, 8 * 8) || !al((void**)&res, RES_SIZE * 8))
      return RSBA_ERR_HIP;
    auto up = [](void* d, const void* h, size_t bytes) { return bytes == 0 || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) == hipSuccess; };
    if (!up(mo, hmo.data(), N * sizeof(MarkerObs)) || !up(ts, hts.data(), N * sizeof(TimeSlots)) || !up(chunk_ptr, cptr.data(), (G + 1) * 4) ||
        !up(time_ptr, tptr.data(), (T + 1) * 4) || !up(slot_ptr, sptr.data(), (T + 1) * 4) || !up(slot_col, scol.data(), scol.size() * 4) ||
        !up(time_full, tfull.data(), T * 4) || !up(col_full, cf.data(), nr * 4) ||
        !up(obs8, hobs.data(), 8 * (size_t)N * 8) || !up(intr, p.intrinsics.data(), p.intrinsics.size() * 8) ||
        !up(params0, p.parameters.data(), nfull * 8))
      return RSBA_ERR_HIP;
    if (hipMemset(res, 0, RES_SIZE * 8) != hipSuccess) return RSBA_ERR_HIP;
    if (split) {
      if (!al((void**)&slot_order, (size_t)nslots * 4) || !al((void**)&x_order, (size_t)nx_threads * 4) || !al((void**)&slot_time, (size_t)nslots * 4) || !al((void**)&sb_ptr, ((size_t)nslots + 1) * 4) ||
          !al((void**)&sb_blk, h_sb_blk.size() * sizeof(int4)) || !al((void**)&xi_ptr, ((size_t)nx + 1) * 4) || !al((void**)&xi_blk, h_xi_blk.size() * sizeof(int4)) ||
          !al((void**)&xi_cc, (size_t)nx * 4) || !al((void**)&xi_cm, (size_t)nx * 4) || !al((void**)&xc_ptr, ((size_t)G + 1) * 4) ||
          !al((void**)&sp, (size_t)nslots * RSBA_SP_STRIDE * 8) || !al((void**)&xout, (size_t)nx * 36 * 8) || !al((void**)&tscal, 4 * (size_t)T * 8))
        return RSBA_ERR_HIP;
      if (split_backsub) {
        std::vector<int> h_bt(N);
        for (int t = 0; t < T; ++t) for (int k = tptr[t]; k < tptr[t + 1]; ++k) h_bt[k] = t;
        if (!al((void**)&blk_time, (size_t)N * 4) || !up(blk_time, h_bt.data(), (size_t)N * 4)) return RSBA_ERR_HIP;
      }
      if (!up(slot_order, h_order.data(), (size_t)nslots * 4) || !up(x_order, h_xorder.data(), (size_t)nx_threads * 4) || !up(slot_time, h_slot_time.data(), (size_t)nslots * 4) ||
          !up(sb_ptr, h_sb_ptr.data(), ((size_t)nslots + 1) * 4) || !up(sb_blk, h_sb_blk.data(), h_sb_blk.size() * sizeof(int4)) ||
          !up(xi_ptr, h_xi_ptr.data(), ((size_t)nx + 1) * 4) || !up(xi_blk, h_xi_blk.data(), h_xi_blk.size() * sizeof(int4)) ||
          !up(xi_cc, h_xi_cc.data(), (size_t)nx * 4) || !up(xi_cm, h_xi_cm.data(), (size_t)nx * 4) || !up(xc_ptr, h_xc_ptr.data(), ((size_t)G + 1) * 4))
        return RSBA_ERR_HIP;
      const void* kacc = acc_tiles == 3 ? (acc_tb == 2 ? (const void*)k_mc_accumulate_mfma<3, 2> : (const void*)k_mc_accumulate_mfma<3, 1>) : acc_tiles == 8 ? (const void*)k_mc_accumulate_mfma<8, 1>
                         : lds_s ? (const void*)k_mc_accumulate<true> : (const void*)k_mc_accumulate<false>;
      if (lds_acc > 48 * 1024 && hipFuncSetAttribute(kacc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc) != hipSuccess) return RSBA_ERR_HIP;
      if (!(getenv("RSBA_MT_FORK") && atoi(getenv("RSBA_MT_FORK")) == 0)) {
        bool ok = true;
        for (auto& q : fork_s) ok = ok && hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess;
        for (auto& q : fork_ev) ok = ok && hipEventCreateWithFlags(&q, hipEventDisableTiming) == hipSuccess;
        if (!ok) return RSBA_ERR_HIP;
      }
    }
    if (backsub_wg) {
      const size_t lds_bw = (size_t)(dmax / 6 + 1) * (CC_STRIDE + 12 + 6) * sizeof(double);
      if (lds_bw > 48 * 1024 && hipFuncSetAttribute((const void*)k_time_backsub_wg<2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bw) != hipSuccess) return RSBA_ERR_HIP;
    }
    if (!split && lds_elim > 48 * 1024 &&
        hipFuncSetAttribute(lds_s ? (const void*)k_time_eliminate<true> : (const void*)k_time_eliminate<false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_elim) != hipSuccess) return RSBA_ERR_HIP;
    if (nr <= RSBA_CHOL_MAXN) {
      const size_t lds_c = std::max((size_t)4 * 1024, CholeskyLdsDoubles(nr)) * sizeof(double);
      if (lds_c > 48 * 1024 &&
          hipFuncSetAttribute((const void*)k_marker_reduced_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return RSBA_ERR_HIP;
      solve_lds = nr <= RSBA_PB * RSBA_MRS_MAXP && MarkerSolveLdsDoubles(nr) * sizeof(double) <= 156 * 1024 && !(getenv("RSBA_MT_SOLVE_LDS") && atoi(getenv("RSBA_MT_SOLVE_LDS")) == 0);
      if (getenv("RSBA_DEBUG")) fprintf(stderr, "rsba: marker reduced system %d columns, solve in LDS: %d (%zu bytes)\n", nr, (int)solve_lds, MarkerSolveLdsDoubles(nr) * sizeof(double));
      if (solve_lds && hipFuncSetAttribute((const void*)k_marker_reduced_solve_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MarkerSolveLdsDoubles(nr) * sizeof(double))) != hipSuccess) return RSBA_ERR_HIP;
    } else {
      if (hipFuncSetAttribute((const void*)k_chol_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CholStepLdsDoubles() * sizeof(double))) != hipSuccess)
        return RSBA_ERR_HIP;
      int dev = 0; hipDeviceProp_t prop;
      const char* e2 = getenv("RSBA_CHOL_TILES");
      if (!(e2 && atoi(e2) == 0) && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        const int m = MultiCholPadded(nr), nrt = (m + 1 + 63) / 64, ntiles = nrt * (nrt + 1) / 2;
        if (ntiles <= 2 * prop.multiProcessorCount) {
          tc_np = m / RSBA_PB; tc_nrt = nrt; tc_tiles = ntiles;
          const size_t nflags = (size_t)tc_np * (nrt + 1) + 1;
          if (!al((void**)&tc_flags, nflags * sizeof(int)) || hipMemset(tc_flags, 0, nflags * sizeof(int)) != hipSuccess) return RSBA_ERR_HIP;
          const size_t hand_bytes = (size_t)2 * nrt * kTileHandDoubles * sizeof(double);   // the diagonal chain's hand-overs: the sentinel everywhere
          {
            const std::vector<int> order = TileOrder(nrt, prop.multiProcessorCount);
            if (!al((void**)&tc_map, order.size() * sizeof(int)) || hipMemcpy(tc_map, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return RSBA_ERR_HIP;
          }
          if (!al((void**)&tc_hand, hand_bytes) || hipMemset(tc_hand, 0xff, hand_bytes) != hipSuccess) return RSBA_ERR_HIP;
          if (hipFuncSetAttribute((const void*)k_chol_tiles_persistent, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(TileCholLdsDoubles() * sizeof(double))) != hipSuccess) return RSBA_ERR_HIP;
        }
      }
    }
    return RSBA_OK;
  }
  int Reset(hipStream_t st) {
    if (hipMemcpyAsync(params[0], params0, nfull * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return RSBA_ERR_HIP;
    if (hipMemcpyAsync(params[1], params0, nfull * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return RSBA_ERR_HIP;
    cur = 0;
    return RSBA_OK;
  }
  void Accept() { cur = 1 - cur; }

  template <typename Timer>
  int Step(hipStream_t st, const rsba_options& o, double radius, bool first, double* res_host, Timer& Tm) {
    IterParams ip;
    ip.radius = radius; ip.min_lm_diagonal = o.min_lm_diagonal; ip.max_lm_diagonal = o.max_lm_diagonal; ip.huber_delta = 0.0;
    ip.first = first ? 1 : 0; ip.jacobi_scaling = o.jacobi_scaling;
    const int x = cur, c = 1 - cur;
    if (hipMemcpyAsync(params[c], params[x], nfull * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return RSBA_ERR_HIP;
    auto chk = [&](const char* what) {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { fprintf(stderr, "rsba: %s launch failed: %s\n", what, hipGetErrorString(e)); return false; }
      return true;
    };
    if (!chk("(before marker step)")) return RSBA_ERR_HIP;
    const RedLayout RL{nr};
    Tm.Begin("k_pose_constants", st);
    k_pose_constants<<<(nfull / 6 + 255) / 256, 256, 0, st>>>(nfull / 6, params[x], posec);
    Tm.End(st);
    ElimArgs ea{nr, dmax, (int)N, chunk_ptr, time_ptr, slot_ptr, slot_col, time_full, col_full, ts, mo, obs8, intr, posec, half_side, params[x], scale_t, tdata, part, ip};
    if (split) {
      SplitArgs sa{nslots, T, nx, ncam_cols, nx_threads, slot_order, x_order, slot_time, slot_col, sb_ptr, sb_blk, time_ptr, time_full, col_full, ts, mo, obs8, intr, posec, half_side,
                   xi_ptr, xi_blk, xi_cc, xi_cm, sp, xout};
      // the three product kernels are independent and none fills the chip: side by side on three streams (one after the other
      // when every kernel is timed)
      const bool fork = fork_s[0] != nullptr && !Tm.enabled();
      hipStream_t s_time = fork ? fork_s[0] : st, s_cross = fork ? fork_s[1] : st;
      if (fork) {
        if (hipEventRecord(fork_ev[0], st) != hipSuccess || hipStreamWaitEvent(s_time, fork_ev[0], 0) != hipSuccess ||
            hipStreamWaitEvent(s_cross, fork_ev[0], 0) != hipSuccess) return RSBA_ERR_HIP;
      }
      Tm.Begin("k_mc_slot_products", st);
      k_mc_slot_products<<<(nslots + 255) / 256, 256, 0, st>>>(sa);
      Tm.End(st);
      Tm.Begin("k_mc_time_products", st);
      k_mc_time_products<<<(T + 3) / 4, 256, 0, s_time>>>(sa, ip, params[x], scale_t, tdata, tscal);
      Tm.End(st);
      if (nx > 0) {
        Tm.Begin("k_mc_cross", st);
        k_mc_cross<<<(nx_threads + 255) / 256, 256, 0, s_cross>>>(sa);
        Tm.End(st);
      }
      if (fork) {
        if (hipEventRecord(fork_ev[1], s_time) != hipSuccess || hipEventRecord(fork_ev[2], s_cross) != hipSuccess ||
            hipStreamWaitEvent(st, fork_ev[1], 0) != hipSuccess || hipStreamWaitEvent(st, fork_ev[2], 0) != hipSuccess) return RSBA_ERR_HIP;
      }
      AccArgs aa{nr, dmax, chunk_ptr, slot_ptr, slot_col, sp, tdata, tscal, xc_ptr, xi_cc, xi_cm, xout, part};
      Tm.Begin("k_mc_accumulate", st);
      if (acc_tiles == 3 && acc_tb == 2) k_mc_accumulate_mfma<3, 2><<<G, RSBA_MT_THREADS, lds_acc, st>>>(aa);
      else if (acc_tiles == 3) k_mc_accumulate_mfma<3, 1><<<G, RSBA_MT_THREADS, lds_acc, st>>>(aa);
      else if (acc_tiles == 8) k_mc_accumulate_mfma<8, 1><<<G, RSBA_MT_THREADS, lds_acc, st>>>(aa);
      else if (lds_s) k_mc_accumulate<true><<<G, RSBA_MT_THREADS, lds_acc, st>>>(aa);
      else k_mc_accumulate<false><<<G, RSBA_MT_THREADS, lds_acc, st>>>(aa);
      Tm.End(st);
      if (!chk("split elimination")) return RSBA_ERR_HIP;
    } else {
      Tm.Begin("k_time_eliminate", st);
      if (lds_s) k_time_eliminate<true><<<G, RSBA_MT_THREADS, lds_elim, st>>>(ea);
      else k_time_eliminate<false><<<G, RSBA_MT_THREADS, lds_elim, st>>>(ea);
      Tm.End(st);
      if (!chk("k_time_eliminate")) return RSBA_ERR_HIP;
    }
    Tm.Begin("k_marker_reduce", st);
    k_marker_reduce<<<(unsigned)((PartLayout{nr}.size() + 63) / 64), 512, 0, st>>>(nr, G, part, red);
    Tm.End(st);
    if (nr <= RSBA_CHOL_MAXN) {
      const size_t lds_c = std::max((size_t)4 * 1024, CholeskyLdsDoubles(nr)) * sizeof(double);
      Tm.Begin("k_marker_reduced_solve", st);
      if (solve_lds) k_marker_reduced_solve_lds<<<1, 512, MarkerSolveLdsDoubles(nr) * sizeof(double), st>>>(nr, red, A, scale_r, col_full, params[x], params[c], delta_r, solve_out, ip);
      else k_marker_reduced_solve<<<1, 512, lds_c, st>>>(nr, red, A, scale_r, col_full, params[x], params[c], delta_r, solve_out, ip);
      Tm.End(st);
    } else {
      Tm.Begin("k_sys_build", st);
      k_sys_build<<<nr + 1, 256, 0, st>>>(red, RL, Wm, nullptr, nullptr, scale_r, ip, 1, ok_flag);
      Tm.End(st);
      if (tc_tiles > 0) {
        Tm.Begin("k_chol_tiles_persistent", st);
        k_chol_tiles_persistent<<<tc_tiles, 256, TileCholLdsDoubles() * sizeof(double), st>>>(
            nr, Wm, A, ok_flag, TileCholFlags{tc_flags, tc_flags + tc_np, tc_flags + (size_t)tc_np * (tc_nrt + 1), tc_nrt, tc_hand, tc_tag & 1, 0, tc_map}, tc_tag + 1, res, TileSysSource{}, TileGate{});
        ++tc_tag;
        Tm.End(st);
      } else {
        const size_t lds_s = CholStepLdsDoubles() * sizeof(double);
        Tm.Begin("k_chol_step(all panels)", st);
        for (int kb = 0; kb < nr; kb += RSBA_PB) {
          const int r0 = kb + std::min(RSBA_PB, nr - kb);
          const int nrt = (nr + 1 - r0 + RSBA_CT - 1) / RSBA_CT;
          k_chol_step<<<nrt * (nrt + 1) / 2, 256, lds_s, st>>>(nr, kb, Wm, A, ok_flag);
        }
        Tm.End(st);
      }
      const size_t lds_f = std::max((size_t)4 * 1024, (size_t)((nr + 63) & ~63) + 3 * RSBA_PB * RSBA_PLD + 64) * sizeof(double);
      Tm.Begin("k_marker_chol_finish", st);
      k_marker_chol_finish<<<1, 1024, lds_f, st>>>(nr, red, A, scale_r, col_full, params[x], params[c], delta_r, solve_out, ok_flag);
      Tm.End(st);
    }
    if (!chk("reduced solve")) return RSBA_ERR_HIP;
    Tm.Begin("k_time_backsub_terms", st);
    if (split_backsub) {
      k_pose_constants_reduced<<<(nr / 6 + 63) / 64, 64, 0, st>>>(nr / 6, col_full, params[c], posec_c);
      k_mc_time_step<<<(8 * T + 255) / 256, 256, 0, st>>>(T, slot_ptr, slot_col, time_full, sp, tdata, delta_r, params[x], params[c], delta_t, posec_c, bp_time);
      // the two halves of the candidate's evaluation side by side (one after the other when every kernel is timed)
      const bool fork2 = fork_s[0] != nullptr && !Tm.enabled();
      hipStream_t s_cost = fork2 ? fork_s[0] : st;
      if (fork2 && (hipEventRecord(fork_ev[0], st) != hipSuccess || hipStreamWaitEvent(s_cost, fork_ev[0], 0) != hipSuccess)) return RSBA_ERR_HIP;
      k_mc_candidate<0><<<ncand_wg, 256, 0, st>>>(N, T, ts, mo, obs8, intr, half_side, posec, posec_c, delta_r, delta_t, blk_time, bp_time);
      k_mc_candidate<1><<<ncand_wg, 256, 0, s_cost>>>(N, T, ts, mo, obs8, intr, half_side, posec, posec_c, delta_r, delta_t, blk_time, bp_time);
      if (fork2 && (hipEventRecord(fork_ev[1], s_cost) != hipSuccess || hipStreamWaitEvent(st, fork_ev[1], 0) != hipSuccess)) return RSBA_ERR_HIP;
    } else if (backsub_wg) {
      k_pose_constants_reduced<<<(nr / 6 + 63) / 64, 64, 0, st>>>(nr / 6, col_full, params[c], posec_c);
      const size_t lds_bw = (size_t)(dmax / 6 + 1) * (CC_STRIDE + 12 + 6) * sizeof(double);   // the shot's tables (k_time_backsub_wg)
      k_time_backsub_wg<2, 256><<<T, 256, lds_bw, st>>>(T, time_ptr, time_full, ts, mo, obs8, intr, half_side, posec, posec_c, tdata, delta_r, params[x],
                                                   params[c], delta_t, bp_time, slot_ptr, slot_col, col_full);
    } else {
      k_time_backsub_terms<<<nb_time, 256, 0, st>>>(T, time_ptr, time_full, ts, mo, obs8, intr, half_side, posec, tdata, delta_r, params[x],
                                                    params[c], delta_t, bp_time);
    }
    Tm.End(st);
    Tm.Begin("k_marker_schur_finish", st);
    k_marker_schur_finish<<<1, 1024, 0, st>>>(nb_time, bp_time, red + RL.scal(), solve_out, res);
    Tm.End(st);
    if (!chk("k_marker_schur_finish")) return RSBA_ERR_HIP;
    if (hipMemcpyAsync(res_host, res, RES_SIZE * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return RSBA_ERR_HIP;
    { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) { fprintf(stderr, "rsba: marker-chain step failed: %s\n", hipGetErrorString(e)); return RSBA_ERR_HIP; } }
    if (tc_tiles > 0 && res_host[RES_STALL] != 0.0) {
      // the tiles of the persistent factorisation were not all running side by side: the multi-launch path then
      fprintf(stderr, "rsba: persistent tiled Cholesky stalled; using the multi-launch factorisation\n");
      tc_tiles = 0;
      if (hipMemset(res, 0, RES_SIZE * sizeof(double)) != hipSuccess) return RSBA_ERR_HIP;
      return Step(st, o, radius, first, res_host, Tm);
    }
    return RSBA_OK;
  }
  int SumSquares(hipStream_t st, double* out) {
    if (Reset(st) != RSBA_OK) return RSBA_ERR_HIP;
    k_marker_eval<<<(N + 63) / 64, 64, 0, st>>>(N, mo, obs8, params[0], intr, half_side, 0, nullptr, nullptr, ss_x);
    std::vector<double> h(N);
    if (hipMemcpyAsync(h.data(), ss_x, N * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return RSBA_ERR_HIP;
    double s = 0; for (double v : h) s += v;
    *out = s;
    return RSBA_OK;
  }
  int Download(rsba_problem* p) {
    if (hipMemcpy(p->parameters.data(), params[cur], nfull * 8, hipMemcpyDeviceToHost) != hipSuccess) return RSBA_ERR_HIP;
    return RSBA_OK;
  }
};

}  // namespace rsba
