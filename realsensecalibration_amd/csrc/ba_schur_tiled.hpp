// Tiled Schur-complement accumulation (schur_impl = 1): the kernel that dominates an LM iteration.
//
//   S = U + D_c^2 - sum_j W_j (V_j + D_pj^2)^-1 W_j'          (schur_eliminator in Ceres' DENSE_SCHUR)
//
// A point with k views contributes k(k+1)/2 upper 6x6 blocks of 108 fp64 FMAs each: 4.5 GFLOP per
// iteration at 64 cameras x 100k points x 20 views — 40x the arithmetic of everything else in the
// iteration, against 104 MB of algorithmic HBM traffic.  The elimination is fp64-FMA bound, not HBM bound,
// and the output (1.18 MB) cannot live in one CU's LDS, so scattering per-point products with atomics
// (schur_impl = 0) runs at the atomic rate.  This kernel keeps every S block in REGISTERS instead:
//
//   * cameras are grouped by 16; a workgroup (256 threads) owns one 16 x 16 tile of camera pairs and one
//     thread owns one pair (a, b): its 6x6 block lives in 36 fp64 registers for the whole launch.
//   * which points contribute to pair (a, b) is static: camMask[cam] is a bitset over points.  A thread
//     walks the set bits of camMask[a] & camMask[b] (its own hit list, ~10% of the points), so lanes are
//     ~fully utilised instead of the 10% a lock-step sweep over points would give.
//   * per point only X, the damped inverse point block and V^-1 g_p (12 doubles) are staged in LDS, 512
//     points per chunk; the two observation-side factors of a hit are recomputed from the thread's own
//     camera constants (R, t, fx, fy in registers).  That doubles the FMA count per hit but keeps the LDS
//     traffic at 12 gathered doubles per hit and the chunk large enough that hit-count imbalance across the
//     lanes of a wave stays below ~25%.
//   * block(a,b) = E_a' (N_a Vinv N_b') E_b with E = [A K | Pj]; the per-camera 3x3 factor K (left Jacobian
//     of SO(3)) is pulled out of the sum over points and applied once per pair by the tile's finisher.
//   * every workgroup writes its 256 x 42 partial sums; they are added inside the same launch, in a fixed order
//     (bitwise reproducible): groups of RSBA_GRP (RSBA_GRP_SMALL up to 64 cameras) segments by their last arriver, tiles by
//     reducer workgroups — a pair tile's one per 3 x 3 quadrant of its blocks, each applying K and writing its part of S itself
//     (ReducerQuadrant); a self tile's by slices of the components, the last one finishing the tile (ReducerSegment).
//   * everything that is a sum over ONE camera's observations (diagonal blocks, damping diagonal, g_c, rhs
//     correction) is a "self" segment of the same launch (SelfSegment).
//   * the launch works through a list ordered by camera group ("stage") and publishes ready[1 + g] when group g's
//     columns of S are complete: the Cholesky may already be waiting for them (pipelined schedule, ba_solver.hip).
//
// The point-side pass (k_point_pass) that precedes it evaluates residuals and Jacobian blocks once per
// observation (point side only), forms the damped inverse point blocks and writes the 12 doubles per point the
// Schur kernel stages.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <vector>

#include "ba_point_kernels.hpp"

struct rsba_solver;

namespace rsba {

class KernelTimer;

#define RSBA_TG 16          // cameras per group
// Flags of a step in TiledSchur::ready (64 ints): [1 + g] stage g published (g < RSBA_MAX_STAGES), then the three below.
#define RSBA_MAX_STAGES 32        // camera groups a pipelined solve can gate on (512 cameras)
#define RSBA_READY_ALLDIAG 40     // = tag once every self tile is finished (first step of a run: every camera's diag U)
#define RSBA_READY_SOLVED 41      // = tag when the reduced system is solved (the back-substitution waits for it inside the kernel)
#define RSBA_READY_STARTED 42     // arrival counter of the factorisation's workgroups ("all resident")
#define RSBA_PRIO(p) __builtin_amdgcn_s_setprio(p)   // wavefront priority (see k_schur_tiles)
#ifndef RSBA_RESIDENT
#define RSBA_RESIDENT 0     // 1: up to 64 cameras too, as many workgroups as the chip holds draw tickets until the work list is through
#endif
// Priority of a hit loop.  RSBA_RESIDENT: by STAGE — a wavefront working on an earlier camera group's columns wins the arbitration against
// its SIMD-mate working on a later one (resident workgroups never change age, and the arbiter serves the older wavefront first: the
// younger workgroup of a CU would sit on its stage's entries until the older one has left the kernel)
#if RSBA_RESIDENT
#define RSBA_PRIO_HITS(stage) do { if ((stage) == 0) RSBA_PRIO(2); else if ((stage) == 1) RSBA_PRIO(1); else RSBA_PRIO(0); } while (0)
#else
#define RSBA_PRIO_HITS(stage) RSBA_PRIO(0)
#endif
#ifndef RSBA_CHUNK
#define RSBA_CHUNK 512      // points per LDS chunk (-DRSBA_CHUNK=256 builds and runs: measured only together with three workgroups per CU, HISTORY.md round 5)
#endif
#define RSBA_CW (RSBA_CHUNK / 64)
// A point's record in the chunk's LDS copy: RSBA_PT_STRIDE (12) doubles at a stride of RSBA_PT_LDS.  Every lane of a hit loop reads the record
// of ITS OWN hit — 64 random points per instruction, four 16-byte reads and an 8-byte one.  At a stride of 12 doubles (24 banks) the
// records start on 8 of the 64 banks; at 14 (-DRSBA_PT_LDS=14: 28 banks, still 16-byte aligned, 80 KB of LDS a workgroup) on 16 —
// measured (round 5, one box, twice each): 0.3492 - 0.3505 ms per step against 0.3479 - 0.3482 at 12.  The hit loop does not wait for
// its LDS reads long enough for the conflicts to show; the larger carve costs what they give.
#ifndef RSBA_PT_LDS
#define RSBA_PT_LDS 12
#endif
// (16-byte piece p = tid + 256 u of the chunk's records as memory holds them — six pieces a record — -> its place in LDS, in pieces: one
//  division per thread, then steps of 256 = 42 records + 4 pieces)
static_assert(RSBA_PT_LDS % 2 == 0 && RSBA_PT_LDS >= 12, "records are staged and read in 16-byte pieces");
struct PtLdsCursor {
  int h, a;
  __device__ __forceinline__ explicit PtLdsCursor(int tid) { const int r = tid / 6; h = tid - r * 6; a = r * (RSBA_PT_LDS / 2) + h; }
  __device__ __forceinline__ void Next() {
    h += 4; a += 42 * (RSBA_PT_LDS / 2) + 4;
    if (h >= 6) { h -= 6; a += RSBA_PT_LDS / 2 - 6; }
  }
};

// (RSBA_PT_STRIDE, the record of a point — X(3) Vinv(6) y(3) — is defined in ba_point_kernels.hpp, whose back-substitution writes it too)
#define RSBA_PART 42        // 36 block + 6 corr

// A segment is a range of 64-point mask words of one tile (not necessarily chunk-aligned: small problems get as many
// workgroups as they have words).
struct SchurSeg {
  int ga, gb, word_begin, word_end, self;
  // in-kernel reduction tree of the pair tiles: segment -> group of RSBA_GRP consecutive segments -> tile -> stage
  int tile, grp, grp_seg0, grp_nseg, tile_grp0, tile_ngrp, stage, stage_ntiles, nred, index, pad2;
  // index: the entry's own number (segs_ordered, the copy in launch order, is what the kernel reads: one load per ticket)
  // self: 0 pair segment, 1 self segment, 2 / 3 reducer of a pair / self tile (word_begin..word_end = its components)
};
// An entry of the work list, field by field into scalar registers.  (Copied as a struct, the entry went through vector
// registers into SCRATCH and were read back from there field by field — and a kernel that uses scratch at all pays for it in
// the launch latency of every workgroup: end of an entry -> start of the next on its slot 7.6 instead of 6.1 us.)
__device__ __forceinline__ SchurSeg LoadSeg(const SchurSeg* __restrict__ p) {
  SchurSeg s;
#define RSBA_F(f) s.f = __builtin_amdgcn_readfirstlane(p->f)
  RSBA_F(ga); RSBA_F(gb); RSBA_F(word_begin); RSBA_F(word_end); RSBA_F(self); RSBA_F(tile); RSBA_F(grp); RSBA_F(grp_seg0); RSBA_F(grp_nseg);
  RSBA_F(tile_grp0); RSBA_F(tile_ngrp); RSBA_F(stage); RSBA_F(stage_ntiles); RSBA_F(nred); RSBA_F(index); RSBA_F(pad2);
#undef RSBA_F
  return s;
}
#define RSBA_GRP 8          // segments per reduction group (more than 64 cameras)
#define RSBA_GRP_SMALL 4    // ... up to 64 cameras
#define RSBA_SELF_SETS 6      // reducers of a self tile: the sets of its 42 components that the K factors do not couple (ReducerSelfSet)
#define RSBA_DIRECT_GROUPS 4  // tiles with at most this many groups are finished by their last group, without reducers

struct TiledSchur {
  int C = 0, P = 0, ngroups = 0, nwords = 0, nchunks = 0, nseg = 0, nseg_pair = 0, grid_pp = 0;
  unsigned long long* cam_mask = nullptr;   // [ngroups*16][nwords]
  SchurSeg* segs = nullptr;                 // [nseg]
  int ntiles = 0;
  double* ptdata = nullptr;                 // [P][12]
  double* partial = nullptr;                // [nseg][42][256]
  double* grp_sum = nullptr;                // [ngrp][42][256] sums of RSBA_GRP consecutive segments
  int* tree_error = nullptr;
  int self_arrivals = 0;                    // arrivals the self tiles make at the first step's counter (SelfTileArrive): one per tile, or one per reducer
  int* error_flag = nullptr;                // where the reducers report a time-out when the solver has a host-mapped result block (else tree_error[0])
  int* grp_flag = nullptr;                  // [ngrp] launch number of the latest complete group sum (reducers)
  int epoch = 0;                            // launches so far
  int* sync_cnt = nullptr;                  // arrival counters, self-resetting: [ngrp] group members, [ntiles] groups done, [16] stage tiles, [ntiles] reducers done
  int nsync = 0, nblocks = 0;               // counters; blocks of the launch (segments + reducers)
  int* ready = nullptr;                     // ready[1 + g] = step tag once stage g (self tile g + pair tiles (g, g' >= g)) is in S
  int* block_seg = nullptr;                 // [nseg] launch order: block -> segment (host-side diagnostics)
  SchurSeg* segs_ordered = nullptr;         // [nblocks] the entries in launch order (what a ticket indexes)
  unsigned ticket_base = 0;                 // where the next launch's tickets start (SchurArgs::ticket_base)
  SchurSeg* segs_ordered_self = nullptr;    // [nblocks_self] the self segments and their reducers only (gradient evaluation)
  int* small_flag = nullptr;                // != 0: some camera takes the small-angle branch this linearisation (point side kernels)
  int nblocks_self = 0;
  int ngrp = 0;
  double* block_scal = nullptr;
  // robust-loss support: sqrt(rho') per observation in camera-major order
  int* cam_prefix = nullptr;                // [ngroups*16][nwords] set bits of cam_mask before each word
  int* cam_ptr = nullptr;                   // [ngroups*16+1] start of each camera's observation list
  int* cm_pos = nullptr;                    // [rows*64] observation (sliced layout, see ObsSliced) -> camera-major position
  double* sq_cm2[2] = {nullptr, nullptr};   // [N] each: sqrt(rho') at x / at the candidate (same double buffering as the points)
  double* lin2[2] = {nullptr, nullptr};     // [P][RSBA_LIN_STRIDE] each: V_j (6), g_pj (3), the point's share of sum rho, at x / at the candidate
  bool has_first_order = false;
  SchurSeg* segs_ordered_first = nullptr;   // the work list of a run's FIRST step (pipelined): every self tile ahead of the pair tiles
  bool lin_valid = false;                   // lin2[cur] holds the linearisation of the current x (set by a completed step of this run)
  // What ptdata / block_scal hold right now: the damped point blocks of state `pt_state` (index into the solver's double
  // buffers) for trust-region radius `pt_radius`, the per-block scalars in `scal_blocks` blocks.  Written by the point
  // pass, by the damping kernel — or by the back-substitution, which damps its candidate for the radius an accepted step
  // with a clamped radius update leads to (three times this step's); the next step then launches no point-side kernel.
  bool pt_valid = false;
  int pt_state = -1, scal_blocks = 0;
  double pt_radius = 0.0;
  double *u_cm = nullptr, *v_cm = nullptr;  // [N] observations in camera-major order (self tiles need the pixel)
  // more than 64 cameras: the pair segments' hit lists (PairSegmentSparse); nullptr below that
  unsigned *hits = nullptr, *hit_off = nullptr;
  int* hit_trips = nullptr;
  size_t hit_entries = 0;

  int Build(int C, int P, const std::vector<int>& pt_ptr, const std::vector<int>& obs_cam, const std::vector<double>& u, const std::vector<double>& v,
            const std::vector<int>& sliced_q /* sliced slot -> CSR position, -1 pads */, bool staged, bool bordered = false);
  // stages of the pipelined solve (TiledSchur::Build: stage_of): camera groups, or 2 B + 1 with the last group as a border
  int nstages = 0;
  int Launch(rsba_solver* s, const IterParams& ip, KernelTimer& T);
  void LaunchPointPass(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st);
  // the point pass of a step whose x already has its linearisation in lin2[cur] (every step but a run's first)
  void LaunchPointDamp(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st);
  void LaunchTiles(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st, int tag, bool first_staged = false, bool ahead = false, long long* ahead_trace = nullptr);
  // the self tiles only: cost, g_c, max |g_p| at x (what HandleSuccessfulStep evaluates at the new point); S is not formed
  void LaunchSelfOnly(rsba_solver* s, const IterParams& ip, KernelTimer& T, hipStream_t st);
  void Free();
};

// One word for the Schur kernel: does ANY camera take AngleAxisRotatePoint's small-angle branch at this linearisation?
// (Workgroup 0 of the point-side kernel, which runs right before the Schur kernel in every step.)
__device__ __forceinline__ void PublishSmallAngleFlag(int C, const double* __restrict__ camc_g, int* __restrict__ small_flag) {
  if (blockIdx.x != 0) return;
  int f = 0;
  for (int c = threadIdx.x; c < C; c += blockDim.x) f |= camc_g[(size_t)c * CC_STRIDE + CC_SMALL] != 0.0 ? 1 : 0;
  f = __syncthreads_or(f);
  if (threadIdx.x == 0) *small_flag = f;
}

// ------------------------------------------------------------------------------------------------
// K_A1: point pass, one thread per point.
// ------------------------------------------------------------------------------------------------
template <bool kStageCamc>
__global__ void __launch_bounds__(256)
k_point_pass(int C, int P, ObsSliced obs, const double* __restrict__ camc_g, const double* __restrict__ pts,
             double* __restrict__ scale_p, double* __restrict__ ptdata, double* __restrict__ block_scal,
             const int* __restrict__ cm_pos /* sliced like obs */, double* __restrict__ sq_cm, double* __restrict__ lin /* [P][RSBA_LIN_STRIDE] */,
             int* __restrict__ small_flag, IterParams ip) {
  extern __shared__ double lds[];
  double* camc_l = lds;      // C x 33 when staged
  const int tid = threadIdx.x;
  PublishSmallAngleFlag(C, camc_g, small_flag);
  if (kStageCamc) {
    // eight loads in flight per thread before the first LDS store (a plain copy loop waits for every load: 8 round trips
    // at the head of every LM step)
    for (int i0 = 0; i0 < C * CC_STRIDE; i0 += 8 * blockDim.x) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x + tid; v[u] = i < C * CC_STRIDE ? camc_g[i] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x + tid; if (i < C * CC_STRIDE) { const int c = i / CC_STRIDE, e = i - c * CC_STRIDE; camc_l[c * RSBA_CC_LDS + e] = v[u]; } }
    }
  }
  __syncthreads();
  const double* camc = kStageCamc ? camc_l : camc_g;
  const int ccs = kStageCamc ? RSBA_CC_LDS : CC_STRIDE;
  double cost = 0, xn = 0, fail = 0, gmax = 0;
  for (int j = blockIdx.x * blockDim.x + tid; j < P; j += gridDim.x * blockDim.x) {
    const int lane = j & 63;
    bool any = false;
    const double X[3] = {pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]};
    double V[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, cost_j = 0.0;
    // The grid is small (1.5 workgroups per CU at 100k points): a thread's ~20 observation records are a chain of
    // dependent round trips to memory unless several are in flight — four slots are loaded ahead of the one in use.
    const int t0 = obs.row_ptr[j >> 6], t1 = obs.row_ptr[(j >> 6) + 1];
    int camq[4]; double2 uvq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t qq = (size_t)(t0 + u) * 64 + lane;
      camq[u] = t0 + u < t1 ? obs.cam[qq] : -1;
      uvq[u] = t0 + u < t1 ? obs.uv[qq] : make_double2(0.0, 0.0);
    }
    for (int t = t0; t < t1; ++t) {
      const size_t q = (size_t)t * 64 + lane;
      const int cam = camq[0];
      const double2 uv = uvq[0];
#pragma unroll
      for (int u = 0; u < 3; ++u) { camq[u] = camq[u + 1]; uvq[u] = uvq[u + 1]; }
      {
        const size_t qq = (size_t)(t + 4) * 64 + lane;
        camq[3] = t + 4 < t1 ? obs.cam[qq] : -1;
        uvq[3] = t + 4 < t1 ? obs.uv[qq] : make_double2(0.0, 0.0);
      }
      if (cam < 0) continue;
      any = true;
      double r[2], jc[12], jp[6], sq;
      ResidualJacobian(camc + (size_t)cam * ccs, X, uv.x, uv.y, r, jc, jp);
      cost_j += LossAndScale(ip.huber_delta, r[0] * r[0] + r[1] * r[1], &sq);
      if (ip.huber_delta != 0.0) sq_cm[cm_pos[q]] = sq;
      if (sq != 1.0) {
        r[0] *= sq; r[1] *= sq;
#pragma unroll
        for (int i = 0; i < 6; ++i) jp[i] *= sq;
      }
      V[0] += jp[0] * jp[0] + jp[3] * jp[3]; V[1] += jp[0] * jp[1] + jp[3] * jp[4]; V[2] += jp[0] * jp[2] + jp[3] * jp[5];
      V[3] += jp[1] * jp[1] + jp[4] * jp[4]; V[4] += jp[1] * jp[2] + jp[4] * jp[5]; V[5] += jp[2] * jp[2] + jp[5] * jp[5];
      gp[0] += jp[0] * r[0] + jp[3] * r[1]; gp[1] += jp[1] * r[0] + jp[4] * r[1]; gp[2] += jp[2] * r[0] + jp[5] * r[1];
    }
    cost += cost_j;
    const bool konst = ip.pt_const != nullptr && ip.pt_const[j] != 0;   // a constant point block (k_fix_const_lin, ba_point_kernels.hpp)
    if (konst) { V[0] = RSBA_CONST_POINT_STIFFNESS; V[1] = 0.0; V[2] = 0.0; V[3] = RSBA_CONST_POINT_STIFFNESS; V[4] = 0.0; V[5] = RSBA_CONST_POINT_STIFFNESS; gp[0] = 0.0; gp[1] = 0.0; gp[2] = 0.0; }
    {
      // the linearisation of the point, kept: a rejected step damps it again with the smaller radius (k_point_damp)
      double* ln = lin + (size_t)j * RSBA_LIN_STRIDE;
#pragma unroll
      for (int i = 0; i < 6; ++i) ln[i] = V[i];
      ln[6] = gp[0]; ln[7] = gp[1]; ln[8] = gp[2]; ln[9] = cost_j;
    }
    double sp[3] = {1.0, 1.0, 1.0};
    if (ip.jacobi_scaling) {
      if (ip.first) { sp[0] = 1.0 / (1.0 + sqrt(V[0])); sp[1] = 1.0 / (1.0 + sqrt(V[3])); sp[2] = 1.0 / (1.0 + sqrt(V[5])); }
      else { sp[0] = scale_p[3 * (size_t)j]; sp[1] = scale_p[3 * (size_t)j + 1]; sp[2] = scale_p[3 * (size_t)j + 2]; }
    }
    if (ip.first) { scale_p[3 * (size_t)j] = sp[0]; scale_p[3 * (size_t)j + 1] = sp[1]; scale_p[3 * (size_t)j + 2] = sp[2]; }
    double Vi[6];
    const bool ok = PointBlockInverse(V, sp, ip.min_lm_diagonal, ip.max_lm_diagonal, ip.radius, Vi);
    if (!ok || konst) {
#pragma unroll
      for (int i = 0; i < 6; ++i) Vi[i] = 0.0;
      if (any && !konst) fail += 1.0;
    }
    double y[3];
    Sym3MulVec(Vi, gp, y);
    double* pd = ptdata + (size_t)j * RSBA_PT_STRIDE;
    pd[0] = X[0]; pd[1] = X[1]; pd[2] = X[2];
#pragma unroll
    for (int i = 0; i < 6; ++i) pd[3 + i] = Vi[i];
    pd[9] = y[0]; pd[10] = y[1]; pd[11] = y[2];
    if (!konst) xn += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
    gmax = fmax(gmax, fmax(fabs(gp[0]), fmax(fabs(gp[1]), fabs(gp[2]))));
  }
  __shared__ double s[4][256];
  s[0][tid] = cost; s[1][tid] = xn; s[2][tid] = fail; s[3][tid] = gmax;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s[0][tid] += s[0][tid + off]; s[1][tid] += s[1][tid + off]; s[2][tid] += s[2][tid + off]; s[3][tid] = fmax(s[3][tid], s[3][tid + off]); }
    __syncthreads();
  }
  if (tid < 4) block_scal[4 * blockIdx.x + tid] = s[tid][0];
}

// ------------------------------------------------------------------------------------------------
// K_A1': the point pass of every LM step but the first.  The linearisation of the points at x is already there:
// V_j, g_pj and the point's share of the cost were written either by k_point_pass (first step) or, for an accepted
// candidate, by the back-substitution kernel, which evaluates the candidate's residuals anyway and now takes the 2x3
// blocks along (k_backsub_candidate: the candidate's residuals ARE the next iteration's).  What is left per step is the
// radius-dependent part: damped inverse point block, V^-1 g_p, and the per-block scalars the Schur kernel folds.
// No observation record is read: 29 doubles in, 12 out per point.
// ------------------------------------------------------------------------------------------------
// `dec` != nullptr: the kernel was queued behind the PREVIOUS step's last kernel, before the host knew that step's outcome;
// state and radius are the device's decision (LmNext, ba_point_kernels.hpp): accepted -> the candidate's buffers (the
// `_c` pointers), rejected -> x's, radius = dec[2]; dec[0] == 0: nothing to do.
__global__ void __launch_bounds__(256)
k_point_damp(int P, const double* pts_x, const double* __restrict__ scale_p, const double* lin_x,
             double* __restrict__ ptdata, double* __restrict__ block_scal, int C, const double* camc_x,
             int* __restrict__ small_flag, IterParams ip, long long* trace = nullptr, const double* __restrict__ dec = nullptr,
             const double* pts_c = nullptr, const double* lin_c = nullptr, const double* camc_c = nullptr) {
  const int tid = threadIdx.x;
  if (trace != nullptr && blockIdx.x == 0 && tid == 0) trace[30] = wall_clock64();
  const double *pts = pts_x, *lin = lin_x, *camc_g = camc_x;
  if (dec != nullptr) {
    if (dec[0] == 0.0) return;
    if (dec[1] != 0.0) { pts = pts_c; lin = lin_c; camc_g = camc_c; }
    ip.radius = dec[2];
  }
  PublishSmallAngleFlag(C, camc_g, small_flag);
  double cost = 0, xn = 0, fail = 0, gmax = 0;
  for (int j = blockIdx.x * blockDim.x + tid; j < P; j += gridDim.x * blockDim.x) {
    const double* ln = lin + (size_t)j * RSBA_LIN_STRIDE;
    double V[6], gp[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) V[i] = ln[i];
    gp[0] = ln[6]; gp[1] = ln[7]; gp[2] = ln[8];
    cost += ln[9];
    const double X[3] = {pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]};
    double sp[3] = {1.0, 1.0, 1.0};
    if (ip.jacobi_scaling) { sp[0] = scale_p[3 * (size_t)j]; sp[1] = scale_p[3 * (size_t)j + 1]; sp[2] = scale_p[3 * (size_t)j + 2]; }
    const bool any = V[0] != 0.0 || V[3] != 0.0 || V[5] != 0.0;   // a point with observations has a non-zero block
    const bool konst = ip.pt_const != nullptr && ip.pt_const[j] != 0;   // a constant point block: no step, not in the norms (k_fix_const_lin)
    if (konst) { gp[0] = 0.0; gp[1] = 0.0; gp[2] = 0.0; }
    double Vi[6];
    const bool ok = PointBlockInverse(V, sp, ip.min_lm_diagonal, ip.max_lm_diagonal, ip.radius, Vi);
    if (!ok || konst) {
#pragma unroll
      for (int i = 0; i < 6; ++i) Vi[i] = 0.0;
      if (any && !konst) fail += 1.0;
    }
    double y[3];
    Sym3MulVec(Vi, gp, y);
    double* pd = ptdata + (size_t)j * RSBA_PT_STRIDE;
    pd[0] = X[0]; pd[1] = X[1]; pd[2] = X[2];
#pragma unroll
    for (int i = 0; i < 6; ++i) pd[3 + i] = Vi[i];
    pd[9] = y[0]; pd[10] = y[1]; pd[11] = y[2];
    if (!konst) xn += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
    gmax = fmax(gmax, fmax(fabs(gp[0]), fmax(fabs(gp[1]), fabs(gp[2]))));
  }
  __shared__ double s[4][256];
  s[0][tid] = cost; s[1][tid] = xn; s[2][tid] = fail; s[3][tid] = gmax;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s[0][tid] += s[0][tid + off]; s[1][tid] += s[1][tid + off]; s[2][tid] += s[2][tid + off]; s[3][tid] = fmax(s[3][tid], s[3][tid + off]); }
    __syncthreads();
  }
  if (tid < 4) block_scal[4 * blockIdx.x + tid] = s[tid][0];
  if (trace != nullptr && blockIdx.x == gridDim.x - 1 && tid == 0) trace[31] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// K_A2: pair kernel.  One workgroup per (tile, segment of chunks); thread (ia, ib) owns camera pair
// (16 ga + ia, 16 gb + ib).  LDS: point data of the current chunk + the 32 visibility bit-rows.
// ------------------------------------------------------------------------------------------------
// acc + a b + c d as two dependent FMAs.  (`acc += a * b + c * d` is three instructions — the sum of products is rounded
// on its own before it is added — and the elimination kernel is bound by the number of fp64 instructions it issues.)
__device__ __forceinline__ double Fma2(double a, double b, double c, double d, double acc) { return fma(c, d, fma(a, b, acc)); }

// (1 / x for the depth of a point in a camera: RcpNewton, ba_math.hpp)

struct SideConst {
  double R[9], t[3], fx, fy;
  bool small;
};

__device__ __forceinline__ void LoadSide(const double* __restrict__ cc, SideConst& s) {
#pragma unroll
  for (int i = 0; i < 9; ++i) s.R[i] = cc[CC_R + i];
  s.t[0] = cc[CC_T]; s.t[1] = cc[CC_T + 1]; s.t[2] = cc[CC_T + 2];
  s.fx = cc[CC_FX]; s.fy = cc[CC_FY];
  s.small = cc[CC_SMALL] != 0.0;
}

// Reduced Jacobian rows of one observation: e0/e1 = rows of [A | Pj] (6 each, entries 4 / 3 are zero),
// n0/n1 = rows of Pj R (2x3).  The camera's K factor is NOT applied here.
__device__ __forceinline__ void SideRows(const SideConst& s, const double X[3], double sq, double e0[6], double e1[6], double n0[3], double n1[3]) {
  const double q0 = s.R[0] * X[0] + s.R[1] * X[1] + s.R[2] * X[2];
  const double q1 = s.R[3] * X[0] + s.R[4] * X[1] + s.R[5] * X[2];
  const double q2 = s.R[6] * X[0] + s.R[7] * X[1] + s.R[8] * X[2];
  const double p0 = q0 + s.t[0], p1 = q1 + s.t[1], p2 = q2 + s.t[2];
  const double iz = RcpNewton(p2);
  const double al = s.fx * iz * sq, be = s.fy * iz * sq;   // sqrt(rho') scales every Jacobian entry of the observation
  const double ga = -al * p0 * iz, de = -be * p1 * iz;
  const double w0 = s.small ? X[0] : q0, w1 = s.small ? X[1] : q1, w2 = s.small ? X[2] : q2;
  e0[0] = w1 * ga; e0[1] = w2 * al - w0 * ga; e0[2] = -w1 * al; e0[3] = al; e0[4] = 0.0; e0[5] = ga;
  e1[0] = w1 * de - w2 * be; e1[1] = -w0 * de; e1[2] = w0 * be; e1[3] = 0.0; e1[4] = be; e1[5] = de;
  n0[0] = al * s.R[0] + ga * s.R[6]; n0[1] = al * s.R[1] + ga * s.R[7]; n0[2] = al * s.R[2] + ga * s.R[8];
  n1[0] = be * s.R[3] + de * s.R[6]; n1[1] = be * s.R[4] + de * s.R[7]; n1[2] = be * s.R[5] + de * s.R[8];
}

// Camera constants the pair kernel keeps in LDS: R(9) t(3) fx fy small + pad = 16 doubles per camera.
#define RSBA_SC_STRIDE 16

struct SideLds {
  const double* c;  // LDS pointer to this side's 16 doubles
  __device__ __forceinline__ double R(int i) const { return c[i]; }
};

// Reduced Jacobian rows from LDS-resident camera constants (same arithmetic as SideRows).
template <bool kSmall>
__device__ __forceinline__ void SideRowsLds(const double* __restrict__ c, const double X[3], double sq, double e0[6], double e1[6],
                                            double n0[3], double n1[3]) {
  const double r0 = c[0], r1 = c[1], r2 = c[2], r3 = c[3], r4 = c[4], r5 = c[5], r6 = c[6], r7 = c[7], r8 = c[8];
  const double q0 = r0 * X[0] + r1 * X[1] + r2 * X[2];
  const double q1 = r3 * X[0] + r4 * X[1] + r5 * X[2];
  const double q2 = r6 * X[0] + r7 * X[1] + r8 * X[2];
  const double p0 = q0 + c[9], p1 = q1 + c[10], p2 = q2 + c[11];
  const double iz = RcpNewton(p2);
  const double al = c[12] * iz * sq, be = c[13] * iz * sq;
  const double ga = -al * p0 * iz, de = -be * p1 * iz;
  const bool small = kSmall && c[14] != 0.0;   // kSmall false: no camera of the tile takes the small-angle branch
  const double w0 = small ? X[0] : q0, w1 = small ? X[1] : q1, w2 = small ? X[2] : q2;
  e0[0] = w1 * ga; e0[1] = w2 * al - w0 * ga; e0[2] = -w1 * al; e0[3] = al; e0[4] = 0.0; e0[5] = ga;
  e1[0] = w1 * de - w2 * be; e1[1] = -w0 * de; e1[2] = w0 * be; e1[3] = 0.0; e1[4] = be; e1[5] = de;
  n0[0] = al * r0 + ga * r6; n0[1] = al * r1 + ga * r7; n0[2] = al * r2 + ga * r8;
  n1[0] = be * r3 + de * r6; n1[1] = be * r4 + de * r7; n1[2] = be * r5 + de * r8;
}

// The same rows with the projective scale factored out, for the pair tiles:  E = D E~,  N = D N~,  D = diag(al, be), so
//   E_a' (N_a Vinv N_b') E_b = E~_a' [ D_a^2 (N~_a Vinv N~_b') D_b^2 ] E~_b
// and the rows lose their products with al / be:  with u = p0 / p2, v = p1 / p2
//   e~0 = [-w1 u, w2 + w0 u, -w1, 1, 0, -u]     e~1 = [-(w1 v + w2), w0 v, w0, 0, 1, -v]     n~0 = R0 - u R2,  n~1 = R1 - v R2
// (33 instructions per side instead of 43; the 1s save four more in Z).  Returned: entries 0, 1, 2, 5 of e~0 / e~1 in
// e0[4] / e1[4], the rows n~, and d2 = (al^2, be^2).
template <bool kSmall>
__device__ __forceinline__ void SideRowsUnscaledLds(const double* __restrict__ c, const double X[3], double sq, double e0[4], double e1[4],
                                                    double n0[3], double n1[3], double d2[2]) {
  // (every multiply-add written out: left to the compiler, a sum of three products is contracted as mul + fma + fma starting from
  //  WHICHEVER product it likes — it differed between the two halves of one unrolled loop — and the pair segments' three
  //  implementations, which must add the same bits, would each round their own way)
  const double r0 = c[0], r1 = c[1], r2 = c[2], r3 = c[3], r4 = c[4], r5 = c[5], r6 = c[6], r7 = c[7], r8 = c[8];
  const double q0 = fma(r2, X[2], fma(r1, X[1], r0 * X[0]));
  const double q1 = fma(r5, X[2], fma(r4, X[1], r3 * X[0]));
  const double q2 = fma(r8, X[2], fma(r7, X[1], r6 * X[0]));
  const double p0 = q0 + c[9], p1 = q1 + c[10], p2 = q2 + c[11];
  const double iz = RcpNewton(p2);
  const double al = c[12] * iz * sq, be = c[13] * iz * sq;
  const double u = p0 * iz, v = p1 * iz;
  const bool small = kSmall && c[14] != 0.0;
  const double w0 = small ? X[0] : q0, w1 = small ? X[1] : q1, w2 = small ? X[2] : q2;
  e0[0] = -w1 * u; e0[1] = fma(w0, u, w2); e0[2] = -w1; e0[3] = -u;
  e1[0] = -fma(w1, v, w2); e1[1] = w0 * v; e1[2] = w0; e1[3] = -v;
  n0[0] = fma(-u, r6, r0); n0[1] = fma(-u, r7, r1); n0[2] = fma(-u, r8, r2);
  n1[0] = fma(-v, r6, r3); n1[1] = fma(-v, r7, r4); n1[2] = fma(-v, r8, r5);
  d2[0] = al * al; d2[1] = be * be;
}

// The 120 pairs ia < ib of a diagonal tile, packed: lane t of waves 0/1 owns pair (kDiagPair[t] >> 4, kDiagPair[t] & 15); the
// lanes of waves 2/3 of a diagonal-tile workgroup have empty hit lists (they only help staging), so a diagonal tile costs two
// waves of arithmetic, not four.
__device__ __constant__ unsigned char kDiagPair[128] = {
#define RSBA_P(a, b) (unsigned char)((a) * 16 + (b))
    RSBA_P(0,1),RSBA_P(0,2),RSBA_P(0,3),RSBA_P(0,4),RSBA_P(0,5),RSBA_P(0,6),RSBA_P(0,7),RSBA_P(0,8),RSBA_P(0,9),RSBA_P(0,10),RSBA_P(0,11),RSBA_P(0,12),RSBA_P(0,13),RSBA_P(0,14),RSBA_P(0,15),
    RSBA_P(1,2),RSBA_P(1,3),RSBA_P(1,4),RSBA_P(1,5),RSBA_P(1,6),RSBA_P(1,7),RSBA_P(1,8),RSBA_P(1,9),RSBA_P(1,10),RSBA_P(1,11),RSBA_P(1,12),RSBA_P(1,13),RSBA_P(1,14),RSBA_P(1,15),
    RSBA_P(2,3),RSBA_P(2,4),RSBA_P(2,5),RSBA_P(2,6),RSBA_P(2,7),RSBA_P(2,8),RSBA_P(2,9),RSBA_P(2,10),RSBA_P(2,11),RSBA_P(2,12),RSBA_P(2,13),RSBA_P(2,14),RSBA_P(2,15),
    RSBA_P(3,4),RSBA_P(3,5),RSBA_P(3,6),RSBA_P(3,7),RSBA_P(3,8),RSBA_P(3,9),RSBA_P(3,10),RSBA_P(3,11),RSBA_P(3,12),RSBA_P(3,13),RSBA_P(3,14),RSBA_P(3,15),
    RSBA_P(4,5),RSBA_P(4,6),RSBA_P(4,7),RSBA_P(4,8),RSBA_P(4,9),RSBA_P(4,10),RSBA_P(4,11),RSBA_P(4,12),RSBA_P(4,13),RSBA_P(4,14),RSBA_P(4,15),
    RSBA_P(5,6),RSBA_P(5,7),RSBA_P(5,8),RSBA_P(5,9),RSBA_P(5,10),RSBA_P(5,11),RSBA_P(5,12),RSBA_P(5,13),RSBA_P(5,14),RSBA_P(5,15),
    RSBA_P(6,7),RSBA_P(6,8),RSBA_P(6,9),RSBA_P(6,10),RSBA_P(6,11),RSBA_P(6,12),RSBA_P(6,13),RSBA_P(6,14),RSBA_P(6,15),
    RSBA_P(7,8),RSBA_P(7,9),RSBA_P(7,10),RSBA_P(7,11),RSBA_P(7,12),RSBA_P(7,13),RSBA_P(7,14),RSBA_P(7,15),
    RSBA_P(8,9),RSBA_P(8,10),RSBA_P(8,11),RSBA_P(8,12),RSBA_P(8,13),RSBA_P(8,14),RSBA_P(8,15),
    RSBA_P(9,10),RSBA_P(9,11),RSBA_P(9,12),RSBA_P(9,13),RSBA_P(9,14),RSBA_P(9,15),
    RSBA_P(10,11),RSBA_P(10,12),RSBA_P(10,13),RSBA_P(10,14),RSBA_P(10,15),
    RSBA_P(11,12),RSBA_P(11,13),RSBA_P(11,14),RSBA_P(11,15),
    RSBA_P(12,13),RSBA_P(12,14),RSBA_P(12,15),
    RSBA_P(13,14),RSBA_P(13,15),
    RSBA_P(14,15),
    0, 0, 0, 0, 0, 0, 0, 0
#undef RSBA_P
};

// v[i] = sum over q < n of block q's entry i (this thread's slot), blocks RSBA_PART*256 doubles apart, in block order.
// Plain loads after the caller's acquire (see TileTreeReduce); two blocks are in flight at a time when the registers
// allow it (the sums sit at the tail of a tile, where only memory-level parallelism shortens them).
// The thread's index as something the compiler cannot see through: the entries' functions run in a loop now (resident
// workgroups), and with a plain threadIdx.x every address they derive from it is computed once in front of the loop and kept —
// 50 to 60 registers more than there are, spilled into the hot loops.
__device__ __forceinline__ int OpaqueTid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  __builtin_assume(t >= 0 && t < 256);
  return t;
}

template <int NV>
__device__ __forceinline__ void TreeSum(const double* __restrict__ in, int n, double* v) {
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  int q = 0;
  if (NV <= 36) {
    for (; q + 1 < n; q += 2) {
      const double* p0 = in + (size_t)q * RSBA_PART * 256;
      const double* p1 = p0 + (size_t)RSBA_PART * 256;
      double x0[NV], x1[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) x0[i] = p0[i * 256];
#pragma unroll
      for (int i = 0; i < NV; ++i) x1[i] = p1[i * 256];
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = (v[i] + x0[i]) + x1[i];
    }
  }
  for (; q < n; ++q) {
    const double* p0 = in + (size_t)q * RSBA_PART * 256;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += p0[i * 256];
  }
}

// Reduction of a tile without further launches, in two levels.
//  1. Every compute workgroup writes its partial block; the last workgroup of a group of RSBA_GRP consecutive segments
//     to arrive adds the group's blocks in segment order (the sums do not depend on who arrives last) into grp_sum and
//     counts the group as done.
//  2. One CU pulls ~30 GB/s out of memory, so a single workgroup adding a tile's 30-40 group sums (2-3 MB) was measured
//     at 65-105 us, at the tail of every stage.  Instead each tile has reducer workgroups (four per pair tile, six per self
//     tile), placed in block order right behind the stage's compute blocks (so everything they wait for has been dispatched
//     before them: they can spin without deadlock).  Each adds, over the tile's groups as they arrive, a set of components the
//     K factors do not couple with the others — a 3 x 3 quadrant of the pairs' blocks, a part of a camera's own sums — applies K
//     and writes its part of S (ReducerQuadrant, ReducerSelfSet).
// The eight XCDs' L2s are not coherent with each other inside a kernel, and an agent-scope fence costs a write-back /
// invalidate of a whole L2 (measured: 2x on this kernel when every workgroup fenced).  So partial sums are written with
// agent-scope (sc1) stores that go through to memory, ordered by waiting for the stores' acknowledgements before a
// counter is bumped; only the workgroups that read other workgroups' sums invalidate their L2 first, and only the tile
// finishers, whose results are written with ordinary stores, write theirs back.
// Where a compute workgroup's 256 x 42 sums go: its partial block — or, a group of ONE segment in a tile with reducers, straight
// into the group's sum (GroupReduce then only waits for the stores and raises the group's flag: no arrival counter, no second
// fetch, no second store — three dependent trips to memory; the host puts such groups at the end of every tile, where the
// stage's flag waits for them).
__device__ __forceinline__ bool SingleSegmentGroup(const SchurSeg& sg) { return sg.grp_nseg == 1 && sg.nred != 0; }
__device__ __forceinline__ double* SegmentOut(const SchurSeg& sg, double* __restrict__ partial, double* __restrict__ grp_sum, int seg_index) {
  return SingleSegmentGroup(sg) ? grp_sum + (size_t)sg.grp * RSBA_PART * 256 : partial + (size_t)seg_index * RSBA_PART * 256;
}

template <int NV>
__device__ __forceinline__ bool GroupReduce(const SchurSeg& sg, const double* __restrict__ partial, double* __restrict__ grp_sum,
                                            int* __restrict__ sync_cnt, int ngrp, double* v, int* __restrict__ grp_flag, int epoch) {
  __shared__ int s_last;
  const int tid = OpaqueTid();
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (SingleSegmentGroup(sg)) {   // the sums are the group's (SegmentOut) and acknowledged
    if (tid == 0) __hip_atomic_store(&grp_flag[sg.grp], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  if (tid == 0) s_last = __hip_atomic_fetch_add(&sync_cnt[sg.grp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sg.grp_nseg - 1;
  __syncthreads();
  if (!s_last) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  TreeSum<NV>(partial + (size_t)sg.grp_seg0 * RSBA_PART * 256 + tid, sg.grp_nseg, v);
  double* gs = grp_sum + (size_t)sg.grp * RSBA_PART * 256 + tid;
#pragma unroll
  for (int i = 0; i < NV; ++i) __hip_atomic_store(&gs[i * 256], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) __hip_atomic_store(&sync_cnt[sg.grp], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(&grp_flag[sg.grp], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the group's sum is acknowledged (s_waitcnt above)
    // (a tile with reducers does not count its groups: the reducers follow the flags)
    s_last = sg.nred == 0 && __hip_atomic_fetch_add(&sync_cnt[ngrp + sg.tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sg.tile_ngrp - 1;
  }
  __syncthreads();
  // tiles of a few groups only (many cameras: 136 pair tiles at 256 cameras) have no reducers: the workgroup that
  // completes the last group adds the group sums itself
  if (sg.nred != 0 || !s_last) return false;
  if (tid == 0) __hip_atomic_store(&sync_cnt[ngrp + sg.tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  TreeSum<NV>(grp_sum + (size_t)sg.tile_grp0 * RSBA_PART * 256 + tid, sg.tile_ngrp, v);
  return true;
}

// The tile's results are written: one arrival at the stage counter; true (whole workgroup) for the last tile of the stage.
__device__ __forceinline__ bool StageArrive(const SchurSeg& sg, int* __restrict__ sync_cnt, int ngrp, int ntiles, bool publish) {
  __shared__ int s_last;
  // the tile's blocks of S must be visible to the waiting Cholesky (another XCD) before the stage is published; with
  // the sequential schedule the kernel boundary does that, and 152 L2 write-backs (256 cameras) are not free
  if (publish) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    s_last = __hip_atomic_fetch_add(&sync_cnt[ngrp + ntiles + sg.stage], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sg.stage_ntiles - 1;
    if (s_last) __hip_atomic_store(&sync_cnt[ngrp + ntiles + sg.stage], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_last != 0;
}

// A stage's flag, ONE store per stage and launch (the last arrival of the stage): a RELEASE at agent scope.  What the stage
// publishes left its workgroups as agent-scope (write-through) stores whose acknowledgements each of them awaited before it
// arrived at the stage counter, so on this hardware the data is in memory before the flag either way — round 3 made the
// reducers' flag stores relaxed on that ground (no measurable difference: 0.420 against 0.417 ms).  But then the hand-over is
// correct by the property of every store on the path, not by the memory model: one plain store among them (FinishLinearize's
// are, covered by StageArrive's fence) would leave a stale line in another XCD's L2 silently.  The release costs one L2
// write-back per stage, off every per-tile path; the consumers acquire behind the flag (WaitFlagWG / WaitFlagPlainWG).
__device__ __forceinline__ void PublishStage(int* flag, int tag) { __hip_atomic_store(flag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// A self tile is finished (its diag U, g_c, diagonal block are written and — StageArrive's fence, pipelined schedule —
// visible): the last of the ngroups self tiles publishes ready[9].  Thread 0 only; call behind StageArrive.
__device__ __forceinline__ void SelfTileArrive(int* __restrict__ sync_cnt, int ngrp, int ntiles, int ngroups, int* __restrict__ ready, int tag) {
  int* cnt = sync_cnt + ngrp + ntiles + RSBA_MAX_STAGES;
  if (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) {
    __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ready[RSBA_READY_ALLDIAG], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Self tile: lane (ia, s) holds slice s of camera 16 ga + ia's 42 sums in v[]; called by the whole workgroup (shuffles).
// Adds the 16 slices in a fixed tree order, then K factors, diagonal S block, diag(U), g_c and the rhs correction.
__device__ __forceinline__ void FinishSelfSlot(int C, int ga, double* v, const double* __restrict__ camc, double* __restrict__ red, RedLayout L,
                                               const double* __restrict__ cam_free) {
  const int tid = OpaqueTid(), ia = tid >> 4, ib = tid & 15, cam_a = RSBA_TG * ga + ia;
  // sum the 16 slices of a row (lanes ia*16 .. ia*16+15 are contiguous inside a wave) in a fixed tree order
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
    for (int i = 0; i < RSBA_PART; ++i) v[i] += __shfl_down(v[i], off, 16);
  }
  if (ib != 0 || cam_a >= C) return;
  if (cam_free != nullptr && cam_free[cam_a] == 0.0) {
    // constant camera: no columns in the reduced program -> zero block, gradient and damping diagonal; the LM floor
    // (min_lm_diagonal / radius) keeps the factorisation positive and its step is exactly zero
#pragma unroll
    for (int i = 0; i < RSBA_PART; ++i) v[i] = 0.0;
  }
  const double* K = camc + (size_t)cam_a * CC_STRIDE + CC_K;
  // symmetric core -> full 6x6, then T' core T with T = blkdiag(K, I)
  double cfull[36];
  { int t = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
#pragma unroll
      for (int q = p; q < 6; ++q) { cfull[6 * p + q] = v[t]; cfull[6 * q + p] = v[t]; ++t; } } }
  double tmp[36], blk[36];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int p = 0; p < 3; ++p) tmp[6 * p + q] = K[0 * 3 + p] * cfull[0 * 6 + q] + K[1 * 3 + p] * cfull[1 * 6 + q] + K[2 * 3 + p] * cfull[2 * 6 + q];
#pragma unroll
    for (int p = 3; p < 6; ++p) tmp[6 * p + q] = cfull[6 * p + q];
  }
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int q = 0; q < 3; ++q) blk[6 * p + q] = tmp[6 * p + 0] * K[0 * 3 + q] + tmp[6 * p + 1] * K[1 * 3 + q] + tmp[6 * p + 2] * K[2 * 3 + q];
#pragma unroll
    for (int q = 3; q < 6; ++q) blk[6 * p + q] = tmp[6 * p + q];
  }
  double* Sd = red + L.S() + (size_t)(6 * cam_a) * L.nc + 6 * cam_a;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int q = p; q < 6; ++q) { const double x = 0.5 * (blk[6 * p + q] + blk[6 * q + p]); Sd[(size_t)p * L.nc + q] = x; Sd[(size_t)q * L.nc + p] = x; }
  }
  // diag(U): rows 0..2 through K, rows 3..5 as they are
  const double u00 = v[21], u01 = v[22], u02 = v[23], u11 = v[24], u12 = v[25], u22 = v[26];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const double k0 = K[0 * 3 + p], k1 = K[1 * 3 + p], k2 = K[2 * 3 + p];
    red[L.diagU() + 6 * cam_a + p] = k0 * (u00 * k0 + u01 * k1 + u02 * k2) + k1 * (u01 * k0 + u11 * k1 + u12 * k2) + k2 * (u02 * k0 + u12 * k1 + u22 * k2);
  }
#pragma unroll
  for (int p = 3; p < 6; ++p) red[L.diagU() + 6 * cam_a + p] = v[27 + p - 3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    red[L.gc() + 6 * cam_a + p] = K[0 * 3 + p] * v[30] + K[1 * 3 + p] * v[31] + K[2 * 3 + p] * v[32];
    red[L.corr() + 6 * cam_a + p] = -(K[0 * 3 + p] * v[36] + K[1 * 3 + p] * v[37] + K[2 * 3 + p] * v[38]);
  }
#pragma unroll
  for (int p = 3; p < 6; ++p) { red[L.gc() + 6 * cam_a + p] = v[30 + p]; red[L.corr() + 6 * cam_a + p] = -v[36 + p]; }
}

// K factors and the S blocks of one camera pair: slot (ia, ib) of tile (ga, gb) with its 36 reduced sums in `core`.
__device__ __forceinline__ void FinishPairSlot(int C, int ga, int gb, int slot, const double* core, const double* __restrict__ camc,
                                               double* __restrict__ red, RedLayout L, const double* __restrict__ cam_free) {
  const int ia = slot >> 4, ib = slot & 15;
  const int cam_a = RSBA_TG * ga + ia, cam_b = RSBA_TG * gb + ib;
  if (cam_a >= C || cam_b >= C || (ga == gb && ia >= ib)) return;
  const double* Ka = camc + (size_t)cam_a * CC_STRIDE + CC_K;
  const double* Kb = camc + (size_t)cam_b * CC_STRIDE + CC_K;
  // rows: Ta' core  (first three rows mixed by Ka')
  double tmp[36];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int p = 0; p < 3; ++p) tmp[6 * p + q] = Ka[0 * 3 + p] * core[0 * 6 + q] + Ka[1 * 3 + p] * core[1 * 6 + q] + Ka[2 * 3 + p] * core[2 * 6 + q];
#pragma unroll
    for (int p = 3; p < 6; ++p) tmp[6 * p + q] = core[6 * p + q];
  }
  // columns: (.) Tb
  double blk[36];
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int q = 0; q < 3; ++q) blk[6 * p + q] = tmp[6 * p + 0] * Kb[0 * 3 + q] + tmp[6 * p + 1] * Kb[1 * 3 + q] + tmp[6 * p + 2] * Kb[2 * 3 + q];
#pragma unroll
    for (int q = 3; q < 6; ++q) blk[6 * p + q] = tmp[6 * p + q];
  }
  if (cam_free != nullptr && cam_free[cam_a] * cam_free[cam_b] == 0.0) {
#pragma unroll
    for (int i = 0; i < 36; ++i) blk[i] = 0.0;   // a constant camera couples to nobody
  }
  // S is written full symmetric: block (a,b) = -blk, block (b,a) = -blk' (nobody else touches off-diagonal blocks)
  double* Sb = red + L.S() + (size_t)(6 * cam_a) * L.nc + 6 * cam_b;
  double* St = red + L.S() + (size_t)(6 * cam_b) * L.nc + 6 * cam_a;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int q = 0; q < 6; ++q) { Sb[(size_t)p * L.nc + q] = -blk[6 * p + q]; St[(size_t)q * L.nc + p] = -blk[6 * p + q]; }
  }
}

// One hit of a camera pair: acc += E~a' [ Da^2 (N~a Vinv N~b') Db^2 ] E~b for the point X with damped inverse block v0..v5 (the
// arithmetic of PairSegment's loop body, operation for operation: the masked, the sparse and the listed pair segments add the
// same bits).  ca / cb: the two cameras' 15 constants (R, t, fx, fy, small-angle flag).
template <bool kSmall>
__device__ __forceinline__ void PairHit(const double* __restrict__ ca, const double* __restrict__ cb, const double X[3], double v0, double v1, double v2,
                                        double v3, double v4, double v5, double sqa, double sqb, double* acc) {
  double ea0[4], ea1[4], na0[3], na1[3], da[2];
  SideRowsUnscaledLds<kSmall>(ca, X, sqa, ea0, ea1, na0, na1, da);
  // t = N~a Vinv (2x3), then the b side, M = D_a^2 (t N~b') D_b^2 (2x2), Z = E~a' M (6x2)
  auto dot3 = [](double a0, double b0, double a1, double b1, double a2, double b2) { return fma(a2, b2, fma(a1, b1, a0 * b0)); };
  const double t00 = dot3(na0[0], v0, na0[1], v1, na0[2], v2), t01 = dot3(na0[0], v1, na0[1], v3, na0[2], v4), t02 = dot3(na0[0], v2, na0[1], v4, na0[2], v5);
  const double t10 = dot3(na1[0], v0, na1[1], v1, na1[2], v2), t11 = dot3(na1[0], v1, na1[1], v3, na1[2], v4), t12 = dot3(na1[0], v2, na1[1], v4, na1[2], v5);
  double eb0[4], eb1[4], nb0[3], nb1[3], db[2];
  SideRowsUnscaledLds<kSmall>(cb, X, sqb, eb0, eb1, nb0, nb1, db);
  const double m00 = dot3(t00, nb0[0], t01, nb0[1], t02, nb0[2]) * (da[0] * db[0]), m01 = dot3(t00, nb1[0], t01, nb1[1], t02, nb1[2]) * (da[0] * db[1]);
  const double m10 = dot3(t10, nb0[0], t11, nb0[1], t12, nb0[2]) * (da[1] * db[0]), m11 = dot3(t10, nb1[0], t11, nb1[1], t12, nb1[2]) * (da[1] * db[1]);
  // Z = E~a' M, acc += Z E~b.  Columns 3 / 4 of E~ are (1, 0)' and (0, 1)': rows 3 / 4 of Z are the rows of M, columns
  // 3 / 4 of the update are Z itself.  Rows / columns 0, 1, 2, 5 take entry 0, 1, 2, 3 of the packed e~.
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const int pp = p == 5 ? 3 : p;
    double z0, z1;
    if (p == 3) { z0 = m00; z1 = m01; }
    else if (p == 4) { z0 = m10; z1 = m11; }
    else { z0 = fma(ea1[pp], m10, ea0[pp] * m00); z1 = fma(ea1[pp], m11, ea0[pp] * m01); }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int qq = q == 5 ? 3 : q;
      if (q == 3) acc[6 * p + q] += z0;
      else if (q == 4) acc[6 * p + q] += z1;
      else acc[6 * p + q] = Fma2(z0, eb0[qq], z1, eb1[qq], acc[6 * p + q]);
    }
  }
}

// Pair tiles (ga <= gb, a != b).  Two workgroups per CU: the accumulators (72 VGPRs) are the only long-lived
// per-lane state; camera constants sit in LDS (broadcast within a 16-lane row / replicated across rows).
struct SchurArgs {
  int C, P, nwords;
  const double* __restrict__ camc;
  const double* __restrict__ cam_free;   // per camera 1.0 / 0.0 (constant), nullptr: all free
  const SchurSeg* __restrict__ segs;
  const unsigned long long* __restrict__ cam_mask;
  const double* __restrict__ ptdata;
  const int* __restrict__ cam_prefix;
  const int* __restrict__ cam_ptr;
  const double* __restrict__ sq_cm;
  const double* __restrict__ u_cm;
  const double* __restrict__ v_cm;
  double* __restrict__ partial;
  double* __restrict__ grp_sum;
  int* __restrict__ sync_cnt;
  int ngrp, ntiles, last_group;
  int self_arrivals;               // how many arrivals complete the self tiles (all_self)
  int all_self;                    // 1: the self tiles' finishers count themselves and the last one publishes ready[9] (first step: every
                                   // camera's diag U is known -> the factorisation's Jacobi scale, see TiledSchur::segs_ordered_first)
  const SchurSeg* __restrict__ segs_ordered;   // the work list in launch order
  const int* __restrict__ small_flag;          // see TiledSchur::small_flag
  int* __restrict__ ready;
  int tag;            // 0: nobody is waiting (sequential schedule)
  int self_only;      // 1: the work list holds the self tiles only (gradient evaluation): no stage bookkeeping
  double* __restrict__ red;
  RedLayout L;
  int nblocks_pp;
  const double* __restrict__ block_scal;
  double* __restrict__ gmax_p;
  int* tree_error;                 // set when a reducer gave up waiting (cannot happen; never hang)
  int* ticket;                     // next entry of the work list (block_seg): counts on from launch to launch, ticket_base is where this one starts
  unsigned ticket_base;
  int total;                       // entries of this launch's work list (the workgroups loop over tickets until they draw one past it)
  int* grp_flag;                   // [ngrp] = epoch once the group's sum is in grp_sum (the reducers add groups as they arrive)
  int epoch;                       // launch number, never 0
  long long* trace;   // diagnostic (RSBA_TRACE=1)
  long long* wg_trace;  // diagnostic (RSBA_TRACE=2): start / end / compute-end stamp of every block
  // more than 64 cameras: the pair segments' hit lists, built once at set-up (PairSegmentSparse); nullptr: the masks are searched
  const unsigned* __restrict__ hits;       // [entry][3]: point, camera-major observation index on the a side, on the b side
  const unsigned* __restrict__ hit_off;    // [pair segment][4 waves]: first entry of the wave's list (64 entries per trip, lane-interleaved)
  const int* __restrict__ hit_trips;       // [pair segment][4 waves]: trips = the longest of the wave's 64 lists
  // launched AHEAD (queued behind the previous step's last kernel and the damping kernel, before the host knew that step's
  // outcome): the state is the device's decision (LmNext, ba_point_kernels.hpp) — dec[1] != 0: the previous step was accepted,
  // the camera constants and sqrt(rho') are the candidate's (the `_alt` pointers).  nullptr: the host chose
  const double* dec = nullptr;
  const double* camc_alt = nullptr;
  const double* sq_cm_alt = nullptr;
};

template <bool kLoss, bool kSmall>
__device__ __forceinline__ void PairSegment(const SchurArgs& a, const SchurSeg& sg, int seg_index, int ticket, double* pt, unsigned long long (*mk)[RSBA_CW], double* sc) {
  const int C = a.C, P = a.P, nwords = a.nwords;
  const double* __restrict__ camc = a.camc;
  const unsigned long long* __restrict__ cam_mask = a.cam_mask;
  const double* __restrict__ ptdata = a.ptdata;
  const int* __restrict__ cam_prefix = a.cam_prefix;
  const int* __restrict__ cam_ptr = a.cam_ptr;
  const double* __restrict__ sq_cm = a.sq_cm;
  double* __restrict__ partial = a.partial;
  const int tid = OpaqueTid();
  const bool diag_tile = sg.ga == sg.gb;
  // off-diagonal tile: lane = (ia, ib) directly.  Diagonal tile: the 120 pairs ia < ib sit in lanes 0..119 of BOTH halves of
  // the workgroup; waves 0/1 walk the even mask words of a chunk, waves 2/3 the odd ones, and the two partial blocks are
  // added through LDS at the end of the entry (with the pairs in two waves only, a diagonal-tile entry held its slot as
  // long as a full tile's for half the arithmetic).
  const int dt = diag_tile ? (tid & 127) : tid;
  const int pr = diag_tile ? (dt < 120 ? kDiagPair[dt] : 0) : tid;
  const int ia = pr >> 4, ib = pr & 15;
  const int cam_a = RSBA_TG * sg.ga + ia, cam_b = RSBA_TG * sg.gb + ib;
  const bool live = cam_a < C && cam_b < C && (!diag_tile || dt < 120);
  const int w0 = diag_tile ? (tid >> 7) : 0, wstep = diag_tile ? 2 : 1;
  // camera constants of the tile's 32 cameras: two values per thread, loaded here, stored with the first chunk's data
  double scv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + 256 * u;
    const int row = i / RSBA_SC_STRIDE, e = i - row * RSBA_SC_STRIDE;
    const int cam = row < RSBA_TG ? RSBA_TG * sg.ga + row : RSBA_TG * sg.gb + (row - RSBA_TG);
    const double* cc = camc + (size_t)(cam < C ? cam : 0) * CC_STRIDE;
    const int src = e < 9 ? CC_R + e : (e < 12 ? CC_T + e - 9 : (e == 12 ? CC_FX : (e == 13 ? CC_FY : CC_SMALL)));
    scv[u] = e < 15 ? cc[src] : 0.0;
  }
  const double* ca = sc + ia * RSBA_SC_STRIDE;
  const double* cb = sc + (RSBA_TG + ib) * RSBA_SC_STRIDE;
  {
    // first chunk: point data and visibility rows through registers, every load (camera constants included) issued before
    // the first LDS store — one round trip.  (The accumulators are not live yet: 48 more registers here are free.)
    // The records travel in 16-byte pieces (six a record): twelve loads and twelve LDS stores a thread instead of twenty-four
    // 8-byte ones — 248 -> 227 registers, 53 -> 30 spilled scalars; the step 0.3498 - 0.3511 -> 0.3471 - 0.3493 ms (round 5, one box, three
    // alternating runs each), the cfg5 shard 0.952 -> 0.948, the cfg4 shard 0.4036 -> 0.4018.
    const int wb = sg.word_begin;
    const int nwc = min(RSBA_CW, sg.word_end - wb);
    const int j0 = wb * 64;
    const int np = max(0, min(nwc * 64, P - j0));
    static_assert(RSBA_PT_STRIDE == 12, "six 16-byte pieces a record");
    constexpr int kPtPerThread = RSBA_CHUNK * 6 / 256;
    double2 pv[kPtPerThread];
    const double2* __restrict__ src2 = reinterpret_cast<const double2*>(ptdata + (size_t)j0 * RSBA_PT_STRIDE);
#pragma unroll
    for (int u = 0; u < kPtPerThread; ++u) { const int i = tid + 256 * u; pv[u] = i < np * 6 ? src2[i] : make_double2(0.0, 0.0); }
    unsigned long long mv;
    {
      const int row = tid >> 3, w = tid & 7;  // 32 rows x 8 words = 256 threads
      const int cam = row < RSBA_TG ? RSBA_TG * sg.ga + row : RSBA_TG * sg.gb + (row - RSBA_TG);
      mv = (cam < C && w < nwc) ? cam_mask[(size_t)cam * nwords + (size_t)wb + w] : 0ull;
    }
    sc[tid] = scv[0]; sc[tid + 256] = scv[1];
    PtLdsCursor pc(tid);
#pragma unroll
    for (int u = 0; u < kPtPerThread; ++u) { const int i = tid + 256 * u; if (i < np * 6) reinterpret_cast<double2*>(pt)[pc.a] = pv[u]; pc.Next(); }
    if ((tid & 7) < RSBA_CW) mk[tid >> 3][tid & 7] = mv;
    __syncthreads();
  }
  double acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;

  for (int wb = sg.word_begin; wb < sg.word_end; wb += RSBA_CW) {
    const int nwc = min(RSBA_CW, sg.word_end - wb);  // words of this LDS chunk
    if (wb != sg.word_begin) {
      // later chunks of a long segment (more than 64 cameras: many tiles, few segments each)
      const int j0 = wb * 64;
      const int np = max(0, min(nwc * 64, P - j0));
      __syncthreads();
      { PtLdsCursor pc(tid); const double2* __restrict__ src2 = reinterpret_cast<const double2*>(ptdata + (size_t)j0 * RSBA_PT_STRIDE);
        for (int i = tid; i < np * 6; i += 256) { reinterpret_cast<double2*>(pt)[pc.a] = src2[i]; pc.Next(); } }
      {
        const int row = tid >> 3, w = tid & 7;
        const int cam = row < RSBA_TG ? RSBA_TG * sg.ga + row : RSBA_TG * sg.gb + (row - RSBA_TG);
        if (w < RSBA_CW) mk[row][w] = (cam < C && w < nwc) ? cam_mask[(size_t)cam * nwords + (size_t)wb + w] : 0ull;
      }
      __syncthreads();
    }
    // every lane walks ITS OWN hit list through the whole chunk: the word index is per lane, so a wave
    // runs max-over-lanes(hits in 512 points) trips, not the sum over words of the per-word maxima
    // one flat loop per lane: (w, h) is the lane's cursor into its hit list; the cursor advance is a tiny inner loop
    // that does not touch the accumulators
    RSBA_PRIO_HITS(sg.stage);
    int w = w0;
    unsigned long long h = live ? (mk[ia][w0] & mk[RSBA_TG + ib][w0]) : 0ull;
    if (live) { while (h == 0ull && w + wstep < RSBA_CW) { w += wstep; h = mk[ia][w] & mk[RSBA_TG + ib][w]; } }
#pragma unroll 1
    while (h != 0ull) {
      const int bit = __ffsll((long long)h) - 1;
      const int wcur = w;
      h &= h - 1;
      while (h == 0ull && w + wstep < RSBA_CW) { w += wstep; h = mk[ia][w] & mk[RSBA_TG + ib][w]; }
      const double* pd = pt + (size_t)(wcur * 64 + bit) * RSBA_PT_LDS;
      const double X[3] = {pd[0], pd[1], pd[2]};
      const double v0 = pd[3], v1 = pd[4], v2 = pd[5], v3 = pd[6], v4 = pd[7], v5 = pd[8];
      double sqa = 1.0, sqb = 1.0;
      if (kLoss) {
        // rank of this point in each camera's own observation list -> its sqrt(rho')
        const unsigned long long below = (1ull << bit) - 1ull;
        const int gw = wb + wcur;
        sqa = sq_cm[cam_ptr[cam_a] + cam_prefix[(size_t)cam_a * nwords + gw] + __popcll(mk[ia][wcur] & below)];
        sqb = sq_cm[cam_ptr[cam_b] + cam_prefix[(size_t)cam_b * nwords + gw] + __popcll(mk[RSBA_TG + ib][wcur] & below)];
      }
      PairHit<kSmall>(ca, cb, X, v0, v1, v2, v3, v4, v5, sqa, sqb, acc);
    }
    RSBA_PRIO(3);
  }
  if (diag_tile) {
    // odd-word half (waves 2/3) -> LDS -> even-word half; the point buffer is free once every lane has left the chunk loop
    __syncthreads();
    if (tid >= 128 && dt < 120) {
#pragma unroll
      for (int i = 0; i < 36; ++i) pt[i * 128 + dt] = acc[i];
    }
    __syncthreads();
    if (tid < 120) {
#pragma unroll
      for (int i = 0; i < 36; ++i) acc[i] += pt[i * 128 + tid];
    }
  }
  // slot of pair (ia, ib) in the workgroup's partial block is ia*16+ib whatever lane computed it
  double* out = SegmentOut(sg, partial, a.grp_sum, seg_index);
  if (!diag_tile || tid < 120) {
#pragma unroll
    for (int i = 0; i < 36; ++i) __hip_atomic_store(&out[i * 256 + pr], acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (a.wg_trace && tid == 0) a.wg_trace[3 * ticket + 2] = wall_clock64();
  double v[36];
  if (!GroupReduce<36>(sg, partial, a.grp_sum, a.sync_cnt, a.ngrp, v, a.grp_flag, a.epoch)) return;
  FinishPairSlot(C, sg.ga, sg.gb, tid, v, camc, a.red, a.L, a.cam_free);
  if (!StageArrive(sg, a.sync_cnt, a.ngrp, a.ntiles, a.tag != 0)) return;
  if (tid == 0 && a.tag) PublishStage(a.ready + 1 + sg.stage, a.tag);
  if (a.trace && tid == 0 && sg.stage < 7) a.trace[17 + sg.stage] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// Pair segment above 64 cameras: SPARSE.  At 256 cameras x 20 views a camera pair shares 0.6 % of the points: a lane of the
// masked search above finds 3 hits in a 512-point chunk, a wave runs as many trips as its busiest lane (~8), and every
// workgroup stages every chunk (48 KB) to find them — the kernel spent its time staging and searching (680 us for 2.9 GFLOP,
// 5 % of the fp64 peak, round 2).  Which points a pair shares never changes, so the lists are built ONCE at set-up
// (TiledSchur::BuildHitLists): per pair segment and wavefront, lane-interleaved — entry n of lane l at [n * 64 + l], as many
// trips as the longest of the 64 lists (one segment is ~1/8 of the points: ~48 hits per lane, busiest lane ~60) — each entry
// the point and the two observations' places in the camera-major arrays (sqrt(rho')).  The point record (X, damped inverse
// block: 72 bytes) is gathered from ptdata one trip ahead of the arithmetic, the entries two.  No chunk staging, no barrier, no
// mask.  Same per-hit arithmetic as PairSegment, same partial blocks, same reduction tree.  A trip takes a wavefront ~1.8 us
// — what it takes in the 64-camera kernel (two wavefronts per SIMD, dependent fp64 chains) — and nothing about the gather
// moved that (round 3, each measured at the config-5 shard and taken out again): records fetched cooperatively, five 16-byte
// pieces per record on consecutive lanes through LDS (19 cache lines per load instruction instead of 64); a second copy of
// the records in 128-byte aligned slots (one line per hit instead of 1.5); the work list cut into one list per XCD so that
// a point range is always gathered through the same L2.
// A diagonal tile's 120 pairs sit in both halves of the workgroup as above; the halves split the hits by the parity of the
// point's 64-point word.
// ------------------------------------------------------------------------------------------------
#define RSBA_HIT_NONE 0xffffffffu
template <bool kLoss, bool kSmall>
__device__ __forceinline__ void PairSegmentSparse(const SchurArgs& a, const SchurSeg& sg, int seg_index, int ticket, double* pt, double* sc) {
  const int C = a.C;
  const double* __restrict__ camc = a.camc;
  const double* __restrict__ ptdata = a.ptdata;
  const double* __restrict__ sq_cm = a.sq_cm;
  double* __restrict__ partial = a.partial;
  const int tid = OpaqueTid(), wv = tid >> 6, ln = tid & 63;
  const bool diag_tile = sg.ga == sg.gb;
  const int dt = diag_tile ? (tid & 127) : tid;
  const int pr = diag_tile ? (dt < 120 ? kDiagPair[dt] : 0) : tid;
  const int ia = pr >> 4, ib = pr & 15;
  // this wave's list: first entry, trips
  const unsigned* __restrict__ hl = a.hits + 3 * ((size_t)a.hit_off[4 * seg_index + wv] + ln);
  const int ntrip = a.hit_trips[4 * seg_index + wv];
  // entries run TWO trips ahead of the arithmetic, point records one: the record's address depends on the entry, and an entry
  // fetched in the same trip as its record is a round trip to memory per trip with nothing to overlap it
  unsigned e0 = RSBA_HIT_NONE, e1 = 0, e2 = 0, f0 = RSBA_HIT_NONE, f1 = 0, f2 = 0;
  if (ntrip > 0) { e0 = hl[0]; e1 = hl[1]; e2 = hl[2]; }
  if (ntrip > 1) { f0 = hl[3 * 64]; f1 = hl[3 * 64 + 1]; f2 = hl[3 * 64 + 2]; }
  bool fv = ntrip > 1;   // (f holds a trip of the list: wave-uniform)
  // camera constants of the tile's 32 cameras: two values per thread
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + 256 * u;
    const int row = i / RSBA_SC_STRIDE, e = i - row * RSBA_SC_STRIDE;
    const int cam = row < RSBA_TG ? RSBA_TG * sg.ga + row : RSBA_TG * sg.gb + (row - RSBA_TG);
    const double* cc = camc + (size_t)(cam < C ? cam : 0) * CC_STRIDE;
    const int src = e < 9 ? CC_R + e : (e < 12 ? CC_T + e - 9 : (e == 12 ? CC_FX : (e == 13 ? CC_FY : CC_SMALL)));
    sc[i] = e < 15 ? cc[src] : 0.0;
  }
  const double* ca = sc + ia * RSBA_SC_STRIDE;
  const double* cb = sc + (RSBA_TG + ib) * RSBA_SC_STRIDE;
  // the first entry's point record and loss factors
  double pn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sqn[2] = {1.0, 1.0};
  auto fetch = [&](unsigned j, unsigned oa, unsigned ob) {
    const double* pd = ptdata + (size_t)j * RSBA_PT_STRIDE;
#pragma unroll
    for (int i = 0; i < 9; ++i) pn[i] = pd[i];
    if (kLoss) { sqn[0] = sq_cm[oa]; sqn[1] = sq_cm[ob]; }
  };
  if (e0 != RSBA_HIT_NONE) fetch(e0, e1, e2);
  __syncthreads();   // (the camera constants)
  double acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
  RSBA_PRIO(0);
#pragma unroll 1
  for (int n = 0; n < ntrip; ++n) {
    const bool hit = e0 != RSBA_HIT_NONE;
    const double X[3] = {pn[0], pn[1], pn[2]};
    const double v0 = pn[3], v1 = pn[4], v2 = pn[5], v3 = pn[6], v4 = pn[7], v5 = pn[8];
    const double sqa = sqn[0], sqb = sqn[1];
    // the next trip's records (its entries came in during the previous trip) and the entries after that go out before this
    // trip's arithmetic
    // (the entries' load is UNCONDITIONAL — past the list's end the last trip's again, told apart by fv when it is rotated in a trip later:
    //  under `if (n + 2 < ntrip)` the compiler rotated f1 / f2 inside the branch and waited for the load it had just issued — vmcnt(0) right
    //  behind it, a trip to memory exposed in every trip of the loop, the entries "two ahead" in name only)
    e0 = fv ? f0 : RSBA_HIT_NONE; e1 = f1; e2 = f2;
    {
      const unsigned* __restrict__ hn = hl + 3 * 64 * (size_t)min(n + 2, ntrip - 1);
      f0 = hn[0]; f1 = hn[1]; f2 = hn[2];
      fv = n + 2 < ntrip;
    }
    if (e0 != RSBA_HIT_NONE) fetch(e0, e1, e2);
    if (!hit) continue;
    PairHit<kSmall>(ca, cb, X, v0, v1, v2, v3, v4, v5, sqa, sqb, acc);
  }
  RSBA_PRIO(3);
  if (diag_tile) {
    // odd-word half (waves 2/3) -> LDS -> even-word half
    __syncthreads();
    if (tid >= 128 && dt < 120) {
#pragma unroll
      for (int i = 0; i < 36; ++i) pt[i * 128 + dt] = acc[i];
    }
    __syncthreads();
    if (tid < 120) {
#pragma unroll
      for (int i = 0; i < 36; ++i) acc[i] += pt[i * 128 + tid];
    }
  }
  double* out = SegmentOut(sg, partial, a.grp_sum, seg_index);
  if (!diag_tile || tid < 120) {
#pragma unroll
    for (int i = 0; i < 36; ++i) __hip_atomic_store(&out[i * 256 + pr], acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (a.wg_trace && tid == 0) a.wg_trace[3 * ticket + 2] = wall_clock64();
  double v[36];
  if (!GroupReduce<36>(sg, partial, a.grp_sum, a.sync_cnt, a.ngrp, v, a.grp_flag, a.epoch)) return;
  FinishPairSlot(C, sg.ga, sg.gb, tid, v, camc, a.red, a.L, a.cam_free);
  if (!StageArrive(sg, a.sync_cnt, a.ngrp, a.ntiles, a.tag != 0)) return;
  if (tid == 0 && a.tag) PublishStage(a.ready + 1 + sg.stage, a.tag);
  if (a.trace && tid == 0 && sg.stage < 7) a.trace[17 + sg.stage] = wall_clock64();
}

// Self tiles: everything that is a sum over ONE camera's observations — the diagonal block U_a - W V^-1 W' (a, a),
// diag(U_a) for the LM damping, the camera gradient g_c and the Schur correction of the right-hand side.
// Camera a sees ~3x the points a pair shares, so its points are dealt to 16 lanes: lane (ia, s) takes the points of
// rank s, s + 16, ... in the camera's list of the chunk (built in LDS, see below).  The residual needs this observation's pixel: u/v are kept in a second,
// camera-major copy (u_cm, v_cm) addressed by the rank of the point in the camera's own list.
// Per-lane sums (42, the K = J_l factor is applied by the tile's finisher, FinishSelfSlot):
//   [0,21)  upper triangle of  E'(I - N V^-1 N')E   = core of  U_a - W V^-1 W'
//   [21,27) upper triangle of the top-left 3x3 of E'E, [27,30) diagonal entries 3..5 of E'E   (-> diag U_a)
//   [30,36) E' r                                           (-> g_c)
//   [36,42) E' N V^-1 g_p                                  (-> -corr)
template <bool kLoss>
__device__ __forceinline__ void SelfSegment(const SchurArgs& a, const SchurSeg& sg, int seg_index, int ticket, double* pt, unsigned long long (*mk)[RSBA_CW],
                                            unsigned short (*lst)[RSBA_CHUNK], int* cnt) {
  const int C = a.C, P = a.P, nwords = a.nwords;
  const double* __restrict__ camc = a.camc;
  const unsigned long long* __restrict__ cam_mask = a.cam_mask;
  const double* __restrict__ ptdata = a.ptdata;
  const int* __restrict__ cam_prefix = a.cam_prefix;
  const int* __restrict__ cam_ptr = a.cam_ptr;
  const double* __restrict__ sq_cm = a.sq_cm;
  const double* __restrict__ u_cm = a.u_cm;
  const double* __restrict__ v_cm = a.v_cm;
  double* __restrict__ partial = a.partial;
  const int tid = OpaqueTid(), ia = tid >> 4, ib = tid & 15;
  const int cam_a = RSBA_TG * sg.ga + ia;
  const bool live = cam_a < C;
  const double* cca = camc + (size_t)(live ? cam_a : 0) * CC_STRIDE;
  SideConst A;
  LoadSide(cca, A);
  const double ppx = cca[CC_PPX], ppy = cca[CC_PPY];
  const int obs0 = live ? cam_ptr[cam_a] : 0;
  double acc[RSBA_PART];
#pragma unroll
  for (int i = 0; i < RSBA_PART; ++i) acc[i] = 0.0;
  for (int wb = sg.word_begin; wb < sg.word_end; wb += RSBA_CW) {
    const int nwc = min(RSBA_CW, sg.word_end - wb);
    const int j0 = wb * 64;
    const int np = max(0, min(nwc * 64, P - j0));
    // every load of the chunk (and, on the first chunk, the camera's constants above) in flight before the first LDS store
    static_assert(RSBA_PT_STRIDE == 12, "six 16-byte pieces a record");
    constexpr int kPtPerThread = RSBA_CHUNK * 6 / 256;
    double2 pv[kPtPerThread];
    const double2* __restrict__ src2 = reinterpret_cast<const double2*>(ptdata + (size_t)j0 * RSBA_PT_STRIDE);
#pragma unroll
    for (int u = 0; u < kPtPerThread; ++u) { const int i = tid + 256 * u; pv[u] = i < np * 6 ? src2[i] : make_double2(0.0, 0.0); }
    unsigned long long mv = 0ull;
    if (tid < RSBA_TG * 8) {
      const int row = tid >> 3, w = tid & 7;
      const int cam = RSBA_TG * sg.ga + row;
      mv = (cam < C && w < nwc) ? cam_mask[(size_t)cam * nwords + (size_t)wb + w] : 0ull;
    }
    const int obs_chunk_pre = live ? cam_prefix[(size_t)cam_a * nwords + wb] : 0;
    __syncthreads();
    PtLdsCursor pc(tid);
#pragma unroll
    for (int u = 0; u < kPtPerThread; ++u) { const int i = tid + 256 * u; if (i < np * 6) reinterpret_cast<double2*>(pt)[pc.a] = pv[u]; pc.Next(); }
    if (tid < RSBA_TG * 8 && (tid & 7) < RSBA_CW) mk[tid >> 3][tid & 7] = mv;
    __syncthreads();
    // The camera's points of this chunk as a list, dealt to its 16 lanes BY RANK (lane s takes entries s, s + 16, ...): every
    // lane of a camera gets the same number of hits to within one (dealing by bit position left the lanes of a wave at 62 %
    // utilisation: max over 64 lanes of a Binomial(128, 0.3) against its mean), the rank of a point in the camera's own
    // observation list is the list index, and u / v / sqrt(rho') are read from consecutive addresses by consecutive lanes.
    // Thread (row, j) expands half a mask word: 16 threads per camera.
    {
      const int row = tid >> 4, j = tid & 15, w = j >> 1, half = j & 1;
      int base = 0;
#pragma unroll
      for (int x = 0; x < RSBA_CW; ++x) base += x < w ? __popcll(mk[row][x]) : 0;
      const unsigned long long m = w < RSBA_CW ? mk[row][w] : 0ull;   // (chunks of fewer than eight words: the threads beyond them idle)
      unsigned int my = half ? (unsigned int)(m >> 32) : (unsigned int)m;
      if (half) base += __popc((unsigned int)m);
      if (j == 2 * RSBA_CW - 1) cnt[row] = base + __popc(my);
      const int p0 = w * 64 + half * 32;
      while (my != 0u) {
        const int b = __ffs((int)my) - 1;
        my &= my - 1u;
        lst[row][base++] = (unsigned short)(p0 + b);
      }
    }
    __syncthreads();
    const int n_a = live ? cnt[ia] : 0;
    const int obs_chunk = live ? obs0 + obs_chunk_pre : 0;
    RSBA_PRIO_HITS(sg.stage);
#pragma unroll 1
    for (int i = ib; i < n_a; i += 16) {
      const double* pd = pt + (size_t)lst[ia][i] * RSBA_PT_LDS;
      const double X[3] = {pd[0], pd[1], pd[2]};
      const double v0 = pd[3], v1 = pd[4], v2 = pd[5], v3 = pd[6], v4 = pd[7], v5 = pd[8];
      const int oi = obs_chunk + i;   // rank of this point in the camera's own observation list
      const double sqa = kLoss ? sq_cm[oi] : 1.0;
      const double uu = u_cm[oi], vv = v_cm[oi];
      // (fetching the next hit's pixel one trip ahead was measured, round 4: the self segments take as long — they wait elsewhere)
      double e0[6], e1[6], n0[3], n1[3];
      SideRows(A, X, sqa, e0, e1, n0, n1);
      // the residual: ProjectResidual (ba_math.hpp), the one sequence of roundings g_p was formed with as well
      double pq[3], izq, rr[2];
      ProjectResidual(A.R, A.t, A.fx, A.fy, ppx, ppy, X, uu, vv, pq, &izq, rr);
      const double r0 = rr[0] * sqa, r1 = rr[1] * sqa;
      // M' = I - N Vinv N'  (2x2 symmetric)
      const double t00 = n0[0] * v0 + n0[1] * v1 + n0[2] * v2, t01 = n0[0] * v1 + n0[1] * v3 + n0[2] * v4, t02 = n0[0] * v2 + n0[1] * v4 + n0[2] * v5;
      const double t10 = n1[0] * v0 + n1[1] * v1 + n1[2] * v2, t11 = n1[0] * v1 + n1[1] * v3 + n1[2] * v4, t12 = n1[0] * v2 + n1[1] * v4 + n1[2] * v5;
      const double m00 = 1.0 - (t00 * n0[0] + t01 * n0[1] + t02 * n0[2]);
      const double m01 = -(t00 * n1[0] + t01 * n1[1] + t02 * n1[2]);
      const double m11 = 1.0 - (t10 * n1[0] + t11 * n1[1] + t12 * n1[2]);
      int t = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        // e0[4] and e1[3] are structural zeros (see SideRows): one product instead of two wherever they enter
        double z0, z1;
        if (p == 3) { z0 = e0[3] * m00; z1 = e0[3] * m01; }
        else if (p == 4) { z0 = e1[4] * m01; z1 = e1[4] * m11; }
        else { z0 = e0[p] * m00 + e1[p] * m01; z1 = e0[p] * m01 + e1[p] * m11; }
#pragma unroll
        for (int q = p; q < 6; ++q) {
          if (q == 3) acc[t] = fma(z0, e0[3], acc[t]);
          else if (q == 4) acc[t] = fma(z1, e1[4], acc[t]);
          else acc[t] = Fma2(z0, e0[q], z1, e1[q], acc[t]);
          ++t;
        }
      }
      // E'E: top-left 3x3 (upper) and diagonal 3..5
      acc[21] = Fma2(e0[0], e0[0], e1[0], e1[0], acc[21]); acc[22] = Fma2(e0[0], e0[1], e1[0], e1[1], acc[22]); acc[23] = Fma2(e0[0], e0[2], e1[0], e1[2], acc[23]);
      acc[24] = Fma2(e0[1], e0[1], e1[1], e1[1], acc[24]); acc[25] = Fma2(e0[1], e0[2], e1[1], e1[2], acc[25]); acc[26] = Fma2(e0[2], e0[2], e1[2], e1[2], acc[26]);
      // e1[3] and e0[4] are structural zeros
      acc[27] = fma(e0[3], e0[3], acc[27]); acc[28] = fma(e1[4], e1[4], acc[28]); acc[29] = Fma2(e0[5], e0[5], e1[5], e1[5], acc[29]);
      const double f0 = n0[0] * pd[9] + n0[1] * pd[10] + n0[2] * pd[11];
      const double f1 = n1[0] * pd[9] + n1[1] * pd[10] + n1[2] * pd[11];
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        if (p == 3) { acc[33] = fma(e0[3], r0, acc[33]); acc[39] = fma(e0[3], f0, acc[39]); }
        else if (p == 4) { acc[34] = fma(e1[4], r1, acc[34]); acc[40] = fma(e1[4], f1, acc[40]); }
        else { acc[30 + p] = Fma2(e0[p], r0, e1[p], r1, acc[30 + p]); acc[36 + p] = Fma2(e0[p], f0, e1[p], f1, acc[36 + p]); }
      }
    }
    RSBA_PRIO(3);
  }
  double* out = SegmentOut(sg, partial, a.grp_sum, seg_index);
#pragma unroll
  for (int i = 0; i < RSBA_PART; ++i) __hip_atomic_store(&out[i * 256 + tid], acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.wg_trace && tid == 0) a.wg_trace[3 * ticket + 2] = wall_clock64();
  double v[RSBA_PART];
  if (!GroupReduce<RSBA_PART>(sg, partial, a.grp_sum, a.sync_cnt, a.ngrp, v, a.grp_flag, a.epoch)) return;
  FinishSelfSlot(C, sg.ga, v, camc, a.red, a.L, a.cam_free);
  if (sg.ga == a.last_group) FinishLinearize(a.nblocks_pp, a.block_scal, a.red, a.L, a.gmax_p, reinterpret_cast<double (*)[4]>(pt));   // (the chunk buffer is free: GroupReduce's barriers)
  if (a.self_only) return;
  const bool stage_done = StageArrive(sg, a.sync_cnt, a.ngrp, a.ntiles, a.tag != 0);
  if (a.all_self && a.tag && tid == 0) SelfTileArrive(a.sync_cnt, a.ngrp, a.ntiles, a.self_arrivals, a.ready, a.tag);
  if (!stage_done) return;
  if (tid == 0 && a.tag) PublishStage(a.ready + 1 + sg.stage, a.tag);
  if (a.trace && tid == 0 && sg.stage < 7) a.trace[17 + sg.stage] = wall_clock64();
}

// A reducer's sum over its tile's groups, in group order: v[k] += group q's component comp(k) of this thread's slot, q = 0, 1, ...
// The groups are taken as they arrive, and the flags of up to 64 groups ahead are looked at in ONE trip to memory (a wavefront,
// a flag per lane): a poll per batch of four, as it used to be, was a second dependent trip per batch — fourteen batches of ~5 us
// were as long as a whole stage, so the reducers ran behind the groups all the time and the stage's flag waited 15 - 27 us for
// them after its last partial block.  Batches of eight, four, or — among the tile's last eight groups — whatever is there; the
// additions are sequential in q whatever the batches, so the sum does not depend on them.  false: gave up (cannot happen, see
// GroupReduce; never hang).
template <int NC, typename CompOf>
__device__ __forceinline__ bool ReduceGroupsInOrder(const SchurArgs& a, const SchurSeg& sg, const double* __restrict__ in, CompOf comp, double* v, int nc_live) {
  __shared__ int s_run;
  const int tid = OpaqueTid();
  const int* gf = a.grp_flag + sg.tile_grp0;
  const int ng = sg.tile_ngrp;
  const long long t_begin = wall_clock64();
  int known = 0, q = 0;
  bool all_ok = true;
  while (q < ng) {
    const int want = ng - q > 8 ? q + 4 : q + 1;
    while (known < want) {
      if (tid < 64) {
        const int idx = known + tid;
        const int up = idx < ng && __hip_atomic_load(&gf[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch ? 1 : 0;
        const unsigned long long m = __ballot(up);
        int run = m == ~0ull ? 64 : __builtin_ctzll(~m);
        if (known + run < want && wall_clock64() - t_begin > RSBA_STALL_TICKS) run = -1;
        if (tid == 0) s_run = run;
      }
      __syncthreads();
      const int run = s_run;
      __syncthreads();
      if (run < 0) { all_ok = false; known = ng; break; }
      known += run;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (orders the group sums' loads below behind the flags' — the loads themselves are agent-scope)
      if (known < want) __builtin_amdgcn_s_sleep(4);
    }
    const int avail = known - q;
    if (avail >= 8) {
      double x[8][NC];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NC; ++k) x[u][k] = k < nc_live ? __hip_atomic_load(&in[(size_t)(q + u) * RSBA_PART * 256 + comp(k) * 256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NC; ++k) v[k] += x[u][k];
      q += 8;
    } else {
      const int nb = avail < 4 ? avail : 4;
      double x[4][NC];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < NC; ++k) x[u][k] = (u < nb && k < nc_live) ? __hip_atomic_load(&in[(size_t)(q + u) * RSBA_PART * 256 + comp(k) * 256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < NC; ++k) if (u < nb) v[k] += x[u][k];
      q += nb;
    }
  }
  return all_ok;
}

// Reducer workgroup of a PAIR tile: quadrant (qr, qc) = (word_begin >> 1, word_begin & 1) of every camera pair's 6 x 6 block.
// The K factors (blkdiag(J_l, I) on either side) couple nothing across the four 3 x 3 quadrants, so a reducer that adds the
// quadrant's nine components over the tile's groups can finish them itself: no tile sum, no election of a last reducer, no
// second fetch — three dependent trips to memory less at the tail of every stage (the tile sum's acknowledgement, the arrival
// counter, the finisher's fetch: ~8 of the 32 - 36 us between a stage's last partial block and its flag).  Its blocks of S
// leave with agent-scope stores whose acknowledgements it awaits before it arrives at the stage counter: no fence either.
// Same sums in the same order as the tile finisher's (groups in order, four at a time), same products.
__device__ __forceinline__ void ReducerQuadrant(const SchurArgs& a, const SchurSeg& sg, int ticket) {
  const int tid = OpaqueTid();
  const int qr = sg.word_begin >> 1, qc = sg.word_begin & 1;
  const int cbase = 18 * qr + 3 * qc;   // component 6 (3 qr + i) + 3 qc + j = cbase + 6 i + j
  const int ia = tid >> 4, ib = tid & 15;
  const int cam_a = RSBA_TG * sg.ga + ia, cam_b = RSBA_TG * sg.gb + ib;
  const bool live = cam_a < a.C && cam_b < a.C && !(sg.ga == sg.gb && ia >= ib);
  // the K factors ahead of the wait (camera constants: written before the launch)
  double Ka[9], Kb[9];
  {
    const double* pa = a.camc + (size_t)(live ? cam_a : 0) * CC_STRIDE + CC_K;
    const double* pb = a.camc + (size_t)(live ? cam_b : 0) * CC_STRIDE + CC_K;
#pragma unroll
    for (int i = 0; i < 9; ++i) { Ka[i] = pa[i]; Kb[i] = pb[i]; }
  }
  double keep = 1.0;
  if (a.cam_free != nullptr && live) keep = a.cam_free[cam_a] * a.cam_free[cam_b] == 0.0 ? 0.0 : 1.0;   // a constant camera couples to nobody
  double v[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = 0.0;
  const double* in = a.grp_sum + (size_t)sg.tile_grp0 * RSBA_PART * 256 + (size_t)cbase * 256 + tid;
  const bool all_ok = ReduceGroupsInOrder<9>(a, sg, in, [](int k) { return 6 * (k / 3) + k % 3; }, v, 9);
  if (a.wg_trace && tid == 0) a.wg_trace[3 * ticket + 2] = wall_clock64();   // the tile's groups are complete and added
  if (!all_ok && tid == 0) __hip_atomic_store(a.tree_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (live) {
    // rows: Ta' core (the first three rows mixed by Ka'), columns: (.) Tb — FinishPairSlot's products, quadrant by quadrant
    double t[9], blk[9];
#pragma unroll
    for (int qq = 0; qq < 3; ++qq)
#pragma unroll
      for (int p = 0; p < 3; ++p) t[3 * p + qq] = qr == 0 ? Ka[0 * 3 + p] * v[0 * 3 + qq] + Ka[1 * 3 + p] * v[1 * 3 + qq] + Ka[2 * 3 + p] * v[2 * 3 + qq] : v[3 * p + qq];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) blk[3 * p + qq] = keep * (qc == 0 ? t[3 * p + 0] * Kb[0 * 3 + qq] + t[3 * p + 1] * Kb[1 * 3 + qq] + t[3 * p + 2] * Kb[2 * 3 + qq] : t[3 * p + qq]);
    double* Sb = a.red + a.L.S() + (size_t)(6 * cam_a + 3 * qr) * a.L.nc + 6 * cam_b + 3 * qc;
    double* St = a.red + a.L.S() + (size_t)(6 * cam_b + 3 * qc) * a.L.nc + 6 * cam_a + 3 * qr;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) {
        __hip_atomic_store(&Sb[(size_t)p * a.L.nc + qq], -blk[3 * p + qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&St[(size_t)qq * a.L.nc + p], -blk[3 * p + qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
  __builtin_amdgcn_s_waitcnt(0);   // the blocks are in memory (StageArrive's barrier collects everybody's)
  if (!StageArrive(sg, a.sync_cnt, a.ngrp, a.ntiles, false)) return;
  if (tid == 0 && a.tag) PublishStage(a.ready + 1 + sg.stage, a.tag);
  if (a.trace && tid == 0 && sg.stage < 7) a.trace[17 + sg.stage] = wall_clock64();
}

// Reducer workgroup of a SELF tile: set word_begin of the tile's 42 components (per camera, in 16 slices: lane (ia, s)).  The K
// factor — T' . T with T = blkdiag(K, I) on the symmetric core, K' on the rotation halves of g_c and the right-hand-side correction,
// K' U K on the rotation block of U — couples nothing across these six sets, so the reducer that adds a set over the tile's groups
// finishes it itself, as ReducerQuadrant does for the pair tiles (no tile sum, no last reducer, no second fetch: the self tile's
// finish was 11 - 13 us behind its reducers, and at the last stage the kernel ended with it):
//   0: core rows / columns 0..2 (6)   1: core rows 0..2 x columns 3..5 (9)   2: core rows / columns 3..5 (6)
//   3: U rotation block (6)   4: g_c and correction, rotation halves (3 + 3)   5: diag U, g_c, correction, translation halves (9)
// Same sums in the same order as FinishSelfSlot's (groups in order, then the sixteen slices by the same shuffle tree), same products.
__device__ const int kSelfSetComp[RSBA_SELF_SETS][9] = {
    {0, 1, 2, 6, 7, 11, 0, 0, 0}, {3, 4, 5, 8, 9, 10, 12, 13, 14}, {15, 16, 17, 18, 19, 20, 0, 0, 0},
    {21, 22, 23, 24, 25, 26, 0, 0, 0}, {30, 31, 32, 36, 37, 38, 0, 0, 0}, {27, 28, 29, 33, 34, 35, 39, 40, 41}};
__device__ __forceinline__ void ReducerSelfSet(const SchurArgs& a, const SchurSeg& sg, int ticket, char* lds) {
  const int tid = OpaqueTid();
  const int set = sg.word_begin, nc = sg.word_end - sg.word_begin;
  const int ia = tid >> 4, ib = tid & 15, cam_a = RSBA_TG * sg.ga + ia;
  const bool live = cam_a < a.C;
  double K[9];
  {
    const double* pk = a.camc + (size_t)(live ? cam_a : 0) * CC_STRIDE + CC_K;   // (camera constants: written before the launch)
#pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = pk[i];
  }
  const bool is_free = !(a.cam_free != nullptr && live && a.cam_free[cam_a] == 0.0);
  int comp[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) comp[k] = kSelfSetComp[set][k];
  double v[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = 0.0;
  const double* in = a.grp_sum + (size_t)sg.tile_grp0 * RSBA_PART * 256 + tid;
  const bool all_ok = ReduceGroupsInOrder<9>(a, sg, in, [&](int k) { return comp[k]; }, v, nc);
  if (a.wg_trace && tid == 0) a.wg_trace[3 * ticket + 2] = wall_clock64();   // the tile's groups are complete and added
  if (!all_ok && tid == 0) __hip_atomic_store(a.tree_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the sixteen slices of a camera (lanes ia * 16 .. ia * 16 + 15, contiguous inside a wave) in the fixed tree order of FinishSelfSlot
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] += __shfl_down(v[k], off, 16);
  }
  if (ib == 0 && live) {
    if (!is_free) {
      // constant camera: zero block, gradient and damping diagonal (FinishSelfSlot)
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = 0.0;
    }
    const size_t nc_s = a.L.nc;
    double* Sd = a.red + a.L.S() + (size_t)(6 * cam_a) * nc_s + 6 * cam_a;
    auto put = [&](double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (set == 0) {
      const double c[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
      double tmp[9], blk[9];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int p = 0; p < 3; ++p) tmp[3 * p + q] = K[0 * 3 + p] * c[0 * 3 + q] + K[1 * 3 + p] * c[1 * 3 + q] + K[2 * 3 + p] * c[2 * 3 + q];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) blk[3 * p + q] = tmp[3 * p + 0] * K[0 * 3 + q] + tmp[3 * p + 1] * K[1 * 3 + q] + tmp[3 * p + 2] * K[2 * 3 + q];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = p; q < 3; ++q) { const double x = 0.5 * (blk[3 * p + q] + blk[3 * q + p]); put(&Sd[(size_t)p * nc_s + q], x); put(&Sd[(size_t)q * nc_s + p], x); }
    } else if (set == 1) {
      // c[k][j] = core (k, 3 + j) = v[3 k + j]: rows through K' on one side of the diagonal, columns through K on the other
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double up = K[0 * 3 + p] * v[0 * 3 + j] + K[1 * 3 + p] * v[1 * 3 + j] + K[2 * 3 + p] * v[2 * 3 + j];   // block (p, 3 + j)
          const double lo = v[0 * 3 + j] * K[0 * 3 + p] + v[1 * 3 + j] * K[1 * 3 + p] + v[2 * 3 + j] * K[2 * 3 + p];   // block (3 + j, p)
          const double x = 0.5 * (up + lo);
          put(&Sd[(size_t)p * nc_s + 3 + j], x); put(&Sd[(size_t)(3 + j) * nc_s + p], x);
        }
    } else if (set == 2) {
      int t = 0;
#pragma unroll
      for (int p = 3; p < 6; ++p)
#pragma unroll
        for (int q = p; q < 6; ++q) { const double x = 0.5 * (v[t] + v[t]); put(&Sd[(size_t)p * nc_s + q], x); put(&Sd[(size_t)q * nc_s + p], x); ++t; }
    } else if (set == 3) {
      const double u00 = v[0], u01 = v[1], u02 = v[2], u11 = v[3], u12 = v[4], u22 = v[5];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const double k0 = K[0 * 3 + p], k1 = K[1 * 3 + p], k2 = K[2 * 3 + p];
        put(&a.red[a.L.diagU() + 6 * cam_a + p], k0 * (u00 * k0 + u01 * k1 + u02 * k2) + k1 * (u01 * k0 + u11 * k1 + u12 * k2) + k2 * (u02 * k0 + u12 * k1 + u22 * k2));
      }
    } else if (set == 4) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        put(&a.red[a.L.gc() + 6 * cam_a + p], K[0 * 3 + p] * v[0] + K[1 * 3 + p] * v[1] + K[2 * 3 + p] * v[2]);
        put(&a.red[a.L.corr() + 6 * cam_a + p], -(K[0 * 3 + p] * v[3] + K[1 * 3 + p] * v[4] + K[2 * 3 + p] * v[5]));
      }
    } else {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        put(&a.red[a.L.diagU() + 6 * cam_a + 3 + p], v[p]);
        put(&a.red[a.L.gc() + 6 * cam_a + 3 + p], v[3 + p]);
        put(&a.red[a.L.corr() + 6 * cam_a + 3 + p], -v[6 + p]);
      }
    }
  }
  const bool lin = set == RSBA_SELF_SETS - 1 && sg.ga == a.last_group;   // this reducer also folds the point pass' per-block scalars
  if (lin) FinishLinearize(a.nblocks_pp, a.block_scal, a.red, a.L, a.gmax_p, reinterpret_cast<double (*)[4]>(lds));
  if (a.self_only) return;
  __builtin_amdgcn_s_waitcnt(0);   // the results are in memory (StageArrive's barrier collects everybody's; its fence: FinishLinearize's plain stores)
  const bool stage_done = StageArrive(sg, a.sync_cnt, a.ngrp, a.ntiles, lin && a.tag != 0);
  if (a.all_self && a.tag && tid == 0) SelfTileArrive(a.sync_cnt, a.ngrp, a.ntiles, a.self_arrivals, a.ready, a.tag);
  if (!stage_done) return;
  if (tid == 0 && a.tag) PublishStage(a.ready + 1 + sg.stage, a.tag);
  if (a.trace && tid == 0 && sg.stage < 7) a.trace[17 + sg.stage] = wall_clock64();
}

// A reducer workgroup of tile sg.tile (see GroupReduce): a quadrant of a pair tile's blocks or a set of a self tile's components, each
// finished by the reducer that adds it; the last arrival of a stage publishes it for the Cholesky that may be waiting for these
// columns.  (Until round 3 a reducer added a slice of the components into a tile sum and the last one fetched it back and
// finished the tile: three dependent trips to memory more at the tail of every stage.)
__device__ __forceinline__ void ReducerSegment(const SchurArgs& a, const SchurSeg& sg, int ticket, char* lds) {
  if (sg.self == 2) ReducerQuadrant(a, sg, ticket);
  else ReducerSelfSet(a, sg, ticket, lds);
}

// K_A2: the Schur elimination kernel.  One workgroup per segment (a range of 64-point words of one tile), in block order
// stage by stage: the self tile of camera group g, then the pair tiles (g, g' >= g) — what the Cholesky needs for group
// g's columns.  Two workgroups
// per CU: the accumulators are the only long-lived per-lane state; point data, visibility rows and camera constants sit
// in LDS.
// kSparse: the instance for more than 64 cameras (PairSegmentSparse instead of PairSegment) — a kernel of its own: compiled into
// one kernel beside the masked search, the sparse path cost the 64-camera kernel 50 us (284 -> 337 us: the same registers, but
// more scalar spills and a longer hot loop around the same arithmetic)
// kMode: RSBA_PAIRS_MASKED / _SPARSE — which pair segment the instance holds (one each: compiled into one kernel beside
// another, a pair segment costs the other registers, scalar spills and a longer hot loop)
#define RSBA_PAIRS_MASKED 0
#define RSBA_PAIRS_SPARSE 1
// chunk records | mask rows 2 KB | camera constants 4 KB | a self tile's lists | counters.  512-point chunks: 81 024 bytes, two
// workgroups per CU
#define RSBA_SCHUR_LDS_BYTES (RSBA_CHUNK * RSBA_PT_LDS * 8 + 2048 + 4096 + 2 * RSBA_TG * RSBA_CHUNK + 96)
template <bool kLoss, int kMode>
__global__ void __launch_bounds__(256, 2)
k_schur_tiles(int* __restrict__ ticket_p, unsigned ticket_base, int total, const int* __restrict__ small_flag_p, const SchurSeg* __restrict__ segs_p, SchurArgs a) {
  // (what a workgroup needs FIRST — the ticket counter, where this launch's tickets start, the work list — comes as separate leading
  //  arguments: the build preloads the first sixteen argument words into scalar registers at wave launch
  //  (-mllvm -amdgpu-kernarg-preload-count=16), a structure passed by value is not among them, and the ticket used to wait for
  //  a scalar load of its own address)
  // One raw LDS buffer, carved by the entry's role: masked / sparse pair segments and self segments: chunk records 48 KB | mask
  // rows 2 KB | camera constants 4 KB | self tiles: a camera's points of the chunk 16 KB | 16 counters
  __shared__ __attribute__((aligned(16))) char lds_raw[RSBA_SCHUR_LDS_BYTES];
  static_assert(RSBA_CHUNK * RSBA_PT_LDS * 8 + 2048 + 4096 + 2 * RSBA_TG * RSBA_CHUNK + 64 <= RSBA_SCHUR_LDS_BYTES, "LDS carve");
  static_assert(RSBA_CHUNK != 512 || 2 * (RSBA_SCHUR_LDS_BYTES + 256) <= 160 * 1024, "two workgroups per CU");
  constexpr bool kSparse = kMode == RSBA_PAIRS_SPARSE;
  constexpr bool kResident = kSparse || RSBA_RESIDENT != 0;   // the workgroups draw tickets until the list is through
  double* pt = reinterpret_cast<double*>(lds_raw);
  unsigned long long (*mk)[RSBA_CW] = reinterpret_cast<unsigned long long (*)[RSBA_CW]>(lds_raw + RSBA_CHUNK * RSBA_PT_LDS * 8);
  double* sc = reinterpret_cast<double*>(lds_raw + RSBA_CHUNK * RSBA_PT_LDS * 8 + 2048);
  unsigned short (*lst)[RSBA_CHUNK] = reinterpret_cast<unsigned short (*)[RSBA_CHUNK]>(lds_raw + RSBA_CHUNK * RSBA_PT_LDS * 8 + 2048 + 4096);
  int* cnt = reinterpret_cast<int*>(lds_raw + RSBA_CHUNK * RSBA_PT_LDS * 8 + 2048 + 4096 + 2 * RSBA_TG * RSBA_CHUNK);
  // Work is handed out by ticket, not by block index: blocks are assigned to the 8 XCDs round-robin and each XCD
  // dispatches its own in order, so an XCD that is a little slower (the one that lends a CU to the Cholesky has 62 slots
  // instead of 64) starts the last blocks of a stage tens of microseconds late, and the stage ends with them.  With
  // tickets the order of the work list is the order in which slots take it up, whoever they are.  (The sums do not
  // depend on who does which segment.)  The last ticket resets the counter for the next launch.
  // What stands between the dispatch of a workgroup and its first hit is a chain of dependent round trips to memory, ~1-2 us
  // each beside 500 other workgroups, per entry of ~45 us: ticket -> entry -> camera constants / point data / visibility
  // rows.  It used to be six long (ticket, block -> segment index, segment, small-angle flags of the tile's cameras,
  // camera constants, point data); now the entry is ONE load off the ticket (segs_ordered), the small-angle flag is one
  // word for the whole problem, fetched with the ticket, and everything an entry stages is in flight before the first
  // LDS store (PairSegment / SelfSegment).
  // kSparse (more than 64 cameras): the workgroups are RESIDENT and draw tickets until the list is through (one launch = as
  // many workgroups as the chip holds, not one per entry): between two entries of a slot lay the end of a workgroup, the
  // dispatch of the next and its first round trips, 12 us per entry of ~85 us at 256 cameras (377 -> 351 us).  Not so at up
  // to 64 cameras: the same loop around the masked search costs that instance 40 us (281 -> 321 us, more scalar spills in
  // the hot loop), more than the 6.8 us per entry it saves.  The counter is never reset: a launch moves it by its draws —
  // the entries, plus one draw past the end per resident workgroup — and the host passes where it starts.
  __shared__ int s_ticket, s_small;
  // Priorities: what a workgroup does outside its hit loop is a handful of instructions between trips to memory (ticket, entry,
  // staging; partial sums, arrival counters) — but its CU-mate, if older, is in a hit loop that issues an fp64 instruction
  // whenever it can, and the arbiter serves the older wavefront first: the newcomer's few hundred prologue instructions were
  // served at the rate the other left slots free (~8 %), 6 - 8 us from the end of one entry to the start of the next on a slot.
  // So: priority 3 outside the hit loops, 0 inside (RSBA_PRIO() compiles to s_setprio).
  RSBA_PRIO(3);
  if (threadIdx.x == 0) s_small = __hip_atomic_load(small_flag_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (RSBA_EXP(a.dec != nullptr) && a.dec[1] != 0.0) { a.camc = a.camc_alt; a.sq_cm = a.sq_cm_alt; }   // (uniform: scalar loads beside the ticket's round trip)
  for (;;) {
    if (threadIdx.x == 0) s_ticket = (int)((unsigned)__hip_atomic_fetch_add(ticket_p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_base);
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(s_ticket);   // (uniform: the entry is read with scalar loads and stays in scalar registers)
    if (kResident && b >= total) break;
    const SchurSeg sg = LoadSeg(segs_p + b);
    const int seg_index = sg.index;
    if (a.trace && b == 0 && threadIdx.x == 0) a.trace[24] = wall_clock64();
    if (a.wg_trace && threadIdx.x == 0) a.wg_trace[3 * b] = wall_clock64();
    if (sg.self >= 2) ReducerSegment(a, sg, b, lds_raw);
    else if (sg.self) SelfSegment<kLoss>(a, sg, seg_index, b, pt, mk, lst, cnt);
    else {
      // two instances of the pair tile: the small-angle selects of the Jacobian rows (12 instructions per hit) are compiled in
      // only when some camera takes that branch this iteration (a rotation of exactly zero: the reference's test2 fixture)
      if (kSparse) {
        if (s_small) PairSegmentSparse<kLoss, true>(a, sg, seg_index, b, pt, sc);
        else PairSegmentSparse<kLoss, false>(a, sg, seg_index, b, pt, sc);
      } else if (s_small) PairSegment<kLoss, true>(a, sg, seg_index, b, pt, mk, sc);
      else PairSegment<kLoss, false>(a, sg, seg_index, b, pt, mk, sc);
    }
    if (a.wg_trace && threadIdx.x == 0) a.wg_trace[3 * b + 1] = wall_clock64();
    if (!kResident) break;   // one entry per workgroup
    __syncthreads();       // (the next entry reuses the staging buffers and s_ticket)
  }
}


// Multi-GPU pipeline, communication stream: k_wait_stage holds the stream until the Schur kernel has published stage g
// locally; then ONE grouped all-reduce sums the stage's row slab of S (complete on every rank once the stage is: the finishers
// write every block and its mirror) and the group's ranges of g_c, the right-hand-side correction and diag U; k_set_flag
// publishes the summed stage to the factorisation, which reads the slab transposed (StageGate::transposed).  One wavefront and
// a handful of registers each: they are dispatched at once beside the chip-filling Schur kernel and the resident
// back-substitution.  (Tried instead, round 3: the lower trapezoid of the group's columns gathered into one buffer and
// scattered back behind the all-reduce — half the bytes and one collective per stage, but the two copy kernels, fifteen
// workgroups of 46 registers each, waited 5 to 26 us for slots on the full chip, every stage, on the path of the gate.)
// (No fence in either kernel: what the Schur kernel published before the flag is acquired by the collective's own kernel
//  start, what the collective wrote is released by its kernel end, and the flag store follows it in stream order.  With
//  fences — an L2 write-back beside the Schur kernel or the prefetching back-substitution — k_set_flag took 5 to 29 us.)
__global__ void __launch_bounds__(64) k_wait_stage(const int* __restrict__ flag, int tag, int* __restrict__ error) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 10 * RSBA_STALL_TICKS) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
}
__global__ void __launch_bounds__(64) k_set_flag(int* __restrict__ flag, int tag) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace rsba
