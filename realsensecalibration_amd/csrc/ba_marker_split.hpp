// Marker-chain model at scale, the elimination of the time blocks SPLIT by what a sum runs over (round 6).
//
// k_time_eliminate (ba_marker_schur.hpp) walks a chunk of times with one 1024-thread workgroup: per tile of 32 residual blocks
// five barriers with a latency chain between each (7.5 us for ~60 k FMAs; 0.76 ms at 8 x 5000 x 16, 0.03 of the fp64 peak).
// The Jacobian rows of a residual block are ~160 FMAs a corner (MarkerCornerResidualJacobian), cheaper than any hand-over
// between threads: so every sum gets the threads that own it and forms the rows it needs itself, in registers, with no barrier:
//
//   k_mc_slot_products   thread per (time, slot) — slot = a camera or marker block the time's residuals touch: over the slot's
//                        residual blocks (block order), W_s = J_t' J_s (6 x 6), g_s = J_s' r, U_ss = J_s' J_s -> a 64-double record
//   k_mc_time_products   wavefront per time: V = sum J_t' J_t, g_t, sum r^2 (lanes over the blocks, fixed butterfly), then the time's
//                        Jacobi scale, damping, E = (V + D)^-1 on 36 lanes, E g_t -> tdata, and the time's scalars
//   k_mc_cross           thread per (chunk of times, camera-marker pair): U_cm = sum J_m' J_c over the pair's residual blocks
//   k_mc_accumulate      workgroup per chunk of times (the chunks of k_time_eliminate, the same partial systems): per time the
//                        records come from memory (one step ahead, in registers), Y = E W, then the partial system takes
//                        U_ss - W'Y, g_s, -W'E g_t — two barriers a time, no Jacobian in sight
//
// then k_marker_reduce as before.  Every sum has a fixed order (blocks in time order inside a slot, a pair, a time's lanes): bitwise
// reproducible, no atomics.  The rows are formed 2.7 times per residual block instead of once (~1.9 k FMAs against the products' 1.5 k).
// Reference: the four functors of Main_Calibration/bundle_adjustment.h:56-343 (rows: ba_math.hpp), Ceres' SCHUR elimination order.
#pragma once

namespace rsba {

// corners of a residual block formed side by side in the product kernels' loops (they run one wavefront a SIMD: the registers are there,
// what is missing is independent work between dependent instructions)
#ifndef RSBA_MC_UNROLL_SLOT
#define RSBA_MC_UNROLL_SLOT 1
#endif
#ifndef RSBA_MC_UNROLL_CROSS
#define RSBA_MC_UNROLL_CROSS 2
#endif
#ifndef RSBA_MC_UNROLL_CAND
#define RSBA_MC_UNROLL_CAND 1
#endif
#define RSBA_SP_STRIDE 64   // doubles of a (time, slot) record: W (time row x, slot column q at 6 x + q: 36) | g_s (6) | U_ss lower triangle (21) | pad
#define RSBA_SP_LDS 65      // its stride in LDS

struct PoseC {
  double R[9], K[9], T[3];
  bool small, on;
};
// kFull: with the left Jacobian and the small-angle flag (a pose whose own 2 x 6 block is wanted), else rotation and translation only
template <bool kFull>
__device__ __forceinline__ PoseC LoadPose(const double* __restrict__ posec, int pose /* -1: the transform is absent from the chain */) {
  PoseC p;
  p.on = pose >= 0;
  const double* q = posec + (size_t)(pose >= 0 ? pose : 0) * CC_STRIDE;
#pragma unroll
  for (int i = 0; i < 9; ++i) { p.R[i] = q[CC_R + i]; p.K[i] = kFull ? q[CC_K + i] : 0.0; }
#pragma unroll
  for (int i = 0; i < 3; ++i) p.T[i] = q[CC_T + i];
  p.small = kFull ? q[CC_SMALL] != 0.0 : false;
  return p;
}

// One corner's residuals and the wanted 2 x 6 blocks of its rows: the arithmetic of MarkerCornerResidualJacobian (ba_math.hpp), operation
// for operation, on pose constants held in registers.  A wanted block's pose is present (the caller owns it).
template <bool kC, bool kT, bool kM>
__device__ __forceinline__ void CornerRows(const PoseC& cam, const PoseC& tim, const PoseC& mar, double fx, double fy, double ppx, double ppy,
                                           double cx, double cy, double u, double v, double r[2], double Jc[2][6], double Jt[2][6], double Jm[2][6]) {
  auto rot = [](const PoseC& p, const double in[3], double q[3], double out[3]) {
    q[0] = p.R[0] * in[0] + p.R[1] * in[1] + p.R[2] * in[2];
    q[1] = p.R[3] * in[0] + p.R[4] * in[1] + p.R[5] * in[2];
    q[2] = p.R[6] * in[0] + p.R[7] * in[1] + p.R[8] * in[2];
    out[0] = q[0] + p.T[0]; out[1] = q[1] + p.T[1]; out[2] = q[2] + p.T[2];
  };
  auto block = [](const PoseC& p, const double Q[6], const double pin[3], const double q[3], double Jb[2][6]) {
    const double w0 = p.small ? pin[0] : q[0], w1 = p.small ? pin[1] : q[1], w2 = p.small ? pin[2] : q[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double q0 = Q[3 * i], q1 = Q[3 * i + 1], q2 = Q[3 * i + 2];
      const double a0 = w1 * q2 - w2 * q1, a1 = w2 * q0 - w0 * q2, a2 = w0 * q1 - w1 * q0;
      Jb[i][0] = a0 * p.K[0] + a1 * p.K[3] + a2 * p.K[6];
      Jb[i][1] = a0 * p.K[1] + a1 * p.K[4] + a2 * p.K[7];
      Jb[i][2] = a0 * p.K[2] + a1 * p.K[5] + a2 * p.K[8];
      Jb[i][3] = q0; Jb[i][4] = q1; Jb[i][5] = q2;
    }
  };
  const double X[3] = {cx, cy, 0.0};
  double qm[3] = {0, 0, 0}, pm[3] = {X[0], X[1], X[2]};
  if (mar.on) rot(mar, X, qm, pm);
  double qt[3], pt[3];
  rot(tim, pm, qt, pt);
  double qc[3] = {0, 0, 0}, pcm[3] = {pt[0], pt[1], pt[2]};
  if (cam.on) rot(cam, pt, qc, pcm);
  const double iz = 1.0 / pcm[2];
  r[0] = fx * pcm[0] * iz + ppx - u;
  r[1] = fy * pcm[1] * iz + ppy - v;
  const double al = fx * iz, be = fy * iz;
  const double ga = -al * pcm[0] * iz, de = -be * pcm[1] * iz;
  double Qt[6];
  if (cam.on) {
    if (kC) { const double Qc[6] = {al, 0.0, ga, 0.0, be, de}; block(cam, Qc, pt, qc, Jc); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { Qt[j] = al * cam.R[j] + ga * cam.R[6 + j]; Qt[3 + j] = be * cam.R[3 + j] + de * cam.R[6 + j]; }
  } else {
    Qt[0] = al; Qt[1] = 0.0; Qt[2] = ga; Qt[3] = 0.0; Qt[4] = be; Qt[5] = de;
  }
  if (kT) block(tim, Qt, pm, qt, Jt);
  if (kM) {
    double Qm[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Qm[3 * i + j] = Qt[3 * i] * tim.R[j] + Qt[3 * i + 1] * tim.R[3 + j] + Qt[3 * i + 2] * tim.R[6 + j];
    block(mar, Qm, X, qm, Jm);
  }
}

__device__ __forceinline__ double CornerX(int k, double hs) { return (k == 0 || k == 3) ? -hs : hs; }
__device__ __forceinline__ double CornerY(int k, double hs) { return k < 2 ? hs : -hs; }

struct SplitArgs {
  int nslots, T, nx, ncam_cols;
  int nx_threads;                       // entries of x_order
  const int* __restrict__ slot_order;   // [nslots] thread -> slot: camera slots first, the longest lists first inside a role (a wavefront's lanes run alike)
  const int* __restrict__ x_order;      // [nx_threads] thread -> 4 item + 2 (the item's list is shared by this lane and its neighbour) + part, of k_mc_cross
  const int* __restrict__ slot_time;    // [nslots]
  const int* __restrict__ slot_col;     // [nslots] first reduced column
  const int* __restrict__ sb_ptr;       // [nslots + 1] the slot's residual blocks ...
  const int4* __restrict__ sb_blk;      // ... in block (= time, then file) order: {residual block, the OTHER block's pose (-1: none), the detecting camera (intrinsics), 0}
  const int* __restrict__ time_ptr;     // [T + 1]
  const int* __restrict__ time_full;    // [T]
  const int* __restrict__ col_full;     // [nr]
  const TimeSlots* __restrict__ ts;
  const MarkerObs* __restrict__ mo;
  const double* __restrict__ obs8;
  const double* __restrict__ intr;
  const double* __restrict__ posec;
  double half_side;
  const int* __restrict__ xi_ptr;       // [nx + 1] residual blocks of a (chunk, camera column, marker column) item
  const int4* __restrict__ xi_blk;      // {residual block, the time's pose, the detecting camera, 0}
  const int* __restrict__ xi_cc;        // [nx] camera column, [nx] marker column
  const int* __restrict__ xi_cm;
  double* __restrict__ sp;              // [nslots][RSBA_SP_STRIDE]
  double* __restrict__ xout;            // [nx][36]: U_cm, marker row qm, camera column qc at 6 qm + qc
};

template <bool kCam>
__device__ __forceinline__ void SlotProducts(const SplitArgs& a, int S, const PoseC& own, const PoseC& tim) {
  double W[36], gs[6], U[21];
#pragma unroll
  for (int i = 0; i < 36; ++i) W[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) gs[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 21; ++i) U[i] = 0.0;
  const double hs = a.half_side;
  // What a residual block's rows need from memory — the other block's pose, the observed corners, the intrinsics — is asked for ONE ENTRY AHEAD,
  // and the entry's own record two ahead: as block -> slots -> column -> pose -> constants it was four dependent trips to memory per entry, on one
  // wavefront a SIMD with nothing to run meanwhile (5 us an entry; the arithmetic is 1.2).
  struct Fetched { PoseC oth; double ob[8], in[4]; };
  auto fetch = [&](const int4 r, Fetched& f) {
    f.oth = LoadPose<false>(a.posec, r.y);
    const double* ob = a.obs8 + 8 * (size_t)r.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.ob[i] = ob[i];
    const double* in = a.intr + 4 * r.z;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.in[i] = in[i];
  };
  int e = a.sb_ptr[S];
  const int e1 = a.sb_ptr[S + 1];
  const int4 none = {0, -1, 0, 0};
  int4 r1 = e < e1 ? a.sb_blk[e] : none;
  Fetched cur;
  fetch(r1, cur);
  r1 = e + 1 < e1 ? a.sb_blk[e + 1] : none;
  for (; e < e1; ++e) {
    const int4 r2 = e + 2 < e1 ? a.sb_blk[e + 2] : none;
    Fetched nxt;
    fetch(r1, nxt);
    const PoseC& oth = cur.oth;
    const double fx = cur.in[0], fy = cur.in[1], ppx = cur.in[2], ppy = cur.in[3];
#pragma unroll RSBA_MC_UNROLL_SLOT
    for (int c = 0; c < 4; ++c) {
      // (the corner's pixels by selects: an index into the fetched array that is not a constant puts the whole record into private memory)
      const double ou = c == 0 ? cur.ob[0] : (c == 1 ? cur.ob[2] : (c == 2 ? cur.ob[4] : cur.ob[6]));
      const double ov = c == 0 ? cur.ob[1] : (c == 1 ? cur.ob[3] : (c == 2 ? cur.ob[5] : cur.ob[7]));
      double r[2], Jo[2][6], Jt[2][6];
      if (kCam) CornerRows<true, true, false>(own, tim, oth, fx, fy, ppx, ppy, CornerX(c, hs), CornerY(c, hs), ou, ov, r, Jo, Jt, nullptr);
      else CornerRows<false, true, true>(oth, tim, own, fx, fy, ppx, ppy, CornerX(c, hs), CornerY(c, hs), ou, ov, r, nullptr, Jt, Jo);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
          for (int q = 0; q < 6; ++q) W[6 * x + q] = fma(Jt[i][x], Jo[i][q], W[6 * x + q]);
#pragma unroll
        for (int q = 0; q < 6; ++q) gs[q] = fma(Jo[i][q], r[i], gs[q]);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int p = 0; p <= q; ++p) U[q * (q + 1) / 2 + p] = fma(Jo[i][q], Jo[i][p], U[q * (q + 1) / 2 + p]);
      }
    }
    cur = nxt;
    r1 = r2;
  }
  double* out = a.sp + (size_t)S * RSBA_SP_STRIDE;
#pragma unroll
  for (int i = 0; i < 36; ++i) out[i] = W[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) out[36 + i] = gs[i];
#pragma unroll
  for (int i = 0; i < 21; ++i) out[42 + i] = U[i];
  out[63] = 0.0;
}

__global__ void __launch_bounds__(256) k_mc_slot_products(SplitArgs a) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.nslots) return;
  const int S = a.slot_order[g], col = a.slot_col[S];
  const PoseC own = LoadPose<true>(a.posec, a.col_full[col] / 6);
  const PoseC tim = LoadPose<true>(a.posec, a.time_full[a.slot_time[S]] / 6);
  if (col < a.ncam_cols) SlotProducts<true>(a, S, own, tim);
  else SlotProducts<false>(a, S, own, tim);
}

// Wavefront per time.  tdata[t]: E (36) | g_t (6) | E g_t (6); tscal[t]: sum r^2, |x_t|^2, 1.0 if V + D is not positive definite, max |g_t|.
__global__ void __launch_bounds__(256) k_mc_time_products(SplitArgs a, IterParams ip, const double* __restrict__ params_x, double* __restrict__ scale_t,
                                                          double* __restrict__ tdata, double* __restrict__ tscal) {
  __shared__ double s_v[4][3 * 36 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wave;
  if (t >= a.T) return;   // (whole wavefronts; no workgroup barrier below)
  const PoseC tim = LoadPose<true>(a.posec, a.time_full[t] / 6);
  double V[21], g[6], ss = 0.0;
#pragma unroll
  for (int i = 0; i < 21; ++i) V[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = 0.0;
  const double hs = a.half_side;
  for (int k = a.time_ptr[t] + lane; k < a.time_ptr[t + 1]; k += 64) {
    const TimeSlots s = a.ts[k];
    const PoseC cam = LoadPose<false>(a.posec, s.col_cam >= 0 ? a.col_full[s.col_cam] / 6 : -1);
    const PoseC mar = LoadPose<false>(a.posec, s.col_marker >= 0 ? a.col_full[s.col_marker] / 6 : -1);
    const double* in = a.intr + 4 * s.camera;
    const double fx = in[0], fy = in[1], ppx = in[2], ppy = in[3];
    const double* ob = a.obs8 + 8 * (size_t)k;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      double r[2], Jt[2][6];
      CornerRows<false, true, false>(cam, tim, mar, fx, fy, ppx, ppy, CornerX(c, hs), CornerY(c, hs), ob[2 * c], ob[2 * c + 1], r, nullptr, Jt, nullptr);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          g[q] = fma(Jt[i][q], r[i], g[q]);
#pragma unroll
          for (int p = 0; p <= q; ++p) V[q * (q + 1) / 2 + p] = fma(Jt[i][q], Jt[i][p], V[q * (q + 1) / 2 + p]);
        }
        ss = fma(r[i], r[i], ss);
      }
    }
  }
  // the lanes by butterfly: the same bits in every lane
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int i = 0; i < 21; ++i) V[i] += __shfl_xor(V[i], off, 64);
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] += __shfl_xor(g[i], off, 64);
    ss += __shfl_xor(ss, off, 64);
  }
  double* Vd = s_v[wave];        // V + D, row-major
  double* Mx = Vd + 36;          // scratch of the inverse
  double* E = Mx + 36;           // (V + D)^-1
  // lane x < 6: the time's Jacobi scale and damping of parameter x (k_time_eliminate's rule)
  double vxx = 0.0, gx = 0.0;
#pragma unroll
  for (int q = 0; q < 6; ++q) if (lane == q) { vxx = V[q * (q + 1) / 2 + q]; gx = g[q]; }
  double sc = 1.0, xv = 0.0;
  if (lane < 6) {
    if (ip.first) { sc = ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(vxx)) : 1.0; scale_t[6 * t + lane] = sc; }
    else sc = scale_t[6 * t + lane];
    xv = params_x[a.time_full[t] + lane];
  }
  const double s2 = sc * sc;
  const double dd = lane < 6 ? fmin(fmax(s2 * vxx, ip.min_lm_diagonal), ip.max_lm_diagonal) / (ip.radius * s2) : 0.0;
  if (lane < 36) {
    const int x = lane / 6, y = lane - 6 * x, hi = x > y ? x : y, lo = x > y ? y : x;
    double vv = 0.0;
#pragma unroll
    for (int i = 0; i < 21; ++i) if (i == hi * (hi + 1) / 2 + lo) vv = V[i];
    const double dx = __shfl(dd, x, 64);
    Vd[lane] = x == y ? vv + dx : vv;
  }
  double xn2 = 0.0, gmax = 0.0;
#pragma unroll
  for (int x = 0; x < 6; ++x) {
    const double xx = __shfl(xv, x, 64);
    xn2 += xx * xx;
    gmax = fmax(gmax, fabs(g[x]));
  }
  RSBA_WAVE_LDS_SYNC();
  const bool ok = InvertSpd6Lanes(lane, Vd, Mx, E);
  double* td = tdata + (size_t)t * 48;
  if (lane < 36) td[lane] = E[lane];
  if (lane < 6) {
    double e = 0.0;
#pragma unroll
    for (int y = 0; y < 6; ++y) e += E[6 * lane + y] * g[y];
    td[36 + lane] = gx;
    td[42 + lane] = e;
  }
  if (lane == 0) { tscal[4 * (size_t)t] = ss; tscal[4 * (size_t)t + 1] = xn2; tscal[4 * (size_t)t + 2] = ok ? 0.0 : 1.0; tscal[4 * (size_t)t + 3] = gmax; }
}

__global__ void __launch_bounds__(256) k_mc_cross(SplitArgs a) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.nx_threads) return;
  // a long item on TWO neighbouring lanes — even and odd entries, the sums meet by one lane exchange (even + odd: a fixed order): the items are
  // fewer than the chip's SIMDs, the kernel lasts as long as its longest one (75 -> 43 us).  (The same for k_mc_slot_products' lists: 92 -> 101 us —
  // that kernel is two rounds of wavefronts deep and balanced by the dispatcher already.)
  const int code = a.x_order[g], it = code >> 2, two = (code >> 1) & 1, part = code & 1;
  const PoseC cam = LoadPose<true>(a.posec, a.col_full[a.xi_cc[it]] / 6);
  const PoseC mar = LoadPose<true>(a.posec, a.col_full[a.xi_cm[it]] / 6);
  double X[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) X[i] = 0.0;
  const double hs = a.half_side;
  // (the time's pose, the corners and the intrinsics one entry ahead, the entry's record two: see SlotProducts)
  struct Fetched { PoseC tim; double ob[8], in[4]; };
  auto fetch = [&](const int4 r, Fetched& f) {
    f.tim = LoadPose<false>(a.posec, r.y);
    const double* ob = a.obs8 + 8 * (size_t)r.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.ob[i] = ob[i];
    const double* in = a.intr + 4 * r.z;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.in[i] = in[i];
  };
  const int step = 1 + two, e1 = a.xi_ptr[it + 1];
  int e = a.xi_ptr[it] + part;
  const int4 none = {0, 0, 0, 0};
  int4 r1 = e < e1 ? a.xi_blk[e] : none;
  Fetched cur;
  fetch(r1, cur);
  r1 = e + step < e1 ? a.xi_blk[e + step] : none;
  for (; e < e1; e += step) {
    const int4 r2 = e + 2 * step < e1 ? a.xi_blk[e + 2 * step] : none;
    Fetched nxt;
    fetch(r1, nxt);
    const PoseC& tim = cur.tim;
    const double fx = cur.in[0], fy = cur.in[1], ppx = cur.in[2], ppy = cur.in[3];
#pragma unroll RSBA_MC_UNROLL_CROSS
    for (int c = 0; c < 4; ++c) {
      const double ou = c == 0 ? cur.ob[0] : (c == 1 ? cur.ob[2] : (c == 2 ? cur.ob[4] : cur.ob[6]));
      const double ov = c == 0 ? cur.ob[1] : (c == 1 ? cur.ob[3] : (c == 2 ? cur.ob[5] : cur.ob[7]));
      double r[2], Jc[2][6], Jm[2][6];
      CornerRows<true, false, true>(cam, tim, mar, fx, fy, ppx, ppy, CornerX(c, hs), CornerY(c, hs), ou, ov, r, Jc, nullptr, Jm);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int qm = 0; qm < 6; ++qm)
#pragma unroll
          for (int qc = 0; qc < 6; ++qc) X[6 * qm + qc] = fma(Jm[i][qm], Jc[i][qc], X[6 * qm + qc]);
    }
    cur = nxt;
    r1 = r2;
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) { const double o = __shfl_xor(X[i], 1, 64); if (two) X[i] += o; }
  if (part != 0) return;
  double* out = a.xout + (size_t)it * 36;
#pragma unroll
  for (int i = 0; i < 36; ++i) out[i] = X[i];
}

struct AccArgs {
  int nr, dmax;
  const int* __restrict__ chunk_ptr;
  const int* __restrict__ slot_ptr;
  const int* __restrict__ slot_col;
  const double* __restrict__ sp;
  const double* __restrict__ tdata;
  const double* __restrict__ tscal;
  const int* __restrict__ xc_ptr;      // [G + 1] items of k_mc_cross per chunk
  const int* __restrict__ xi_cc;
  const int* __restrict__ xi_cm;
  const double* __restrict__ xout;
  double* __restrict__ part;
};
#define RSBA_ACC_PF 11   // record doubles a thread fetches ahead: 170 slots x 64 / 1024 threads

__host__ __device__ inline size_t AccLdsBytes(int dmax, size_t s_doubles) {
  const int smax = dmax / 6;
  return (size_t)(2 * smax * RSBA_SP_LDS + 6 * dmax + 96 + s_doubles) * sizeof(double) + (size_t)2 * ((smax + 2) & ~1) * sizeof(int);
}

template <bool kLdsS>
__global__ void __launch_bounds__(RSBA_MT_THREADS) k_mc_accumulate(AccArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nr = a.nr, dmax = a.dmax, smax = dmax / 6;
  const PartLayout RL{nr};
  double* Rec = lds;                                      // [2][smax][RSBA_SP_LDS]: the records of a time's slots, two times by parity
  double* Y = Rec + 2 * (size_t)smax * RSBA_SP_LDS;        // [6][d]
  double* Ev = Y + 6 * dmax;                              // [2][48]: E | g_t | E g_t
  int* scolb = (int*)(Ev + 96);                           // [2][smax rounded]
  const int sround = (smax + 2) & ~1;
  double* Sl = (double*)(scolb + 2 * sround);
  double* P = a.part + (size_t)blockIdx.x * RL.size();
  double* Sacc = kLdsS ? Sl : P + RL.S();
  const size_t nacc = RL.packed() + 3 * (size_t)nr;
  for (size_t e = tid; e < nacc; e += RSBA_MT_THREADS) Sacc[e] = 0.0;
  for (size_t e = nacc + tid; e < RL.size(); e += RSBA_MT_THREADS) P[e] = 0.0;
  double cost = 0.0, xn2 = 0.0, gmax = 0.0, fail = 0.0;   // thread 0: the chunk's times in order
  __threadfence_block();
  __syncthreads();
  // the chunk's camera-marker blocks of U: one item per pair, every entry once
  {
    const int x0 = a.xc_ptr[blockIdx.x], x1 = a.xc_ptr[blockIdx.x + 1];
    for (int e = tid; e < 36 * (x1 - x0); e += RSBA_MT_THREADS) {
      const int it = x0 + e / 36, q = e - 36 * (e / 36), qm = q / 6, qc = q - 6 * qm;
      const int gr = a.xi_cm[it] + qm, gc = a.xi_cc[it] + qc;
      Sacc[(size_t)gr * (gr + 1) / 2 + gc] += a.xout[(size_t)it * 36 + q];
    }
  }
  const int t0 = a.chunk_ptr[blockIdx.x], t1 = a.chunk_ptr[blockIdx.x + 1];
  double pf[RSBA_ACC_PF], pe = 0.0, ps[4] = {0.0, 0.0, 0.0, 0.0};
  int pcol = 0, pns = 0;
  auto issue = [&](int t) {
    const int s0 = a.slot_ptr[t];
    pns = a.slot_ptr[t + 1] - s0;
    const double* src = a.sp + (size_t)s0 * RSBA_SP_STRIDE;
#pragma unroll
    for (int u = 0; u < RSBA_ACC_PF; ++u) { const int e = tid + RSBA_MT_THREADS * u; pf[u] = e < pns * RSBA_SP_STRIDE ? src[e] : 0.0; }
    if (tid < 48) pe = a.tdata[(size_t)t * 48 + tid];
    if (tid < pns) pcol = a.slot_col[s0 + tid];
    if (tid == 0) { ps[0] = a.tscal[4 * (size_t)t]; ps[1] = a.tscal[4 * (size_t)t + 1]; ps[2] = a.tscal[4 * (size_t)t + 2]; ps[3] = a.tscal[4 * (size_t)t + 3]; }
  };
  auto commit = [&](int par) {
    double* R = Rec + (size_t)par * smax * RSBA_SP_LDS;
#pragma unroll
    for (int u = 0; u < RSBA_ACC_PF; ++u) { const int e = tid + RSBA_MT_THREADS * u; if (e < pns * RSBA_SP_STRIDE) R[(e >> 6) * RSBA_SP_LDS + (e & 63)] = pf[u]; }
    if (tid < 48) Ev[48 * par + tid] = pe;
    if (tid < pns) scolb[par * sround + tid] = pcol;
    if (tid == 0) { cost += ps[0]; xn2 += ps[1]; fail += ps[2]; gmax = fmax(gmax, ps[3]); }
  };
  if (t0 < t1) { issue(t0); commit(0); }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int par = (t - t0) & 1;
    const int nslot = a.slot_ptr[t + 1] - a.slot_ptr[t], d = 6 * nslot;
    const double* R = Rec + (size_t)par * smax * RSBA_SP_LDS;
    const double* E = Ev + 48 * par;
    const int* scol = scolb + par * sround;
    if (t + 1 < t1) issue(t + 1);
    // Y = E W
    for (int e = tid; e < 6 * d; e += RSBA_MT_THREADS) {
      const int x = e / d, col = e - x * d, s = col / 6, q = col - 6 * s;
      double sum = 0.0;
#pragma unroll
      for (int y = 0; y < 6; ++y) sum += E[6 * x + y] * R[s * RSBA_SP_LDS + 6 * y + q];
      Y[e] = sum;
    }
    __syncthreads();
    // U_ss - W'Y into the partial system: an item = a row of a 6 x 6 block (rs, cs <= rs), lower triangle only
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
        for (int x = 0; x < 6; ++x) wr[x] = R[rs * RSBA_SP_LDS + 6 * x + rq];
        const int gr = scol[rs] + rq;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          if (rs == cs && c > rq) continue;
          double wy = 0.0;
#pragma unroll
          for (int x = 0; x < 6; ++x) wy += wr[x] * Y[x * d + 6 * cs + c];
          double v = Sacc[(size_t)gr * (gr + 1) / 2 + scol[cs] + c];
          if (rs == cs) {
            const double u = R[rs * RSBA_SP_LDS + 42 + rq * (rq + 1) / 2 + c];
            v += u;
            if (c == rq) Sacc[RL.diagU() + gr] += u;
          }
          Sacc[(size_t)gr * (gr + 1) / 2 + scol[cs] + c] = v - wy;
        }
      }
    }
    for (int e = tid; e < d; e += RSBA_MT_THREADS) {
      const int s = e / 6, q = e - 6 * s, gcol = scol[s] + q;
      double c = 0.0;
#pragma unroll
      for (int x = 0; x < 6; ++x) c += R[s * RSBA_SP_LDS + 6 * x + q] * E[42 + x];
      Sacc[RL.gc() + gcol] += R[s * RSBA_SP_LDS + 36 + q];
      Sacc[RL.corr() + gcol] -= c;
    }
    if (t + 1 < t1) commit(1 - par);
    __threadfence_block();
    __syncthreads();
  }
  if (kLdsS) for (size_t e = tid; e < nacc; e += RSBA_MT_THREADS) P[e] = Sl[e];
  if (tid == 0) { P[RL.scal() + 0] = cost; P[RL.scal() + 1] = xn2; P[RL.scal() + 2] = fail; P[RL.scal() + 3] = gmax; }
}

// k_mc_accumulate for reduced systems of up to 240 columns: W'Y on the matrix cores, the chunk's sum of it in the wavefronts' registers.
// The VALU form above reads 10 LDS words per entry of a time's block of W'Y and adds every entry into the chunk's sum in LDS (6 us a
// time at 22 slots: LDS-bound).  Here a time's W goes into a dense 8 x nrp strip in GLOBAL reduced columns (absent slots: zeros; row 6:
// g_s, which meets a zero row of Y; row 7: zeros), Y = E W likewise, and every 16 x 16 tile of the lower triangle belongs to one
// wavefront for the whole chunk: two v_mfma_f64_16x16x4_f64 per tile and time, operands straight from the strips, the sum never leaves
// the accumulators.  U_ss, g_s and -W'E g_t have their own small sums in LDS (added by the owner of a COLUMN, time after time: the
// slot that holds a column changes from time to time); at the end  S = U_ss + U_cm - sum W'Y  is put together in the partial system.
// kTB times per step, two barriers a step: [Y = E W, the small sums, zero the other strips] | [MFMA, the next step's records into the
// other strips] — the records' trip from memory (~2 us, the whole of a one-time step) is paid once per kTB times.
typedef double d4_acc __attribute__((ext_vector_type(4)));
__host__ __device__ inline int AccMfmaTiles(int nr) { const int nt = (nr + 15) / 16; return nt * (nt + 1) / 2; }
__host__ __device__ inline size_t AccMfmaLdsBytes(int nr, int tb) {
  const int nrp = ((nr + 15) / 16) * 16;
  return (size_t)(3 * tb * 8 * nrp + 2 * tb * 48 + 2 * tb * 4 + 2 * tb * (nr / 6) * 21 + 2 * nr + (nr / 6) * 21) * sizeof(double);
}

template <int kTiles /* tiles a wavefront owns at most */, int kTB /* times per step */>
__global__ void __launch_bounds__(RSBA_MT_THREADS) k_mc_accumulate_mfma(AccArgs a) {
  constexpr int kPf = 3;   // record doubles of one time a thread fetches ahead: at most 40 slots (240 columns) x 64 / 1024 threads
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nr = a.nr;
  const int nt = (nr + 15) / 16, nrp = 16 * nt, ntiles = nt * (nt + 1) / 2, strip = 8 * nrp, nub = (nr / 6) * 21;
  const PartLayout RL{nr};
  double* Wg = lds;                         // [2][kTB][8][nrp]
  double* Yg = Wg + 2 * kTB * strip;        // [kTB][8][nrp]
  double* Ev = Yg + kTB * strip;            // [2][kTB][48]: E | g_t | E g_t
  double* Ts = Ev + 2 * kTB * 48;           // [2][kTB][4]: the times' scalars
  double* Ug = Ts + 2 * kTB * 4;            // [2][kTB][nr / 6][21]: U_ss by reduced block
  double* gacc = Ug + 2 * kTB * nub;        // [nr] sum of g_s
  double* cacc = gacc + nr;                 // [nr] -sum W'E g_t
  double* Uacc = cacc + nr;                 // [nr / 6][21] sum of U_ss
  double* P = a.part + (size_t)blockIdx.x * RL.size();
  for (int e = tid; e < 3 * kTB * strip; e += RSBA_MT_THREADS) Wg[e] = 0.0;   // (rows 6, 7 of Y and row 7 of every strip stay zero)
  for (int e = tid; e < 2 * kTB * nub + 2 * nr + nub; e += RSBA_MT_THREADS) Ug[e] = 0.0;
  double cost = 0.0, xn2 = 0.0, gmax = 0.0, fail = 0.0;
  // the wavefront's tiles: t = wave + 16 u -> (ti >= tj)
  int ti[kTiles], tj[kTiles];
  d4_acc acc[kTiles];
#pragma unroll
  for (int u = 0; u < kTiles; ++u) {
    const int tl = wave + 16 * u;
    int r = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
    while (r * (r + 1) / 2 > tl) --r;
    while ((r + 1) * (r + 2) / 2 <= tl) ++r;
    ti[u] = __builtin_amdgcn_readfirstlane(tl < ntiles ? r : -1); tj[u] = __builtin_amdgcn_readfirstlane(tl - r * (r + 1) / 2);   // (wave-uniform: scalar registers)
    acc[u] = d4_acc{0.0, 0.0, 0.0, 0.0};
  }
  const int t0 = a.chunk_ptr[blockIdx.x], t1 = a.chunk_ptr[blockIdx.x + 1], nstep = (t1 - t0 + kTB - 1) / kTB;
  double pf[kTB][kPf], pe = 0.0, psv = 0.0;
  int pcs[kTB][kPf], pns[kTB];
  auto issue = [&](int b) {
#pragma unroll
    for (int j = 0; j < kTB; ++j) {
      const int t = t0 + b * kTB + j;
      const int s0 = t < t1 ? a.slot_ptr[t] : 0;
      pns[j] = t < t1 ? a.slot_ptr[t + 1] - s0 : 0;
      const double* src = a.sp + (size_t)s0 * RSBA_SP_STRIDE;
#pragma unroll
      for (int u = 0; u < kPf; ++u) {
        const int e = tid + RSBA_MT_THREADS * u;
        const bool in = e < pns[j] * RSBA_SP_STRIDE;
        pf[j][u] = in ? src[e] : 0.0;
        pcs[j][u] = in ? a.slot_col[s0 + (e >> 6)] : 0;
      }
    }
    if (tid < 48 * kTB) { const int j = tid / 48, t = t0 + b * kTB + j; pe = t < t1 ? a.tdata[(size_t)t * 48 + (tid - 48 * j)] : 0.0; }
    if (tid < 4 * kTB) { const int t = t0 + b * kTB + (tid >> 2); psv = t < t1 ? a.tscal[4 * (size_t)t + (tid & 3)] : 0.0; }
  };
  // the fetched records into the strips of `par` (zeroed a phase ago): every (time, slot, entry) has one owner
  auto commit = [&](int par) {
#pragma unroll
    for (int j = 0; j < kTB; ++j) {
      double* Wp = Wg + (size_t)(par * kTB + j) * strip;
      double* Up = Ug + (size_t)(par * kTB + j) * nub;
#pragma unroll
      for (int u = 0; u < kPf; ++u) {
        const int e = tid + RSBA_MT_THREADS * u;
        if (e < pns[j] * RSBA_SP_STRIDE) {
          const int i = e & 63, col = pcs[j][u];
          if (i < 36) { const int x = i / 6; Wp[x * nrp + col + (i - 6 * x)] = pf[j][u]; }
          else if (i < 42) Wp[6 * nrp + col + i - 36] = pf[j][u];
          else if (i < 63) Up[(col / 6) * 21 + i - 42] = pf[j][u];
        }
      }
    }
    if (tid < 48 * kTB) Ev[par * kTB * 48 + tid] = pe;
    if (tid < 4 * kTB) Ts[par * kTB * 4 + tid] = psv;
  };
  __syncthreads();
  if (nstep > 0) { issue(0); commit(0); }
  __syncthreads();
  const int mi = lane & 15, mk = lane >> 4;
  int yx[2], ycol[2];   // this thread's entries of Y = E W: 6 x nrp <= 1440 entries, two a thread at most
#pragma unroll
  for (int u = 0; u < 2; ++u) { const int e = tid + RSBA_MT_THREADS * u; yx[u] = e < 6 * nrp ? e / nrp : -1; ycol[u] = e - max(yx[u], 0) * nrp; }
  for (int b = 0; b < nstep; ++b) {
    const int par = b & 1;
    const double* Wb = Wg + (size_t)par * kTB * strip;
    double* Wo = Wg + (size_t)(1 - par) * kTB * strip;
    const double* Eb = Ev + par * kTB * 48;
    if (b + 1 < nstep) issue(b + 1);
    // (a thread's entries (x, column) of Y are the same in every step: found once, in front of the loop — two integer divisions per
    //  entry and step were a third of this phase's instructions)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (yx[u] >= 0) {
#pragma unroll
        for (int j = 0; j < kTB; ++j) {
          const double* Wp = Wb + (size_t)j * strip;
          const double* E = Eb + 48 * j + 6 * yx[u];
          double sum = 0.0;
#pragma unroll
          for (int y = 0; y < 6; ++y) sum += E[y] * Wp[y * nrp + ycol[u]];
          Yg[(size_t)j * strip + yx[u] * nrp + ycol[u]] = sum;
        }
      }
    }
    for (int e = tid; e < kTB * strip; e += RSBA_MT_THREADS) Wo[e] = 0.0;   // (whole strips: row 7 is zero already)
    for (int e = tid; e < kTB * nub; e += RSBA_MT_THREADS) Ug[(size_t)(1 - par) * kTB * nub + e] = 0.0;
    for (int col = tid; col < nr; col += RSBA_MT_THREADS) {
      double g = gacc[col], c = cacc[col];
#pragma unroll
      for (int j = 0; j < kTB; ++j) {
        const double* Wp = Wb + (size_t)j * strip;
        const double* E = Eb + 48 * j;
        double wc = 0.0;
#pragma unroll
        for (int x = 0; x < 6; ++x) wc += Wp[x * nrp + col] * E[42 + x];
        g += Wp[6 * nrp + col];
        c -= wc;
      }
      gacc[col] = g; cacc[col] = c;
    }
    for (int e = tid; e < nub; e += RSBA_MT_THREADS) {
      double u = Uacc[e];
#pragma unroll
      for (int j = 0; j < kTB; ++j) u += Ug[(size_t)(par * kTB + j) * nub + e];
      Uacc[e] = u;
    }
    if (tid == 0) {
#pragma unroll
      for (int j = 0; j < kTB; ++j) { const double* q = Ts + (par * kTB + j) * 4; cost += q[0]; xn2 += q[1]; fail += q[2]; gmax = fmax(gmax, q[3]); }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kTiles; ++u) {
      if (ti[u] < 0) continue;
#pragma unroll
      for (int j = 0; j < kTB; ++j) {
        const double* Wp = Wb + (size_t)j * strip;
        const double* Yp = Yg + (size_t)j * strip;
        const double a0 = Wp[mk * nrp + 16 * ti[u] + mi], b0 = Yp[mk * nrp + 16 * tj[u] + mi];
        const double a1 = Wp[(4 + mk) * nrp + 16 * ti[u] + mi], b1 = Yp[(4 + mk) * nrp + 16 * tj[u] + mi];
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[u], 0, 0, 0);
      }
      if (kTiles > 3) __builtin_amdgcn_sched_barrier(0);   // (else all eight tiles' operands are fetched first: 64 registers beside the 64 accumulators)
    }
    if (b + 1 < nstep) commit(1 - par);
    __syncthreads();
  }
  // the partial system: S = -sum W'Y from the accumulators (D[row = mk + 4 tt][column = mi]), then U_ss and U_cm on top
  for (size_t e = RL.packed() + 3 * (size_t)nr + tid; e < RL.size(); e += RSBA_MT_THREADS) P[e] = 0.0;
#pragma unroll
  for (int u = 0; u < kTiles; ++u) {
    if (ti[u] < 0) continue;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int gr = 16 * ti[u] + mk + 4 * tt, gc = 16 * tj[u] + mi;
      if (gr < nr && gc <= gr) P[RL.S() + (size_t)gr * (gr + 1) / 2 + gc] = -acc[u][tt];
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int e = tid; e < nub; e += RSBA_MT_THREADS) {
    const int bq = e / 21, tri = e - 21 * bq;
    int rq = (int)((sqrtf(8.0f * (float)tri + 1.0f) - 1.0f) * 0.5f);
    while (rq * (rq + 1) / 2 > tri) --rq;
    while ((rq + 1) * (rq + 2) / 2 <= tri) ++rq;
    const int cq = tri - rq * (rq + 1) / 2, gr = 6 * bq + rq, gc = 6 * bq + cq;
    P[RL.S() + (size_t)gr * (gr + 1) / 2 + gc] += Uacc[e];
    if (rq == cq) P[RL.diagU() + gr] = Uacc[e];
  }
  for (int e = tid; e < nr; e += RSBA_MT_THREADS) { P[RL.gc() + e] = gacc[e]; P[RL.corr() + e] = cacc[e]; }
  {
    const int x0 = a.xc_ptr[blockIdx.x], x1 = a.xc_ptr[blockIdx.x + 1];
    for (int e = tid; e < 36 * (x1 - x0); e += RSBA_MT_THREADS) {
      const int it = x0 + e / 36, q = e - 36 * (e / 36), qm = q / 6, qc = q - 6 * qm;
      const int gr = a.xi_cm[it] + qm, gc = a.xi_cc[it] + qc;
      P[RL.S() + (size_t)gr * (gr + 1) / 2 + gc] += a.xout[(size_t)it * 36 + q];
    }
  }
  if (tid == 0) { P[RL.scal() + 0] = cost; P[RL.scal() + 1] = xn2; P[RL.scal() + 2] = fail; P[RL.scal() + 3] = gmax; }
}

// The back-substitution of the time blocks, split the same way (k_time_backsub_wg: a workgroup per time, three barriers, 128 us).
//   k_mc_time_step   eight lanes per time: W_t delta_r = sum over the time's slots of W_s delta_s from the slots' RECORDS (no Jacobian),
//                    delta_t = -E (g_t + W_t delta_r), the candidate pose and its rotation matrix (-> posec_c), |delta_t|^2, |x_t + delta_t|^2
//   k_mc_candidate   thread per residual block: its rows once more for the model cost change -(J d).(r + J d / 2), the candidate's corners
//                    through the candidate's rotation matrices; a workgroup's sums in a fixed order -> one entry of bp_time
// bp_time: [T] (|delta_t|^2, |x_t + delta_t|^2, 0, 0), then one entry per workgroup of k_mc_candidate (0, 0, model cost change, sum r_c^2).
__global__ void __launch_bounds__(256) k_mc_time_step(int T, const int* __restrict__ slot_ptr, const int* __restrict__ slot_col, const int* __restrict__ time_full,
                                                      const double* __restrict__ sp, const double* __restrict__ tdata, const double* __restrict__ delta_r,
                                                      const double* __restrict__ params_x, double* __restrict__ params_c, double* __restrict__ delta_t,
                                                      double* __restrict__ posec_c, double* __restrict__ bp_time) {
  const int g = blockIdx.x * 256 + threadIdx.x, t = g >> 3, l = g & 7;
  if (t >= T) return;   // (whole groups of eight lanes)
  double h[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int s0 = slot_ptr[t], s1 = slot_ptr[t + 1];
  for (int S = s0 + l; S < s1; S += 8) {
    const double* rec = sp + (size_t)S * RSBA_SP_STRIDE;
    const double* d = delta_r + slot_col[S];
    double dd[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) dd[q] = d[q];
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int q = 0; q < 6; ++q) h[x] = fma(rec[6 * x + q], dd[q], h[x]);
  }
#pragma unroll
  for (int x = 0; x < 6; ++x) {
    h[x] += __shfl_xor(h[x], 1, 8); h[x] += __shfl_xor(h[x], 2, 8); h[x] += __shfl_xor(h[x], 4, 8);
  }
  if (l != 0) return;
  const double* td = tdata + (size_t)t * 48;
  const int tf = time_full[t];
  double hh[6], tc[6], d2 = 0.0, xc2 = 0.0;
#pragma unroll
  for (int x = 0; x < 6; ++x) hh[x] = h[x] + td[36 + x];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double sum = 0.0;
#pragma unroll
    for (int y = 0; y < 6; ++y) sum += td[6 * a + y] * hh[y];
    const double d = -sum;
    tc[a] = params_x[tf + a] + d;
    delta_t[6 * t + a] = d;
    params_c[tf + a] = tc[a];
    d2 += d * d; xc2 += tc[a] * tc[a];
  }
  const double zero4[4] = {0.0, 0.0, 0.0, 0.0};
  double cct[CC_STRIDE];
  CameraConstants(tc, zero4, cct);
  double* pc = posec_c + (size_t)(tf / 6) * CC_STRIDE;
#pragma unroll
  for (int q = 0; q < 9; ++q) pc[CC_R + q] = cct[CC_R + q];
#pragma unroll
  for (int q = 0; q < 3; ++q) pc[CC_T + q] = cct[CC_T + q];
  bp_time[4 * (size_t)t] = d2; bp_time[4 * (size_t)t + 1] = xc2; bp_time[4 * (size_t)t + 2] = 0.0; bp_time[4 * (size_t)t + 3] = 0.0;
}

// kPart 0: the model cost change (the block's rows at x: 340 registers, one wavefront a SIMD); 1: the candidate's residuals (rotation
// matrices and translations only: four wavefronts a SIMD) — two launches side by side instead of one kernel with the registers of both.
template <int kPart>
__global__ void __launch_bounds__(256) k_mc_candidate(int N, int T, const TimeSlots* __restrict__ ts, const MarkerObs* __restrict__ mo, const double* __restrict__ obs8,
                                                      const double* __restrict__ intr, double half_side, const double* __restrict__ posec,
                                                      const double* __restrict__ posec_c, const double* __restrict__ delta_r, const double* __restrict__ delta_t,
                                                      const int* __restrict__ blk_time, double* __restrict__ bp_time) {
  __shared__ double s_w[4];
  const int k = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double sum = 0.0;
  if (k < N) {
    const TimeSlots s = ts[k];
    const MarkerObs o = mo[k];
    const double* in = intr + 4 * s.camera;
    const double fx = in[0], fy = in[1], ppx = in[2], ppy = in[3];
    const double* ob = obs8 + 8 * (size_t)k;
    if (kPart == 0) {
      const PoseC cam = LoadPose<true>(posec, o.full_cam >= 0 ? o.full_cam / 6 : -1);
      const PoseC tim = LoadPose<true>(posec, o.full_time / 6);
      const PoseC mar = LoadPose<true>(posec, o.full_marker >= 0 ? o.full_marker / 6 : -1);
      double dl[18];
      const int t = blk_time[k];
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        dl[x] = s.col_cam >= 0 ? delta_r[s.col_cam + x] : 0.0;
        dl[6 + x] = delta_t[6 * t + x];
        dl[12 + x] = s.col_marker >= 0 ? delta_r[s.col_marker + x] : 0.0;
      }
#pragma unroll RSBA_MC_UNROLL_CAND
      for (int c = 0; c < 4; ++c) {
        const double cx = CornerX(c, half_side), cy = CornerY(c, half_side), u = ob[2 * c], v = ob[2 * c + 1];
        double r[2], Jc[2][6], Jt[2][6], Jm[2][6];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 6; ++q) Jc[i][q] = 0.0;   // (an absent camera transform: CornerRows leaves the block alone)
        CornerRows<true, true, true>(cam, tim, mar, fx, fy, ppx, ppy, cx, cy, u, v, r, Jc, Jt, Jm);
        // (an absent marker transform: its block is formed from the stand-in pose's constants and meets dl[12..17] = 0)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double m = 0.0;
#pragma unroll
          for (int x = 0; x < 6; ++x) m += Jc[i][x] * dl[x];
#pragma unroll
          for (int x = 0; x < 6; ++x) m += Jt[i][x] * dl[6 + x];
#pragma unroll
          for (int x = 0; x < 6; ++x) m += Jm[i][x] * dl[12 + x];
          sum -= m * (r[i] + 0.5 * m);
        }
      }
    } else {
      // the candidate's corners through the candidate's transforms (rotation matrices: k_pose_constants_reduced, k_mc_time_step)
      const PoseC ccam = LoadPose<false>(posec_c, o.full_cam >= 0 ? o.full_cam / 6 : -1);
      const PoseC ctim = LoadPose<false>(posec_c, o.full_time / 6);
      const PoseC cmar = LoadPose<false>(posec_c, o.full_marker >= 0 ? o.full_marker / 6 : -1);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double u = ob[2 * c], v = ob[2 * c + 1];
        double pt[3] = {CornerX(c, half_side), CornerY(c, half_side), 0.0};
        auto apply = [&](const PoseC& p) {
          const double a0 = p.R[0] * pt[0] + p.R[1] * pt[1] + p.R[2] * pt[2], a1 = p.R[3] * pt[0] + p.R[4] * pt[1] + p.R[5] * pt[2], a2 = p.R[6] * pt[0] + p.R[7] * pt[1] + p.R[8] * pt[2];
          pt[0] = a0 + p.T[0]; pt[1] = a1 + p.T[1]; pt[2] = a2 + p.T[2];
        };
        if (cmar.on) apply(cmar);
        apply(ctim);
        if (ccam.on) apply(ccam);
        const double r0 = fx * pt[0] / pt[2] + ppx - u, r1 = fy * pt[1] / pt[2] + ppy - v;
        sum += r0 * r0 + r1 * r1;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (lane == 0) s_w[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double* out = bp_time + 4 * ((size_t)T + blockIdx.x);
    const double tot = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
    if (kPart == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = tot; }
    else out[3] = tot;
  }
}

}  // namespace rsba
