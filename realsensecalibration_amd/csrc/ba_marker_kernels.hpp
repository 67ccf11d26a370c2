// Marker-chain model on the device (BASELINE config 1; the model Main_Calibration actually solves).
//
// Residual: 4 marker corners (-h,+h) (+h,+h) (+h,-h) (-h,-h) pushed through up to three rigid
// transforms marker -> base marker -> base camera -> target camera, then the pinhole projection
// (the four functors of /root/reference/Main_Calibration/bundle_adjustment.h:56-343 and their wiring in
// bundle_adjustment_manager.cpp:21-88; Test2 variant: Test2_BundleAdjustment/main.cpp:64-96).
// Derivatives are taken with forward-mode dual numbers on the GPU, i.e. the same arithmetic Ceres'
// AutoDiffCostFunction<.,8,6,6,6> performs, including the first-order branch of AngleAxisRotatePoint.
//
// The problem is tiny (68 residual blocks, 114 parameters on the committed data) and does not shard:
// one kernel evaluates the blocks (one thread each), one workgroup assembles the dense normal
// equations in a fixed order (bitwise reproducible), damps, factors (the same blocked Cholesky the
// reduced camera system uses) and forms the candidate; a third evaluates the candidate.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ba_point_kernels.hpp"
#include "ba_problem.hpp"

namespace rsba {

template <int N>
struct DJet {
  double a;
  double v[N];
};
template <int N> __device__ __forceinline__ DJet<N> JConst(double x) { DJet<N> r; r.a = x; for (int i = 0; i < N; ++i) r.v[i] = 0.0; return r; }
template <int N> __device__ __forceinline__ DJet<N> JVar(double x, int k) { DJet<N> r = JConst<N>(x); r.v[k] = 1.0; return r; }
template <int N> __device__ __forceinline__ DJet<N> operator+(const DJet<N>& f, const DJet<N>& g) { DJet<N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
template <int N> __device__ __forceinline__ DJet<N> operator-(const DJet<N>& f, const DJet<N>& g) { DJet<N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
template <int N> __device__ __forceinline__ DJet<N> operator*(const DJet<N>& f, const DJet<N>& g) { DJet<N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
template <int N> __device__ __forceinline__ DJet<N> operator/(const DJet<N>& f, const DJet<N>& g) {
  DJet<N> r; const double gi = 1.0 / g.a; const double fg = f.a * gi; r.a = fg;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - fg * g.v[i]) * gi; return r; }
template <int N> __device__ __forceinline__ DJet<N> JSqrt(const DJet<N>& f) { DJet<N> r; const double t = sqrt(f.a); const double s = 1.0 / (2.0 * t); r.a = t; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }
template <int N> __device__ __forceinline__ DJet<N> JCos(const DJet<N>& f) { DJet<N> r; r.a = cos(f.a); const double s = -sin(f.a); for (int i = 0; i < N; ++i) r.v[i] = s * f.v[i]; return r; }
template <int N> __device__ __forceinline__ DJet<N> JSin(const DJet<N>& f) { DJet<N> r; r.a = sin(f.a); const double c = cos(f.a); for (int i = 0; i < N; ++i) r.v[i] = c * f.v[i]; return r; }
__device__ __forceinline__ double JConstD(double x) { return x; }

// Rotate p by the angle-axis w (in place), both branches; T = double or DJet<N>.
template <int N>
__device__ void RotateJet(const DJet<N> w[3], DJet<N> p[3]) {
  const DJet<N> t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (t2.a > DBL_EPSILON) {
    const DJet<N> th = JSqrt(t2), c = JCos(th), s = JSin(th), it = JConst<N>(1.0) / th;
    const DJet<N> k0 = w[0] * it, k1 = w[1] * it, k2 = w[2] * it;
    const DJet<N> x0 = k1 * p[2] - k2 * p[1], x1 = k2 * p[0] - k0 * p[2], x2 = k0 * p[1] - k1 * p[0];
    const DJet<N> tmp = (k0 * p[0] + k1 * p[1] + k2 * p[2]) * (JConst<N>(1.0) - c);
    const DJet<N> r0 = p[0] * c + x0 * s + k0 * tmp, r1 = p[1] * c + x1 * s + k1 * tmp, r2 = p[2] * c + x2 * s + k2 * tmp;
    p[0] = r0; p[1] = r1; p[2] = r2;
  } else {
    const DJet<N> r0 = p[0] + (w[1] * p[2] - w[2] * p[1]), r1 = p[1] + (w[2] * p[0] - w[0] * p[2]), r2 = p[2] + (w[0] * p[1] - w[1] * p[0]);
    p[0] = r0; p[1] = r1; p[2] = r2;
  }
}
__device__ inline void RotateD(const double w[3], double p[3]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double r0, r1, r2;
  if (t2 > DBL_EPSILON) {
    const double th = sqrt(t2), c = cos(th), s = sin(th), it = 1.0 / th;
    const double k0 = w[0] * it, k1 = w[1] * it, k2 = w[2] * it;
    const double x0 = k1 * p[2] - k2 * p[1], x1 = k2 * p[0] - k0 * p[2], x2 = k0 * p[1] - k1 * p[0];
    const double tmp = (k0 * p[0] + k1 * p[1] + k2 * p[2]) * (1.0 - c);
    r0 = p[0] * c + x0 * s + k0 * tmp; r1 = p[1] * c + x1 * s + k1 * tmp; r2 = p[2] * c + x2 * s + k2 * tmp;
  } else {
    r0 = p[0] + (w[1] * p[2] - w[2] * p[1]); r1 = p[1] + (w[2] * p[0] - w[0] * p[2]); r2 = p[2] + (w[0] * p[1] - w[1] * p[0]);
  }
  p[0] = r0; p[1] = r1; p[2] = r2;
}

// Per-observation wiring: offsets (in doubles) of the three blocks inside the FULL parameter array, or -1
// when the functor variant has no such block; and offsets inside the ACTIVE vector.
struct MarkerObs {
  int full_cam, full_time, full_marker;  // -1: block not part of this residual
  int act_cam, act_time, act_marker;
  int camera;                            // intrinsics index
  int pad;
};

// Evaluate residual blocks.  with_jacobian: J (8 x 18, columns camera|time|marker) and r (8) are stored.
__global__ void __launch_bounds__(64) k_marker_eval(int N, const MarkerObs* __restrict__ mo, const double* __restrict__ obs8,
                              const double* __restrict__ params, const double* __restrict__ intr, double half_side,
                              int with_jacobian, double* __restrict__ Jbuf, double* __restrict__ rbuf,
                              double* __restrict__ sumsq_per_obs) {
  // Jacobian rows leave through LDS: a thread's 36 doubles per corner would otherwise be 64 scattered 288-byte pieces per
  // store instruction (1152 B between neighbouring lanes); staged, consecutive lanes write consecutive words
  __shared__ double stage[64 * 37];
  const int i0 = blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
  const bool live = i < N;
  const int ii = live ? i : N - 1;   // lanes past the end keep going for the barriers, on the last block's data
  const MarkerObs o = mo[ii];
  const double fx = intr[4 * o.camera], fy = intr[4 * o.camera + 1], ppx = intr[4 * o.camera + 2], ppy = intr[4 * o.camera + 3];
  const double cx[4] = {-half_side, half_side, half_side, -half_side};
  const double cy[4] = {half_side, half_side, -half_side, -half_side};
  double ss = 0.0;
  if (!with_jacobian) {
    for (int k = 0; k < 4; ++k) {
      double p[3] = {cx[k], cy[k], 0.0};
      if (o.full_marker >= 0) { const double* m = params + o.full_marker; RotateD(m, p); p[0] += m[3]; p[1] += m[4]; p[2] += m[5]; }
      { const double* t = params + o.full_time; RotateD(t, p); p[0] += t[3]; p[1] += t[4]; p[2] += t[5]; }
      if (o.full_cam >= 0) { const double* c = params + o.full_cam; RotateD(c, p); p[0] += c[3]; p[1] += c[4]; p[2] += c[5]; }
      const double r0 = fx * p[0] / p[2] + ppx - obs8[8 * (size_t)ii + 2 * k];
      const double r1 = fy * p[1] / p[2] + ppy - obs8[8 * (size_t)ii + 2 * k + 1];
      ss += r0 * r0 + r1 * r1;
    }
  } else {
    typedef DJet<18> J;
    J cam[6], tim[6], mar[6];
    for (int q = 0; q < 6; ++q) {
      cam[q] = o.full_cam >= 0 ? JVar<18>(params[o.full_cam + q], q) : JConst<18>(0.0);
      tim[q] = JVar<18>(params[o.full_time + q], 6 + q);
      mar[q] = o.full_marker >= 0 ? JVar<18>(params[o.full_marker + q], 12 + q) : JConst<18>(0.0);
    }
    for (int k = 0; k < 4; ++k) {
      J p[3] = {JConst<18>(cx[k]), JConst<18>(cy[k]), JConst<18>(0.0)};
      if (o.full_marker >= 0) { RotateJet<18>(mar, p); p[0] = p[0] + mar[3]; p[1] = p[1] + mar[4]; p[2] = p[2] + mar[5]; }
      RotateJet<18>(tim, p); p[0] = p[0] + tim[3]; p[1] = p[1] + tim[4]; p[2] = p[2] + tim[5];
      if (o.full_cam >= 0) { RotateJet<18>(cam, p); p[0] = p[0] + cam[3]; p[1] = p[1] + cam[4]; p[2] = p[2] + cam[5]; }
      const J xp = JConst<18>(fx) * p[0] / p[2] + JConst<18>(ppx);
      const J yp = JConst<18>(fy) * p[1] / p[2] + JConst<18>(ppy);
      const double r0 = xp.a - obs8[8 * (size_t)ii + 2 * k], r1 = yp.a - obs8[8 * (size_t)ii + 2 * k + 1];
      if (live) { rbuf[8 * (size_t)i + 2 * k] = r0; rbuf[8 * (size_t)i + 2 * k + 1] = r1; }
      for (int q = 0; q < 18; ++q) { stage[threadIdx.x * 37 + q] = xp.v[q]; stage[threadIdx.x * 37 + 18 + q] = yp.v[q]; }
      __syncthreads();
      {
        const int nlive = min(64, N - i0);
        for (int e = threadIdx.x; e < nlive * 36; e += 64) { const int t = e / 36, q = e - 36 * t; Jbuf[(size_t)(i0 + t) * 144 + 36 * k + q] = stage[t * 37 + q]; }
      }
      __syncthreads();
      ss += r0 * r0 + r1 * r1;
    }
  }
  if (live) sumsq_per_obs[i] = ss;
}

// One workgroup: normal equations in a fixed order, Jacobi scale, LM damping, Cholesky, step, candidate.
//   A : (n+1) x n work matrix; H/g accumulated by thread-per-entry loops over the observations in order.
template <int kThreads>
__global__ void __launch_bounds__(kThreads)
k_marker_system(int N, int n, const MarkerObs* __restrict__ mo, const double* __restrict__ Jbuf, const double* __restrict__ rbuf,
                const double* __restrict__ sumsq_per_obs, double* __restrict__ A, double* __restrict__ scale,
                double* __restrict__ grad, const int* __restrict__ act_to_full, const double* __restrict__ params_x,
                double* __restrict__ params_c, double* __restrict__ delta_act, double* __restrict__ res, IterParams ip) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  __shared__ int s_ok;
  // per-observation column map: active offset of each of the 18 local columns (or -1)
  auto col_of = [&](const MarkerObs& o, int q) { const int b = q / 6; const int base = b == 0 ? o.act_cam : (b == 1 ? o.act_time : o.act_marker); return base < 0 ? -1 : base + (q - 6 * b); };
  // H (unscaled) lower+upper, and g: each thread owns entries and walks the observations in order
  for (size_t e = tid; e < (size_t)n * n; e += nt) A[e] = 0.0;
  for (int i = tid; i < n; i += nt) { A[(size_t)n * n + i] = 0.0; }
  __syncthreads();
  for (int ob = 0; ob < N; ++ob) {
    const MarkerObs o = mo[ob];
    // 18 x 18 local block, 324 entries over the threads
    for (int e = tid; e < 324; e += nt) {
      const int a = e / 18, b = e - 18 * a;
      const int ca = col_of(o, a), cb = col_of(o, b);
      if (ca < 0 || cb < 0) continue;
      double s = 0.0;
      for (int r = 0; r < 8; ++r) s += Jbuf[(size_t)(8 * ob + r) * 18 + a] * Jbuf[(size_t)(8 * ob + r) * 18 + b];
      A[(size_t)ca * n + cb] += s;
    }
    for (int a = tid; a < 18; a += nt) {
      const int ca = col_of(o, a);
      if (ca < 0) continue;
      double s = 0.0;
      for (int r = 0; r < 8; ++r) s += Jbuf[(size_t)(8 * ob + r) * 18 + a] * rbuf[8 * (size_t)ob + r];
      A[(size_t)n * n + ca] += s;
    }
    __threadfence_block();
    __syncthreads();
  }
  // gradient (unscaled), scale (iteration 0), damping
  for (int i = tid; i < n; i += nt) {
    grad[i] = A[(size_t)n * n + i];
    if (ip.first) scale[i] = ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(A[(size_t)i * n + i])) : 1.0;
  }
  __threadfence_block();
  __syncthreads();
  for (size_t e = tid; e < (size_t)n * n; e += nt) {
    const int i = (int)(e / n), j = (int)(e - (size_t)i * n);
    double v = A[e] * (scale[i] * scale[j]);
    if (i == j) v += fmin(fmax(v, ip.min_lm_diagonal), ip.max_lm_diagonal) / ip.radius;
    A[e] = v;
  }
  for (int i = tid; i < n; i += nt) A[(size_t)n * n + i] *= scale[i];
  __threadfence_block();
  __syncthreads();
  double* ysol = A + (size_t)n * n;
  if (kThreads == 512) CholeskySolvePanelLDS(n, A, ysol, &s_ok, lds, PanelSource{nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr, nullptr, 0});
  else CholeskySolveBlocked(n, A, ysol, &s_ok, lds);
  __syncthreads();
  // step, candidate, norms, cost at x
  double* scr = lds;
  double d2 = 0, x2 = 0, xc2 = 0, gm = 0;
  for (int i = tid; i < n; i += nt) {
    const double d = -scale[i] * ysol[i];
    delta_act[i] = d;
    const double x = params_x[act_to_full[i]], xc = x + d;
    params_c[act_to_full[i]] = xc;
    d2 += d * d; x2 += x * x; xc2 += xc * xc; gm = fmax(gm, fabs(grad[i]));
  }
  double cs = 0;
  for (int i = tid; i < N; i += nt) cs += sumsq_per_obs[i];
  scr[tid] = d2; scr[nt + tid] = x2; scr[2 * nt + tid] = xc2; scr[3 * nt + tid] = gm; scr[4 * nt + tid] = cs;
  __syncthreads();
  for (int off = nt / 2; off > 0; off >>= 1) {
    if (tid < off) {
      scr[tid] += scr[tid + off]; scr[nt + tid] += scr[nt + tid + off]; scr[2 * nt + tid] += scr[2 * nt + tid + off];
      scr[3 * nt + tid] = fmax(scr[3 * nt + tid], scr[3 * nt + tid + off]); scr[4 * nt + tid] += scr[4 * nt + tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    res[RES_COST_X] = 0.5 * scr[4 * nt]; res[RES_GMAX] = scr[3 * nt]; res[RES_XNORM2] = scr[nt];
    res[RES_CHOL_OK] = s_ok ? 1.0 : 0.0; res[RES_STEP2] = scr[0]; res[RES_XCNORM2] = scr[2 * nt]; res[RES_POINT_FAIL] = 0.0;
  }
}

// Model cost change -(J d).(r + J d / 2) and candidate cost, one workgroup, fixed order.
__global__ void __launch_bounds__(256)
k_marker_candidate(int N, const MarkerObs* __restrict__ mo, const double* __restrict__ Jbuf, const double* __restrict__ rbuf,
                   const double* __restrict__ delta_act, const double* __restrict__ sumsq_c, double* __restrict__ res) {
  __shared__ double s[2][256];
  const int tid = threadIdx.x;
  double mcc = 0, cc = 0;
  for (int i = tid; i < N; i += blockDim.x) {
    const MarkerObs o = mo[i];
    for (int r = 0; r < 8; ++r) {
      double mr = 0.0;
      for (int q = 0; q < 18; ++q) {
        const int b = q / 6; const int base = b == 0 ? o.act_cam : (b == 1 ? o.act_time : o.act_marker);
        if (base >= 0) mr += Jbuf[(size_t)(8 * i + r) * 18 + q] * delta_act[base + (q - 6 * b)];
      }
      mcc -= mr * (rbuf[8 * (size_t)i + r] + 0.5 * mr);
    }
    cc += sumsq_c[i];
  }
  s[0][tid] = mcc; s[1][tid] = cc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (tid < off) { s[0][tid] += s[0][tid + off]; s[1][tid] += s[1][tid + off]; } __syncthreads(); }
  if (tid == 0) {
    res[RES_MCC] = s[0][0];
    double c = 0.5 * s[1][0];
    if (!(c == c) || !(fabs(c) <= DBL_MAX)) c = DBL_MAX;
    res[RES_COST_C] = c; res[RES_SUMSQ_C] = s[1][0];
  }
}

class KernelTimer;

struct MarkerDevice {
  int N = 0, n = 0, nfull = 0;
  double half_side = 0;
  MarkerObs* mo = nullptr;
  double *obs8 = nullptr, *intr = nullptr, *params[2] = {nullptr, nullptr}, *params0 = nullptr;
  double *Jbuf = nullptr, *rbuf = nullptr, *ss_x = nullptr, *ss_c = nullptr, *A = nullptr, *scale = nullptr, *grad = nullptr,
         *delta = nullptr, *res = nullptr;
  int* act_to_full = nullptr;
  int cur = 0;

  void Free() {
    void* ptrs[] = {mo, obs8, intr, params[0], params[1], params0, Jbuf, rbuf, ss_x, ss_c, A, scale, grad, delta, res, act_to_full};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    mo = nullptr;
  }
  int Upload(const rsba_problem& p) {
    N = (int)p.num_observations; nfull = (int)p.parameters.size(); half_side = p.marker_side / 2;
    const int nblocks = p.num_cameras + p.num_times + p.num_markers;
    std::vector<char> used(nblocks, 0);
    for (int i = 0; i < N; ++i) { if (p.uses_camera(i)) used[p.camera_block(i)] = 1; used[p.time_block(i)] = 1; if (p.uses_marker(i)) used[p.marker_block(i)] = 1; }
    // a constant block (rsba_problem_set_parameter_block_constant) keeps its transform in every residual that names it (full_*) and has no
    // columns (act_* = -1): it is not in the program, as Ceres has it
    for (int b = 0; b < nblocks && b < (int)p.block_constant.size(); ++b) if (p.block_constant[b]) used[b] = 0;
    std::vector<int> act(nblocks, -1), a2f;
    int na = 0;
    for (int b = 0; b < nblocks; ++b) if (used[b]) { act[b] = 6 * na++; for (int q = 0; q < 6; ++q) a2f.push_back(6 * b + q); }
    n = 6 * na;
    std::vector<MarkerObs> h(N);
    for (int i = 0; i < N; ++i) {
      MarkerObs& o = h[i];
      o.full_cam = p.uses_camera(i) ? 6 * p.camera_block(i) : -1; o.full_time = 6 * p.time_block(i);
      o.full_marker = p.uses_marker(i) ? 6 * p.marker_block(i) : -1;
      o.act_cam = p.uses_camera(i) ? act[p.camera_block(i)] : -1; o.act_time = act[p.time_block(i)];
      o.act_marker = p.uses_marker(i) ? act[p.marker_block(i)] : -1;
      o.camera = p.camera_index[i]; o.pad = 0;
    }
    auto al = [](void** q, size_t bytes) { return hipMalloc(q, std::max<size_t>(bytes, 8)) == hipSuccess; };
    if (!al((void**)&mo, N * sizeof(MarkerObs)) || !al((void**)&obs8, 8 * (size_t)N * 8) || !al((void**)&intr, p.intrinsics.size() * 8) ||
        !al((void**)&params[0], nfull * 8) || !al((void**)&params[1], nfull * 8) || !al((void**)&params0, nfull * 8) ||
        !al((void**)&Jbuf, (size_t)N * 8 * 18 * 8) || !al((void**)&rbuf, (size_t)N * 8 * 8) || !al((void**)&ss_x, N * 8) || !al((void**)&ss_c, N * 8) ||
        !al((void**)&A, (size_t)(n + 2) * n * 8) || !al((void**)&scale, n * 8) || !al((void**)&grad, n * 8) || !al((void**)&delta, n * 8) ||
        !al((void**)&res, RES_SIZE * 8) || !al((void**)&act_to_full, n * sizeof(int)))
      return RSBA_ERR_HIP;
    if (hipMemcpy(mo, h.data(), N * sizeof(MarkerObs), hipMemcpyHostToDevice) != hipSuccess) return RSBA_ERR_HIP;
    if (hipMemcpy(obs8, p.observations.data(), 8 * (size_t)N * 8, hipMemcpyHostToDevice) != hipSuccess) return RSBA_ERR_HIP;
    if (hipMemcpy(intr, p.intrinsics.data(), p.intrinsics.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return RSBA_ERR_HIP;
    if (hipMemcpy(params0, p.parameters.data(), nfull * 8, hipMemcpyHostToDevice) != hipSuccess) return RSBA_ERR_HIP;
    if (hipMemcpy(act_to_full, a2f.data(), n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return RSBA_ERR_HIP;
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
  int Step(hipStream_t st, const rsba_options& o, double radius, bool first, double* res_host, Timer& T) {
    IterParams ip;
    ip.radius = radius; ip.min_lm_diagonal = o.min_lm_diagonal; ip.max_lm_diagonal = o.max_lm_diagonal; ip.huber_delta = 0.0;
    ip.first = first ? 1 : 0; ip.jacobi_scaling = o.jacobi_scaling;
    const int x = cur, c = 1 - cur;
    // the candidate array must carry the untouched blocks too
    if (hipMemcpyAsync(params[c], params[x], nfull * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return RSBA_ERR_HIP;
    auto chk = [&](const char* what) {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { fprintf(stderr, "rsba: %s launch failed: %s\n", what, hipGetErrorString(e)); return false; }
      return true;
    };
    if (!chk("(before marker step)")) return RSBA_ERR_HIP;
    T.Begin("k_marker_eval", st);
    k_marker_eval<<<(N + 63) / 64, 64, 0, st>>>(N, mo, obs8, params[x], intr, half_side, 1, Jbuf, rbuf, ss_x);
    T.End(st);
    if (!chk("k_marker_eval")) return RSBA_ERR_HIP;
    size_t lds = (size_t)std::max(2 * RSBA_TB * (RSBA_NB + 1) + RSBA_NB * (RSBA_NB + 1), 5 * 1024) * sizeof(double);
    if (n <= RSBA_CHOL_MAXN) lds = std::max(lds, CholeskyLdsDoubles(n) * sizeof(double));
    T.Begin("k_marker_system", st);
    if (n <= RSBA_CHOL_MAXN) {
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_marker_system<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      k_marker_system<512><<<1, 512, lds, st>>>(N, n, mo, Jbuf, rbuf, ss_x, A, scale, grad, act_to_full, params[x], params[c], delta, res, ip);
    } else {
      k_marker_system<1024><<<1, 1024, lds, st>>>(N, n, mo, Jbuf, rbuf, ss_x, A, scale, grad, act_to_full, params[x], params[c], delta, res, ip);
    }
    T.End(st);
    if (!chk("k_marker_system")) return RSBA_ERR_HIP;
    T.Begin("k_marker_eval", st);
    k_marker_eval<<<(N + 63) / 64, 64, 0, st>>>(N, mo, obs8, params[c], intr, half_side, 0, nullptr, nullptr, ss_c);
    T.End(st);
    T.Begin("k_marker_candidate", st);
    k_marker_candidate<<<1, 256, 0, st>>>(N, mo, Jbuf, rbuf, delta, ss_c, res);
    T.End(st);
    if (!chk("k_marker_candidate")) return RSBA_ERR_HIP;
    if (hipMemcpyAsync(res_host, res, RES_SIZE * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return RSBA_ERR_HIP;
    { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) { fprintf(stderr, "rsba: marker-chain step failed: %s\n", hipGetErrorString(e)); return RSBA_ERR_HIP; } }
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
