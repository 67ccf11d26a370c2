// Per-observation arithmetic of the point model, shared by every HIP kernel.
//
// Replaces, on the device, what Ceres' AutoDiffCostFunction<ReprojectionError,2,6,3> computes with
// Jets for /root/reference/Test1_BundleAdjustment/bundle_adjustmenter.cpp:122-141: the residual and its
// 2x6 / 2x3 Jacobian blocks, written out analytically.
//
//   q = R(w) X,  p = q + t,  r = (fx p0/p2 + ppx - u, fy p1/p2 + ppy - v)
//   d r / d p    = Pj = [[fx/p2, 0, -fx p0/p2^2], [0, fy/p2, -fy p1/p2^2]]
//   d r / d X    = Pj R
//   d r / d t    = Pj
//   d r / d w    = Pj (-[q]x) Jl(w)         theta^2 >  DBL_EPSILON   (Jl = left Jacobian of SO(3))
//                = Pj (-[X]x)               theta^2 <= DBL_EPSILON   (AngleAxisRotatePoint's first-order
//                                            branch: the derivative AutoDiff sees is that of X + w x X)
// R, Jl and the branch flag depend only on the camera, so they are computed once per camera per
// linearisation (CameraConstants) instead of once per observation.
//
// The header also compiles as plain C++ (g++) so the CPU test-suite can check these formulas against the
// oracle's dual numbers without a GPU.
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RSBA_HD __host__ __device__ __forceinline__
#else
#define RSBA_HD inline
#endif

namespace rsba {

// Per-camera constants, 32 doubles (256 B) per camera.
enum {
  CC_R = 0,      // 9: rotation matrix, row-major (I + [w]x in the small-angle branch)
  CC_K = 9,      // 9: left Jacobian Jl(w) (identity in the small-angle branch)
  CC_T = 18,     // 3: translation
  CC_FX = 21, CC_FY = 22, CC_PPX = 23, CC_PPY = 24,
  CC_SMALL = 25, // 1.0 when theta^2 <= DBL_EPSILON
  CC_STRIDE = 32
};

RSBA_HD void CameraConstants(const double* cam6, const double* intr4, double* cc) {
  const double wx = cam6[0], wy = cam6[1], wz = cam6[2];
  const double theta2 = wx * wx + wy * wy + wz * wz;
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double c = cos(theta), s = sin(theta);
    const double it = 1.0 / theta;
    const double kx = wx * it, ky = wy * it, kz = wz * it;
    const double sh = sin(0.5 * theta);
    const double c1 = 2.0 * sh * sh;  // 1 - cos(theta) without cancellation
    cc[CC_R + 0] = c + c1 * kx * kx;      cc[CC_R + 1] = c1 * kx * ky - s * kz; cc[CC_R + 2] = c1 * kx * kz + s * ky;
    cc[CC_R + 3] = c1 * kx * ky + s * kz; cc[CC_R + 4] = c + c1 * ky * ky;      cc[CC_R + 5] = c1 * ky * kz - s * kx;
    cc[CC_R + 6] = c1 * kx * kz - s * ky; cc[CC_R + 7] = c1 * ky * kz + s * kx; cc[CC_R + 8] = c + c1 * kz * kz;
    const double a = s * it;        // sin(theta)/theta
    const double b = 1.0 - a;       // 1 - sin(theta)/theta
    const double d = c1 * it;       // (1 - cos(theta))/theta
    cc[CC_K + 0] = a + b * kx * kx;      cc[CC_K + 1] = b * kx * ky - d * kz; cc[CC_K + 2] = b * kx * kz + d * ky;
    cc[CC_K + 3] = b * kx * ky + d * kz; cc[CC_K + 4] = a + b * ky * ky;      cc[CC_K + 5] = b * ky * kz - d * kx;
    cc[CC_K + 6] = b * kx * kz - d * ky; cc[CC_K + 7] = b * ky * kz + d * kx; cc[CC_K + 8] = a + b * kz * kz;
    cc[CC_SMALL] = 0.0;
  } else {
    cc[CC_R + 0] = 1.0; cc[CC_R + 1] = -wz; cc[CC_R + 2] = wy;
    cc[CC_R + 3] = wz;  cc[CC_R + 4] = 1.0; cc[CC_R + 5] = -wx;
    cc[CC_R + 6] = -wy; cc[CC_R + 7] = wx;  cc[CC_R + 8] = 1.0;
    cc[CC_K + 0] = 1.0; cc[CC_K + 1] = 0.0; cc[CC_K + 2] = 0.0;
    cc[CC_K + 3] = 0.0; cc[CC_K + 4] = 1.0; cc[CC_K + 5] = 0.0;
    cc[CC_K + 6] = 0.0; cc[CC_K + 7] = 0.0; cc[CC_K + 8] = 1.0;
    cc[CC_SMALL] = 1.0;
  }
  cc[CC_T + 0] = cam6[3]; cc[CC_T + 1] = cam6[4]; cc[CC_T + 2] = cam6[5];
  cc[CC_FX] = intr4[0]; cc[CC_FY] = intr4[1]; cc[CC_PPX] = intr4[2]; cc[CC_PPY] = intr4[3];
  for (int i = CC_SMALL + 1; i < CC_STRIDE; ++i) cc[i] = 0.0;
}

// 1 / x for the depth of a point in a camera.  Device: v_rcp_f64 and two Newton steps, five instructions where the IEEE
// division sequence (scale, rcp, two Newton steps, quotient, residual, fmas, fixup) is eleven; within an ulp of it.
RSBA_HD double RcpNewton(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
#else
  return 1.0 / x;
#endif
}

// t^3 rounded once (to within 0.5 + 2^-50 ulp): the products' rounding errors are recovered with fused multiply-adds and added
// back.  Ceres' radius update is radius / max(1/3, 1 - pow(2 rho - 1, 3)) (trust_region_minimizer.cc; the oracle calls
// std::pow); glibc's pow is correctly rounded, a device pow is not the same function, and (t * t) * t carries two roundings.
// This sequence is the same on the host (MinimizeLoop) and on the device (DecideStep) and gives pow's bits
// (tests/test_host_math.py::test_cube_is_pow).
RSBA_HD double Cube(double t) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double t2 = t * t;
  const double e = fma(t, t, -t2);     // t^2 = t2 + e exactly
  const double p = t2 * t;
  const double e2 = fma(t2, t, -p);    // t2 t = p + e2 exactly
  return p + fma(e, t, e2);
}

// THE residual of one observation: p = R X + t, iz = 1 / p2, r = (fx p0 iz + ppx - u, fy p1 iz + ppy - v), as ONE fixed
// sequence of roundings (every fused multiply-add is written out, so the compiler's contraction has nothing to decide).
// Every kernel whose residual enters a gradient calls this function: g_c and the right-hand side in the Schur kernel's self
// tiles, g_p and sqrt(rho') in the point pass and in the back-substitution's pass over the candidate (whose linearisation is
// the next iteration's).  With every block free (Test1_BundleAdjustment/main.cpp:76-79) the cost has a 7-dof gauge orbit, the
// gradient is orthogonal to it only if g_c and g_p are sums over the SAME residuals, and at a trust-region radius of 1e11 a
// mismatch of a few ulps of the projection (~1e-13 px, coherent over a camera's observations) between the kernel that forms
// g_c and the one that forms g_p is amplified into a drift of 1e-5 along the orbit (measured: the round-2 back-substitution
// formed p from rows pre-multiplied by fx; tests/test_gpu_parity.py::test_huber_and_rejected_steps, raw parameters).
RSBA_HD void ProjectResidual(const double* R, const double* t, double fx, double fy, double ppx, double ppy, const double X[3],
                             double u, double v, double p[3], double* iz_out, double r[2]) {
  p[0] = fma(R[0], X[0], fma(R[1], X[1], fma(R[2], X[2], t[0])));
  p[1] = fma(R[3], X[0], fma(R[4], X[1], fma(R[5], X[2], t[1])));
  p[2] = fma(R[6], X[0], fma(R[7], X[1], fma(R[8], X[2], t[2])));
  const double iz = RcpNewton(p[2]);
  *iz_out = iz;
  r[0] = fma(fx * p[0], iz, ppx) - u;
  r[1] = fma(fy * p[1], iz, ppy) - v;
}

// Residual only (operator()<double>), used for the candidate cost.
RSBA_HD void Residual(const double* cc, const double X[3], double u, double v, double r[2]) {
  double p[3], iz;
  ProjectResidual(cc + CC_R, cc + CC_T, cc[CC_FX], cc[CC_FY], cc[CC_PPX], cc[CC_PPY], X, u, v, p, &iz, r);
}

// Residual + the point-side Jacobian block only (2x3 row-major): what the linearisation of a point needs of an
// observation (V = sum Jp'Jp, g_p = sum Jp'r); same arithmetic as ResidualJacobian.
RSBA_HD void ResidualPointJacobian(const double* cc, const double X[3], double u, double v, double r[2], double jp[6]) {
  const double* R = cc + CC_R;
  double p[3], iz;
  ProjectResidual(R, cc + CC_T, cc[CC_FX], cc[CC_FY], cc[CC_PPX], cc[CC_PPY], X, u, v, p, &iz, r);
  const double al = cc[CC_FX] * iz, be = cc[CC_FY] * iz;
  const double ga = -al * p[0] * iz, de = -be * p[1] * iz;
  jp[0] = al * R[0] + ga * R[6]; jp[1] = al * R[1] + ga * R[7]; jp[2] = al * R[2] + ga * R[8];
  jp[3] = be * R[3] + de * R[6]; jp[4] = be * R[4] + de * R[7]; jp[5] = be * R[5] + de * R[8];
}

// Residual + Jacobian blocks.  jc: 2x6 row-major (d/d rvec, d/d tvec), jp: 2x3 row-major.
RSBA_HD void ResidualJacobian(const double* cc, const double X[3], double u, double v, double r[2],
                              double jc[12], double jp[6]) {
  const double* R = cc + CC_R;
  const double* K = cc + CC_K;
  const double q0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double q1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double q2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  double p[3], iz;
  ProjectResidual(R, cc + CC_T, cc[CC_FX], cc[CC_FY], cc[CC_PPX], cc[CC_PPY], X, u, v, p, &iz, r);
  const double p0 = p[0], p1 = p[1];
  const double al = cc[CC_FX] * iz, be = cc[CC_FY] * iz;
  const double ga = -al * p0 * iz, de = -be * p1 * iz;
  // d r / d X = Pj R
  jp[0] = al * R[0] + ga * R[6]; jp[1] = al * R[1] + ga * R[7]; jp[2] = al * R[2] + ga * R[8];
  jp[3] = be * R[3] + de * R[6]; jp[4] = be * R[4] + de * R[7]; jp[5] = be * R[5] + de * R[8];
  // rows of -Pj [w]x are w x Pj_i, with w = q (Rodrigues branch) or X (first-order branch)
  const bool small = cc[CC_SMALL] != 0.0;
  const double w0 = small ? X[0] : q0, w1 = small ? X[1] : q1, w2 = small ? X[2] : q2;
  const double a0 = w1 * ga, a1 = w2 * al - w0 * ga, a2 = -w1 * al;
  const double b0 = w1 * de - w2 * be, b1 = -w0 * de, b2 = w0 * be;
  jc[0] = a0 * K[0] + a1 * K[3] + a2 * K[6];
  jc[1] = a0 * K[1] + a1 * K[4] + a2 * K[7];
  jc[2] = a0 * K[2] + a1 * K[5] + a2 * K[8];
  jc[3] = al; jc[4] = 0.0; jc[5] = ga;
  jc[6] = b0 * K[0] + b1 * K[3] + b2 * K[6];
  jc[7] = b0 * K[1] + b1 * K[4] + b2 * K[7];
  jc[8] = b0 * K[2] + b1 * K[5] + b2 * K[8];
  jc[9] = 0.0; jc[10] = be; jc[11] = de;
}

// ------------------------------------------------------------------------------------------------
// Marker-chain functors (Main_Calibration/bundle_adjustment.h:56-343), ONE CORNER, analytically.
//
//   X = (cx, cy, 0)  --marker-->  p_m = R_m X + t_m  --time-->  p_t = R_t p_m + t_t  --camera-->  p_c = R_c p_t + t_c
//   r = (fx p_c0 / p_c2 + ppx - u,  fy p_c1 / p_c2 + ppy - v)
// with the marker and / or the camera transform absent in three of the four functors (pc_marker / pc_cam == nullptr).
// J is 2 x 18, columns camera | time | marker (rvec3, tvec3 each), zero for an absent block.  With Pj = d r / d p_c and
// Q the 2 x 3 sensitivity of r to the point ENTERING a transform (Q_c = Pj, Q_t = Pj R_c, Q_m = Pj R_c R_t):
//   d r / d t = Q,      d r / d w = [w* x Q_0 ; w* x Q_1] Jl(w)      (rows of -Q [w*]x are w* x Q_i)
// where w* is the rotated point R p (Rodrigues branch) or the unrotated p (AngleAxisRotatePoint's first-order branch,
// theta^2 <= DBL_EPSILON: AutoDiff differentiates p + w x p there), and Jl the left Jacobian of SO(3) (identity in that
// branch) — exactly the point model's rule (above), applied once per transform of the chain.  Pose constants come in
// CameraConstants' layout (CC_R, CC_K = Jl, CC_T, CC_SMALL; the intrinsics slots are not used).
// Replaces the DJet<18> evaluation of k_marker_eval where the marker-chain model runs at scale: ~1.2k FMAs per residual
// block instead of ~20k, cheap enough to be recomputed wherever a Jacobian row is needed instead of being stored.
// ------------------------------------------------------------------------------------------------
RSBA_HD void MarkerCornerResidualJacobian(const double* pc_cam, const double* pc_time, const double* pc_marker, const double* intr4,
                                          double cx, double cy, double u, double v, double r[2], double J[36]) {
  auto rot = [](const double* pc, const double in[3], double q[3], double out[3]) {
    const double* R = pc + CC_R;
    q[0] = R[0] * in[0] + R[1] * in[1] + R[2] * in[2];
    q[1] = R[3] * in[0] + R[4] * in[1] + R[5] * in[2];
    q[2] = R[6] * in[0] + R[7] * in[1] + R[8] * in[2];
    out[0] = q[0] + pc[CC_T]; out[1] = q[1] + pc[CC_T + 1]; out[2] = q[2] + pc[CC_T + 2];
  };
  // the 2 x 6 block of one transform from Q (2 x 3), the point that entered it (pin) and its rotated image (q)
  auto block = [](const double* pc, const double Q[6], const double pin[3], const double q[3], double* Jb /* row stride 18 */) {
    const bool small = pc[CC_SMALL] != 0.0;
    const double w0 = small ? pin[0] : q[0], w1 = small ? pin[1] : q[1], w2 = small ? pin[2] : q[2];
    const double* K = pc + CC_K;
    for (int i = 0; i < 2; ++i) {
      const double q0 = Q[3 * i], q1 = Q[3 * i + 1], q2 = Q[3 * i + 2];
      const double a0 = w1 * q2 - w2 * q1, a1 = w2 * q0 - w0 * q2, a2 = w0 * q1 - w1 * q0;   // w* x Q_i
      Jb[18 * i + 0] = a0 * K[0] + a1 * K[3] + a2 * K[6];
      Jb[18 * i + 1] = a0 * K[1] + a1 * K[4] + a2 * K[7];
      Jb[18 * i + 2] = a0 * K[2] + a1 * K[5] + a2 * K[8];
      Jb[18 * i + 3] = q0; Jb[18 * i + 4] = q1; Jb[18 * i + 5] = q2;
    }
  };
  const double X[3] = {cx, cy, 0.0};
  double qm[3] = {0, 0, 0}, pm[3] = {X[0], X[1], X[2]};
  if (pc_marker) rot(pc_marker, X, qm, pm);
  double qt[3], pt[3];
  rot(pc_time, pm, qt, pt);
  double qc[3] = {0, 0, 0}, pcm[3] = {pt[0], pt[1], pt[2]};
  if (pc_cam) rot(pc_cam, pt, qc, pcm);
  const double fx = intr4[0], fy = intr4[1], ppx = intr4[2], ppy = intr4[3];
  const double iz = 1.0 / pcm[2];
  r[0] = fx * pcm[0] * iz + ppx - u;
  r[1] = fy * pcm[1] * iz + ppy - v;
  const double al = fx * iz, be = fy * iz;
  const double ga = -al * pcm[0] * iz, de = -be * pcm[1] * iz;
  for (int i = 0; i < 36; ++i) J[i] = 0.0;
  // camera transform
  double Qt[6];   // sensitivity to p_t
  if (pc_cam) {
    const double Qc[6] = {al, 0.0, ga, 0.0, be, de};
    block(pc_cam, Qc, pt, qc, J + 0);
    const double* R = pc_cam + CC_R;
    for (int j = 0; j < 3; ++j) { Qt[j] = al * R[j] + ga * R[6 + j]; Qt[3 + j] = be * R[3 + j] + de * R[6 + j]; }
  } else {
    Qt[0] = al; Qt[1] = 0.0; Qt[2] = ga; Qt[3] = 0.0; Qt[4] = be; Qt[5] = de;
  }
  // time transform
  block(pc_time, Qt, pm, qt, J + 6);
  // marker transform
  if (pc_marker) {
    const double* R = pc_time + CC_R;
    double Qm[6];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) Qm[3 * i + j] = Qt[3 * i] * R[j] + Qt[3 * i + 1] * R[3 + j] + Qt[3 * i + 2] * R[6 + j];
    block(pc_marker, Qm, X, qm, J + 12);
  }
}

// One of the three 2 x 6 blocks of MarkerCornerResidualJacobian's rows (part 0: camera block + the residuals, 1: time
// block, 2: marker block), the same arithmetic: k_time_eliminate stages a corner on three lanes.  The intrinsics come by
// value (the caller fetched them a tile ahead).  J has the row stride 18 of the full rows; a block whose pose is not a
// parameter is written as zeros.
RSBA_HD void MarkerCornerJacobianPart(int part, const double* pc_cam, const double* pc_time, const double* pc_marker,
                                                double fx, double fy, double ppx, double ppy, double cx, double cy, double u, double v,
                                                double* r, double* J) {
  auto rot = [](const double* pc, const double in[3], double q[3], double out[3]) {
    const double* R = pc + CC_R;
    q[0] = R[0] * in[0] + R[1] * in[1] + R[2] * in[2];
    q[1] = R[3] * in[0] + R[4] * in[1] + R[5] * in[2];
    q[2] = R[6] * in[0] + R[7] * in[1] + R[8] * in[2];
    out[0] = q[0] + pc[CC_T]; out[1] = q[1] + pc[CC_T + 1]; out[2] = q[2] + pc[CC_T + 2];
  };
  auto block = [](const double* pc, const double Q[6], const double pin[3], const double q[3], double* Jb) {
    const bool small = pc[CC_SMALL] != 0.0;
    const double w0 = small ? pin[0] : q[0], w1 = small ? pin[1] : q[1], w2 = small ? pin[2] : q[2];
    const double* K = pc + CC_K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double q0 = Q[3 * i], q1 = Q[3 * i + 1], q2 = Q[3 * i + 2];
      const double a0 = w1 * q2 - w2 * q1, a1 = w2 * q0 - w0 * q2, a2 = w0 * q1 - w1 * q0;
      Jb[18 * i + 0] = a0 * K[0] + a1 * K[3] + a2 * K[6];
      Jb[18 * i + 1] = a0 * K[1] + a1 * K[4] + a2 * K[7];
      Jb[18 * i + 2] = a0 * K[2] + a1 * K[5] + a2 * K[8];
      Jb[18 * i + 3] = q0; Jb[18 * i + 4] = q1; Jb[18 * i + 5] = q2;
    }
  };
  auto zeros = [](double* Jb) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) Jb[18 * i + j] = 0.0;
  };
  const double X[3] = {cx, cy, 0.0};
  double qm[3] = {0, 0, 0}, pm[3] = {X[0], X[1], X[2]};
  if (pc_marker) rot(pc_marker, X, qm, pm);
  double qt[3], pt[3];
  rot(pc_time, pm, qt, pt);
  double qc[3] = {0, 0, 0}, pcm[3] = {pt[0], pt[1], pt[2]};
  if (pc_cam) rot(pc_cam, pt, qc, pcm);
  const double iz = 1.0 / pcm[2];
  const double al = fx * iz, be = fy * iz;
  const double ga = -al * pcm[0] * iz, de = -be * pcm[1] * iz;
  if (part == 0) {
    r[0] = fx * pcm[0] * iz + ppx - u;
    r[1] = fy * pcm[1] * iz + ppy - v;
    if (pc_cam) { const double Qc[6] = {al, 0.0, ga, 0.0, be, de}; block(pc_cam, Qc, pt, qc, J + 0); }
    else zeros(J + 0);
    return;
  }
  double Qt[6];
  if (pc_cam) {
    const double* R = pc_cam + CC_R;
#pragma unroll
    for (int j = 0; j < 3; ++j) { Qt[j] = al * R[j] + ga * R[6 + j]; Qt[3 + j] = be * R[3 + j] + de * R[6 + j]; }
  } else {
    Qt[0] = al; Qt[1] = 0.0; Qt[2] = ga; Qt[3] = 0.0; Qt[4] = be; Qt[5] = de;
  }
  if (part == 1) { block(pc_time, Qt, pm, qt, J + 6); return; }
  if (pc_marker) {
    const double* R = pc_time + CC_R;
    double Qm[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Qm[3 * i + j] = Qt[3 * i] * R[j] + Qt[3 * i + 1] * R[3 + j] + Qt[3 * i + 2] * R[6 + j];
    block(pc_marker, Qm, X, qm, J + 12);
  } else {
    zeros(J + 12);
  }
}

// ceres::HuberLoss / CauchyLoss + Corrector for rho'' <= 0 (both): returns rho(s) and the factor sqrt(rho'(s)) that
// scales the residual and both Jacobian blocks.  delta > 0: Huber with a = delta; delta < 0: Cauchy with a = -delta
// (the sign is this implementation's internal encoding of rsba_options::loss_type); 0: no loss.
RSBA_HD double LossAndScale(double delta, double s, double* sqrt_rho1) {
  if (delta > 0.0) {
    if (s > delta * delta) {
      const double rt = sqrt(s);
      double rho1 = delta / rt;
      if (rho1 < DBL_MIN) rho1 = DBL_MIN;
      *sqrt_rho1 = sqrt(rho1);
      return 2.0 * delta * rt - delta * delta;
    }
  } else if (delta < 0.0) {
    // loss_function.cc CauchyLoss: b = a^2, c = 1 / b; rho = b log(1 + s c), rho' = max(min, 1 / (1 + s c))
    const double b = delta * delta, sum = s * (1.0 / b) + 1.0;
    double rho1 = 1.0 / sum;
    if (rho1 < DBL_MIN) rho1 = DBL_MIN;
    *sqrt_rho1 = sqrt(rho1);
    return b * log(sum);
  }
  *sqrt_rho1 = 1.0;
  return s;
}

// Symmetric 3x3 stored as (00, 01, 02, 11, 12, 22).
// Effective inverse of a point's damped Hessian block in unscaled coordinates:
//   Vs = diag(s) V diag(s);  D2 = clamp(diag(Vs), lo, hi) / radius;  Minv = (Vs + D2)^-1 via LLT;
//   out = diag(s) Minv diag(s)        (so that  W_s (Vs+D2)^-1 W_s' = s_c [W out W'] s_c)
// Returns false when the block is not positive definite / not finite (Ceres: linear solver failure).
RSBA_HD bool PointBlockInverse(const double V[6], const double s[3], double lo, double hi, double radius,
                               double out[6]) {
  double m00 = s[0] * s[0] * V[0], m01 = s[0] * s[1] * V[1], m02 = s[0] * s[2] * V[2];
  double m11 = s[1] * s[1] * V[3], m12 = s[1] * s[2] * V[4], m22 = s[2] * s[2] * V[5];
  const double ir = 1.0 / radius;
  m00 += fmin(fmax(m00, lo), hi) * ir;
  m11 += fmin(fmax(m11, lo), hi) * ir;
  m22 += fmin(fmax(m22, lo), hi) * ir;
  // LLT
  if (!(m00 > 0.0)) return false;
  const double l00 = sqrt(m00), i00 = 1.0 / l00;
  const double l10 = m01 * i00, l20 = m02 * i00;
  const double d11 = m11 - l10 * l10;
  if (!(d11 > 0.0)) return false;
  const double l11 = sqrt(d11), i11 = 1.0 / l11;
  const double l21 = (m12 - l20 * l10) * i11;
  const double d22 = m22 - l20 * l20 - l21 * l21;
  if (!(d22 > 0.0)) return false;
  const double l22 = sqrt(d22), i22 = 1.0 / l22;
  // Linv (lower): rows
  const double a00 = i00;
  const double a10 = -l10 * i00 * i11, a11 = i11;
  const double a20 = -(l20 * a00 + l21 * a10) * i22, a21 = -l21 * a11 * i22, a22 = i22;
  // Minv = Linv' Linv
  const double n00 = a00 * a00 + a10 * a10 + a20 * a20;
  const double n01 = a10 * a11 + a20 * a21;
  const double n02 = a20 * a22;
  const double n11 = a11 * a11 + a21 * a21;
  const double n12 = a21 * a22;
  const double n22 = a22 * a22;
  out[0] = s[0] * s[0] * n00; out[1] = s[0] * s[1] * n01; out[2] = s[0] * s[2] * n02;
  out[3] = s[1] * s[1] * n11; out[4] = s[1] * s[2] * n12; out[5] = s[2] * s[2] * n22;
  const double chk = out[0] + out[3] + out[5];
  return chk == chk && fabs(chk) <= DBL_MAX;
}

RSBA_HD void Sym3MulVec(const double M[6], const double x[3], double y[3]) {
  y[0] = M[0] * x[0] + M[1] * x[1] + M[2] * x[2];
  y[1] = M[1] * x[0] + M[3] * x[1] + M[4] * x[2];
  y[2] = M[2] * x[0] + M[4] * x[1] + M[5] * x[2];
}

}  // namespace rsba
