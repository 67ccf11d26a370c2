// HIP kernels of the point-model LM iteration (gfx950, wave64).
//
// These are the steps that are implicit inside ceres::Solve for the reference's point model
// (/root/reference/Test1_BundleAdjustment/main.cpp:63-87 with DENSE_SCHUR): residual + Jacobian
// evaluation, Schur elimination of the point blocks into the reduced camera system, the dense
// Cholesky solve, back-substitution and the candidate cost.  Nothing here is a translation of
// reference code: the reference has no such code (it links Ceres).
//
// Data layout in HBM (all fp64 / int32, struct-of-arrays, observations sorted by point then camera):
//   obs_u[N], obs_v[N], obs_cam[N], pt_ptr[P+1]   20 B per observation + 4 B per point
//   cam[6C], intr[4C], camc[32C]                  camera blocks and their per-linearisation constants
//   pts[3P], scale_p[3P], scale_c[6C]             points, Jacobi scales (fixed at iteration 0)
//   red[]                                         the all-reduce payload: S | gc | corr | diagU | scalars
#pragma once
#include <hip/hip_runtime.h>

#include "ba_cholesky.hpp"
#include "ba_math.hpp"

namespace rsba {

// ---- layout of the reduction payload `red` (doubles) for nc = 6C
//   [0, nc*nc)            S   : U - sum W Vinv W'   (camera-unscaled, undamped; upper blocks valid)
//   [+0, +nc)             gc  : J_c' r              (camera gradient)
//   [+nc, +2nc)           corr: -sum W Vinv g_p     (Schur correction of the rhs)
//   [+2nc, +3nc)          diagU: diag(J_c' J_c)
//   [+3nc, +3nc+8)        scalars: 0 cost(sum rho, not halved) 1 |X|^2 (points) 2 fail count
struct RedLayout {
  int nc;
  __host__ __device__ size_t S() const { return 0; }
  __host__ __device__ size_t gc() const { return (size_t)nc * nc; }
  __host__ __device__ size_t corr() const { return (size_t)nc * nc + nc; }
  __host__ __device__ size_t diagU() const { return (size_t)nc * nc + 2 * (size_t)nc; }
  __host__ __device__ size_t scal() const { return (size_t)nc * nc + 3 * (size_t)nc; }
  __host__ __device__ size_t size() const { return (size_t)nc * nc + 3 * (size_t)nc + 8; }
};

// Everything the device-side of one iteration reads that is a scalar.
struct IterParams {
  double radius;
  double min_lm_diagonal, max_lm_diagonal;
  double huber_delta;
  int first;           // 1: iteration 0 -> compute and store the Jacobi scales
  int jacobi_scaling;
  const double* cam_free = nullptr;   // per camera 1.0 / 0.0 (constant block: SetParameterBlockConstant); nullptr: all free
  const unsigned char* pt_const = nullptr;   // per point (the solver's internal order) != 0: constant block; nullptr: all free
};
// The linearisation record of a CONSTANT point block (Problem::SetParameterBlockConstant on a point, round 6): Ceres removes the block
// from the program — no columns, no step.  Here its record says "infinitely stiff, no gradient": V = kConstPointStiffness I, g_p = 0,
// so that every consumer of the record that forms (V + D)^-1 (g_p + ...) — the back-substitution kernels, untouched by this — gets a step
// of ~1e-200 x (a few thousand): X + step == X exactly, and its share of the model cost change is what the cameras' steps alone give.
// The kernels that WRITE damped blocks (k_point_pass, k_point_damp) set the inverse block and V^-1 g_p to exact zeros for such a point and
// leave it out of |x|, max |g| and the failed-block count.
#define RSBA_CONST_POINT_STIFFNESS 1e200
__global__ void __launch_bounds__(256) k_fix_const_lin(int P, const unsigned char* __restrict__ pt_const, double* __restrict__ lin, int lin_stride) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < P; j += gridDim.x * blockDim.x) {
    if (pt_const[j] == 0) continue;
    double* ln = lin + (size_t)j * lin_stride;
    ln[0] = RSBA_CONST_POINT_STIFFNESS; ln[1] = 0.0; ln[2] = 0.0; ln[3] = RSBA_CONST_POINT_STIFFNESS; ln[4] = 0.0; ln[5] = RSBA_CONST_POINT_STIFFNESS;
    ln[6] = 0.0; ln[7] = 0.0; ln[8] = 0.0;
  }
}


// Result block the host reads back once per iteration (and RCCL reduces in part).
enum {
  RES_COST_X = 0,        // 1/2 sum rho at x
  RES_GMAX = 1,          // max |gradient|
  RES_XNORM2 = 2,        // |x|^2
  RES_CHOL_OK = 3,       // 1.0 when the reduced system factorised
  RES_MCC = 4,           // model cost change
  RES_COST_C = 5,        // 1/2 sum rho at the candidate
  RES_STEP2 = 6,         // |delta|^2
  RES_XCNORM2 = 7,       // |x + delta|^2
  RES_POINT_FAIL = 8,    // number of point blocks that were not positive definite
  RES_SUMSQ_C = 9,       // sum of squared raw residuals at the candidate (for the RMS metric)
  RES_STALL = 10,        // pipelined solve only: 1.0 when the Cholesky gave up waiting for its columns
  RES_WAIT_TIMEOUT = 11, // pipelined solve only: 1.0 when the back-substitution gave up waiting for the solve
  // the trust-region decision as the device took it for the damping kernel queued behind the step (LmNext below)
  RES_DEC_GO = 12, RES_DEC_ACCEPT = 13, RES_DEC_RADIUS = 14,
  RES_TIME_UP = 15,      // several ranks + max_solver_time_in_seconds: != 0 when RANK 0's clock had run out as it launched this step (summed with the
                         // candidate scalars: every rank stops on the same iteration)
  RES_SIZE = 24          // (the last word is the sequence number the host polls)
};

__device__ __forceinline__ double WaveSum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double WaveMax(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ void AtomicMaxNonNeg(double* addr, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// ------------------------------------------------------------------------------------------------
// K0: per-camera constants (R, left Jacobian, t, intrinsics) for one set of camera blocks.
// ------------------------------------------------------------------------------------------------
__global__ void k_camera_constants(int C, const double* __restrict__ cam, const double* __restrict__ intr,
                                   double* __restrict__ camc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double cc[CC_STRIDE];
  CameraConstants(cam + 6 * c, intr + 4 * c, cc);
#pragma unroll
  for (int i = 0; i < CC_STRIDE; ++i) camc[(size_t)c * CC_STRIDE + i] = cc[i];
}

// ------------------------------------------------------------------------------------------------
// K_A (reference Schur kernel, schur_impl = 0): one wavefront per point, one lane per observation.
//   phase 1: residual + Jacobian blocks per lane; wave reductions give V = sum Jp'Jp, g_p, cost;
//            camera-side sums (U = Jc'Jc upper 21, g_c 6, corr 6) go to LDS accumulators per block.
//   phase 2: every lane a holds Y_a = W_a Vinv; W_b is broadcast lane by lane and the 6x6 block
//            -Y_a W_b' is added to S(cam_a, cam_b) for cam_a <= cam_b with global fp64 atomics.
// Correct for any problem with at most 64 views per point; slow (atomic-bound): it exists as the
// on-device cross-check of the tiled kernel and as the first parity-green path.
// ------------------------------------------------------------------------------------------------
#define RSBA_ACC_PER_CAM 33  // 21 (U upper) + 6 (gc) + 6 (corr)
// LDS copies of the per-camera constants use an odd stride: with the global stride (32 doubles = one full bank row)
// every lane reading element e of a different camera would hit the same bank.
#define RSBA_CC_LDS (CC_STRIDE + 1)

template <bool kStageCamc>
__global__ void __launch_bounds__(256)
k_linearize_schur_ref(int C, int P, const double* __restrict__ obs_u, const double* __restrict__ obs_v,
                      const int* __restrict__ obs_cam, const int* __restrict__ pt_ptr,
                      const double* __restrict__ camc_g, const double* __restrict__ pts,
                      double* __restrict__ scale_p, double* __restrict__ red, RedLayout L,
                      double* __restrict__ gmax_out, double* __restrict__ block_scal /* gridDim.x x 4 */,
                      IterParams ip) {
  extern __shared__ double lds[];
  double* acc = lds;                                      // C x 33
  double* camc_l = lds + (size_t)C * RSBA_ACC_PER_CAM;    // C x 32 when staged
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  for (int i = tid; i < C * RSBA_ACC_PER_CAM; i += blockDim.x) acc[i] = 0.0;
  if (kStageCamc) for (int i = tid; i < C * CC_STRIDE; i += blockDim.x) { const int c = i / CC_STRIDE, e = i - c * CC_STRIDE; camc_l[c * RSBA_CC_LDS + e] = camc_g[i]; }
  __syncthreads();
  const double* camc = kStageCamc ? camc_l : camc_g;
  const int ccs = kStageCamc ? RSBA_CC_LDS : CC_STRIDE;

  double cost_acc = 0.0, xn_acc = 0.0, fail_acc = 0.0, gmax = 0.0;
  for (int j = blockIdx.x * nwave + wave; j < P; j += gridDim.x * nwave) {
    const int b = pt_ptr[j], k = pt_ptr[j + 1] - b;
    const double X[3] = {pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]};
    const bool act = lane < k;
    int cam = 0;
    double r[2] = {0, 0}, jc[12], jp[6];
#pragma unroll
    for (int i = 0; i < 12; ++i) jc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) jp[i] = 0.0;
    double rho = 0.0;
    if (act) {
      cam = obs_cam[b + lane];
      ResidualJacobian(camc + (size_t)cam * ccs, X, obs_u[b + lane], obs_v[b + lane], r, jc, jp);
      if (ip.cam_free != nullptr && ip.cam_free[cam] == 0.0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) jc[i] = 0.0;   // a constant camera has no columns
      }
      double sq;
      rho = LossAndScale(ip.huber_delta, r[0] * r[0] + r[1] * r[1], &sq);
      if (sq != 1.0) {
        r[0] *= sq; r[1] *= sq;
#pragma unroll
        for (int i = 0; i < 12; ++i) jc[i] *= sq;
#pragma unroll
        for (int i = 0; i < 6; ++i) jp[i] *= sq;
      }
    }
    // point-side sums over the wave
    double V[6], gp[3];
    V[0] = WaveSum(jp[0] * jp[0] + jp[3] * jp[3]);
    V[1] = WaveSum(jp[0] * jp[1] + jp[3] * jp[4]);
    V[2] = WaveSum(jp[0] * jp[2] + jp[3] * jp[5]);
    V[3] = WaveSum(jp[1] * jp[1] + jp[4] * jp[4]);
    V[4] = WaveSum(jp[1] * jp[2] + jp[4] * jp[5]);
    V[5] = WaveSum(jp[2] * jp[2] + jp[5] * jp[5]);
    gp[0] = WaveSum(jp[0] * r[0] + jp[3] * r[1]);
    gp[1] = WaveSum(jp[1] * r[0] + jp[4] * r[1]);
    gp[2] = WaveSum(jp[2] * r[0] + jp[5] * r[1]);
    const double cost_j = WaveSum(rho);
    double sp[3] = {1.0, 1.0, 1.0};
    if (ip.jacobi_scaling) {
      if (ip.first) {
        sp[0] = 1.0 / (1.0 + sqrt(V[0])); sp[1] = 1.0 / (1.0 + sqrt(V[3])); sp[2] = 1.0 / (1.0 + sqrt(V[5]));
        if (lane == 0) { scale_p[3 * (size_t)j] = sp[0]; scale_p[3 * (size_t)j + 1] = sp[1]; scale_p[3 * (size_t)j + 2] = sp[2]; }
      } else {
        sp[0] = scale_p[3 * (size_t)j]; sp[1] = scale_p[3 * (size_t)j + 1]; sp[2] = scale_p[3 * (size_t)j + 2];
      }
    } else if (ip.first && lane == 0) {
      scale_p[3 * (size_t)j] = 1.0; scale_p[3 * (size_t)j + 1] = 1.0; scale_p[3 * (size_t)j + 2] = 1.0;
    }
    double Vi[6];
    const bool ok = PointBlockInverse(V, sp, ip.min_lm_diagonal, ip.max_lm_diagonal, ip.radius, Vi);
    if (lane == 0) {
      cost_acc += cost_j;
      xn_acc += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
      if (!ok && k > 0) fail_acc += 1.0;
      gmax = fmax(gmax, fmax(fabs(gp[0]), fmax(fabs(gp[1]), fabs(gp[2]))));
    }
    if (!ok) {
#pragma unroll
      for (int i = 0; i < 6; ++i) Vi[i] = 0.0;
    }
    // W = Jc' Jp (6x3), Y = W Vinv (6x3)
    double W[18], Y[18];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) W[3 * a + c3] = jc[a] * jp[c3] + jc[6 + a] * jp[3 + c3];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      Y[3 * a + 0] = W[3 * a] * Vi[0] + W[3 * a + 1] * Vi[1] + W[3 * a + 2] * Vi[2];
      Y[3 * a + 1] = W[3 * a] * Vi[1] + W[3 * a + 1] * Vi[3] + W[3 * a + 2] * Vi[4];
      Y[3 * a + 2] = W[3 * a] * Vi[2] + W[3 * a + 1] * Vi[4] + W[3 * a + 2] * Vi[5];
    }
    if (act) {
      double* a33 = acc + (size_t)cam * RSBA_ACC_PER_CAM;
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c6 = a; c6 < 6; ++c6) { unsafeAtomicAdd(&a33[t], jc[a] * jc[c6] + jc[6 + a] * jc[6 + c6]); ++t; }
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) unsafeAtomicAdd(&a33[21 + a], jc[a] * r[0] + jc[6 + a] * r[1]);
#pragma unroll
      for (int a = 0; a < 6; ++a) unsafeAtomicAdd(&a33[27 + a], -(Y[3 * a] * gp[0] + Y[3 * a + 1] * gp[1] + Y[3 * a + 2] * gp[2]));
    }
    // Schur blocks: -Y_a W_b'
    for (int bb = 0; bb < k; ++bb) {
      const int camb = __shfl(cam, bb, 64);
      double Wb[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) Wb[i] = __shfl(W[i], bb, 64);
      if (act && cam <= camb) {
        double* Sb = red + L.S() + (size_t)(6 * cam) * L.nc + 6 * camb;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c6 = 0; c6 < 6; ++c6) {
            const double v = Y[3 * a] * Wb[3 * c6] + Y[3 * a + 1] * Wb[3 * c6 + 1] + Y[3 * a + 2] * Wb[3 * c6 + 2];
            unsafeAtomicAdd(&Sb[(size_t)a * L.nc + c6], -v);
          }
        }
      }
    }
  }
  __syncthreads();
  // flush the camera accumulators: U into the diagonal blocks of S (+ diagU), gc, corr
  for (int i = tid; i < C * RSBA_ACC_PER_CAM; i += blockDim.x) {
    const int c = i / RSBA_ACC_PER_CAM, e = i - c * RSBA_ACC_PER_CAM;
    const double v = acc[i];
    if (v == 0.0) continue;
    if (e < 21) {
      int a = 0, rem = e;
      while (rem >= 6 - a) { rem -= 6 - a; ++a; }
      const int c6 = a + rem;
      unsafeAtomicAdd(&red[L.S() + (size_t)(6 * c + a) * L.nc + 6 * c + c6], v);
      if (a == c6) unsafeAtomicAdd(&red[L.diagU() + 6 * c + a], v);
    } else if (e < 27) {
      unsafeAtomicAdd(&red[L.gc() + 6 * c + (e - 21)], v);
    } else {
      unsafeAtomicAdd(&red[L.corr() + 6 * c + (e - 27)], v);
    }
  }
  // per-block scalars (deterministic second stage in k_finish_linearize)
  __shared__ double sred[4][4];
  if (lane == 0) { sred[wave][0] = cost_acc; sred[wave][1] = xn_acc; sred[wave][2] = fail_acc; sred[wave][3] = gmax; }
  __syncthreads();
  if (tid == 0) {
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int w = 0; w < nwave; ++w) { c0 += sred[w][0]; c1 += sred[w][1]; c2 += sred[w][2]; c3 = fmax(c3, sred[w][3]); }
    block_scal[4 * blockIdx.x + 0] = c0; block_scal[4 * blockIdx.x + 1] = c1; block_scal[4 * blockIdx.x + 2] = c2;
    block_scal[4 * blockIdx.x + 3] = c3;
  }
  (void)gmax_out;
}

// Second stage of the scalar reductions of K_A, fixed order -> red scalars (sum part) and gmax (max part).
// `s`: 256 x 4 doubles of LDS nobody else is using (the Schur kernel passes its staging buffer: a static array here
// would come on top of the kernel's own 80 KB)
__device__ __forceinline__ void FinishLinearize(int nblocks, const double* __restrict__ block_scal, double* __restrict__ red,
                                                RedLayout L, double* __restrict__ gmax_p, double (*s)[4]) {
  const int tid = threadIdx.x;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int i = tid; i < nblocks; i += blockDim.x) {
    c0 += block_scal[4 * i]; c1 += block_scal[4 * i + 1]; c2 += block_scal[4 * i + 2]; c3 = fmax(c3, block_scal[4 * i + 3]);
  }
  s[tid][0] = c0; s[tid][1] = c1; s[tid][2] = c2; s[tid][3] = c3;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if (tid < off) { s[tid][0] += s[tid + off][0]; s[tid][1] += s[tid + off][1]; s[tid][2] += s[tid + off][2]; s[tid][3] = fmax(s[tid][3], s[tid + off][3]); }
    __syncthreads();
  }
  if (tid == 0) {
    red[L.scal() + 0] = s[0][0]; red[L.scal() + 1] = s[0][1]; red[L.scal() + 2] = s[0][2];
    for (int q = 3; q < 8; ++q) red[L.scal() + q] = 0.0;
    *gmax_p = s[0][3];
  }
}
__global__ void k_finish_linearize(int nblocks, const double* __restrict__ block_scal, double* __restrict__ red,
                                   RedLayout L, double* __restrict__ gmax_p) {
  __shared__ double s[256][4];
  FinishLinearize(nblocks, block_scal, red, L, gmax_p, s);
}

// Camera step from the solution of the scaled system: delta_c = -s_c y, candidate cameras and their constants, the
// camera parts of the norms, gradient max, and the scalars of res[].  One workgroup; `lds` needs 4 * blockDim.x doubles.
__device__ __forceinline__ void CameraStepEpilogue(int C, const double* __restrict__ red, RedLayout L, const double* __restrict__ scale_c,
                                                   const double* __restrict__ ysol, const double* __restrict__ cam_x,
                                                   double* __restrict__ cam_c, const double* __restrict__ intr, double* __restrict__ camc_c,
                                                   double* __restrict__ dcam, const double* __restrict__ gmax_p, double* __restrict__ res,
                                                   int ok, double* lds, const double* __restrict__ cam_free = nullptr) {
  const int n = L.nc, tid = threadIdx.x, nt = blockDim.x;
  // 5. camera step, candidate cameras, norms, gradient max over the camera part
  double d2 = 0, x2 = 0, xc2 = 0, gm = 0;
  for (int i = tid; i < n; i += nt) {
    const double d = -scale_c[i] * ysol[i];
    dcam[i] = d;
    const double x = cam_x[i], xc = x + d;
    cam_c[i] = xc;
    // a constant camera is not a parameter of the reduced program: Ceres' norms do not see it (its step is 0 anyway)
    if (cam_free == nullptr || cam_free[i / 6] != 0.0) { d2 += d * d; x2 += x * x; xc2 += xc * xc; }
    gm = fmax(gm, fabs(red[L.gc() + i]));
  }
  // small fixed-order reduction through LDS
  double* scr = lds;
  scr[tid] = d2; scr[nt + tid] = x2; scr[2 * nt + tid] = xc2; scr[3 * nt + tid] = gm;
  __syncthreads();
  for (int off = nt / 2; off > 0; off >>= 1) {
    if (tid < off) { scr[tid] += scr[tid + off]; scr[nt + tid] += scr[nt + tid + off]; scr[2 * nt + tid] += scr[2 * nt + tid + off]; scr[3 * nt + tid] = fmax(scr[3 * nt + tid], scr[3 * nt + tid + off]); }
    __syncthreads();
  }
  for (int c = tid; c < C; c += nt) {
    double cc[CC_STRIDE];
    CameraConstants(cam_c + 6 * c, intr + 4 * c, cc);
    // (a camera's 32 constants are 256 contiguous bytes per lane: 16-byte stores, half the instructions)
    typedef double d2s_t __attribute__((ext_vector_type(2)));
    d2s_t* out2 = reinterpret_cast<d2s_t*>(camc_c + (size_t)c * CC_STRIDE);
#pragma unroll
    for (int i = 0; i < CC_STRIDE / 2; ++i) { d2s_t v = {cc[2 * i], cc[2 * i + 1]}; out2[i] = v; }
  }
  if (tid == 0) {
    res[RES_COST_X] = 0.5 * red[L.scal() + 0];
    res[RES_GMAX] = fmax(*gmax_p, scr[3 * nt]);
    res[RES_XNORM2] = red[L.scal() + 1] + scr[nt];
    res[RES_POINT_FAIL] = red[L.scal() + 2];
    res[RES_CHOL_OK] = (ok && red[L.scal() + 2] == 0.0) ? 1.0 : 0.0;
    // camera parts of the step / candidate norms; the point parts are added by k_finish_candidate
    res[RES_STEP2] = scr[0];
    res[RES_XCNORM2] = scr[2 * nt];
  }
}

// ------------------------------------------------------------------------------------------------
// K_C: reduced camera system.  One workgroup (1024 threads).
//   1. Jacobi scale of the camera columns (iteration 0) from diagU
//   2. S_s = s_c S s_c + clamp(s_c^2 diagU)/radius, mirrored to a full symmetric matrix; rhs_s = s_c (gc + corr)
//   3. blocked right-looking Cholesky (32-wide panels, 64x64 trailing tiles through LDS); the rhs rides
//      along as row n of the panel so the forward substitution is part of the factorisation
//   4. blocked back-substitution, delta_c = -s_c y, candidate cameras and their constants
// A is the (n+1) x n row-major work matrix (row n = rhs).
// ------------------------------------------------------------------------------------------------
#define RSBA_NB 32
#define RSBA_TB 64

__device__ void CholeskySolveBlocked(int n, double* __restrict__ A, double* __restrict__ x_out, int* ok_out,
                                     double* lds /* >= 2*64*33 + 32*33 doubles */) {
  const int tid = threadIdx.x, nt = blockDim.x;
  double* L11 = lds;                       // 32 x 33
  double* tA = lds + RSBA_NB * (RSBA_NB + 1);     // 64 x 33
  double* tB = tA + RSBA_TB * (RSBA_NB + 1);      // 64 x 33
  __shared__ int s_ok;
  if (tid == 0) s_ok = 1;
  __syncthreads();
  const int rows = n + 1;  // including the rhs row
  for (int kb = 0; kb < n; kb += RSBA_NB) {
    const int nb = min(RSBA_NB, n - kb);
    // (a) diagonal block -> LDS, factor with one wave
    for (int i = tid; i < nb * nb; i += nt) { const int r = i / nb, c = i - r * nb; L11[r * (RSBA_NB + 1) + c] = A[(size_t)(kb + r) * n + kb + c]; }
    __syncthreads();
    if (tid < 64) {
      for (int j = 0; j < nb; ++j) {
        double d = L11[j * (RSBA_NB + 1) + j];
        if (!(d > 0.0) || !(d <= DBL_MAX)) { if (tid == 0) s_ok = 0; d = 1.0; }
        const double l = sqrt(d), il = 1.0 / l;
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) L11[j * (RSBA_NB + 1) + j] = l;
        double lij = 0.0;
        if (tid > j && tid < nb) { lij = L11[tid * (RSBA_NB + 1) + j] * il; L11[tid * (RSBA_NB + 1) + j] = lij; }
        __builtin_amdgcn_wave_barrier();
        if (tid > j && tid < nb) {
          for (int c = j + 1; c <= tid; ++c) L11[tid * (RSBA_NB + 1) + c] -= lij * L11[c * (RSBA_NB + 1) + j];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    for (int i = tid; i < nb * nb; i += nt) { const int r = i / nb, c = i - r * nb; if (c <= r) A[(size_t)(kb + r) * n + kb + c] = L11[r * (RSBA_NB + 1) + c]; }
    // (b) panel below (rows kb+nb .. n, the rhs row included): row <- row L11^-T, one thread per row
    for (int r = kb + nb + tid; r < rows; r += nt) {
      double* row = A + (size_t)r * n + kb;
      double v[RSBA_NB];
#pragma unroll
      for (int c = 0; c < RSBA_NB; ++c) v[c] = c < nb ? row[c] : 0.0;
#pragma unroll
      for (int c = 0; c < RSBA_NB; ++c) {
        if (c < nb) {
          double s = v[c];
#pragma unroll
          for (int q = 0; q < RSBA_NB; ++q) if (q < c) s -= v[q] * L11[c * (RSBA_NB + 1) + q];
          v[c] = s / L11[c * (RSBA_NB + 1) + c];
        }
      }
#pragma unroll
      for (int c = 0; c < RSBA_NB; ++c) if (c < nb) row[c] = v[c];
    }
    __threadfence_block();
    __syncthreads();
    // (c) trailing update, lower tiles only: A[I, J] -= Lp[I] Lp[J]'  for tiles I >= J over rows kb+nb..n
    const int t0 = kb + nb;
    const int m = rows - t0;  // trailing rows incl. rhs row
    if (m > 0) {
      const int ntile = (m + RSBA_TB - 1) / RSBA_TB;
      // 1024 threads: each computes a 2x2 patch of a 64x64 tile
      const int tr = (tid >> 5), tc = (tid & 31);  // rows 2*tr.., cols 2*tc..
      for (int ti = 0; ti < ntile; ++ti) {
        for (int i = tid; i < RSBA_TB * RSBA_NB; i += nt) {
          const int r = i / RSBA_NB, c = i - r * RSBA_NB; const int gr = t0 + ti * RSBA_TB + r;
          tA[r * (RSBA_NB + 1) + c] = (gr < rows && c < nb) ? A[(size_t)gr * n + kb + c] : 0.0;
        }
        for (int tj = 0; tj <= ti; ++tj) {
          __syncthreads();
          for (int i = tid; i < RSBA_TB * RSBA_NB; i += nt) {
            const int r = i / RSBA_NB, c = i - r * RSBA_NB; const int gr = t0 + tj * RSBA_TB + r;
            tB[r * (RSBA_NB + 1) + c] = (gr < rows && c < nb) ? A[(size_t)gr * n + kb + c] : 0.0;
          }
          __syncthreads();
          if (nt >= 1024 || tid < 1024) {
            double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
#pragma unroll 8
            for (int q = 0; q < RSBA_NB; ++q) {
              const double a0 = tA[(2 * tr) * (RSBA_NB + 1) + q], a1 = tA[(2 * tr + 1) * (RSBA_NB + 1) + q];
              const double b0 = tB[(2 * tc) * (RSBA_NB + 1) + q], b1 = tB[(2 * tc + 1) * (RSBA_NB + 1) + q];
              c00 += a0 * b0; c01 += a0 * b1; c10 += a1 * b0; c11 += a1 * b1;
            }
            const int gr0 = t0 + ti * RSBA_TB + 2 * tr, gc0 = t0 + tj * RSBA_TB + 2 * tc;
            if (gr0 < rows && gc0 < n && gc0 <= gr0) A[(size_t)gr0 * n + gc0] -= c00;
            if (gr0 < rows && gc0 + 1 < n && gc0 + 1 <= gr0) A[(size_t)gr0 * n + gc0 + 1] -= c01;
            if (gr0 + 1 < rows && gc0 < n && gc0 <= gr0 + 1) A[(size_t)(gr0 + 1) * n + gc0] -= c10;
            if (gr0 + 1 < rows && gc0 + 1 < n && gc0 + 1 <= gr0 + 1) A[(size_t)(gr0 + 1) * n + gc0 + 1] -= c11;
          }
        }
        __syncthreads();
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  // Row n of A now holds y' with L y = rhs.  Back-substitution L' x = y, blocked from the bottom.
  double* y = A + (size_t)n * n;
  double* part = tA;  // 32 x 33 partial dot products
  for (int kb = ((n - 1) / RSBA_NB) * RSBA_NB; kb >= 0; kb -= RSBA_NB) {
    const int nb = min(RSBA_NB, n - kb);
    const int below = n - (kb + nb);
    // y[kb+c] -= sum_{i >= kb+nb} L[i][kb+c] x[i]     (x already stored in y for those i)
    {
      const int c = tid & 31, slice = tid >> 5;  // 32 slices of rows
      double s = 0.0;
      if (c < nb) for (int i = kb + nb + slice; i < n; i += 32) s += A[(size_t)i * n + kb + c] * y[i];
      part[slice * (RSBA_NB + 1) + c] = s;
    }
    __syncthreads();
    if (tid < nb && below > 0) { double s = 0.0; for (int q = 0; q < 32; ++q) s += part[q * (RSBA_NB + 1) + tid]; y[kb + tid] -= s; }
    __syncthreads();
    if (tid < 64) {
      // solve L11' x = y for this block, sequentially from the last row
      for (int j = nb - 1; j >= 0; --j) {
        double xj = 0.0;
        if (tid == 0) { xj = y[kb + j] / A[(size_t)(kb + j) * n + kb + j]; y[kb + j] = xj; }
        xj = __shfl(xj, 0, 64);
        if (tid < j) y[kb + tid] -= A[(size_t)(kb + j) * n + kb + tid] * xj;
        __builtin_amdgcn_wave_barrier();
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) x_out[i] = y[i];
  __syncthreads();
  if (tid == 0) *ok_out = s_ok;
}

// Whole workgroup: everything the kernel wrote (camera step, candidate cameras and their constants, res[]) is made
// visible, then gate.done = tag releases the back-substitution that is waiting for it (pipelined schedule).
__device__ __forceinline__ void SolveDone(const StageGate& gate) {
  if (gate.done == nullptr) return;
  // every wave's stores are performed at the barrier (they are in this XCD's L2); ONE release then writes the L2 back
  // and sets the flag — a fence in each of the eight waves was eight write-backs in a row at the tail of the LM step
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(gate.done, gate.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(512)
k_reduced_system_solve(int C, double* __restrict__ red, RedLayout L, double* __restrict__ A /* (nc+2) x nc */,
                       double* __restrict__ S_copy /* may be null */, double* __restrict__ rhs_copy,
                       double* __restrict__ scale_c, const double* __restrict__ cam_x, double* __restrict__ cam_c,
                       const double* __restrict__ intr, double* __restrict__ camc_c, double* __restrict__ dcam,
                       const double* __restrict__ gmax_p, double* __restrict__ res, IterParams ip, int sym_full,
                       int* __restrict__ chol_ok, StageGate gate) {
  extern __shared__ double lds[];
  const int n = L.nc, tid = threadIdx.x, nt = blockDim.x;
  __shared__ int s_ok;
#ifdef RSBA_PROFILE_PHASES
  long long _k0 = clock64();
#endif
  // the raw matrix is scaled, damped and mirrored by the panel loads themselves when it is already full symmetric
  const bool fused = sym_full && S_copy == nullptr;
  // pipelined: launched ahead of the linearisation.  Camera group g's gate (ready[1 + g]) says its diagonal blocks,
  // damping diagonal, right-hand side and pair blocks are in place.  The first iteration defines the Jacobi scale from
  // the whole damping diagonal, so it waits for every group before it starts; later iterations start with group 0.
  const bool gated = gate.ready != nullptr;
  if (gated) {
    if (gate.trace && tid == 0) gate.trace[0] = wall_clock64();
    AnnounceResident(gate);
    if (ip.first) {
      for (int g = 0; g * gate.cols < n; ++g)
        if (!WaitReady(gate.ready + 1 + g, gate.tag, gate.waited, gate.budget)) { if (tid == 0) res[RES_STALL] = 1.0; SolveDone(gate); return; }
    }
    if (gate.trace && tid == 0) gate.trace[1] = wall_clock64();
  }
  {
    // 1. camera Jacobi scale
    for (int i = tid; i < n; i += nt) {
      if (ip.first) scale_c[i] = ip.jacobi_scaling ? 1.0 / (1.0 + sqrt(red[L.diagU() + i])) : 1.0;
    }
    if (tid == 0) *chol_ok = 1;
    __threadfence_block();
    __syncthreads();
    // 2. scaled, damped, mirrored system: one wave per row, lanes along the columns
    if (!fused) {
      const int lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
      for (int i = wv; i < n; i += nw) {
        const int bi = i / 6;
        const double si = scale_c[i];
        for (int j = lane; j < n; j += 64) {
          const int bj = j / 6;
          // upper blocks are the valid ones; inside a diagonal block the upper triangle
          const bool upper = (bi < bj) || (bi == bj && i <= j);
          const double raw = upper ? red[L.S() + (size_t)i * n + j] : red[L.S() + (size_t)j * n + i];
          double v = raw * (si * scale_c[j]);  // product of the scales first: bitwise symmetric
          if (i == j) {
            const double d = si * si * red[L.diagU() + i];
            v += fmin(fmax(d, ip.min_lm_diagonal), ip.max_lm_diagonal) / ip.radius;
          }
          A[(size_t)i * n + j] = v;
          if (S_copy) S_copy[(size_t)i * n + j] = v;
        }
      }
    }
    // right-hand side; the gated solve writes each group's entries when the group's gate opens
    if (!gated) {
      for (int i = tid; i < n; i += nt) {
        const double v = scale_c[i] * (red[L.gc() + i] + red[L.corr() + i]);
        A[(size_t)n * n + i] = v;
        if (rhs_copy) rhs_copy[i] = v;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
#ifdef RSBA_PROFILE_PHASES
  if (tid == 0) g_phase_cycles[8] += clock64() - _k0;
#endif
  // 3-4. factor + solve; y = solution of the scaled system, reused from row n of A
  double* ysol = A + (size_t)n * n;
  CholeskySolvePanelLDS(n, A, ysol, &s_ok, lds,
                        fused ? PanelSource{red + L.S(), scale_c, red + L.diagU(), ip.min_lm_diagonal, ip.max_lm_diagonal, 1.0 / ip.radius,
                                            gated ? red + L.gc() : nullptr, gated ? red + L.corr() : nullptr, sym_full == 2 ? 1 : 0}
                              : PanelSource{nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr, nullptr, 0},
                        gate);
  __syncthreads();
  if (s_ok < 0) { if (tid == 0) res[RES_STALL] = 1.0; SolveDone(gate); return; }
  int ok = 1;
  if (tid == 0) {
    res[RES_STALL] = 0.0;
    if (!s_ok) *chol_ok = 0;
    ok = *chol_ok;
  }
#ifdef RSBA_PROFILE_PHASES
  if (tid == 0) { g_phase_cycles[9] += clock64() - _k0; }
#endif
  // 5. camera step, candidate cameras, norms, gradient max over the camera part
  CameraStepEpilogue(C, red, L, scale_c, ysol, cam_x, cam_c, intr, camc_c, dcam, gmax_p, res, ok, lds, ip.cam_free);
  if (gate.trace && tid == 0) gate.trace[15] = wall_clock64();
  SolveDone(gate);
}

// Per-point linearisation record kept between LM steps (see k_point_damp in ba_schur_tiled.hpp):
// V_j = sum Jp'Jp (6, symmetric), g_pj = sum Jp'r (3), the point's share of sum rho (1).
#define RSBA_LIN_STRIDE 10
// Per-point record the Schur kernel stages (ptdata): X (3), the damped inverse point block (6), V^-1 g_p (3).
#define RSBA_PT_STRIDE 12

// Observations as the point-centric kernels walk them: sliced ELL.  A slice is 64 consecutive points (one wavefront);
// slot t of lane l sits at (row_ptr[slice] + t) * 64 + l, slots in camera order, cam < 0 pads a point with fewer views
// than the widest of its slice.  A wavefront's loads are then contiguous (1 KB of pixels, 256 B of camera indices per
// slot) instead of 64 separate cache lines of the point-major CSR arrays.
struct ObsSliced {
  const int* __restrict__ row_ptr;   // [ceil(P/64) + 1]
  const double2* __restrict__ uv;    // [rows * 64]
  const int* __restrict__ cam;       // [rows * 64]
};

// ------------------------------------------------------------------------------------------------
// K_B: back-substitution + candidate.  One thread per point; a second pass over the observation
// records.  delta_p = -Vinv (g_p + sum Jp' Jc delta_c); model-cost-change terms accumulate in the same
// loop; then the cost of the candidate (cameras' constants at x + delta already in camc_c).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void FinishCandidate(int nblocks, const double* __restrict__ block_part, double* __restrict__ small_red,
                                                double* __restrict__ res, double* host, double seq, const double* res_stall = nullptr);
__device__ __forceinline__ void FinishCandidateIn(double (*s)[256], int nblocks, const double* __restrict__ block_part, double* __restrict__ small_red,
                                                  double* __restrict__ res, double* host, double seq, const double* res_stall = nullptr);

// The step's accept / reject decision and next radius, taken on the device by the workgroup that completes the result
// block — the arithmetic of MinimizeLoop, operation for operation and without fused multiply-adds, so that the host, which
// takes the same decision from the same numbers a few microseconds later, arrives at the same bits.  It exists for ONE
// consumer: k_point_damp for the NEXT step is queued right behind this step's last kernel and damps the point blocks of
// whichever state the decision names, while the host is still reading the result and launching the factorisation and
// the Schur kernel — the damping kernel leaves the step's critical path (~6 us of 0.43 ms).  dec[0] = 1 (go), dec[1] =
// accepted, dec[2] = next radius; a host that stops, or decides differently, simply does not use what was damped.
struct LmNext {
  double* dec = nullptr;     // device, 4 doubles; nullptr: no decision wanted
  double radius = 0, decrease_factor = 0, min_relative_decrease = 0, max_radius = 0;
};
__device__ __forceinline__ void DecideStep(const LmNext& lm, double x_cost, double cand_cost, double mcc, double step2, double chol_ok,
                                           double* accept, double* next_radius) {
  const bool solved = chol_ok != 0.0 && isfinite(mcc) && isfinite(step2);
  *accept = 0.0;
  *next_radius = __ddiv_rn(lm.radius, lm.decrease_factor);
  if (!(solved && mcc > 0.0)) return;
  const double rd = __ddiv_rn(__dsub_rn(x_cost, cand_cost), mcc);
  if (!(rd > lm.min_relative_decrease)) return;
  const double t = __dsub_rn(__dmul_rn(2.0, rd), 1.0);
  const double t3 = Cube(t);
  *accept = 1.0;
  *next_radius = fmin(lm.max_radius, __ddiv_rn(lm.radius, fmax(1.0 / 3.0, __dsub_rn(1.0, t3))));
}

// kFused (tiled path): the linearisation of the points at x is READ (lin_x: V, g_p as the Schur kernel used them) instead
// of accumulated again, and the second pass, which evaluates the candidate's residuals anyway, takes the candidate's 2x3
// blocks along and WRITES its linearisation (lin_c; sqrt(rho') into sq_cm_c): if the step is accepted the next LM
// iteration starts from it without another pass over the observation records (k_point_damp).
struct FusedLin {
  const double* __restrict__ lin_x;   // [P][RSBA_LIN_STRIDE]
  double* __restrict__ lin_c;
  const int* __restrict__ cm_pos;     // sliced slot -> camera-major position (robust loss only)
  double* __restrict__ sq_cm_c;
  long long* trace;                   // diagnostic (RSBA_TRACE=1): [28] workgroup 0 past the solve's flag, [29] result posted
  long long* bs_wg = nullptr;         // diagnostic (RSBA_TRACE=4): [workgroup][4] past the flag, tables staged, pass at x done, block sums out
  LmNext lm;                          // k_backsub_candidate_proj, single GPU: see LmNext
  // k_backsub_candidate_proj with a communicator: the last workgroup leaves this rank's sums (and stall flags) in small_red for
  // the all-reduce and touches neither the result block nor the host — k_publish_result completes the step behind the collective
  int sums_only = 0;
};

template <bool kStage, bool kFused>
__global__ void __launch_bounds__(256)
k_backsub_candidate(int C, int P, ObsSliced obs,
                    const double* __restrict__ camc_xg, const double* __restrict__ camc_cg,
                    const double* __restrict__ dcam_g, const double* __restrict__ pts_x, double* __restrict__ pts_c,
                    const double* __restrict__ scale_p, double* __restrict__ block_part /* gridDim.x x 8 */, IterParams ip,
                    int* __restrict__ done_cnt, double* __restrict__ small_red, double* __restrict__ res, double* host, double seq,
                    const int* __restrict__ solve_done, int solve_tag, long long* __restrict__ waited, int* __restrict__ wait_timeout,
                    FusedLin fl) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  // camera constants at x and at the candidate, and the camera step: 70 doubles per camera, LDS-resident when they fit
  const double* camc_x = camc_xg;
  const double* camc_c = camc_cg;
  const double* dcam = dcam_g;
  // Pipelined schedule: the kernel is launched behind the Schur kernel, while the Cholesky is still running on its own
  // CU and stream; the chip is idle by then, so the workgroups just sit here (camera constants at x already staged) until
  // the solve publishes its tag — no cross-stream event, no launch latency after the solve.  0.5 s budget: never hang.
  auto wait_solve = [&]() {
    if (solve_done == nullptr) return;
    if (tid == 0) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(solve_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != solve_tag) {
        __builtin_amdgcn_s_sleep(127);
        // (the solve was not running beside us — e.g. its CU was taken: what follows is computed from stale data, and the
        //  host must repeat the step with the sequential schedule)
        if (wall_clock64() - t0 > RSBA_STALL_TICKS) { __hip_atomic_store(wait_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      if (waited != nullptr && blockIdx.x == 0) *waited += wall_clock64() - t0;   // the kernel's span minus this is its own work
      if (fl.trace != nullptr && blockIdx.x == 0) fl.trace[28] = wall_clock64();
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  };
  if (kStage) {
    double* lx = lds;
    double* lc = lds + (size_t)C * RSBA_CC_LDS;
    double* ld = lc + (size_t)C * RSBA_CC_LDS;
    // copies with eight loads in flight per thread (a plain copy loop waits for every load; what comes after the solve is
    // the tail of the LM step and was eight dependent round trips to the solve's output)
    auto stage_camc = [&](const double* __restrict__ src, double* dst) {
      for (int i0 = 0; i0 < C * CC_STRIDE; i0 += 8 * blockDim.x) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x + tid; v[u] = i < C * CC_STRIDE ? src[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x + tid; if (i < C * CC_STRIDE) { const int c = i / CC_STRIDE, e = i - c * CC_STRIDE; dst[c * RSBA_CC_LDS + e] = v[u]; } }
      }
    };
    stage_camc(camc_xg, lx);
    wait_solve();
    {
      double dv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int i = u * blockDim.x + tid; dv[u] = i < 6 * C ? dcam_g[i] : 0.0; }
      stage_camc(camc_cg, lc);
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int i = u * blockDim.x + tid; if (i < 6 * C) ld[i] = dv[u]; }
      for (int i = 2 * blockDim.x + tid; i < 6 * C; i += blockDim.x) ld[i] = dcam_g[i];
    }
    __syncthreads();
    camc_x = lx; camc_c = lc; dcam = ld;
  } else {
    wait_solve();
  }
  double mcc = 0, cost_c = 0, dp2 = 0, xc2 = 0, ss_c = 0;
  for (int j = blockIdx.x * blockDim.x + tid; j < P; j += gridDim.x * blockDim.x) {
    const int lane = j & 63, tb = obs.row_ptr[j >> 6], te = obs.row_ptr[(j >> 6) + 1];
    bool any = false;
    const double X[3] = {pts_x[3 * (size_t)j], pts_x[3 * (size_t)j + 1], pts_x[3 * (size_t)j + 2]};
    double V[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, bv[3] = {0, 0, 0}, a1 = 0, a2 = 0;
    if (kFused) {
      const double* ln = fl.lin_x + (size_t)j * RSBA_LIN_STRIDE;
#pragma unroll
      for (int i = 0; i < 6; ++i) V[i] = ln[i];
      gp[0] = ln[6]; gp[1] = ln[7]; gp[2] = ln[8];
    }
    // observation records four slots ahead of the one in use (k_point_pass: a thread's records are otherwise a chain of
    // dependent round trips to memory; the grid is 1.5 workgroups per CU)
    int camq[4]; double2 uvq[4];
    auto fill = [&](int u, int t) {
      const size_t qq = (size_t)t * 64 + lane;
      camq[u] = t < te ? obs.cam[qq] : -1;
      uvq[u] = t < te ? obs.uv[qq] : make_double2(0.0, 0.0);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) fill(u, tb + u);
    for (int t = tb; t < te; ++t) {
      const int cam = camq[0];
      const double2 uv = uvq[0];
#pragma unroll
      for (int u = 0; u < 3; ++u) { camq[u] = camq[u + 1]; uvq[u] = uvq[u + 1]; }
      fill(3, t + 4);
      if (cam < 0) continue;
      any = true;
      double r[2], jc[12], jp[6];
      ResidualJacobian(camc_x + (size_t)cam * (kStage ? RSBA_CC_LDS : CC_STRIDE), X, uv.x, uv.y, r, jc, jp);
      double sq;
      LossAndScale(ip.huber_delta, r[0] * r[0] + r[1] * r[1], &sq);
      const double* dc = dcam + 6 * cam;
      double e0 = 0, e1 = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) { e0 += jc[a] * dc[a]; e1 += jc[6 + a] * dc[a]; }
      if (sq != 1.0) {
        r[0] *= sq; r[1] *= sq; e0 *= sq; e1 *= sq;
#pragma unroll
        for (int i = 0; i < 6; ++i) jp[i] *= sq;
      }
      if (!kFused) {
        V[0] += jp[0] * jp[0] + jp[3] * jp[3]; V[1] += jp[0] * jp[1] + jp[3] * jp[4]; V[2] += jp[0] * jp[2] + jp[3] * jp[5];
        V[3] += jp[1] * jp[1] + jp[4] * jp[4]; V[4] += jp[1] * jp[2] + jp[4] * jp[5]; V[5] += jp[2] * jp[2] + jp[5] * jp[5];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) { if (!kFused) gp[a] += jp[a] * r[0] + jp[3 + a] * r[1]; bv[a] += jp[a] * e0 + jp[3 + a] * e1; }
      a1 += e0 * r[0] + e1 * r[1];
      a2 += e0 * e0 + e1 * e1;
    }
    const double sp[3] = {scale_p[3 * (size_t)j], scale_p[3 * (size_t)j + 1], scale_p[3 * (size_t)j + 2]};
    double Vi[6];
    const bool ok = PointBlockInverse(V, sp, ip.min_lm_diagonal, ip.max_lm_diagonal, ip.radius, Vi);
    double t[3] = {gp[0] + bv[0], gp[1] + bv[1], gp[2] + bv[2]}, dp[3] = {0, 0, 0};
    if (ok && any) { Sym3MulVec(Vi, t, dp); dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2]; }
    const double Xc[3] = {X[0] + dp[0], X[1] + dp[1], X[2] + dp[2]};
    pts_c[3 * (size_t)j] = Xc[0]; pts_c[3 * (size_t)j + 1] = Xc[1]; pts_c[3 * (size_t)j + 2] = Xc[2];
    double Vd[3];
    Sym3MulVec(V, dp, Vd);
    mcc -= a1 + (dp[0] * gp[0] + dp[1] * gp[1] + dp[2] * gp[2]) + 0.5 * a2 + (dp[0] * bv[0] + dp[1] * bv[1] + dp[2] * bv[2]) +
           0.5 * (dp[0] * Vd[0] + dp[1] * Vd[1] + dp[2] * Vd[2]);
    dp2 += dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
    xc2 += Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2];
    double Vc[6] = {0, 0, 0, 0, 0, 0}, gc[3] = {0, 0, 0}, costj_c = 0.0;   // the candidate's linearisation (kFused)
#pragma unroll
    for (int u = 0; u < 4; ++u) fill(u, tb + u);
    for (int t = tb; t < te; ++t) {
      const int cam = camq[0];
      const double2 uv = uvq[0];
#pragma unroll
      for (int u = 0; u < 3; ++u) { camq[u] = camq[u + 1]; uvq[u] = uvq[u + 1]; }
      fill(3, t + 4);
      if (cam < 0) continue;
      if (kFused) {
        double r[2], jp[6], sq;
        ResidualPointJacobian(camc_c + (size_t)cam * (kStage ? RSBA_CC_LDS : CC_STRIDE), Xc, uv.x, uv.y, r, jp);
        const double s = r[0] * r[0] + r[1] * r[1];
        costj_c += LossAndScale(ip.huber_delta, s, &sq);
        ss_c += s;
        if (ip.huber_delta != 0.0) fl.sq_cm_c[fl.cm_pos[(size_t)t * 64 + lane]] = sq;
        if (sq != 1.0) {
          r[0] *= sq; r[1] *= sq;
#pragma unroll
          for (int i = 0; i < 6; ++i) jp[i] *= sq;
        }
        Vc[0] += jp[0] * jp[0] + jp[3] * jp[3]; Vc[1] += jp[0] * jp[1] + jp[3] * jp[4]; Vc[2] += jp[0] * jp[2] + jp[3] * jp[5];
        Vc[3] += jp[1] * jp[1] + jp[4] * jp[4]; Vc[4] += jp[1] * jp[2] + jp[4] * jp[5]; Vc[5] += jp[2] * jp[2] + jp[5] * jp[5];
        gc[0] += jp[0] * r[0] + jp[3] * r[1]; gc[1] += jp[1] * r[0] + jp[4] * r[1]; gc[2] += jp[2] * r[0] + jp[5] * r[1];
      } else {
        double r[2];
        Residual(camc_c + (size_t)cam * (kStage ? RSBA_CC_LDS : CC_STRIDE), Xc, uv.x, uv.y, r);
        const double s = r[0] * r[0] + r[1] * r[1];
        double sq;
        costj_c += LossAndScale(ip.huber_delta, s, &sq);
        ss_c += s;
      }
    }
    cost_c += costj_c;
    if (kFused) {
      double* ln = fl.lin_c + (size_t)j * RSBA_LIN_STRIDE;
#pragma unroll
      for (int i = 0; i < 6; ++i) ln[i] = Vc[i];
      ln[6] = gc[0]; ln[7] = gc[1]; ln[8] = gc[2]; ln[9] = costj_c;
    }
  }
  // block reduction in a fixed order
  __shared__ double s[5][256];
  s[0][tid] = mcc; s[1][tid] = cost_c; s[2][tid] = dp2; s[3][tid] = xc2; s[4][tid] = ss_c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int q = 0; q < 5; ++q) s[q][tid] += s[q][tid + off];
    }
    __syncthreads();
  }
  // (agent-scope store: the last workgroup may sit on another XCD, and a full fence per workgroup costs an L2 write-back)
  if (tid < 5) __hip_atomic_store(&block_part[8 * blockIdx.x + tid], s[tid][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Single GPU (done_cnt != nullptr): the last workgroup to arrive adds the per-block partials (in block order, so the
  // sums do not depend on who is last) and posts the result: no second launch.  With RCCL the sums have to be
  // all-reduced first, and k_finish_candidate / k_publish_result do it.
  if (done_cnt != nullptr) {
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      s_last = __hip_atomic_fetch_add(done_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
      if (s_last) __hip_atomic_store(done_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (tid == 0 && wait_timeout != nullptr) {
        res[RES_WAIT_TIMEOUT] = (double)__hip_atomic_load(wait_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(wait_timeout, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      FinishCandidate((int)gridDim.x, block_part, small_red, res, host, seq);
      if (fl.trace != nullptr && tid == 0) fl.trace[29] = wall_clock64();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K_B, projective form (the default when the point linearisation is kept between steps and the cameras' rows fit LDS).
//
// k_backsub_candidate above took 53 us of the 0.465 ms step at 64 cameras x 100k points, all of them behind the solve: per
// observation and pass it read ~25 camera constants from LDS (R, Jl, t, intrinsics, the camera step; every lane its own
// camera, so nothing is broadcast: 16 KB per observation slot and wavefront, 13 us of LDS time per CU with two workgroups
// and no bank conflict, three times that measured) and spent ~135 instructions on residual, both Jacobian blocks and
// Jc dc.  Nothing of that needs the camera block itself:
//   h = A [X; 1],  A = [fx R0 | fx t0 ;  fy R1 | fy t1 ;  R2 | t2]   (3 x 4 per camera)
//   r = (h0 / h2 + ppx - u, h1 / h2 + ppy - v),      d r_i / d X = (A_i - (h_i / h2) A_2)[0:3] / h2
// and the change of r along the camera step dc = (dw, dt) is the same expression with the DERIVATIVE of h along the step,
//   dh = B [X; 1],  B = the rows of A with R replaced by [theta]x W and t by dt,   theta = Jl(w) dw,
//   W = R (Rodrigues branch: d(R X) = theta x (R X)) or I (AngleAxisRotatePoint's first-order branch: d = theta x X),
//   Jc dc = ((dh0 - (h0 / h2) dh2) / h2, (dh1 - (h1 / h2) dh2) / h2)
// — 26 doubles per camera for the pass at x, 14 for the pass at the candidate (the principal point kept apart: folded
// into A it would cancel again in every Jacobian entry), built once per workgroup from the camera
// constants (A before the solve's flag is up, B and the candidate's A behind it).  Per observation:
// ~56 + ~51 instructions instead of ~2 x 135; the division is v_rcp_f64 + two Newton steps (RcpNewton).  The first eight
// observation records, the point and its linearisation are in flight before the kernel waits for the solve.
// Same sums, same order over a point's observations; the results differ from the other form in the last bits only.
// ------------------------------------------------------------------------------------------------
#define RSBA_PJ_NX 26      // rows of the table at x: A (12), B (12), ppx, ppy
#define RSBA_PJ_NC 16      // rows of the table at the candidate: R (9), t (3), fx, fy, ppx, ppy — ProjectResidual's operands
#define RSBA_BS_REG_LOSS 9  // observation records a lane of a robust instance keeps in registers (ten without a loss)

// (1 / x: RcpNewton, ba_math.hpp)

// Loads of what the solve has just published (camera step, candidate's camera constants): agent scope, so that they
// are served by the memory side without an acquire fence — a fence per workgroup invalidates the XCD's L2 each time, and
// 391 of them at the same moment made this stage 10 us long.
__device__ __forceinline__ double LoadFresh(const double* p) {
  return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One entry per ds_read_b64.  The compiler pairs neighbouring entries into ds_read2st64_b64, which the LDS serves in four
// groups of 16 lanes, banked modulo 32 dwords, at half the bytes per clock (MI355X_MICROARCH.md, LDS): 16 lanes with 16
// different cameras conflict two- to three-fold, and the 13 paired reads of a slot cost ~260 LDS cycles per wavefront —
// with six wavefronts per CU the passes ran at ~1400 cycles per slot, all of it LDS time.  ds_read_b64 is served in two
// groups of 32 lanes, banked modulo 64: conflict-free while the cameras of a half-wave differ by less than 32.
#define RSBA_LDS_RD(dst, k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "i"((k) * kCpad * 8))
template <int kCpad>
__device__ __forceinline__ void ReadRows26(unsigned ad, double* a) {
  RSBA_LDS_RD(a[0], 0); RSBA_LDS_RD(a[1], 1); RSBA_LDS_RD(a[2], 2); RSBA_LDS_RD(a[3], 3); RSBA_LDS_RD(a[4], 4); RSBA_LDS_RD(a[5], 5);
  RSBA_LDS_RD(a[6], 6); RSBA_LDS_RD(a[7], 7); RSBA_LDS_RD(a[8], 8); RSBA_LDS_RD(a[9], 9); RSBA_LDS_RD(a[10], 10); RSBA_LDS_RD(a[11], 11);
  RSBA_LDS_RD(a[12], 12); RSBA_LDS_RD(a[13], 13); RSBA_LDS_RD(a[14], 14); RSBA_LDS_RD(a[15], 15); RSBA_LDS_RD(a[16], 16); RSBA_LDS_RD(a[17], 17);
  RSBA_LDS_RD(a[18], 18); RSBA_LDS_RD(a[19], 19); RSBA_LDS_RD(a[20], 20); RSBA_LDS_RD(a[21], 21); RSBA_LDS_RD(a[22], 22); RSBA_LDS_RD(a[23], 23);
  RSBA_LDS_RD(a[24], 24); RSBA_LDS_RD(a[25], 25);
  // (the values are operands of the wait: nothing that uses them moves above it)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
               "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]));
  asm volatile("" : "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]),
               "+v"(a[22]), "+v"(a[23]), "+v"(a[24]), "+v"(a[25]));
}
template <int kCpad>
__device__ __forceinline__ void ReadRows16(unsigned ad, double* a) {
  RSBA_LDS_RD(a[0], 0); RSBA_LDS_RD(a[1], 1); RSBA_LDS_RD(a[2], 2); RSBA_LDS_RD(a[3], 3); RSBA_LDS_RD(a[4], 4); RSBA_LDS_RD(a[5], 5);
  RSBA_LDS_RD(a[6], 6); RSBA_LDS_RD(a[7], 7); RSBA_LDS_RD(a[8], 8); RSBA_LDS_RD(a[9], 9); RSBA_LDS_RD(a[10], 10); RSBA_LDS_RD(a[11], 11);
  RSBA_LDS_RD(a[12], 12); RSBA_LDS_RD(a[13], 13); RSBA_LDS_RD(a[14], 14); RSBA_LDS_RD(a[15], 15);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
               "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}
#undef RSBA_LDS_RD

// LDS of a workgroup, in doubles: the two tables, then the records of kLds slots (a double2 and an int per lane and slot).
template <int kCpad, int kLds>
struct BacksubProjLds {
  static constexpr int kTables = (RSBA_PJ_NX + RSBA_PJ_NC) * kCpad;
  static constexpr int kUv = 4 * kLds * 64 * 2;     // [wave][slot][lane] double2
  static constexpr int kCam = 4 * kLds * 64 / 2;    // [wave][slot][lane] int
  static constexpr size_t kBytes = (size_t)(kTables + kUv + kCam) * sizeof(double);
};

// The tables are element-major — entry k of camera c at [k * kCpad + c] — so that the lanes of a wavefront, each
// reading entry k of ITS camera, hit 64 different banks as long as their cameras differ by less than 32 (the records of a
// point are in camera order, so the cameras of one slot cluster), the same camera is a broadcast, and k is an immediate
// offset.
//
// Observation records: a lane's first kReg slots stay in registers, the next kLds in LDS (lane-private: [wave]
// [slot][lane]); all of them are fetched BEFORE the kernel waits for the solve and serve both passes.  (All twenty in
// registers do not fit beside the passes' ~190 registers with two workgroups per CU: the compiler put the pixel pairs into
// scratch and fetched one per slot, each a trip to memory — 16 + 20 us for the two passes.)  Slots beyond that are
// streamed from memory (correct, slow: points seen by more than kReg + kLds cameras of a slice).
//
// Work is dealt by SLICE (64 points, one wavefront), wave w of workgroup b taking slices b + G (w + 4 i): with G = two
// workgroups per CU every CU gets six or seven wavefronts of work; one workgroup per 256 points is 1.5 per CU, i.e. half
// the CUs with twice the LDS traffic of the others.
template <int kCpad, int kReg, int kLds, bool kLoss>
__global__ void __launch_bounds__(256, kCpad <= 128 ? 2 : 1)
k_backsub_candidate_proj(int C, int P, ObsSliced obs,
                         const double* __restrict__ camc_xg, const double* __restrict__ camc_cg,
                         const double* __restrict__ dcam_g, const double* __restrict__ pts_x, double* __restrict__ pts_c,
                         const double* __restrict__ scale_p, double* __restrict__ block_part /* gridDim.x x 8 */, IterParams ip,
                         int* __restrict__ done_cnt, double* __restrict__ small_red, double* __restrict__ res, double* host, double seq,
                         const int* __restrict__ solve_done, int solve_tag, long long* __restrict__ waited, int* __restrict__ wait_timeout,
                         FusedLin fl) {
  using L = BacksubProjLds<kCpad, kLds>;
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  double* lx = lds;                                  // [RSBA_PJ_NX][kCpad]
  double* lc = lds + RSBA_PJ_NX * kCpad;             // [RSBA_PJ_NC][kCpad]
  double2* uvl = reinterpret_cast<double2*>(lds + L::kTables) + (size_t)wv * kLds * 64 + lane;   // + slot * 64
  int* caml = reinterpret_cast<int*>(lds + L::kTables + L::kUv) + (size_t)wv * kLds * 64 + lane;
  // Until B is built, what it takes of the constants at x — Jl (9), W (9), fx, fy: 20 rows — sits in the rows of the
  // candidate's table (0..13) and of B (14..19 -> rows 12..17 of the table at x).  Thread c reads column c of all of it
  // before it writes column c of B and of the candidate's table, so nothing is overwritten unread.
  auto lk = [&](int k, int c) -> double& { return k < RSBA_PJ_NC ? lc[k * kCpad + c] : lx[(12 + k - RSBA_PJ_NC) * kCpad + c]; };
  const int nslices = (P + 63) >> 6;
  int slice = blockIdx.x + (int)gridDim.x * wv;
  int j = slice * 64 + lane;
  // --- everything that does not depend on the solve: the cameras' rows at x, what B takes of the constants at x, then this
  // wavefront's first slice — points, their linearisation, and the observation records of every lane (the workgroups sit
  // here for ~100 us of the pipelined step: whatever they hold by then is not fetched behind the solve, where the step
  // waits for this kernel alone)
  for (int c = tid; c < C; c += blockDim.x) {
    const double* cc = camc_xg + (size_t)c * CC_STRIDE;
    const double fx = cc[CC_FX], fy = cc[CC_FY];
    const bool small = cc[CC_SMALL] != 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      lx[i * kCpad + c] = fx * cc[CC_R + i];
      lx[(4 + i) * kCpad + c] = fy * cc[CC_R + 3 + i];
      lx[(8 + i) * kCpad + c] = cc[CC_R + 6 + i];
    }
    lx[3 * kCpad + c] = fx * cc[CC_T];
    lx[7 * kCpad + c] = fy * cc[CC_T + 1];
    lx[11 * kCpad + c] = cc[CC_T + 2];
    lx[24 * kCpad + c] = cc[CC_PPX];
    lx[25 * kCpad + c] = cc[CC_PPY];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      lk(i, c) = cc[CC_K + i];
      lk(9 + i, c) = small ? ((i == 0 || i == 4 || i == 8) ? 1.0 : 0.0) : cc[CC_R + i];   // W: R, or I in the first-order branch
    }
    lk(18, c) = fx; lk(19, c) = fy;
  }
  __builtin_amdgcn_sched_barrier(0);
  int tb = 0, te = 0;
  double X[3] = {0, 0, 0}, V[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, sp[3] = {1, 1, 1};
  int camq[kReg]; double2 uvq[kReg];
  auto load_point = [&]() {
    tb = obs.row_ptr[slice]; te = obs.row_ptr[slice + 1];
    if (j < P) {
      const double* ln = fl.lin_x + (size_t)j * RSBA_LIN_STRIDE;
#pragma unroll
      for (int i = 0; i < 6; ++i) V[i] = ln[i];
      gp[0] = ln[6]; gp[1] = ln[7]; gp[2] = ln[8];
#pragma unroll
      for (int i = 0; i < 3; ++i) { X[i] = pts_x[3 * (size_t)j + i]; sp[i] = scale_p[3 * (size_t)j + i]; }
    }
    // (the slots of a slice's missing points are padding: cam < 0)
#pragma unroll
    for (int u = 0; u < kReg; ++u) {
      const size_t qq = (size_t)(tb + u) * 64 + lane;
      camq[u] = tb + u < te ? obs.cam[qq] : -1;
      uvq[u] = tb + u < te ? obs.uv[qq] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u0 = 0; u0 < kLds; u0 += 5) {
      int cv[5]; double2 uv[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int t = tb + kReg + u0 + u;
        const size_t qq = (size_t)t * 64 + lane;
        cv[u] = (u0 + u < kLds && t < te) ? obs.cam[qq] : -1;
        uv[u] = (u0 + u < kLds && t < te) ? obs.uv[qq] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < 5; ++u) if (u0 + u < kLds) { caml[(u0 + u) * 64] = cv[u]; uvl[(u0 + u) * 64] = uv[u]; }
    }
  };
  if (slice < nslices) load_point();
  // --- the solve (pipelined schedule: this kernel is launched behind the Schur kernel and sits here until the
  // factorisation's workgroup 0 has published the camera step; see k_backsub_candidate)
  if (solve_done != nullptr) {
    if (tid == 0) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(solve_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != solve_tag) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > RSBA_STALL_TICKS) { __hip_atomic_store(wait_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      if (waited != nullptr && blockIdx.x == 0) *waited += wall_clock64() - t0;
      if (fl.trace != nullptr && blockIdx.x == 0) fl.trace[28] = wall_clock64();
      if (fl.bs_wg != nullptr) fl.bs_wg[4 * blockIdx.x] = wall_clock64();
    }
  }
  __syncthreads();
  // the solve's half of the result block, in registers of every workgroup: whoever turns out to be the last one completes
  // and posts it without another trip to memory for it
  const double res_pre = (done_cnt != nullptr && tid < RES_SIZE) ? LoadFresh(res + tid) : 0.0;
  // (sequential schedule: the kernel boundary has made the solve's output visible; the agent-scope loads cost nothing extra)
  for (int c = tid; c < C; c += blockDim.x) {
    // this camera's column of the constants at x, the camera step, the candidate's constants — all loads first
    double kk[20], dc[6], v[16];
    const double* cc = camc_cg + (size_t)c * CC_STRIDE;
#pragma unroll
    for (int i = 0; i < 6; ++i) dc[i] = LoadFresh(dcam_g + 6 * (size_t)c + i);
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = LoadFresh(cc + CC_R + i);
#pragma unroll
    for (int i = 0; i < 3; ++i) v[9 + i] = LoadFresh(cc + CC_T + i);
    v[12] = LoadFresh(cc + CC_FX); v[13] = LoadFresh(cc + CC_FY); v[14] = LoadFresh(cc + CC_PPX); v[15] = LoadFresh(cc + CC_PPY);
#pragma unroll
    for (int i = 0; i < 20; ++i) kk[i] = lk(i, c);
    // B: the rows of A with R -> [theta]x W, t -> dt;  theta = Jl dw
    const double th0 = kk[0] * dc[0] + kk[1] * dc[1] + kk[2] * dc[2];
    const double th1 = kk[3] * dc[0] + kk[4] * dc[1] + kk[5] * dc[2];
    const double th2 = kk[6] * dc[0] + kk[7] * dc[1] + kk[8] * dc[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double w0 = kk[9 + i], w1 = kk[12 + i], w2 = kk[15 + i];   // column i of W
      lx[(12 + i) * kCpad + c] = kk[18] * (th1 * w2 - th2 * w1);
      lx[(16 + i) * kCpad + c] = kk[19] * (th2 * w0 - th0 * w2);
      lx[(20 + i) * kCpad + c] = th0 * w1 - th1 * w0;
    }
    lx[15 * kCpad + c] = kk[18] * dc[3];
    lx[19 * kCpad + c] = kk[19] * dc[4];
    lx[23 * kCpad + c] = dc[5];
    // the candidate's constants as ProjectResidual takes them
#pragma unroll
    for (int i = 0; i < 12; ++i) lc[i * kCpad + c] = v[i];
    lc[12 * kCpad + c] = v[12];
    lc[13 * kCpad + c] = v[13];
    lc[14 * kCpad + c] = v[14];
    lc[15 * kCpad + c] = v[15];
  }
  __syncthreads();
  if (fl.trace != nullptr && blockIdx.x == 0 && tid == 0) fl.trace[32] = wall_clock64();
  if (fl.bs_wg != nullptr && tid == 0) fl.bs_wg[4 * blockIdx.x + 1] = wall_clock64();
  const unsigned lx_ad = (unsigned)(size_t)(lds_double*)lx, lc_ad = (unsigned)(size_t)(lds_double*)lc;   // LDS byte addresses
  double mcc = 0, cost_c = 0, dp2 = 0, xc2 = 0, ss_c = 0;
  for (bool first = true; slice < nslices; slice += 4 * (int)gridDim.x, j = slice * 64 + lane, first = false) {
    if (!first) load_point();
    const int nslot = te - tb;   // the same for the 64 lanes
    bool any = false;
    double bv[3] = {0, 0, 0}, a1 = 0, a2 = 0;
    // pass at x: b = sum Jp' (Jc dc), a1 = sum (Jc dc)' r, a2 = sum |Jc dc|^2
    // (no branch around a slot: a padding record — cam < 0 — reads camera 0 and its contribution is selected away; with a
    //  branch per slot the compiler merges the accumulators behind it with ~40 register copies per slot)
    auto at_x = [&](int cam, double2 uv) {
      const bool valid = cam >= 0;
      if (kLoss && !valid) return;   // (the robust variant has no registers to spare for the select form: it branches)
      any = any || valid;
      double av[RSBA_PJ_NX];
      ReadRows26<kCpad>(lx_ad + 8u * (unsigned)(valid ? cam : 0), av);
#define RSBA_A(k) av[k]
      const double h0 = fma(RSBA_A(0), X[0], fma(RSBA_A(1), X[1], fma(RSBA_A(2), X[2], RSBA_A(3))));
      const double h1 = fma(RSBA_A(4), X[0], fma(RSBA_A(5), X[1], fma(RSBA_A(6), X[2], RSBA_A(7))));
      const double h2 = fma(RSBA_A(8), X[0], fma(RSBA_A(9), X[1], fma(RSBA_A(10), X[2], RSBA_A(11))));
      const double d0 = fma(RSBA_A(12), X[0], fma(RSBA_A(13), X[1], fma(RSBA_A(14), X[2], RSBA_A(15))));
      const double d1 = fma(RSBA_A(16), X[0], fma(RSBA_A(17), X[1], fma(RSBA_A(18), X[2], RSBA_A(19))));
      const double d2 = fma(RSBA_A(20), X[0], fma(RSBA_A(21), X[1], fma(RSBA_A(22), X[2], RSBA_A(23))));
      const double iz = RcpNewton(h2);
      const double pr0 = h0 * iz, pr1 = h1 * iz;
      double r[2] = {pr0 + RSBA_A(24) - uv.x, pr1 + RSBA_A(25) - uv.y};
      double jp[6] = {fma(-pr0, RSBA_A(8), RSBA_A(0)) * iz, fma(-pr0, RSBA_A(9), RSBA_A(1)) * iz, fma(-pr0, RSBA_A(10), RSBA_A(2)) * iz,
                      fma(-pr1, RSBA_A(8), RSBA_A(4)) * iz, fma(-pr1, RSBA_A(9), RSBA_A(5)) * iz, fma(-pr1, RSBA_A(10), RSBA_A(6)) * iz};
      double e0 = fma(-pr0, d2, d0) * iz, e1 = fma(-pr1, d2, d1) * iz;
      if (kLoss) {
        double sq;
        LossAndScale(ip.huber_delta, r[0] * r[0] + r[1] * r[1], &sq);
        if (sq != 1.0) {
          r[0] *= sq; r[1] *= sq; e0 *= sq; e1 *= sq;
#pragma unroll
          for (int i = 0; i < 6; ++i) jp[i] *= sq;
        }
      }
      // every term of the three sums carries e0 or e1
      e0 = valid ? e0 : 0.0; e1 = valid ? e1 : 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) bv[a] += jp[a] * e0 + jp[3 + a] * e1;
      a1 += e0 * r[0] + e1 * r[1];
      a2 += e0 * e0 + e1 * e1;
    };
#pragma unroll
    for (int u = 0; u < kReg; ++u) { at_x(camq[u], uvq[u]); __builtin_amdgcn_sched_barrier(0); }
    {
      const int nl = min(kLds, nslot - kReg);
#pragma unroll 1
      for (int u = 0; u < nl; ++u) at_x(caml[u * 64], uvl[u * 64]);
    }
    for (int t = tb + kReg + kLds; t < te; ++t) {   // (points with more views than a lane keeps)
      const size_t qq = (size_t)t * 64 + lane;
      at_x(obs.cam[qq], obs.uv[qq]);
    }
    if (fl.trace != nullptr && blockIdx.x == 0 && tid == 0) fl.trace[33] = wall_clock64();
    if (fl.bs_wg != nullptr && tid == 0) fl.bs_wg[4 * blockIdx.x + 2] = wall_clock64();
    double Vi[6];
    const bool ok = PointBlockInverse(V, sp, ip.min_lm_diagonal, ip.max_lm_diagonal, ip.radius, Vi);
    double tv[3] = {gp[0] + bv[0], gp[1] + bv[1], gp[2] + bv[2]}, dp[3] = {0, 0, 0};
    if (ok && any) { Sym3MulVec(Vi, tv, dp); dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2]; }
    const double Xc[3] = {X[0] + dp[0], X[1] + dp[1], X[2] + dp[2]};
    double Vc[6] = {0, 0, 0, 0, 0, 0}, gc[3] = {0, 0, 0}, costj_c = 0.0;
    if (j < P) {
      pts_c[3 * (size_t)j] = Xc[0]; pts_c[3 * (size_t)j + 1] = Xc[1]; pts_c[3 * (size_t)j + 2] = Xc[2];
      double Vd[3];
      Sym3MulVec(V, dp, Vd);
      mcc -= a1 + (dp[0] * gp[0] + dp[1] * gp[1] + dp[2] * gp[2]) + 0.5 * a2 + (dp[0] * bv[0] + dp[1] * bv[1] + dp[2] * bv[2]) +
             0.5 * (dp[0] * Vd[0] + dp[1] * Vd[1] + dp[2] * Vd[2]);
      dp2 += dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
      xc2 += Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2];
    }
    // pass at the candidate: its cost and its linearisation
    auto at_c = [&](int cam, double2 uv, int t) {
      const bool valid = cam >= 0;
      if (kLoss && !valid) return;
      double av[RSBA_PJ_NC];
      ReadRows16<kCpad>(lc_ad + 8u * (unsigned)(valid ? cam : 0), av);
#undef RSBA_A
      // the residual every gradient of the next iteration is made of: ProjectResidual's roundings, not the projective rows'
      double p[3], iz, r[2];
      ProjectResidual(av, av + 9, av[12], av[13], av[14], av[15], Xc, uv.x, uv.y, p, &iz, r);
      if (!valid) { iz = 0.0; r[0] = 0.0; r[1] = 0.0; }   // a padding record: zero rows, and a zero residual
      const double pr0 = p[0] * iz, pr1 = p[1] * iz, al = av[12] * iz, be = av[13] * iz;
      double jp[6] = {fma(-pr0, av[6], av[0]) * al, fma(-pr0, av[7], av[1]) * al, fma(-pr0, av[8], av[2]) * al,
                      fma(-pr1, av[6], av[3]) * be, fma(-pr1, av[7], av[4]) * be, fma(-pr1, av[8], av[5]) * be};
      const double s = r[0] * r[0] + r[1] * r[1];
      ss_c += s;
      if (kLoss) {
        double sq;
        const double rho = LossAndScale(ip.huber_delta, s, &sq);
        costj_c += valid ? rho : 0.0;
        if (valid) fl.sq_cm_c[fl.cm_pos[(size_t)t * 64 + lane]] = sq;
        if (sq != 1.0) {
          r[0] *= sq; r[1] *= sq;
#pragma unroll
          for (int i = 0; i < 6; ++i) jp[i] *= sq;
        }
      } else {
        costj_c += s;
      }
      Vc[0] += jp[0] * jp[0] + jp[3] * jp[3]; Vc[1] += jp[0] * jp[1] + jp[3] * jp[4]; Vc[2] += jp[0] * jp[2] + jp[3] * jp[5];
      Vc[3] += jp[1] * jp[1] + jp[4] * jp[4]; Vc[4] += jp[1] * jp[2] + jp[4] * jp[5]; Vc[5] += jp[2] * jp[2] + jp[5] * jp[5];
      gc[0] += jp[0] * r[0] + jp[3] * r[1]; gc[1] += jp[1] * r[0] + jp[4] * r[1]; gc[2] += jp[2] * r[0] + jp[5] * r[1];
    };
#pragma unroll
    for (int u = 0; u < kReg; ++u) { at_c(camq[u], uvq[u], tb + u); __builtin_amdgcn_sched_barrier(0); }
    {
      const int nl = min(kLds, nslot - kReg);
#pragma unroll 1
      for (int u = 0; u < nl; ++u) at_c(caml[u * 64], uvl[u * 64], tb + kReg + u);
    }
    for (int t = tb + kReg + kLds; t < te; ++t) {
      const size_t qq = (size_t)t * 64 + lane;
      at_c(obs.cam[qq], obs.uv[qq], t);
    }
    if (j < P) {
      cost_c += costj_c;
      double* ln = fl.lin_c + (size_t)j * RSBA_LIN_STRIDE;
#pragma unroll
      for (int i = 0; i < 6; ++i) ln[i] = Vc[i];
      ln[6] = gc[0]; ln[7] = gc[1]; ln[8] = gc[2]; ln[9] = costj_c;
    }
  }
  // block reduction in a fixed order (in the records' LDS, once every wavefront is through with its own), then (single
  // GPU) the last workgroup adds the blocks and posts the result
  if (fl.trace != nullptr && blockIdx.x == 0 && tid == 0) fl.trace[34] = wall_clock64();
  static_assert(L::kUv >= 7 * 256, "the block sums reuse the records' LDS");
  __syncthreads();
  double (*s)[256] = reinterpret_cast<double (*)[256]>(lds + L::kTables);
  {
    // every wavefront its 64 sums by butterfly, thread q the four wavefronts' sums of quantity q
    const double w5[5] = {WaveSum(mcc), WaveSum(cost_c), WaveSum(dp2), WaveSum(xc2), WaveSum(ss_c)};
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 5; ++q) s[q][wv] = w5[q];
    }
  }
  __syncthreads();
  if (tid < 5) s[tid][0] = ((s[tid][0] + s[tid][1]) + s[tid][2]) + s[tid][3];
  __syncthreads();
  if (tid < 5) __hip_atomic_store(&block_part[8 * blockIdx.x + tid], s[tid][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fl.bs_wg != nullptr && tid == 0) fl.bs_wg[4 * blockIdx.x + 3] = wall_clock64();
  if (done_cnt != nullptr) {
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      s_last = __hip_atomic_fetch_add(done_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
      if (s_last) __hip_atomic_store(done_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (fl.trace != nullptr && blockIdx.x == 0 && tid == 0) fl.trace[35] = wall_clock64();
    if (s_last) {
      if (fl.trace != nullptr && tid == 0) fl.trace[36] = wall_clock64();
      // the blocks' sums (agent-scope loads: no fence), thread t blocks t, t + 256, ...; butterflies; four wavefronts
      double v[5] = {0, 0, 0, 0, 0};
      for (int b = tid; b < (int)gridDim.x; b += 256) {
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] += LoadFresh(block_part + 8 * b + q);
      }
      const double timed_out = (tid == 0 && wait_timeout != nullptr) ? (double)__hip_atomic_load(wait_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      __syncthreads();   // (s is read above by nobody any more: the block sums went out before the counter)
#pragma unroll
      for (int q = 0; q < 5; ++q) { const double w = WaveSum(v[q]); if (lane == 0) s[q][wv] = w; }
      if (tid == 0) { s[5][0] = timed_out; if (wait_timeout != nullptr) __hip_atomic_store(wait_timeout, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      __syncthreads();
      if (tid < RES_SIZE - 1) {
        // PublishResult's arithmetic, one value per thread, on the copy of the solve's half taken behind the flag
        const double sm[5] = {((s[0][0] + s[0][1]) + s[0][2]) + s[0][3], ((s[1][0] + s[1][1]) + s[1][2]) + s[1][3], ((s[2][0] + s[2][1]) + s[2][2]) + s[2][3],
                              ((s[3][0] + s[3][1]) + s[3][2]) + s[3][3], ((s[4][0] + s[4][1]) + s[4][2]) + s[4][3]};
        if (tid < 5) small_red[tid] = sm[tid];
        if (fl.sums_only) {
          // summed over the ranks: a stall anywhere — the factorisation's in-kernel waits, this kernel's wait for the solve — is everybody's
          if (tid == RES_STALL) small_red[5] = res_pre + s[5][0];
        } else {
        double c = 0.5 * sm[1];
        if (!(c == c) || !(fabs(c) <= DBL_MAX)) c = DBL_MAX;  // Ceres: failed evaluation -> max double
        double out = res_pre;
        if (tid == RES_MCC) out = sm[0];
        else if (tid == RES_COST_C) out = c;
        else if (tid == RES_STEP2) out = res_pre + sm[2];
        else if (tid == RES_XCNORM2) out = res_pre + sm[3];
        else if (tid == RES_SUMSQ_C) out = sm[4];
        else if (tid == RES_WAIT_TIMEOUT && wait_timeout != nullptr) out = s[5][0];
        if (fl.lm.dec != nullptr) s[6][tid] = out;
        if (fl.lm.dec == nullptr || tid < RES_DEC_GO) {
          res[tid] = out;
          if (host != nullptr) __hip_atomic_store(&host[tid], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        }
      }
      if (fl.lm.dec != nullptr && !fl.sums_only) {
        __syncthreads();
        if (tid == 0) {
          double acc, nr;
          DecideStep(fl.lm, s[6][RES_COST_X], s[6][RES_COST_C], s[6][RES_MCC], s[6][RES_STEP2], s[6][RES_CHOL_OK], &acc, &nr);
          const double d3[3] = {1.0, acc, nr};
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            fl.lm.dec[q] = d3[q];
            res[RES_DEC_GO + q] = d3[q];
            if (host != nullptr) __hip_atomic_store(&host[RES_DEC_GO + q], d3[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          // whose decision it is: a factorisation launched ahead of the next step is already resident and waits for this (AheadSel)
          __hip_atomic_store(fl.lm.dec + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (host != nullptr && tid < 64 && !fl.sums_only) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the values before the sequence number
        if (tid == 0) __hip_atomic_store(&host[RES_SIZE - 1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (fl.trace != nullptr && tid == 0) fl.trace[29] = wall_clock64();
    }
  }
}

// Second stage: fixed-order sum of the per-block partials into the small reduction payload
// small[0..4] = {mcc, cost_c (sum rho), |dp|^2, |Xc|^2, sum sq residuals}
__device__ __forceinline__ void PublishResult(const double* __restrict__ small_red, double* __restrict__ res);
__device__ __forceinline__ void PostToHost(const double* __restrict__ res, double* host, double seq);
// One workgroup: small_red[0..4] = fixed-order sums of the per-block partials; with res != nullptr (single GPU) the
// result block is completed and posted to the host straight away.
// (s: 5 x 256 doubles of LDS the caller can spare)
// (WaveSum: a butterfly over the 64 lanes, the same value in every lane, a fixed order)
__device__ __forceinline__ void FinishCandidateIn(double (*s)[256], int nblocks, const double* __restrict__ block_part, double* __restrict__ small_red,
                                                  double* __restrict__ res, double* host, double seq, const double* res_stall) {
  const int tid = threadIdx.x;
  // the first four wavefronts add the blocks (thread t: blocks t, t + 256, ...), each wavefront its 64 sums by butterfly,
  // thread q the four wavefronts' sums of quantity q: two barriers instead of the nine of a 256-wide tree (this runs
  // behind the last workgroup of the step's last kernel)
  if (tid < 256) {
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = tid; i < nblocks; i += 256) {
#pragma unroll
      for (int q = 0; q < 5; ++q) v[q] += block_part[8 * i + q];
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) { const double w = WaveSum(v[q]); if ((tid & 63) == 0) s[q][tid >> 6] = w; }
  }
  __syncthreads();
  if (tid < 5) s[tid][0] = ((s[tid][0] + s[tid][1]) + s[tid][2]) + s[tid][3];
  __syncthreads();
  if (tid < 5) small_red[tid] = s[tid][0];
  if (tid == 5 && res_stall != nullptr) small_red[5] = res_stall[RES_STALL];   // summed over the ranks: a stall anywhere is everybody's
  if (res != nullptr) {
    __syncthreads();
    if (tid == 0) { double sr[5]; for (int q = 0; q < 5; ++q) sr[q] = s[q][0]; PublishResult(sr, res); }
    __syncthreads();
    PostToHost(res, host, seq);
  }
}
__device__ __forceinline__ void FinishCandidate(int nblocks, const double* __restrict__ block_part, double* __restrict__ small_red,
                                                double* __restrict__ res, double* host, double seq, const double* res_stall) {
  __shared__ double s[5][256];
  FinishCandidateIn(s, nblocks, block_part, small_red, res, host, seq, res_stall);
}
__global__ void __launch_bounds__(256)
k_finish_candidate(int nblocks, const double* __restrict__ block_part, double* __restrict__ small_red, double* __restrict__ res,
                   double* host, double seq, const double* res_stall, long long* trace = nullptr, int* wait_timeout = nullptr) {
  if (trace && threadIdx.x == 0) trace[26] = wall_clock64();
  FinishCandidate(nblocks, block_part, small_red, res, host, seq, res_stall);
  if (threadIdx.x == 5 && res_stall != nullptr && wait_timeout != nullptr) {   // multi-GPU pipeline: a timed-out wait counts as a stall
    small_red[5] += (double)__hip_atomic_load(wait_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(wait_timeout, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Folds the (all-reduced) point-side sums into the result block the host reads.
__device__ __forceinline__ void PublishResult(const double* __restrict__ small_red, double* __restrict__ res) {
  res[RES_MCC] = small_red[0];
  double c = 0.5 * small_red[1];
  if (!(c == c) || !(fabs(c) <= DBL_MAX)) c = DBL_MAX;  // Ceres: failed evaluation -> max double
  res[RES_COST_C] = c;
  res[RES_STEP2] += small_red[2];
  res[RES_XCNORM2] += small_red[3];
  res[RES_SUMSQ_C] = small_red[4];
}
// The step's result block goes straight into the host's pinned, coherent buffer, sequence number last: the host polls
// that word instead of paying for a copy kernel and a stream synchronisation per LM iteration.  Called by a whole
// wavefront or workgroup: fifteen lanes write one value each (serially they cost ~5 us of PCIe round trips), lane 0 the
// sequence number after them.
__device__ __forceinline__ void PostToHost(const double* __restrict__ res, double* host, double seq) {
  if (host == nullptr) return;
  const int tid = threadIdx.x;
  if (tid < 64) {
    if (tid < RES_SIZE - 1) __hip_atomic_store(&host[tid], res[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the values before the sequence number
    if (tid == 0) __hip_atomic_store(&host[RES_SIZE - 1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// (lm.dec != nullptr: the step's decision for the damping kernel queued behind this one, as the single-GPU back-substitution
//  takes it — here from the all-reduced sums, i.e. the same bits on every rank)
__global__ void k_publish_result(const double* __restrict__ small_red, double* __restrict__ res, double* host, double seq, int stall_summed,
                                 long long* trace = nullptr, LmNext lm = LmNext()) {
  if (trace && threadIdx.x == 0 && blockIdx.x == 0) trace[27] = wall_clock64();
  if (blockIdx.x != 0) return;
  if (threadIdx.x == 0) {
    PublishResult(small_red, res);
    if (stall_summed) { res[RES_STALL] = small_red[5]; res[RES_TIME_UP] = small_red[6]; }
    if (lm.dec != nullptr) {
      double acc, nr;
      DecideStep(lm, res[RES_COST_X], res[RES_COST_C], res[RES_MCC], res[RES_STEP2], res[RES_CHOL_OK], &acc, &nr);
      const double d3[3] = {1.0, acc, nr};
      for (int q = 0; q < 3; ++q) { lm.dec[q] = d3[q]; res[RES_DEC_GO + q] = d3[q]; }
    }
  }
  __syncthreads();
  PostToHost(res, host, seq);
}

// Gradient evaluation at x without a solve (TrustRegionMinimizer::HandleSuccessfulStep's EvaluateGradientAndJacobian when
// the run ends right after an accepted step): cost and max |gradient| from the linearisation payload, posted to the host.
__global__ void __launch_bounds__(256)
k_gradient_result(const double* __restrict__ red, RedLayout L, const double* __restrict__ gmax_p, double* __restrict__ res,
                  double* host, double seq) {
  __shared__ double sm[256];
  const int tid = threadIdx.x;
  double gm = 0.0;
  for (int i = tid; i < L.nc; i += 256) gm = fmax(gm, fabs(red[L.gc() + i]));
  sm[tid] = gm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (tid < off) sm[tid] = fmax(sm[tid], sm[tid + off]); __syncthreads(); }
  if (tid == 0) { res[RES_COST_X] = 0.5 * red[L.scal() + 0]; res[RES_GMAX] = fmax(*gmax_p, sm[0]); }
  __syncthreads();
  PostToHost(res, host, seq);
}

// Cost only at the current point-model parameters (used by rsba_reprojection_error).
__global__ void __launch_bounds__(256)
k_cost_only(int P, ObsSliced obs, const double* __restrict__ camc, const double* __restrict__ pts,
            double* __restrict__ block_part, double huber_delta) {
  const int tid = threadIdx.x;
  double cost = 0, ss = 0;
  for (int j = blockIdx.x * blockDim.x + tid; j < P; j += gridDim.x * blockDim.x) {
    const double X[3] = {pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]};
    const int lane = j & 63;
    for (int t = obs.row_ptr[j >> 6]; t < obs.row_ptr[(j >> 6) + 1]; ++t) {
      const size_t q = (size_t)t * 64 + lane;
      const int cam = obs.cam[q];
      if (cam < 0) continue;
      const double2 uv = obs.uv[q];
      double r[2], sq;
      Residual(camc + (size_t)cam * CC_STRIDE, X, uv.x, uv.y, r);
      const double s = r[0] * r[0] + r[1] * r[1];
      cost += LossAndScale(huber_delta, s, &sq);
      ss += s;
    }
  }
  __shared__ double s[2][256];
  s[0][tid] = cost; s[1][tid] = ss;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (tid < off) { s[0][tid] += s[0][tid + off]; s[1][tid] += s[1][tid + off]; } __syncthreads(); }
  if (tid == 0) { block_part[8 * blockIdx.x + 1] = s[0][0]; block_part[8 * blockIdx.x + 4] = s[1][0]; block_part[8 * blockIdx.x] = 0; block_part[8 * blockIdx.x + 2] = 0; block_part[8 * blockIdx.x + 3] = 0; }
}

}  // namespace rsba
