// Initial-guess math of the reference's front end without OpenCV: what
// /root/reference/Main_Calibration/correspondencer.cpp does between the ArUco detections and correspondence.txt.
//
//   BaseFromMarkerDetection   :119-127   pose of the base marker in the main camera from the detection of another marker
//   MarkerFromCamera          :137-147   pose of marker i in the main camera from the base pose and the board geometry
//   MarkerCornersInCamera     :5-39      GetCornersInCameraWorld: the four corners of a marker, top-left first
//   SolvePnPEPnP              :192-195   cv::solvePnP(..., SOLVEPNP_EPNP): EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009)
//                                        as OpenCV's calib3d/src/epnp.cpp runs it — control points by PCA, the 12 x 12
//                                        M'M, the three beta approximations with five Gauss-Newton steps each, Horn's
//                                        absolute orientation, the solution with the smallest reprojection error.
//                                        OpenCV is a dependency of the reference that is not vendored (no version pin in
//                                        the repository; 3.x/4.x epnp.cpp are the same algorithm): the restatement is
//                                        anchored on the reference's own output, the camera rows of the committed
//                                        Common/Correspondence/hongo/correspondence.txt (tests/test_initial_guess.py).
//                                        Zero lens distortion only (the committed intrinsics carry zeros; undistortPoints
//                                        followed by epnp's re-projection with K is then the identity on the pixels).
//
// Host code, double precision, no device work: the problem sizes are a few hundred points per camera.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "ba_problem.hpp"

namespace rsba {

namespace {

void MatMul3(const double* A, const double* B, double* C) {  // C = A B
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
void Transpose3(const double* A, double* T) {
  double R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = A[3 * j + i];
  memcpy(T, R, sizeof(R));
}
void MatVec3(const double* A, const double* x, double* y) {
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
  memcpy(y, t, sizeof(t));
}

// Cyclic Jacobi for a symmetric n x n matrix (row-major, destroyed): eigenvalues descending, eigenvectors as ROWS of V.
void JacobiEigenSym(int n, double* A, double* evals, double* V) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
    if (off <= 1e-60 || off <= 1e-32 * diag) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vpk = V[p * n + k], vqk = V[q * n + k];
          V[p * n + k] = c * vpk - s * vqk; V[q * n + k] = s * vpk + c * vqk;
        }
      }
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return A[a * n + a] > A[b * n + b]; });
  std::vector<double> Vs((size_t)n * n);
  for (int i = 0; i < n; ++i) { evals[i] = A[order[i] * n + order[i]]; memcpy(&Vs[(size_t)i * n], &V[(size_t)order[i] * n], n * sizeof(double)); }
  memcpy(V, Vs.data(), Vs.size() * sizeof(double));
}

// U' and the singular values of a 3 x 3 matrix exactly as cvSVD(A, W, U', 0, CV_SVD_MODIFY_A | CV_SVD_U_T) delivers them in
// OpenCV 4.0.1 (modules/core/src/lapack.cpp, JacobiSVDImpl_<double>: one-sided Hestenes Jacobi on the rows of A', pairs
// (i, j) in cyclic order, rotation angle from (a - b, 2p) with the branch on a < b, singular values sorted by selection).
// OpenCV is a third-party dependency of the reference and is not vendored; this restates the published routine for
// ONE reason: epnp.cpp takes the principal axes of the model points from this call (choose_control_points), their SIGNS
// are whatever the rotation sequence leaves, and with noisy detections the EPnP pose depends on them.  With these signs
// the camera rows the reference committed in Common/Correspondence/hongo/correspondence.txt (written by
// Correspondencer::CalculateTransforms, correspondencer.cpp:192-195) are reproduced to their six printed digits.
void JacobiSvdUt3(const double* A, double* Ut /* rows: left singular vectors */, double* W) {
  double At[9], w2[3];
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) At[3 * i + k] = A[3 * k + i];
  for (int i = 0; i < 3; ++i) w2[i] = At[3 * i] * At[3 * i] + At[3 * i + 1] * At[3 * i + 1] + At[3 * i + 2] * At[3 * i + 2];
  const double eps = DBL_EPSILON * 10;
  for (int iter = 0; iter < 30; ++iter) {
    bool changed = false;
    for (int i = 0; i < 2; ++i)
      for (int j = i + 1; j < 3; ++j) {
        double* Ai = At + 3 * i; double* Aj = At + 3 * j;
        double a = w2[i], b = w2[j], p = Ai[0] * Aj[0] + Ai[1] * Aj[1] + Ai[2] * Aj[2];
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = std::hypot(p, beta);
        double c, sn;
        if (beta < 0) { const double delta = (gamma - beta) * 0.5; sn = std::sqrt(delta / gamma); c = p / (gamma * sn * 2); }
        else { c = std::sqrt((gamma + beta) / (gamma * 2)); sn = p / (gamma * c * 2); }
        a = b = 0;
        for (int k = 0; k < 3; ++k) {
          const double t0 = c * Ai[k] + sn * Aj[k], t1 = -sn * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        w2[i] = a; w2[j] = b;
        changed = true;
      }
    if (!changed) break;
  }
  for (int i = 0; i < 3; ++i) W[i] = std::sqrt(At[3 * i] * At[3 * i] + At[3 * i + 1] * At[3 * i + 1] + At[3 * i + 2] * At[3 * i + 2]);
  for (int i = 0; i < 2; ++i) {
    int j = i;
    for (int k = i + 1; k < 3; ++k) if (W[j] < W[k]) j = k;
    if (i != j) { std::swap(W[i], W[j]); for (int k = 0; k < 3; ++k) std::swap(At[3 * i + k], At[3 * j + k]); }
  }
  for (int i = 0; i < 3; ++i) {
    const double sc = W[i] > DBL_MIN ? 1.0 / W[i] : 0.0;   // (a vanishing direction: coplanar points, refused further down)
    for (int k = 0; k < 3; ++k) Ut[3 * i + k] = At[3 * i + k] * sc;
  }
}

// Minimum-norm least squares x = argmin |A x - b|, A m x k (k <= 5), through the eigen-decomposition of A'A
// (what cvSolve(..., CV_SVD) returns for these well-posed 6 x k systems).
void LeastSquares(int m, int k, const double* A, const double* b, double* x) {
  double AtA[25], Atb[5], ev[5], V[25];
  for (int i = 0; i < k; ++i) {
    for (int j = 0; j < k; ++j) { double s = 0; for (int r = 0; r < m; ++r) s += A[r * k + i] * A[r * k + j]; AtA[i * k + j] = s; }
    double s = 0; for (int r = 0; r < m; ++r) s += A[r * k + i] * b[r]; Atb[i] = s;
  }
  JacobiEigenSym(k, AtA, ev, V);
  for (int i = 0; i < k; ++i) x[i] = 0.0;
  for (int e = 0; e < k; ++e) {
    if (!(ev[e] > 1e-28 * std::max(ev[0], 1e-300))) continue;
    double proj = 0; for (int i = 0; i < k; ++i) proj += V[e * k + i] * Atb[i];
    for (int i = 0; i < k; ++i) x[i] += V[e * k + i] * proj / ev[e];
  }
}

// 3 x 3 SVD A = U diag(s) V' through the eigen-decomposition of A'A; U completed to a rotation basis when rank-deficient.
void Svd3(const double* A, double* U, double* s, double* V /* columns are right singular vectors */) {
  double AtA[9], ev[3], Vr[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) AtA[3 * i + j] = A[i] * A[j] + A[3 + i] * A[3 + j] + A[6 + i] * A[6 + j];
  JacobiEigenSym(3, AtA, ev, Vr);
  for (int e = 0; e < 3; ++e) {
    s[e] = std::sqrt(std::max(ev[e], 0.0));
    for (int i = 0; i < 3; ++i) V[3 * i + e] = Vr[3 * e + i];
  }
  for (int e = 0; e < 3; ++e) {
    double u[3];
    const double v[3] = {V[e], V[3 + e], V[6 + e]};
    MatVec3(A, v, u);
    if (s[e] > 1e-14 * std::max(s[0], 1e-300)) for (int i = 0; i < 3; ++i) U[3 * i + e] = u[i] / s[e];
    else if (e == 2) {  // third column: cross product of the first two
      U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    } else for (int i = 0; i < 3; ++i) U[3 * i + e] = i == e ? 1.0 : 0.0;
  }
}

struct EPnP {
  int n;
  double fu, fv, uc, vc;
  const double* pws;   // 3 n
  const double* us;    // 2 n
  std::vector<double> alphas, pcs;
  double cws[4][3], ccs[4][3];

  void ChooseControlPoints() {
    for (int k = 0; k < 3; ++k) { double s = 0; for (int i = 0; i < n; ++i) s += pws[3 * i + k]; cws[0][k] = s / n; }
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dc[3], uct[9];
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] += (pws[3 * i + a] - cws[0][a]) * (pws[3 * i + b] - cws[0][b]);
    // principal axes with the signs OpenCV's SVD gives them (epnp.cpp choose_control_points: cvSVD of PW0'PW0, U transposed)
    JacobiSvdUt3(C, uct, dc);
    for (int i = 1; i < 4; ++i) {
      const double k = std::sqrt(std::max(dc[i - 1], 0.0) / n);
      for (int j = 0; j < 3; ++j) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
    }
  }
  bool ComputeBarycentric() {
    double CC[9], inv[9];
    for (int i = 0; i < 3; ++i) for (int j = 1; j < 4; ++j) CC[3 * i + j - 1] = cws[j][i] - cws[0][i];
    const double det = CC[0] * (CC[4] * CC[8] - CC[5] * CC[7]) - CC[1] * (CC[3] * CC[8] - CC[5] * CC[6]) + CC[2] * (CC[3] * CC[7] - CC[4] * CC[6]);
    if (!(std::fabs(det) > 0.0)) return false;   // coplanar points give a flat control tetrahedron: OpenCV's SVD inverse
    inv[0] = (CC[4] * CC[8] - CC[5] * CC[7]) / det; inv[1] = (CC[2] * CC[7] - CC[1] * CC[8]) / det; inv[2] = (CC[1] * CC[5] - CC[2] * CC[4]) / det;
    inv[3] = (CC[5] * CC[6] - CC[3] * CC[8]) / det; inv[4] = (CC[0] * CC[8] - CC[2] * CC[6]) / det; inv[5] = (CC[2] * CC[3] - CC[0] * CC[5]) / det;
    inv[6] = (CC[3] * CC[7] - CC[4] * CC[6]) / det; inv[7] = (CC[1] * CC[6] - CC[0] * CC[7]) / det; inv[8] = (CC[0] * CC[4] - CC[1] * CC[3]) / det;
    alphas.resize(4 * (size_t)n);
    for (int i = 0; i < n; ++i) {
      double* a = &alphas[4 * (size_t)i];
      for (int j = 0; j < 3; ++j)
        a[1 + j] = inv[3 * j] * (pws[3 * i] - cws[0][0]) + inv[3 * j + 1] * (pws[3 * i + 1] - cws[0][1]) + inv[3 * j + 2] * (pws[3 * i + 2] - cws[0][2]);
      a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    return true;
  }
  void ComputeCcs(const double* betas, const double* ut) {
    for (int i = 0; i < 4; ++i) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
    for (int i = 0; i < 4; ++i) {
      const double* v = ut + 12 * (11 - i);
      for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) ccs[j][k] += betas[i] * v[3 * j + k];
    }
  }
  void ComputePcs() {
    pcs.resize(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
      const double* a = &alphas[4 * (size_t)i];
      for (int j = 0; j < 3; ++j) pcs[3 * (size_t)i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
  }
  void EstimateRt(double* R, double* t) {
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { pc0[j] += pcs[3 * (size_t)i + j]; pw0[j] += pws[3 * (size_t)i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= n; pw0[j] /= n; }
    double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) ABt[3 * j + k] += (pcs[3 * (size_t)i + j] - pc0[j]) * (pws[3 * (size_t)i + k] - pw0[k]);
    double U[9], s[3], V[9];
    Svd3(ABt, U, s, V);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int i = 0; i < 3; ++i) t[i] = pc0[i] - (R[3 * i] * pw0[0] + R[3 * i + 1] * pw0[1] + R[3 * i + 2] * pw0[2]);
  }
  double ReprojectionError(const double* R, const double* t) const {
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
      const double* p = pws + 3 * (size_t)i;
      const double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0], Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
      const double inv = 1.0 / (R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2]);
      const double ue = uc + fu * X * inv, ve = vc + fv * Y * inv;
      sum += std::sqrt((us[2 * i] - ue) * (us[2 * i] - ue) + (us[2 * i + 1] - ve) * (us[2 * i + 1] - ve));
    }
    return sum / n;
  }
  double ComputeRt(const double* ut, const double* betas, double* R, double* t) {
    ComputeCcs(betas, ut);
    ComputePcs();
    if (pcs[2] < 0.0) {  // solve_for_sign
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) ccs[i][j] = -ccs[i][j];
      for (double& v : pcs) v = -v;
    }
    EstimateRt(R, t);
    return ReprojectionError(R, t);
  }
  static void GaussNewton(const double* L, const double* rho, double* betas) {
    for (int it = 0; it < 5; ++it) {
      double A[24], b[6], x[4];
      for (int i = 0; i < 6; ++i) {
        const double* l = L + 10 * i;
        A[4 * i] = 2 * l[0] * betas[0] + l[1] * betas[1] + l[3] * betas[2] + l[6] * betas[3];
        A[4 * i + 1] = l[1] * betas[0] + 2 * l[2] * betas[1] + l[4] * betas[2] + l[7] * betas[3];
        A[4 * i + 2] = l[3] * betas[0] + l[4] * betas[1] + 2 * l[5] * betas[2] + l[8] * betas[3];
        A[4 * i + 3] = l[6] * betas[0] + l[7] * betas[1] + l[8] * betas[2] + 2 * l[9] * betas[3];
        b[i] = rho[i] - (l[0] * betas[0] * betas[0] + l[1] * betas[0] * betas[1] + l[2] * betas[1] * betas[1] + l[3] * betas[0] * betas[2] +
                         l[4] * betas[1] * betas[2] + l[5] * betas[2] * betas[2] + l[6] * betas[0] * betas[3] + l[7] * betas[1] * betas[3] +
                         l[8] * betas[2] * betas[3] + l[9] * betas[3] * betas[3]);
      }
      LeastSquares(6, 4, A, b, x);
      for (int k = 0; k < 4; ++k) betas[k] += x[k];
    }
  }

  bool Solve(double* R_out, double* t_out) {
    ChooseControlPoints();
    if (!ComputeBarycentric()) return false;
    std::vector<double> MtM(144, 0.0);
    for (int i = 0; i < n; ++i) {
      const double* a = &alphas[4 * (size_t)i];
      double m1[12], m2[12];
      for (int j = 0; j < 4; ++j) {
        m1[3 * j] = a[j] * fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = a[j] * (uc - us[2 * i]);
        m2[3 * j] = 0.0; m2[3 * j + 1] = a[j] * fv; m2[3 * j + 2] = a[j] * (vc - us[2 * i + 1]);
      }
      for (int p = 0; p < 12; ++p) for (int q = 0; q < 12; ++q) MtM[12 * p + q] += m1[p] * m1[q] + m2[p] * m2[q];
    }
    double D[12], Ut[144];
    JacobiEigenSym(12, MtM.data(), D, Ut);   // rows of Ut: eigenvectors, eigenvalues descending (cvSVD's U_T)
    // L (6 x 10) and rho
    double L[60], rho[6], dv[4][6][3];
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    for (int i = 0; i < 4; ++i) {
      const double* v = Ut + 12 * (11 - i);
      for (int j = 0; j < 6; ++j) for (int k = 0; k < 3; ++k) dv[i][j][k] = v[3 * pa[j] + k] - v[3 * pb[j] + k];
    }
    auto dot = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
    for (int i = 0; i < 6; ++i) {
      double* r = L + 10 * i;
      r[0] = dot(dv[0][i], dv[0][i]); r[1] = 2 * dot(dv[0][i], dv[1][i]); r[2] = dot(dv[1][i], dv[1][i]);
      r[3] = 2 * dot(dv[0][i], dv[2][i]); r[4] = 2 * dot(dv[1][i], dv[2][i]); r[5] = dot(dv[2][i], dv[2][i]);
      r[6] = 2 * dot(dv[0][i], dv[3][i]); r[7] = 2 * dot(dv[1][i], dv[3][i]); r[8] = 2 * dot(dv[2][i], dv[3][i]);
      r[9] = dot(dv[3][i], dv[3][i]);
      double dsq = 0; for (int k = 0; k < 3; ++k) dsq += (cws[pa[i]][k] - cws[pb[i]][k]) * (cws[pa[i]][k] - cws[pb[i]][k]);
      rho[i] = dsq;
    }
    double betas[4][4], Rs[4][9], ts[4][3], err[4];
    {  // approximation 1: betas10 = [B11 B12 B13 B14]
      double A[24], x[4];
      static const int cols[4] = {0, 1, 3, 6};
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) A[4 * i + j] = L[10 * i + cols[j]];
      LeastSquares(6, 4, A, rho, x);
      const double sg = x[0] < 0 ? -1.0 : 1.0;
      betas[1][0] = std::sqrt(sg * x[0]);
      for (int j = 1; j < 4; ++j) betas[1][j] = sg * x[j] / betas[1][0];
    }
    {  // approximation 2: [B11 B12 B22]
      double A[18], x[3];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) A[3 * i + j] = L[10 * i + j];
      LeastSquares(6, 3, A, rho, x);
      if (x[0] < 0) { betas[2][0] = std::sqrt(-x[0]); betas[2][1] = x[2] < 0 ? std::sqrt(-x[2]) : 0.0; }
      else { betas[2][0] = std::sqrt(x[0]); betas[2][1] = x[2] > 0 ? std::sqrt(x[2]) : 0.0; }
      if (x[1] < 0) betas[2][0] = -betas[2][0];
      betas[2][2] = betas[2][3] = 0.0;
    }
    {  // approximation 3: [B11 B12 B22 B13 B23]
      double A[30], x[5];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 5; ++j) A[5 * i + j] = L[10 * i + j];
      LeastSquares(6, 5, A, rho, x);
      if (x[0] < 0) { betas[3][0] = std::sqrt(-x[0]); betas[3][1] = x[2] < 0 ? std::sqrt(-x[2]) : 0.0; }
      else { betas[3][0] = std::sqrt(x[0]); betas[3][1] = x[2] > 0 ? std::sqrt(x[2]) : 0.0; }
      if (x[1] < 0) betas[3][0] = -betas[3][0];
      betas[3][2] = x[3] / betas[3][0];
      betas[3][3] = 0.0;
    }
    int best = 1;
    for (int N = 1; N <= 3; ++N) {
      GaussNewton(L, rho, betas[N]);
      err[N] = ComputeRt(Ut, betas[N], Rs[N], ts[N]);
      if (!(err[N] == err[N])) err[N] = HUGE_VAL;
    }
    if (err[2] < err[1]) best = 2;
    if (err[3] < err[best]) best = 3;
    memcpy(R_out, Rs[best], 9 * sizeof(double));
    memcpy(t_out, ts[best], 3 * sizeof(double));
    return err[best] < HUGE_VAL;
  }
};

}  // namespace

// correspondencer.cpp:119-127: R = R_cam R_mb', t = R_cam R_mb' (-t_mb) + t_cam
void BaseFromMarkerDetection(const double marker_from_camera[6], const double marker_from_base[6], double base_from_camera[6]) {
  double Rc[9], Rb[9], Rbt[9], R[9], t[3];
  Rodrigues(marker_from_camera, Rc);
  Rodrigues(marker_from_base, Rb);
  Transpose3(Rb, Rbt);
  MatMul3(Rc, Rbt, R);
  const double neg[3] = {-marker_from_base[3], -marker_from_base[4], -marker_from_base[5]};
  MatVec3(R, neg, t);
  RotationToAngleAxis(R, base_from_camera);
  for (int k = 0; k < 3; ++k) base_from_camera[3 + k] = t[k] + marker_from_camera[3 + k];
}

// correspondencer.cpp:137-147: R = R_base R_mb, t = R_base t_mb + t_base
void MarkerFromCamera(const double base_from_camera[6], const double marker_from_base[6], double marker_from_camera[6]) {
  double Rb[9], Rm[9], R[9], t[3];
  Rodrigues(base_from_camera, Rb);
  Rodrigues(marker_from_base, Rm);
  MatMul3(Rb, Rm, R);
  MatVec3(Rb, marker_from_base + 3, t);
  RotationToAngleAxis(R, marker_from_camera);
  for (int k = 0; k < 3; ++k) marker_from_camera[3 + k] = t[k] + base_from_camera[3 + k];
}

// correspondencer.cpp:5-39: tvec -/+ E +/- F with E, F the first two columns of R times half the side
void MarkerCornersInCamera(const double pose[6], double marker_side, double out12[12]) {
  double R[9];
  Rodrigues(pose, R);
  const double h = marker_side / 2;
  const double E[3] = {R[0] * h, R[3] * h, R[6] * h}, F[3] = {R[1] * h, R[4] * h, R[7] * h};
  static const double se[4] = {-1, 1, 1, -1}, sf[4] = {1, 1, -1, -1};
  for (int c = 0; c < 4; ++c) for (int k = 0; k < 3; ++k) out12[3 * c + k] = pose[3 + k] + se[c] * E[k] + sf[c] * F[k];
}

int SolvePnPEPnP(int n, const double* object_points, const double* image_points, const double intrinsics4[4], double pose[6]) {
  if (n < 4 || !object_points || !image_points || !intrinsics4 || !pose) return RSBA_ERR_ARG;   // correspondencer.cpp:185-190
  EPnP e;
  e.n = n; e.pws = object_points; e.us = image_points;
  e.fu = intrinsics4[0]; e.fv = intrinsics4[1]; e.uc = intrinsics4[2]; e.vc = intrinsics4[3];
  double R[9], t[3];
  if (!e.Solve(R, t)) return RSBA_ERR_UNSUPPORTED;   // degenerate (coplanar) point set
  RotationToAngleAxis(R, pose);
  memcpy(pose + 3, t, sizeof(t));
  return RSBA_OK;
}

// Correspondencer::GetCorrespondencePoints' object points + CalculateTransforms (:137-176, :178-205) on a loaded
// marker-chain problem: the time and marker blocks already hold the base poses and the board geometry; every camera
// but the first gets its EPnP pose from the corners (in the main camera's frame) of all the markers it detected.
int InitialCameraPoses(rsba_problem* p) {
  if (!p || !p->is_marker_chain()) return RSBA_ERR_ARG;
  const int C = p->num_cameras, T = p->num_times;
  std::vector<std::vector<double>> obj(C), img(C);
  for (int64_t i = 0; i < p->num_observations; ++i) {
    const int c = p->camera_index[i], t = p->time_index[i], m = p->marker_index[i];
    double pose[6], corners[12];
    MarkerFromCamera(&p->parameters[6 * (size_t)(C + t)], &p->parameters[6 * (size_t)(C + T + m)], pose);
    MarkerCornersInCamera(pose, p->marker_side, corners);
    obj[c].insert(obj[c].end(), corners, corners + 12);
    img[c].insert(img[c].end(), &p->observations[8 * (size_t)i], &p->observations[8 * (size_t)i] + 8);
  }
  for (int k = 0; k < 6; ++k) p->parameters[k] = 0.0;   // :180-181
  for (int c = 1; c < C; ++c) {
    const int n = (int)(img[c].size() / 2);
    if (n < 4) return RSBA_ERR_FORMAT;   // "The correspondence points are too few." (:185-190)
    const int rc = SolvePnPEPnP(n, obj[c].data(), img[c].data(), &p->intrinsics[4 * (size_t)c], &p->parameters[6 * (size_t)c]);
    if (rc != RSBA_OK) return rc;
  }
  return RSBA_OK;
}

}  // namespace rsba
