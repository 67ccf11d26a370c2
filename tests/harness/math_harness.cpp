// Test-only: exposes the product's per-observation arithmetic (ba_math.hpp, compiled for the host)
// so the CPU suite can compare it with the oracle's dual numbers.
#include "../../realsensecalibration_amd/csrc/ba_math.hpp"

extern "C" {
void h_camera_constants(const double* cam6, const double* intr4, double* cc32) { rsba::CameraConstants(cam6, intr4, cc32); }
void h_residual_jacobian(const double* cam6, const double* intr4, const double* X, const double* uv, double* r, double* jc, double* jp) {
  double cc[rsba::CC_STRIDE];
  rsba::CameraConstants(cam6, intr4, cc);
  rsba::ResidualJacobian(cc, X, uv[0], uv[1], r, jc, jp);
}
void h_residual(const double* cam6, const double* intr4, const double* X, const double* uv, double* r) {
  double cc[rsba::CC_STRIDE];
  rsba::CameraConstants(cam6, intr4, cc);
  rsba::Residual(cc, X, uv[0], uv[1], r);
}
int h_point_block_inverse(const double* V6, const double* s3, double lo, double hi, double radius, double* out6) {
  return rsba::PointBlockInverse(V6, s3, lo, hi, radius, out6) ? 1 : 0;
}
double h_loss(double delta, double s, double* sq) { return rsba::LossAndScale(delta, s, sq); }
}
