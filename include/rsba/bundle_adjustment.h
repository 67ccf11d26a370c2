// C++ mirror of the reference's problem container, over the C ABI of librsba.so.
//
// Same class name, namespace and accessors as RSCalibration::BALProblem in
// /root/reference/Main_Calibration/bundle_adjustment.h:18-54 (loadFile, parameters, camera_parameters,
// marker_transform, mutable_*_from_*, getPoint3dCoordinates, ...), so that bundle_adjustment_manager.cpp
// compiles against it unchanged apart from the Ceres calls it no longer needs.  What is gone: the four
// templated reprojection functors (:56-343) — their arithmetic now runs in HIP kernels behind rsba_solve —
// and cv::Mat (intrinsics are {fx, fy, ppx, ppy} per camera).
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rsba.h"

namespace RSCalibration {

struct Point3d { double x, y, z; };
using Intrinsics = std::array<double, 4>;  // fx, fy, ppx, ppy  (K(0,0), K(1,1), K(0,2), K(1,2))

class BALProblem {
 public:
  BALProblem() = default;
  BALProblem(const BALProblem&) = delete;
  BALProblem& operator=(const BALProblem&) = delete;
  ~BALProblem() { rsba_problem_free(p_); }

  // bundle_adjustment.cpp:132-187.  The reference reads MARKER_SIDE and the intrinsics elsewhere; here they are
  // handed in because the functors that used to hold them are gone.  Returns false when the file cannot be
  // opened (as the reference does) or is malformed.
  bool loadFile(const char* filename, double marker_side, const std::vector<Intrinsics>& intrinsics,
                int model = RSBA_MODEL_MARKER_CHAIN) {
    rsba_problem_free(p_);
    p_ = nullptr;
    std::vector<double> k;
    for (const auto& i : intrinsics) k.insert(k.end(), i.begin(), i.end());
    return rsba_problem_load_correspondence(filename, model, marker_side, k.data(), &p_) == RSBA_OK;
  }
  // Test1 BALProblem::LoadFile (Test1_BundleAdjustment/bundle_adjustmenter.cpp:55-85)
  bool LoadFile(const char* filename, const Intrinsics& intrinsics) {
    rsba_problem_free(p_);
    p_ = nullptr;
    return rsba_problem_load_points_file(filename, intrinsics.data(), &p_) == RSBA_OK;
  }

  int num_cameras() const { return rsba_problem_num_cameras(p_); }
  int num_times() const { return rsba_problem_num_times(p_); }
  int num_observations() const { return (int)rsba_problem_num_observations(p_); }
  int num_observations_per_time_camera(int time_idx, int camera_idx) const { return rsba_problem_num_observations_per_time_camera(p_, time_idx, camera_idx); }
  const double* observations() const { return rsba_problem_observations(p_); }
  int num_parameters() const { return (int)rsba_problem_num_parameters(p_); }
  const double* parameters() const { return rsba_problem_parameters(p_); }
  int camera_idx(int observation_id) const { return rsba_problem_camera_idx(p_, observation_id); }
  int marker_idx(int observation_id) const { return rsba_problem_marker_idx(p_, observation_id); }
  double* camera_parameters(int camera_idx) { return rsba_problem_camera_parameters(p_, camera_idx); }
  double* marker_transform(int marker_idx) { return rsba_problem_marker_transform(p_, marker_idx); }
  double* mutable_camera_transform_from_base_camera(int i) { return rsba_problem_parameters(p_) + 6 * rsba_problem_camera_idx(p_, i); }
  double* mutable_base_marker_transform_from_base_camera(int i) { return rsba_problem_parameters(p_) + 6 * num_cameras() + 6 * rsba_problem_time_idx(p_, i); }
  double* mutable_marker_transform_from_base_marker(int i) { return rsba_problem_parameters(p_) + 6 * num_cameras() + 6 * num_times() + 6 * rsba_problem_marker_idx(p_, i); }
  // Test1 accessors
  double* mutable_cameras() { return rsba_problem_parameters(p_); }
  double* mutable_points() { return rsba_problem_parameters(p_) + 6 * num_cameras(); }
  double* mutable_camera_for_observation(int i) { return mutable_cameras() + 6 * rsba_problem_camera_idx(p_, i); }
  double* mutable_point_for_observation(int i) { return mutable_points() + 3 * rsba_problem_point_idx(p_, i); }

  void getPoint3dCoordinates(std::vector<Point3d>& points) {
    std::vector<double> buf(12 * (size_t)num_observations());
    if (rsba_problem_point3d_coordinates(p_, buf.data()) != RSBA_OK) return;
    for (size_t i = 0; i < buf.size() / 3; ++i) points.push_back(Point3d{buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]});
  }

  rsba_problem* handle() { return p_; }

 private:
  rsba_problem* p_ = nullptr;
};

}  // namespace RSCalibration
