// C++ mirror of the OpenCV-free part of RSCalibration::Correspondencer
// (/root/reference/Main_Calibration/correspondencer.h:17-31, correspondencer.cpp): everything between the ArUco
// detections and correspondence.txt.  Detection itself (aruco::detectMarkers, estimatePoseSingleMarkers, :74-81) and the
// image windows stay with OpenCV on the host; this class takes their outputs as plain arrays.
//   Transform            my_struct.h: {rvec, tvec}
//   GetCornersInCameraWorld   correspondencer.cpp:5-39
//   BaseFromDetection         :104-127   (the base marker's pose at one time, from the lowest-id marker camera 0 saw)
//   MarkerFromCamera          :132-147
//   CalculateTransforms       :178-205   (camera 0 = identity, the others by solvePnP(..., SOLVEPNP_EPNP))
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rsba.h"

namespace RSCalibration {

struct Transform {
  std::array<double, 3> rvec{{0, 0, 0}}, tvec{{0, 0, 0}};
};
struct Point3d { double x, y, z; };
struct Point2d { double x, y; };

class Correspondencer {
 public:
  // intrinsics: {fx, fy, ppx, ppy} per camera in SERIAL_NUMBERS order (my_const.h:15); zero distortion
  Correspondencer(const std::vector<std::array<double, 4>>& camera_intrinsics, double marker_side)
      : intrinsics_(camera_intrinsics), marker_side_(marker_side) {}

  std::vector<Point3d> GetCornersInCameraWorld(const Transform& t) const {
    double pose[6], out[12];
    Pack(t, pose);
    Check(rsba_marker_corners_in_camera(pose, marker_side_, out), "GetCornersInCameraWorld");
    std::vector<Point3d> ret(4);
    for (int i = 0; i < 4; ++i) ret[i] = Point3d{out[3 * i], out[3 * i + 1], out[3 * i + 2]};
    return ret;
  }
  // the detected marker IS the base marker: its pose is the base pose (:104-111); otherwise :112-127
  static Transform BaseFromDetection(const Transform& marker_from_camera, const Transform& marker_from_base) {
    double a[6], b[6], out[6];
    Pack(marker_from_camera, a); Pack(marker_from_base, b);
    Check(rsba_base_pose_from_marker_detection(a, b, out), "BaseFromDetection");
    return Unpack(out);
  }
  static Transform MarkerFromCamera(const Transform& base_from_camera, const Transform& marker_from_base) {
    double a[6], b[6], out[6];
    Pack(base_from_camera, a); Pack(marker_from_base, b);
    Check(rsba_marker_pose_in_camera(a, b, out), "MarkerFromCamera");
    return Unpack(out);
  }
  // object_points[c] / image_points[c]: all corners camera c saw, over all times (what GetCorrespondencePoints collects)
  void CalculateTransforms(const std::vector<std::vector<Point3d>>& object_points, const std::vector<std::vector<Point2d>>& image_points,
                           std::vector<Transform>& cameras) const {
    cameras.assign(object_points.size(), Transform());
    for (size_t c = 1; c < object_points.size(); ++c) {
      if (image_points[c].size() < 4) throw std::runtime_error("The correspondence points are too few.");   // :185-190
      double pose[6];
      static_assert(sizeof(Point3d) == 3 * sizeof(double) && sizeof(Point2d) == 2 * sizeof(double), "packed points");
      Check(rsba_solve_pnp_epnp((int32_t)image_points[c].size(), &object_points[c][0].x, &image_points[c][0].x, intrinsics_[c].data(), pose),
            "CalculateTransforms");
      cameras[c] = Unpack(pose);
    }
  }

 private:
  static void Pack(const Transform& t, double* p) { for (int k = 0; k < 3; ++k) { p[k] = t.rvec[k]; p[3 + k] = t.tvec[k]; } }
  static Transform Unpack(const double* p) { Transform t; for (int k = 0; k < 3; ++k) { t.rvec[k] = p[k]; t.tvec[k] = p[3 + k]; } return t; }
  static void Check(int rc, const char* what) { if (rc != RSBA_OK) throw std::runtime_error(std::string(what) + ": " + rsba_error_string(rc)); }
  std::vector<std::array<double, 4>> intrinsics_;
  double marker_side_;
};

}  // namespace RSCalibration
