// Host-side problem container: the role of BALProblem in the reference
// (/root/reference/Main_Calibration/bundle_adjustment.h:18-54 for the marker-chain model,
//  /root/reference/Test1_BundleAdjustment/bundle_adjustmenter.cpp:14-104 for the point model).
// Owns flat arrays in the reference's layouts; the solver writes its result back into `parameters`.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rsba.h"

struct rsba_problem {
  int32_t model = RSBA_MODEL_POINTS;
  int32_t num_cameras = 0;
  int32_t num_points = 0;   // point model
  int32_t num_times = 0;    // marker-chain
  int32_t num_markers = 0;  // marker-chain
  int64_t num_observations = 0;
  double marker_side = 0.0;  // marker-chain (my_const.h:9)
  std::vector<int32_t> camera_index;  // per observation
  std::vector<int32_t> point_index;   // point model
  std::vector<int32_t> time_index;    // marker-chain
  std::vector<int32_t> marker_index;  // marker-chain
  std::vector<int32_t> obs_per_time_camera;  // T x C (marker-chain)
  std::vector<double> observations;   // 2 (points) or 8 (marker-chain) per observation
  std::vector<double> parameters;     // [C x 6 | P x 3]  or  [C | T | M] x 6
  std::vector<double> intrinsics;     // 4 per camera: fx, fy, ppx, ppy
  std::vector<uint8_t> camera_constant;  // point model: 1 = problem.SetParameterBlockConstant(camera); empty = none
  std::vector<uint8_t> point_constant;   // point model: the same for point blocks (round 6)
  std::vector<uint8_t> block_constant;   // marker-chain models: per block of [C cameras | T times | M markers] (round 6); empty = none

  int64_t num_parameters() const { return (int64_t)parameters.size(); }
  int obs_dim() const { return model == RSBA_MODEL_POINTS ? 2 : 8; }
  bool is_marker_chain() const { return model != RSBA_MODEL_POINTS; }
  // wiring rules of bundle_adjustment_manager.cpp:26-87 / Test2 main.cpp:64-96
  bool uses_camera(int64_t i) const { return camera_index[i] != 0; }
  bool uses_marker(int64_t i) const { return model == RSBA_MODEL_MARKER_CHAIN_TEST2 ? true : marker_index[i] != 0; }
  int camera_block(int64_t i) const { return camera_index[i]; }
  int time_block(int64_t i) const { return num_cameras + time_index[i]; }
  int marker_block(int64_t i) const { return num_cameras + num_times + marker_index[i]; }
};

namespace rsba {

int LoadPointsFile(const char* path, const double* intrinsics4, rsba_problem** out);
int LoadCorrespondence(const char* path, int32_t model, double marker_side, const double* intrinsics,
                       rsba_problem** out);
int ReadIntrinsicsXml(const char* path, double* out4);
void Rodrigues(const double rvec[3], double R[9]);
void RotationToAngleAxis(const double R[9], double aa[3]);
int LoadReprojectionCheck(const char* correspondence_txt, const char* point3d_txt, const char* camera_transform_xml,
                          const double* intrinsics, rsba_problem** out);
void AngleAxisRotatePointHost(const double aa[3], const double pt[3], double out[3]);
void MarkerCorners3d(const rsba_problem& p, double* out /* 12 per observation */);
// ba_initial_guess.cpp: the front end's math between the detections and correspondence.txt (correspondencer.cpp)
void BaseFromMarkerDetection(const double marker_from_camera[6], const double marker_from_base[6], double base_from_camera[6]);
void MarkerFromCamera(const double base_from_camera[6], const double marker_from_base[6], double marker_from_camera[6]);
void MarkerCornersInCamera(const double pose[6], double marker_side, double out12[12]);
int SolvePnPEPnP(int n, const double* object_points, const double* image_points, const double intrinsics4[4], double pose[6]);
int InitialCameraPoses(rsba_problem* p);
int WriteOutputs(const rsba_problem& p, const char* camera_transform_xml, const char* extrinsics_dir,
                 const char* point3d_txt);

}  // namespace rsba
