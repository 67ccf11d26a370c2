// C++ mirror of RSCalibration::BAManager (/root/reference/Main_Calibration/bundle_adjustment_manager.h:7-17):
// constructor loads correspondence.txt, StartBA() runs the solve that used to be ceres::Solve with
// DENSE_SCHUR (bundle_adjustment_manager.cpp:16-96), Write() exports Camera_Transform.xml, mat{i}.txt and
// point3d.txt (:98-175).  Differences that are deliberate: no exit()/system("PAUSE") — failures throw or are
// returned; paths and MARKER_SIDE are arguments instead of compile-time constants (my_const.h:9-16).
#pragma once
#include <cstdio>
#include <map>
#include <string>

#include "bundle_adjustment.h"

namespace RSCalibration {

struct BAPaths {
  std::string correspondence = "../Common/Correspondence/hongo/correspondence.txt";   // bundle_adjustment_manager.cpp:8
  std::string camera_transform_xml = "../Common/Correspondence/hongo/Camera_Transform.xml";  // :108
  std::string extrinsics_dir = "../Common/Calibration/Extrinsics";                     // :140
  std::string point3d = "../Common/Correspondence/hongo/point3d.txt";                  // :158
};

class BAManager {
 public:
  // camera_intrinsics: one {fx, fy, ppx, ppy} per camera index, in SERIAL_NUMBERS order (my_const.h:15);
  // the reference takes map<string, Mat> keyed by serial and indexes it with SERIAL_NUMBERS[camera_idx].
  BAManager(const std::vector<Intrinsics>& camera_intrinsics, double marker_side = 0.0148, const BAPaths& paths = BAPaths(),
            int model = RSBA_MODEL_MARKER_CHAIN)
      : paths_(paths) {
    if (!bal_problem.loadFile(paths_.correspondence.c_str(), marker_side, camera_intrinsics, model))
      throw std::runtime_error("unable to open correspondence file " + paths_.correspondence);
    rsba_options_default(&options);
    options.minimizer_progress_to_stdout = 1;  // bundle_adjustment_manager.cpp:92
  }

  // ceres::Solve(options, &problem, &summary) + FullReport (bundle_adjustment_manager.cpp:90-95)
  int StartBA() {
    const int rc = rsba_solve(bal_problem.handle(), &options, &summary);
    if (rc != RSBA_OK) { fprintf(stderr, "rsba_solve: %s\n", rsba_error_string(rc)); return rc; }
    static const char* kTerm[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
    printf("\nSolver Summary\nIterations: %d (successful %d, unsuccessful %d)\nCost: initial %.6e  final %.6e\n"
           "Time in minimizer: %.6f s (setup %.6f s)\nTermination: %s\n",
           summary.num_iterations, summary.num_successful_steps, summary.num_unsuccessful_steps, summary.initial_cost,
           summary.final_cost, summary.minimizer_seconds, summary.setup_seconds, kTerm[summary.termination_type]);
    return rc;
  }

  void Write() {
    printf("Marker Transform\n");  // bundle_adjustment_manager.cpp:100-107
    const int markers = rsba_problem_num_markers(bal_problem.handle());
    for (int m = 0; m < markers; ++m) {
      const double* t = bal_problem.marker_transform(m);
      printf("%d Rvec: %g %g %g tvec: %g %g %g\n", m, t[0], t[1], t[2], t[3], t[4], t[5]);
    }
    const int rc = rsba_write_outputs(bal_problem.handle(), paths_.camera_transform_xml.c_str(), paths_.extrinsics_dir.c_str(),
                                      paths_.point3d.c_str());
    if (rc != RSBA_OK) throw std::runtime_error(std::string("BAManager::Write: ") + rsba_error_string(rc));
  }

  // ReprojectionCheck::Reproject's two printed numbers (reprojection_check.cpp:100-101), from the parameters
  double ReprojectionRms(double* error = nullptr) {
    double e = 0, rms = 0;
    rsba_reprojection_error(bal_problem.handle(), &options, &e, &rms);
    if (error) *error = e;
    return rms;
  }

  // ReprojectionCheck::Reproject as main.cpp:41-43 runs it after Write(): from the files just written
  // (reprojection_check.cpp:5-101: 6-digit point3d.txt, Camera_Transform.xml, float32 corners)
  double ReprojectionRmsFromFiles(const std::vector<Intrinsics>& camera_intrinsics, double* error = nullptr) {
    std::vector<double> k;
    for (const Intrinsics& i : camera_intrinsics) k.insert(k.end(), i.begin(), i.end());
    double e = 0, rms = 0;
    const int rc = rsba_reprojection_check_files(paths_.correspondence.c_str(), paths_.point3d.c_str(), paths_.camera_transform_xml.c_str(),
                                                 k.data(), &e, &rms);
    if (rc != RSBA_OK) throw std::runtime_error(std::string("ReprojectionCheck: ") + rsba_error_string(rc));
    if (error) *error = e;
    return rms;
  }

  BALProblem bal_problem;
  rsba_options options;
  rsba_summary summary{};

 private:
  BAPaths paths_;
};

}  // namespace RSCalibration
