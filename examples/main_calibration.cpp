// The BA half of the reference's Main_Calibration/main.cpp (:27-43) on the MI355X path: intrinsics ->
// BAManager -> StartBA -> Write -> reprojection RMS.  The OpenCV front end (detection, solvePnP, main.cpp:6-25)
// stays with the reference; its output correspondence.txt is this program's input.
//
//   g++ -O2 -std=c++17 -Iinclude examples/main_calibration.cpp -Lrealsensecalibration_amd -lrsba
//       -Wl,-rpath,$PWD/realsensecalibration_amd -o main_calibration   (one line)
//   ./main_calibration <Common dir> <output dir>
#include <cstdio>
#include <string>

#include "rsba/bundle_adjustment_manager.h"

using namespace RSCalibration;

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <dir with hongo/correspondence.txt and intrinsics/*.xml> <output dir>\n", argv[0]); return 2; }
  const std::string in = argv[1], out = argv[2];
  const char* serials[4] = {"821312061029", "816612062327", "821212062536", "821212061326"};  // my_const.h:15
  std::vector<Intrinsics> K(4);
  for (int i = 0; i < 4; ++i) {
    const std::string f = in + "/intrinsics/" + serials[i] + ".xml";
    if (rsba_read_intrinsics_xml(f.c_str(), K[i].data()) != RSBA_OK) { fprintf(stderr, "File can not be opened: %s\n", f.c_str()); return 1; }
  }
  BAPaths paths;
  paths.correspondence = in + "/hongo/correspondence.txt";
  paths.camera_transform_xml = out + "/Camera_Transform.xml";
  paths.extrinsics_dir = out;
  paths.point3d = out + "/point3d.txt";
  try {
    BAManager ba_manager(K, 0.0148, paths);
    if (ba_manager.StartBA() != RSBA_OK) return 1;
    ba_manager.Write();
    double err = 0;
    const double rms = ba_manager.ReprojectionRms(&err);
    printf("Reprojection Error (After BA): %.9g\nAverage Reprojection Error per One Coordinate: %.9g\n", err, rms);
    // the reference's own check re-reads what Write() produced (6-digit text): main.cpp:41-43
    double err_files = 0;
    const double rms_files = ba_manager.ReprojectionRmsFromFiles(K, &err_files);
    printf("From the written files: %.9g %.9g\n", err_files, rms_files);
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
