// AddressSanitizer / UBSan exercise of the HOST side of librsba (file readers, problem container, writers, the front
// end's initial-guess math): the three host translation units are compiled with -fsanitize=address,undefined together
// with this driver and run by tests/test_host_sanitize.py.  The device side (ba_solver.hip) is not part of the build:
// rsba::DeviceCount is stubbed to 0, and no solve entry point is called.  Test infrastructure only.
//   usage: host_sanitize_driver <tests/golden> <scratch dir>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rsba.h"

namespace rsba { int DeviceCount() { return 0; } }

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); exit(2); } } while (0)

static void WriteFile(const std::string& path, const std::string& text) {
  FILE* f = fopen(path.c_str(), "w");
  CHECK(f != nullptr);
  fputs(text.c_str(), f);
  fclose(f);
}

int main(int argc, char** argv) {
  CHECK(argc == 3);
  const std::string G = argv[1], T = argv[2];
  CHECK(rsba_version() == RSBA_VERSION && rsba_device_count() == 0);
  for (int c = -1; c < 10; ++c) CHECK(rsba_error_string(c) != nullptr);

  // ---- intrinsics XML (my_io.cpp:5-31) + correspondence.txt (bundle_adjustment.cpp:132-187)
  const char* serials[4] = {"821312061029", "816612062327", "821212062536", "821212061326"};
  double intr[16];
  for (int i = 0; i < 4; ++i) CHECK(rsba_read_intrinsics_xml((G + "/intrinsics/" + serials[i] + ".xml").c_str(), intr + 4 * i) == RSBA_OK);
  CHECK(rsba_read_intrinsics_xml((G + "/intrinsics/none.xml").c_str(), intr) == RSBA_ERR_IO);
  rsba_problem* p = nullptr;
  CHECK(rsba_problem_load_correspondence((G + "/hongo/correspondence.txt").c_str(), RSBA_MODEL_MARKER_CHAIN, 0.0148, intr, &p) == RSBA_OK);
  CHECK(rsba_problem_num_times(p) == 6 && rsba_problem_num_cameras(p) == 4 && rsba_problem_num_markers(p) == 11);
  CHECK(rsba_problem_num_observations(p) == 68 && rsba_problem_num_parameters(p) == 126 && rsba_problem_num_points(p) == 0);
  for (int64_t i = -1; i <= 68; ++i) {   // one past either end: accessors must refuse, not read
    (void)rsba_problem_camera_idx(p, i); (void)rsba_problem_time_idx(p, i); (void)rsba_problem_marker_idx(p, i); (void)rsba_problem_point_idx(p, i);
  }
  for (int t = -1; t <= 6; ++t) for (int c = -1; c <= 4; ++c) (void)rsba_problem_num_observations_per_time_camera(p, t, c);
  CHECK(rsba_problem_camera_parameters(p, 4) == nullptr && rsba_problem_camera_parameters(p, 3) != nullptr);
  CHECK(rsba_problem_marker_transform(p, 11) == nullptr && rsba_problem_marker_transform(p, -1) == nullptr);
  std::vector<double> corners(12 * 68);
  CHECK(rsba_problem_point3d_coordinates(p, corners.data()) == RSBA_OK);
  // writers (bundle_adjustment_manager.cpp:98-175)
  CHECK(rsba_write_outputs(p, (T + "/Camera_Transform.xml").c_str(), T.c_str(), (T + "/point3d.txt").c_str()) == RSBA_OK);
  CHECK(rsba_write_outputs(p, nullptr, nullptr, nullptr) == RSBA_OK);
  CHECK(rsba_write_outputs(p, (T + "/no/such/dir/x.xml").c_str(), nullptr, nullptr) == RSBA_ERR_IO);
  // Correspondencer::CalculateTransforms on the loaded problem (EPnP per camera)
  CHECK(rsba_problem_initial_camera_poses(p) == RSBA_OK);
  rsba_problem_free(p);
  p = nullptr;
  CHECK(rsba_problem_load_correspondence((G + "/test2/correspondence_test.txt").c_str(), RSBA_MODEL_MARKER_CHAIN_TEST2, 0.048, intr, &p) == RSBA_OK);
  CHECK(rsba_write_outputs(p, (T + "/Camera_Transform2.xml").c_str(), nullptr, (T + "/point3d2.txt").c_str()) == RSBA_OK);
  rsba_problem_free(p);
  p = nullptr;

  // ---- malformed / short inputs: return codes, no reads past the end
  CHECK(rsba_problem_load_correspondence((G + "/hongo/nope.txt").c_str(), RSBA_MODEL_MARKER_CHAIN, 0.0148, intr, &p) == RSBA_ERR_IO);
  WriteFile(T + "/short.txt", "2 2 1 3\n0 1 1\n1 1 0\n0 0 0 1 2 3 4 5 6 7 8\n0 1 0 1 2 3");
  CHECK(rsba_problem_load_correspondence((T + "/short.txt").c_str(), RSBA_MODEL_MARKER_CHAIN, 0.0148, intr, &p) == RSBA_ERR_FORMAT);
  WriteFile(T + "/badidx.txt", "1 1 1 1\n0 1\n0 7 0 1 2 3 4 5 6 7 8\n0 0 0 0 0 0\n0 0 0 0 0 0\n0 0 0 0 0 0\n");
  CHECK(rsba_problem_load_correspondence((T + "/badidx.txt").c_str(), RSBA_MODEL_MARKER_CHAIN, 0.0148, intr, &p) != RSBA_OK);
  WriteFile(T + "/neg.txt", "-1 4 2 5\n");
  CHECK(rsba_problem_load_correspondence((T + "/neg.txt").c_str(), RSBA_MODEL_MARKER_CHAIN, 0.0148, intr, &p) != RSBA_OK);
  WriteFile(T + "/empty.txt", "");
  CHECK(rsba_problem_load_correspondence((T + "/empty.txt").c_str(), RSBA_MODEL_MARKER_CHAIN, 0.0148, intr, &p) == RSBA_ERR_FORMAT);
  CHECK(rsba_problem_load_points_file((T + "/empty.txt").c_str(), intr, &p) == RSBA_ERR_FORMAT);

  // ---- Test1 point file (bundle_adjustmenter.cpp:55-85), both header forms
  CHECK(rsba_problem_load_points_file((G + "/two_cam_data.txt").c_str(), intr, &p) == RSBA_OK);
  CHECK(rsba_problem_model(p) == RSBA_MODEL_POINTS && rsba_problem_num_cameras(p) == 1);   // the committed file holds one camera
  const int64_t n1 = rsba_problem_num_observations(p);
  CHECK(n1 == rsba_problem_num_points(p));
  for (int64_t i = -1; i <= n1; ++i) { (void)rsba_problem_point_idx(p, i); (void)rsba_problem_camera_idx(p, i); }
  CHECK(rsba_problem_set_camera_constant(p, 0, 1) == RSBA_OK && rsba_problem_set_camera_constant(p, 1, 1) == RSBA_ERR_ARG);
  rsba_problem_free(p);
  p = nullptr;
  WriteFile(T + "/pts3.txt", "2 2 3\n0 0 1.5 2.5\n1 0 3.5 4.5\n1 1 5.5 6.5\n0 0 0\n0 0 1\n0 0 0\n0 0 2\n0.1 0.2 3\n0.3 0.1 4\n");
  CHECK(rsba_problem_load_points_file((T + "/pts3.txt").c_str(), intr, &p) == RSBA_OK);
  CHECK(rsba_problem_num_observations(p) == 3 && rsba_problem_num_points(p) == 2);
  rsba_problem_free(p);
  p = nullptr;
  WriteFile(T + "/pts_bad.txt", "2 2 3\n0 0 1.5 2.5\n1 5 3.5 4.5\n1 1 5.5 6.5\n0 0 0\n0 0 1\n0 0 0\n0 0 2\n0.1 0.2 3\n0.3 0.1 4\n");
  CHECK(rsba_problem_load_points_file((T + "/pts_bad.txt").c_str(), intr, &p) != RSBA_OK);

  // ---- problem from arrays: index validation
  {
    const int32_t cam[3] = {0, 1, 1}, pt[3] = {0, 0, 1};
    const double obs[6] = {1, 2, 3, 4, 5, 6}, par[18] = {0}, k[8] = {600, 600, 320, 240, 600, 600, 320, 240};
    CHECK(rsba_problem_create_points(2, 2, 3, cam, pt, obs, par, k, &p) == RSBA_OK);
    rsba_problem_free(p);
    p = nullptr;
    const int32_t bad[3] = {0, 2, 1};
    CHECK(rsba_problem_create_points(2, 2, 3, bad, pt, obs, par, k, &p) != RSBA_OK);
    CHECK(rsba_problem_create_points(2, 2, 3, nullptr, pt, obs, par, k, &p) == RSBA_ERR_ARG);
    CHECK(rsba_problem_create_points(0, 0, 0, cam, pt, obs, par, k, &p) != RSBA_ERR_HIP);
    if (p) { rsba_problem_free(p); p = nullptr; }
  }

  // ---- the front end's pose algebra and EPnP (correspondencer.cpp:5-39, 119-147, 192-195)
  {
    const double a[6] = {0.1, -0.2, 0.3, 0.05, 0.02, 0.6}, b[6] = {-0.3, 0.1, 0.2, 0.01, -0.04, 0.1};
    double c[6], d[6], corners4[12];
    CHECK(rsba_base_pose_from_marker_detection(a, b, c) == RSBA_OK && rsba_marker_pose_in_camera(c, b, d) == RSBA_OK);
    for (int i = 0; i < 6; ++i) CHECK(std::fabs(d[i] - a[i]) < 1e-12);
    CHECK(rsba_marker_corners_in_camera(a, 0.0148, corners4) == RSBA_OK);
    CHECK(rsba_base_pose_from_marker_detection(nullptr, b, c) == RSBA_ERR_ARG);
    // EPnP on eight exact projections of a non-planar set
    const double K[4] = {630, 625, 318, 237};
    double obj[24], img[16], pose[6];
    const double truth[6] = {0.2, -0.1, 0.15, 0.03, -0.02, 1.2};
    for (int i = 0; i < 8; ++i) { obj[3 * i] = (i & 1) ? 0.1 : -0.1; obj[3 * i + 1] = (i & 2) ? 0.12 : -0.08; obj[3 * i + 2] = (i & 4) ? 0.09 : -0.11; }
    {
      // Rodrigues by hand
      const double th = std::sqrt(truth[0] * truth[0] + truth[1] * truth[1] + truth[2] * truth[2]);
      const double kx = truth[0] / th, ky = truth[1] / th, kz = truth[2] / th, cs = std::cos(th), sn = std::sin(th), c1 = 1 - cs;
      const double R[9] = {cs + c1 * kx * kx, c1 * kx * ky - sn * kz, c1 * kx * kz + sn * ky, c1 * kx * ky + sn * kz, cs + c1 * ky * ky, c1 * ky * kz - sn * kx,
                           c1 * kx * kz - sn * ky, c1 * ky * kz + sn * kx, cs + c1 * kz * kz};
      for (int i = 0; i < 8; ++i) {
        const double* X = obj + 3 * i;
        const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + truth[3], y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + truth[4], z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + truth[5];
        img[2 * i] = K[0] * x / z + K[2]; img[2 * i + 1] = K[1] * y / z + K[3];
      }
    }
    CHECK(rsba_solve_pnp_epnp(8, obj, img, K, pose) == RSBA_OK);
    for (int i = 0; i < 6; ++i) CHECK(std::fabs(pose[i] - truth[i]) < 1e-6);
    CHECK(rsba_solve_pnp_epnp(3, obj, img, K, pose) != RSBA_OK);
    for (int i = 0; i < 8; ++i) obj[3 * i + 2] = 0.0;   // coplanar: refused
    CHECK(rsba_solve_pnp_epnp(8, obj, img, K, pose) == RSBA_ERR_UNSUPPORTED);
  }
  rsba_options o;
  rsba_options_default(&o);
  CHECK(o.max_num_iterations == 50);
  printf("host sanitize driver: ok\n");
  return 0;
}
