// C ABI: problem container, options, file formats (no device code here).  Declared in include/rsba.h.
#include <climits>
#include <cstdint>
#include <cstring>

#include "ba_problem.hpp"
#include "ba_solver.hpp"

extern "C" {

int rsba_version(void) { return RSBA_VERSION; }
int rsba_device_count(void) { return rsba::DeviceCount(); }

const char* rsba_error_string(int code) {
  switch (code) {
    case RSBA_OK: return "ok";
    case RSBA_ERR_IO: return "unable to open file";
    case RSBA_ERR_FORMAT: return "malformed file";
    case RSBA_ERR_ARG: return "bad argument";
    case RSBA_ERR_HIP: return "HIP runtime error";
    case RSBA_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
    case RSBA_ERR_COMM: return "RCCL error";
    case RSBA_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}

int rsba_problem_create_points(int32_t C, int32_t P, int64_t N, const int32_t* camera_index, const int32_t* point_index,
                               const double* observations, const double* parameters, const double* intrinsics, rsba_problem** out) {
  if (C <= 0 || P <= 0 || N < 0 || !camera_index || !point_index || !observations || !parameters || !intrinsics || !out) return RSBA_ERR_ARG;
  for (int64_t i = 0; i < N; ++i)
    if (camera_index[i] < 0 || camera_index[i] >= C || point_index[i] < 0 || point_index[i] >= P) return RSBA_ERR_ARG;
  rsba_problem* p = new rsba_problem();
  p->model = RSBA_MODEL_POINTS; p->num_cameras = C; p->num_points = P; p->num_observations = N;
  p->camera_index.assign(camera_index, camera_index + N);
  p->point_index.assign(point_index, point_index + N);
  p->observations.assign(observations, observations + 2 * N);
  p->parameters.assign(parameters, parameters + 6 * (size_t)C + 3 * (size_t)P);
  p->intrinsics.assign(intrinsics, intrinsics + 4 * (size_t)C);
  *out = p;
  return RSBA_OK;
}

int rsba_problem_create_marker_chain(int32_t model, int32_t C, int32_t T, int32_t M, int64_t N, const int32_t* time_index,
                                     const int32_t* camera_index, const int32_t* marker_index, const double* observations,
                                     const double* parameters, const double* intrinsics, double marker_side, rsba_problem** out) {
  if ((model != RSBA_MODEL_MARKER_CHAIN && model != RSBA_MODEL_MARKER_CHAIN_TEST2) || C <= 0 || T <= 0 || M <= 0 || N < 0 || N > INT32_MAX ||
      !time_index || !camera_index || !marker_index || !observations || !parameters || !intrinsics || !(marker_side > 0.0) || !out)
    return RSBA_ERR_ARG;
  for (int64_t i = 0; i < N; ++i)
    if (time_index[i] < 0 || time_index[i] >= T || camera_index[i] < 0 || camera_index[i] >= C || marker_index[i] < 0 || marker_index[i] >= M)
      return RSBA_ERR_ARG;
  rsba_problem* p = new rsba_problem();
  p->model = model; p->num_cameras = C; p->num_times = T; p->num_markers = M; p->num_observations = N; p->marker_side = marker_side;
  p->time_index.assign(time_index, time_index + N);
  p->camera_index.assign(camera_index, camera_index + N);
  p->marker_index.assign(marker_index, marker_index + N);
  p->observations.assign(observations, observations + 8 * N);
  p->parameters.assign(parameters, parameters + 6 * ((size_t)C + T + M));
  p->intrinsics.assign(intrinsics, intrinsics + 4 * (size_t)C);
  p->obs_per_time_camera.assign((size_t)T * C, 0);
  for (int64_t i = 0; i < N; ++i) p->obs_per_time_camera[(size_t)time_index[i] * C + camera_index[i]]++;
  *out = p;
  return RSBA_OK;
}

int rsba_problem_load_points_file(const char* path, const double* intrinsics4, rsba_problem** out) { return rsba::LoadPointsFile(path, intrinsics4, out); }
int rsba_problem_load_correspondence(const char* path, int32_t model, double marker_side, const double* intrinsics, rsba_problem** out) {
  return rsba::LoadCorrespondence(path, model, marker_side, intrinsics, out);
}
int rsba_problem_set_camera_constant(rsba_problem* p, int32_t camera_idx, int32_t constant) {
  if (!p || camera_idx < 0 || camera_idx >= p->num_cameras) return RSBA_ERR_ARG;
  if (p->model != RSBA_MODEL_POINTS) return RSBA_ERR_UNSUPPORTED;   // the marker-chain wiring already fixes camera 0 / marker 0
  if (p->camera_constant.empty()) p->camera_constant.assign(p->num_cameras, 0);
  p->camera_constant[camera_idx] = constant ? 1 : 0;
  return RSBA_OK;
}
int rsba_problem_set_point_constant(rsba_problem* p, int32_t point_idx, int32_t constant) {
  if (!p || point_idx < 0 || point_idx >= p->num_points) return RSBA_ERR_ARG;
  if (p->model != RSBA_MODEL_POINTS) return RSBA_ERR_UNSUPPORTED;
  if (p->point_constant.empty()) p->point_constant.assign(p->num_points, 0);
  p->point_constant[point_idx] = constant ? 1 : 0;
  return RSBA_OK;
}
int rsba_problem_set_parameter_block_constant(rsba_problem* p, int64_t parameter_offset, int32_t constant) {
  if (!p || parameter_offset < 0 || parameter_offset >= (int64_t)p->parameters.size()) return RSBA_ERR_ARG;
  if (p->model == RSBA_MODEL_POINTS) {
    const int64_t cam_end = 6LL * p->num_cameras;
    if (parameter_offset < cam_end) return parameter_offset % 6 ? RSBA_ERR_ARG : rsba_problem_set_camera_constant(p, (int32_t)(parameter_offset / 6), constant);
    return (parameter_offset - cam_end) % 3 ? RSBA_ERR_ARG : rsba_problem_set_point_constant(p, (int32_t)((parameter_offset - cam_end) / 3), constant);
  }
  if (parameter_offset % 6) return RSBA_ERR_ARG;
  const size_t nblocks = (size_t)p->num_cameras + p->num_times + p->num_markers;
  if (p->block_constant.empty()) p->block_constant.assign(nblocks, 0);
  p->block_constant[(size_t)(parameter_offset / 6)] = constant ? 1 : 0;
  return RSBA_OK;
}
void rsba_problem_free(rsba_problem* p) { delete p; }

int rsba_base_pose_from_marker_detection(const double* marker_from_camera, const double* marker_from_base, double* base_from_camera) {
  if (!marker_from_camera || !marker_from_base || !base_from_camera) return RSBA_ERR_ARG;
  rsba::BaseFromMarkerDetection(marker_from_camera, marker_from_base, base_from_camera);
  return RSBA_OK;
}
int rsba_marker_pose_in_camera(const double* base_from_camera, const double* marker_from_base, double* marker_from_camera) {
  if (!base_from_camera || !marker_from_base || !marker_from_camera) return RSBA_ERR_ARG;
  rsba::MarkerFromCamera(base_from_camera, marker_from_base, marker_from_camera);
  return RSBA_OK;
}
int rsba_marker_corners_in_camera(const double* pose, double marker_side, double* out12) {
  if (!pose || !out12 || !(marker_side > 0.0)) return RSBA_ERR_ARG;
  rsba::MarkerCornersInCamera(pose, marker_side, out12);
  return RSBA_OK;
}
int rsba_solve_pnp_epnp(int32_t n, const double* object_points, const double* image_points, const double* intrinsics4, double* pose) {
  return rsba::SolvePnPEPnP(n, object_points, image_points, intrinsics4, pose);
}
int rsba_problem_initial_camera_poses(rsba_problem* p) { return rsba::InitialCameraPoses(p); }

int32_t rsba_problem_model(const rsba_problem* p) { return p ? p->model : -1; }
int32_t rsba_problem_num_cameras(const rsba_problem* p) { return p ? p->num_cameras : 0; }
int32_t rsba_problem_num_points(const rsba_problem* p) { return p ? p->num_points : 0; }
int32_t rsba_problem_num_times(const rsba_problem* p) { return p ? p->num_times : 0; }
int32_t rsba_problem_num_markers(const rsba_problem* p) { return p ? p->num_markers : 0; }
int64_t rsba_problem_num_observations(const rsba_problem* p) { return p ? p->num_observations : 0; }
int64_t rsba_problem_num_parameters(const rsba_problem* p) { return p ? p->num_parameters() : 0; }
int32_t rsba_problem_num_observations_per_time_camera(const rsba_problem* p, int32_t t, int32_t c) {
  if (!p || !p->is_marker_chain() || t < 0 || t >= p->num_times || c < 0 || c >= p->num_cameras) return 0;
  return p->obs_per_time_camera[t * p->num_cameras + c] * 4;  // bundle_adjustment.cpp:29-32
}
const double* rsba_problem_observations(const rsba_problem* p) { return p ? p->observations.data() : nullptr; }
double* rsba_problem_parameters(rsba_problem* p) { return p ? p->parameters.data() : nullptr; }
int32_t rsba_problem_camera_idx(const rsba_problem* p, int64_t i) { return (p && i >= 0 && i < p->num_observations) ? p->camera_index[i] : -1; }
int32_t rsba_problem_point_idx(const rsba_problem* p, int64_t i) { return (p && !p->is_marker_chain() && i >= 0 && i < p->num_observations) ? p->point_index[i] : -1; }
int32_t rsba_problem_time_idx(const rsba_problem* p, int64_t i) { return (p && p->is_marker_chain() && i >= 0 && i < p->num_observations) ? p->time_index[i] : -1; }
int32_t rsba_problem_marker_idx(const rsba_problem* p, int64_t i) { return (p && p->is_marker_chain() && i >= 0 && i < p->num_observations) ? p->marker_index[i] : -1; }
double* rsba_problem_camera_parameters(rsba_problem* p, int32_t c) { return (p && c >= 0 && c < p->num_cameras) ? p->parameters.data() + 6 * c : nullptr; }
double* rsba_problem_marker_transform(rsba_problem* p, int32_t m) {
  return (p && p->is_marker_chain() && m >= 0 && m < p->num_markers) ? p->parameters.data() + 6 * (p->num_cameras + p->num_times + m) : nullptr;
}
int rsba_problem_point3d_coordinates(const rsba_problem* p, double* out) {
  if (!p || !out || !p->is_marker_chain()) return RSBA_ERR_ARG;
  rsba::MarkerCorners3d(*p, out);
  return RSBA_OK;
}

void rsba_options_default(rsba_options* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50; o->max_num_consecutive_invalid_steps = 5; o->jacobi_scaling = 1; o->minimizer_progress_to_stdout = 0;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8; o->huber_delta = 0.0;
  o->device = -1; o->schur_impl = 1; o->profile_kernels = 0; o->rank = 0; o->world_size = 1; o->comm_unique_id = nullptr; o->stream = nullptr;
  o->max_solver_time_in_seconds = 1e9;
}

int rsba_read_intrinsics_xml(const char* path, double* out4) { return rsba::ReadIntrinsicsXml(path, out4); }
int rsba_write_outputs(rsba_problem* p, const char* xml, const char* dir, const char* p3d) {
  if (!p) return RSBA_ERR_ARG;
  return rsba::WriteOutputs(*p, xml, dir, p3d);
}

}  // extern "C"
