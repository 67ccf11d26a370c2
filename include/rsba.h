/*
 * rsba.h — C ABI of the MI355X-native bundle-adjustment path (librsba.so).
 *
 * Drop-in boundary for the hot path of ajingu/RealSenseCalibration: everything that
 * Main_Calibration/bundle_adjustment.h + bundle_adjustment_manager.h hand to Ceres
 * (problem container, reprojection-error cost, Levenberg-Marquardt + DENSE_SCHUR solve, result
 * export).  Plain pointers and sizes only; no C++/torch types.  Each entry point names the
 * reference interface it replaces (paths relative to the reference repository root).
 *
 * Parameter layouts are the reference's own:
 *   point model        [C cameras x (rvec3, tvec3) | P points x xyz]
 *                      Test1_BundleAdjustment/bundle_adjustmenter.cpp:35-53
 *   marker-chain model [C cameras | T times | M markers] x (rvec3, tvec3)
 *                      Main_Calibration/bundle_adjustment.cpp:64-87
 * Solutions are written in place into the problem's parameter array, exactly as Ceres writes
 * through the raw pointers BALProblem hands out; blocks that no residual references are untouched.
 *
 * All compute runs on the GPU (gfx950).  There is no CPU fallback: every solve entry point returns
 * RSBA_ERR_NO_DEVICE when no HIP device is present.
 */
#ifndef RSBA_H_
#define RSBA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSBA_VERSION 100

/* ---- return codes (the reference exits the process instead: bundle_adjustment_manager.cpp:8-13) */
enum {
  RSBA_OK = 0,
  RSBA_ERR_IO = 1,        /* file could not be opened (BALProblem::loadFile returns false)   */
  RSBA_ERR_FORMAT = 2,    /* short or malformed file (reference: fscanfOrDie / unchecked)    */
  RSBA_ERR_ARG = 3,       /* NULL / out-of-range argument                                    */
  RSBA_ERR_HIP = 4,       /* a HIP runtime call failed                                       */
  RSBA_ERR_NO_DEVICE = 5, /* no gfx950 device visible: the product has no CPU path           */
  RSBA_ERR_COMM = 6,      /* RCCL failure                                                    */
  RSBA_ERR_UNSUPPORTED = 7
};

/* ---- models */
enum {
  RSBA_MODEL_POINTS = 0,             /* ReprojectionError<2,6,3>, Test1_BundleAdjustment/bundle_adjustmenter.cpp:106-148 */
  RSBA_MODEL_MARKER_CHAIN = 1,       /* four functors + wiring of Main_Calibration/bundle_adjustment_manager.cpp:21-88  */
  RSBA_MODEL_MARKER_CHAIN_TEST2 = 2  /* two functors + wiring of Test2_BundleAdjustment/main.cpp:64-96                  */
};

/* ---- ceres::TerminationType as Summary reports it (bundle_adjustment_manager.cpp:95 FullReport) */
enum { RSBA_CONVERGENCE = 0, RSBA_NO_CONVERGENCE = 1, RSBA_FAILURE = 2 };
enum {
  RSBA_STOP_NONE = 0, RSBA_STOP_GRADIENT = 1, RSBA_STOP_PARAMETER = 2, RSBA_STOP_FUNCTION = 3,
  RSBA_STOP_MAX_ITERATIONS = 4, RSBA_STOP_MIN_RADIUS = 5, RSBA_STOP_INVALID_STEPS = 6,
  RSBA_STOP_INITIAL_FAILURE = 7, RSBA_STOP_MAX_TIME = 8
};

enum rsba_loss { RSBA_LOSS_HUBER = 0, RSBA_LOSS_CAUCHY = 1 };

typedef struct rsba_problem rsba_problem; /* BALProblem (bundle_adjustment.h:18-54) */
typedef struct rsba_solver rsba_solver;   /* device-resident state of one ceres::Solve call */

/* ceres::Solver::Options as bundle_adjustment_manager.cpp:90-92 leaves it (linear_solver_type =
 * DENSE_SCHUR, everything else Ceres 1.14 defaults), plus the knobs this implementation adds. */
typedef struct rsba_options {
  int32_t max_num_iterations;                /* 50 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;                    /* 1 */
  int32_t minimizer_progress_to_stdout;      /* 0 (the reference sets true, :92) */
  double initial_trust_region_radius;        /* 1e4 */
  double max_trust_region_radius;            /* 1e16 */
  double min_trust_region_radius;            /* 1e-32 */
  double min_relative_decrease;              /* 1e-3 */
  double min_lm_diagonal;                    /* 1e-6 */
  double max_lm_diagonal;                    /* 1e32 */
  double function_tolerance;                 /* 1e-6 */
  double gradient_tolerance;                 /* 1e-10 */
  double parameter_tolerance;                /* 1e-8 */
  double huber_delta;                        /* the loss function's parameter a; 0 = no loss (the reference passes NULL, :38) */
  /* --- implementation knobs --- */
  int32_t device;          /* HIP device ordinal, -1 = current                                  */
  int32_t schur_impl;      /* point model: 0 = reference kernel (global atomics), 1 = tiled (default).          */
                           /* marker-chain: 0 = dense normal equations in one workgroup, 1 = eliminate the time */
                           /* blocks when the dense system has more than 384 unknowns, 2 = always eliminate     */
  int32_t profile_kernels; /* HIP events on the solver stream (rsba_solver_kernel_stats): 1 = around every kernel,
                              2 = only around the Schur pair kernel and the reduced-system solve */
  int32_t rank;            /* multi-GPU: this process' rank, 0..world_size-1                     */
  int32_t world_size;      /* 1 = single GPU.  >1: the problem handed in is this rank's point   */
                           /* shard (all cameras, its own points); the reduced camera system is */
                           /* all-reduced over RCCL each iteration                               */
  int32_t loss_type;       /* with huber_delta = a > 0: RSBA_LOSS_HUBER (0) = ceres::HuberLoss(a),       */
                           /* RSBA_LOSS_CAUCHY (1) = ceres::CauchyLoss(a); both have rho'' <= 0, so the  */
                           /* corrector scales residual and Jacobians by sqrt(rho') (corrector.cc)      */
  const void* comm_unique_id; /* world_size > 1: the 128-byte id from rsba_comm_unique_id (rank 0's) */
  void* stream;               /* hipStream_t to run on, NULL = a private stream                    */
  double max_solver_time_in_seconds; /* 1e9 (Ceres' default; Solver::Options, left alone by bundle_adjustment_manager.cpp:90-92): checked
                                        once per iteration IN FRONT OF the iteration limit and for the first time right behind
                                        iteration 0, against minimiser + set-up (preprocessor) time, as TrustRegionMinimizer's
                                        FinalizeIterationAndCheckIfMinimizerCanContinue does -> NO_CONVERGENCE, RSBA_STOP_MAX_TIME;
                                        a budget of 0 returns the start with no step taken.  Several ranks (round 6): RANK 0's clock decides for all of them —
                                        what it says when a step is launched is all-reduced with that step's candidate scalars, so every rank
                                        stops on the same iteration, one step behind the clock.
                                        Solver::Options::use_nonmonotonic_steps has no field: the reference leaves it false, and the
                                        device-side step decision (DecideStep) implements the monotonic rule only. */
} rsba_options;

typedef struct rsba_summary {
  int32_t termination_type; /* RSBA_CONVERGENCE / NO_CONVERGENCE / FAILURE */
  int32_t stop_reason;      /* RSBA_STOP_* */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_iterations; /* successful + unsuccessful, iteration 0 excluded */
  int32_t reserved;
  double initial_cost;
  double final_cost;
  double minimizer_seconds; /* wall time of the LM loop only (inputs already on the device) */
  double setup_seconds;     /* ordering + upload */
} rsba_summary;

/* one row per iteration (0 = initial evaluation), Ceres' progress table columns */
typedef struct rsba_iteration {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t linear_solver_iterations; /* ls_iter: 1 for the direct DENSE_SCHUR solve, 0 for row 0 */
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  double iteration_time_in_seconds;  /* iter_time */
  double cumulative_time_in_seconds; /* total_time, since the start of the minimiser loop */
} rsba_iteration;

/* Which schedule a solver runs and what has gone wrong so far (bench.py prints it; a first run on new hardware reads it) */
typedef struct rsba_schedule_info {
  int32_t schedule;        /* 0 sequential, 1 pipelined (factorisation gated on the Schur kernel's stages), 2 pipelined multi-GPU */
  int32_t stalls;          /* steps whose in-kernel wait ran out of its budget and were repeated sequentially */
  int32_t fallbacks;       /* permanent fallbacks taken (third pipeline stall; one-workgroup / multi-launch factorisation) */
  int32_t comm_nranks;     /* ranks of the communicator (1: none) */
  int32_t chol_workgroups; /* workgroups of the reduced system's factorisation (resident tiles above 64 cameras; 33 .. 64 cameras on one rank: six + the border's) */
  int32_t schur_impl;      /* as run: 0 when a shard with duplicate observations fell back to the atomic kernel */
  char comm_kind[16];      /* "none" | "rccl" | "loopback" | "shm" */
} rsba_schedule_info;

typedef struct rsba_kernel_stat {
  char name[48];
  int64_t launches;
  double total_ms; /* HIP-event time on the solver stream, profile_kernels = 1 only */
} rsba_kernel_stat;

int rsba_version(void);
int rsba_device_count(void);
const char* rsba_error_string(int code);

/* ------------------------------------------------------------------ problem container */
/* Point model from arrays (what Test1's BALProblem::LoadFile builds, bundle_adjustmenter.cpp:55-85).
 * intrinsics: 4 doubles per camera (fx, fy, ppx, ppy) = K(0,0), K(1,1), K(0,2), K(1,2)
 * (ReprojectionError ctor, :113-120).  Arrays are copied. */
int rsba_problem_create_points(int32_t num_cameras, int32_t num_points, int64_t num_observations,
                               const int32_t* camera_index, const int32_t* point_index,
                               const double* observations /* 2 per observation */,
                               const double* parameters /* 6C + 3P */,
                               const double* intrinsics /* 4C */, rsba_problem** out);

/* Marker-chain problem from arrays (what BALProblem::loadFile, bundle_adjustment.cpp:132-187, leaves in memory): one row
 * per detected marker = residual block; observations 8 per row (4 corners x (u, v)); parameters [C | T | M] x 6
 * (rvec, tvec), camera 0 / marker 0 being the fixed base blocks of RSBA_MODEL_MARKER_CHAIN. */
int rsba_problem_create_marker_chain(int32_t model, int32_t num_cameras, int32_t num_times, int32_t num_markers,
                                     int64_t num_observations, const int32_t* time_index, const int32_t* camera_index,
                                     const int32_t* marker_index, const double* observations, const double* parameters,
                                     const double* intrinsics, double marker_side, rsba_problem** out);

/* ceres::Problem::SetParameterBlockConstant on a camera block of the point model (not used by the reference; SURVEY
 * §8f rank 4): the camera keeps its value, has no columns in the linear system and does not count in the norms of the
 * convergence tests.  Fixing one camera removes the gauge freedom of a free network. */
int rsba_problem_set_camera_constant(rsba_problem* p, int32_t camera_idx, int32_t constant);
/* ... on a POINT block (round 6): the point keeps its value, is not eliminated (no block in the Schur complement, no step), its
 * observations still count in the cost and in their cameras' blocks, and it is left out of the norms of the convergence tests —
 * Ceres removes a constant block from the program.  The tiled Schur kernel only (schur_impl != 0; rsba_solver_create returns
 * RSBA_ERR_UNSUPPORTED with the atomic kernel, also when duplicate observations select it). */
int rsba_problem_set_point_constant(rsba_problem* p, int32_t point_idx, int32_t constant);
/* ceres::Problem::SetParameterBlockConstant(values) as the reference would call it — with the block's place in the parameter array
 * (`parameter_offset` = values - parameters_: a multiple of 6 for a pose block, 6 C + 3 j for point j of the point model).  Point model: the
 * two calls above.  Marker-chain models (round 6): any camera / time / marker block of [C | T | M] (bundle_adjustment.cpp:64-87) — the
 * block keeps its transform in every residual that names it and leaves the program; the solve then runs the dense path (the
 * time-eliminating one does not know constant blocks). */
int rsba_problem_set_parameter_block_constant(rsba_problem* p, int64_t parameter_offset, int32_t constant);

/* Test1 file "two_cam_data.txt": `C P`, P rows `cam pt u v` (one observation per point,
 * bundle_adjustmenter.cpp:62-64), C x (rvec row, tvec row), P rows xyz.  Also accepts the extended
 * first line `C P N` with N observation rows.  One intrinsics 4-vector is used for every
 * observation, as Test1_BundleAdjustment/main.cpp:73-74 does. */
int rsba_problem_load_points_file(const char* path, const double* intrinsics4, rsba_problem** out);

/* BALProblem::loadFile for correspondence.txt (bundle_adjustment.cpp:132-187).
 * model: RSBA_MODEL_MARKER_CHAIN or RSBA_MODEL_MARKER_CHAIN_TEST2; marker_side: my_const.h:9;
 * intrinsics: 4 per camera index, in SERIAL_NUMBERS order (my_const.h:15). */
int rsba_problem_load_correspondence(const char* path, int32_t model, double marker_side,
                                     const double* intrinsics /* 4C */, rsba_problem** out);

/* ------------------------------------------------------------------ initial guesses (the reference's front end)
 * What Correspondencer computes between the ArUco detections and correspondence.txt, without OpenCV.  Poses are
 * 6 doubles (rvec, tvec), p_out = R(rvec) p_in + tvec.  Host code; zero lens distortion (the committed intrinsics). */
/* correspondencer.cpp:119-127: the base marker's pose in the main camera from the detection of another marker and
 * that marker's pose in the base marker's frame (my_io GetMarkerGeometry). */
int rsba_base_pose_from_marker_detection(const double* marker_from_camera, const double* marker_from_base,
                                         double* base_from_camera);
/* correspondencer.cpp:137-147: marker i in the main camera = base pose o (marker i in the base marker's frame). */
int rsba_marker_pose_in_camera(const double* base_from_camera, const double* marker_from_base, double* marker_from_camera);
/* Correspondencer::GetCornersInCameraWorld (correspondencer.cpp:5-39): top-left, top-right, bottom-right, bottom-left. */
int rsba_marker_corners_in_camera(const double* pose, double marker_side, double* out12);
/* cv::solvePnP(object, image, K, dist = 0, rvec, tvec, false, SOLVEPNP_EPNP) as correspondencer.cpp:192-195 calls it.
 * n >= 4 points (the reference exits below 4, :185-190); RSBA_ERR_UNSUPPORTED for a coplanar point set. */
int rsba_solve_pnp_epnp(int32_t n, const double* object_points /* 3n */, const double* image_points /* 2n */,
                        const double* intrinsics4, double* pose);
/* Correspondencer::CalculateTransforms (correspondencer.cpp:178-205) on a marker-chain problem whose time and marker
 * blocks are filled: camera 0 := identity, every other camera := EPnP over the corners of all markers it detected. */
int rsba_problem_initial_camera_poses(rsba_problem* p);

void rsba_problem_free(rsba_problem* p);

/* BALProblem accessors (bundle_adjustment.h:36-53) */
int32_t rsba_problem_model(const rsba_problem* p);
int32_t rsba_problem_num_cameras(const rsba_problem* p);
int32_t rsba_problem_num_points(const rsba_problem* p);  /* point model; 0 otherwise */
int32_t rsba_problem_num_times(const rsba_problem* p);   /* marker-chain; 0 otherwise */
int32_t rsba_problem_num_markers(const rsba_problem* p); /* marker-chain; 0 otherwise */
int64_t rsba_problem_num_observations(const rsba_problem* p);
int64_t rsba_problem_num_parameters(const rsba_problem* p);
/* count x 4 corners, as BALProblem::num_observations_per_time_camera returns (bundle_adjustment.cpp:29-32) */
int32_t rsba_problem_num_observations_per_time_camera(const rsba_problem* p, int32_t time_idx, int32_t camera_idx);
const double* rsba_problem_observations(const rsba_problem* p);
double* rsba_problem_parameters(rsba_problem* p); /* mutable: results land here */
int32_t rsba_problem_camera_idx(const rsba_problem* p, int64_t observation);
int32_t rsba_problem_point_idx(const rsba_problem* p, int64_t observation);  /* point model */
int32_t rsba_problem_time_idx(const rsba_problem* p, int64_t observation);   /* marker-chain */
int32_t rsba_problem_marker_idx(const rsba_problem* p, int64_t observation); /* marker-chain */
double* rsba_problem_camera_parameters(rsba_problem* p, int32_t camera_idx);
double* rsba_problem_marker_transform(rsba_problem* p, int32_t marker_idx);
/* BALProblem::getPoint3dCoordinates (bundle_adjustment.cpp:89-130): 4 corners x xyz per observation */
int rsba_problem_point3d_coordinates(const rsba_problem* p, double* out /* 12 per observation */);

/* ------------------------------------------------------------------ solve */
void rsba_options_default(rsba_options* o);

/* BAManager::StartBA / ceres::Solve (bundle_adjustment_manager.cpp:16-96; Test1 main.cpp:63-87):
 * upload, minimise on the GPU, write the solution back into the problem's parameter array. */
int rsba_solve(rsba_problem* p, const rsba_options* o, rsba_summary* summary);

/* The same in three steps, so a caller (bench.py) can time the minimiser with inputs resident in HBM. */
int rsba_solver_create(rsba_problem* p, const rsba_options* o, rsba_solver** out);
int rsba_solver_run(rsba_solver* s, rsba_summary* summary); /* LM loop; restarts from the uploaded state */
int rsba_solver_download(rsba_solver* s);                    /* device state -> problem parameters */
/* ceres::Solve takes its Solver::Options per call (bundle_adjustment_manager.cpp:90-94): the next rsba_solver_run of this
 * solver stops after max_num_iterations and records kernel times as profile_kernels says (rsba_options); the kernel
 * statistics collected so far are dropped.  bench.py warms a solver up with W iterations, then times K on the same one. */
int rsba_solver_configure_run(rsba_solver* s, int32_t max_num_iterations, int32_t profile_kernels);
int rsba_solver_iterations(const rsba_solver* s, rsba_iteration* out, int32_t capacity); /* rows written */
int rsba_solver_kernel_stats(const rsba_solver* s, rsba_kernel_stat* out, int32_t capacity);
/* final 1/2 sum rho and sum of squared raw residuals of the last run (all ranks' total) */
/* Summary::FullReport() of the latest run (the reference prints it, bundle_adjustment_manager.cpp:95): problem sizes,
 * costs, iteration counts, times and the termination message.  snprintf semantics: returns the length needed. */
int rsba_solver_full_report(const rsba_solver* s, char* buf, int32_t capacity);
int rsba_solver_final_costs(const rsba_solver* s, double* cost, double* sum_sq_residuals);
void rsba_solver_destroy(rsba_solver* s);

/* Stage-level entry (tests): one linearisation of the point model at the current parameters with a
 * given trust-region radius.  Any output may be NULL.
 *   S       (6C)^2 reduced camera matrix, Jacobi-scaled, LM-damped, full symmetric, row-major
 *   rhs     6C
 *   delta   6C + 3P: the LM step in parameter space (x_candidate - x)
 *   scalars [0] cost at x, [1] model cost change, [2] max|gradient|, [3] 1 if the Cholesky succeeded,
 *           [4] cost at x + delta, [5] |delta|, [6] |x| */
int rsba_points_linearize_and_step(rsba_problem* p, const rsba_options* o, double radius, double* S,
                                   double* rhs, double* delta, double* scalars /* 8 */);

/* Stage-level entry (tests): the payload the multi-GPU path all-reduces after one linearisation of this problem
 * (a rank's point shard, or the whole problem): S (6C)^2 unscaled / undamped, full symmetric | g_c (6C) | rhs
 * correction (6C) | diag U (6C) | 8 scalars (cost sum, |points|^2, failed point blocks, ...), followed by the one value
 * that is max-reduced (max |g_p|).  Additive over disjoint point shards: the sum of the shards' payloads is the
 * payload of the union.  payload == NULL: only *count (doubles needed) is returned. */
int rsba_points_linearize_payload(rsba_problem* p, const rsba_options* o, double radius, double* payload, int64_t capacity,
                                  int64_t* count);

/* ------------------------------------------------------------------ multi-GPU bootstrap */
/* ncclGetUniqueId: rank 0 calls this and ships the 128 bytes to the other ranks (bench.py does it
 * through torch.distributed); every rank then passes it in rsba_options.comm_unique_id. */
int rsba_comm_unique_id(void* out128);
/* The same 128 bytes for a LOOPBACK group: world_size solver objects of ONE process on ONE GPU, each created and run by its own
 * host thread with this id in rsba_options.comm_unique_id (and its rank).  The collectives of the multi-GPU schedule are then
 * sums over the group's solvers on that GPU instead of ncclAllReduce over xGMI (csrc/ba_comm.hpp): the whole N > 1 schedule
 * — sharded upload, three collectives per LM step, the summed stall flag — on a one-GPU box.  The ranks take turns on the
 * device, so it measures nothing; it is how the multi-rank code path is tested where only one GPU is visible.  The reference
 * has no counterpart (single-threaded: Main_Calibration/bundle_adjustment_manager.cpp:90-92). */
int rsba_comm_loopback_id(void* out128);
/* The same 128 bytes for a SHARED-MEMORY group (round 5): world_size PROCESSES of one host — on one GPU or several — whose
 * collectives are staged through a POSIX shared-memory segment named after `name` ([A-Za-z0-9_.-], at most 80 characters, the
 * same string on every rank, unique per group and run) and added on the host in rank order.  It is how one process per rank,
 * bench.py's own launcher and the id bootstrap run end to end where RCCL cannot (two ranks on one device); sequential multi-GPU
 * schedule only.  No counterpart in the reference (single process, single thread). */
int rsba_comm_shm_id(const char* name, void* out128);
/* Destroys every RCCL communicator this process still holds (ncclCommDestroy); call once, after the last solver is destroyed and
 * before the process tears the HIP runtime down.  Optional: communicators otherwise live until exit. */
void rsba_comm_finalize(void);
/* ncclCommCount of the solver's communicator: the number of ranks its all-reduces really span (1 without a
 * communicator).  bench.py prints it as `rccl_nranks` and refuses to report a line when it differs from --gpus. */
int rsba_solver_comm_nranks(const rsba_solver* s);
/* The schedule in effect and the stalls / fallbacks so far (rsba_schedule_info).  The reference has no counterpart: Ceres runs
 * one thread (bundle_adjustment_manager.cpp:90-92). */
int rsba_solver_schedule_info(const rsba_solver* s, rsba_schedule_info* out);

/* ------------------------------------------------------------------ files either side of the path */
/* IO::GetIntrinsics (my_io.cpp:5-31) without OpenCV: reads <intrinsics> 3x3 from an OpenCV
 * FileStorage XML and returns fx, fy, ppx, ppy. */
int rsba_read_intrinsics_xml(const char* path, double* out4);

/* BAManager::Write (bundle_adjustment_manager.cpp:98-175).  Any path may be NULL to skip that file.
 *   camera_transform_xml : R{i} 3x3 (Main) or rvec 3x1 (Test2 variant, main.cpp:128) + t{i}
 *   extrinsics_dir       : mat{i}.txt = [R^T | -R^T t] one value per line (:135-149)
 *   point3d_txt          : `4N T C`, count rows, 4N xyz rows (:154-174) */
int rsba_write_outputs(rsba_problem* p, const char* camera_transform_xml, const char* extrinsics_dir,
                       const char* point3d_txt);

/* ReprojectionCheck::Reproject's metric (reprojection_check.cpp:76-101) from the current
 * parameters, evaluated on the GPU: error = sum((du^2+dv^2)/2), rms = sqrt(2 error / (2 n_points)). */
int rsba_reprojection_error(rsba_problem* p, const rsba_options* o, double* error, double* rms);

/* ReprojectionCheck::Reproject end to end from the files it reads (reprojection_check.cpp:5-101): the 6-digit 3D
 * corners of point3d.txt, R{i} (3x3, or the 3x1 rvec of the Test2 variant) and t{i} of Camera_Transform.xml, and the
 * detected corners — taken from correspondence.txt and rounded to float32 as the reference holds them (Point2f,
 * :78) — projected on the GPU with zero distortion (:69).  intrinsics: fx, fy, ppx, ppy per camera.  On the
 * committed hongo files this prints the reference's 143.64 / 0.726696 (vs 0.726670 from the unrounded parameters). */
int rsba_reprojection_check_files(const char* correspondence_txt, const char* point3d_txt, const char* camera_transform_xml,
                                  const double* intrinsics, double* error, double* rms);

#ifdef __cplusplus
}
#endif
#endif /* RSBA_H_ */
