// ORACLE — TEST INFRASTRUCTURE ONLY (see ba_oracle.hpp).  C entry points for ctypes.
#include "ba_oracle.hpp"

#include <chrono>
#include <cstdio>

using namespace oracle;

extern "C" {

struct OracleOptions {
  int max_num_iterations;
  int jacobi_scaling;
  int num_threads;
  int reserved;
  double initial_trust_region_radius;
  double function_tolerance;
  double gradient_tolerance;
  double parameter_tolerance;
  double min_lm_diagonal;
  double max_lm_diagonal;
  double huber_delta;
};

struct OracleSummary {
  int termination;
  int stop_reason;
  int num_successful_steps;
  int num_unsuccessful_steps;
  int num_iterations;
  int reserved;
  double initial_cost;
  double final_cost;
  double minimizer_seconds;
};

void oracle_options_default(OracleOptions* o) {
  Options d;
  o->max_num_iterations = d.max_num_iterations;
  o->jacobi_scaling = d.jacobi_scaling;
  o->num_threads = 1;
  o->reserved = 0;
  o->initial_trust_region_radius = d.initial_trust_region_radius;
  o->function_tolerance = d.function_tolerance;
  o->gradient_tolerance = d.gradient_tolerance;
  o->parameter_tolerance = d.parameter_tolerance;
  o->min_lm_diagonal = d.min_lm_diagonal;
  o->max_lm_diagonal = d.max_lm_diagonal;
  o->huber_delta = 0.0;
}

static Options ToOptions(const OracleOptions* o) {
  Options d;
  if (!o) return d;
  d.max_num_iterations = o->max_num_iterations;
  d.jacobi_scaling = o->jacobi_scaling;
  d.num_threads = o->num_threads > 0 ? o->num_threads : 1;
  d.initial_trust_region_radius = o->initial_trust_region_radius;
  d.function_tolerance = o->function_tolerance;
  d.gradient_tolerance = o->gradient_tolerance;
  d.parameter_tolerance = o->parameter_tolerance;
  d.min_lm_diagonal = o->min_lm_diagonal;
  d.max_lm_diagonal = o->max_lm_diagonal;
  d.huber_delta = o->huber_delta;
  return d;
}

// iter_log: up to max_log rows of 8 doubles:
// iteration, cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius, flags(valid + 2*successful)
static void FillSummary(const Summary& s, double seconds, OracleSummary* out, double* iter_log, int max_log) {
  if (out) {
    out->termination = s.termination; out->stop_reason = s.stop_reason;
    out->num_successful_steps = s.num_successful_steps; out->num_unsuccessful_steps = s.num_unsuccessful_steps;
    out->num_iterations = s.num_iterations; out->reserved = 0;
    out->initial_cost = s.initial_cost; out->final_cost = s.final_cost; out->minimizer_seconds = seconds;
  }
  if (iter_log) {
    for (int i = 0; i < (int)s.iterations.size() && i < max_log; ++i) {
      const IterationSummary& it = s.iterations[i];
      double* row = iter_log + 8 * i;
      row[0] = it.iteration; row[1] = it.cost; row[2] = it.cost_change; row[3] = it.gradient_max_norm;
      row[4] = it.step_norm; row[5] = it.relative_decrease; row[6] = it.trust_region_radius;
      row[7] = it.step_is_valid + 2 * it.step_is_successful;
    }
  }
}

void oracle_angle_axis_rotate_point(const double* aa, const double* pt, double* out) {
  AngleAxisRotatePoint<double>(aa, pt, out);
}

void oracle_rodrigues(const double* rvec, double* R) { Rodrigues(rvec, R); }

// Point functor value + AutoDiff Jacobian blocks (row-major 2x6, 2x3).
void oracle_point_residual_jacobian(const double* camera, const double* point, const double* intr4,
                                    const double* uv, double* r2, double* jc12, double* jp6) {
  typedef Jet<9> J9;
  Intrinsics K{intr4[0], intr4[1], intr4[2], intr4[3]};
  J9 cam[6], pt[3], jr[2];
  for (int k = 0; k < 6; ++k) cam[k] = J9(camera[k], k);
  for (int k = 0; k < 3; ++k) pt[k] = J9(point[k], 6 + k);
  PointReprojectionError<J9>(cam, pt, K, uv[0], uv[1], jr);
  for (int r = 0; r < 2; ++r) {
    r2[r] = jr[r].a;
    for (int k = 0; k < 6; ++k) jc12[6 * r + k] = jr[r].v[k];
    for (int k = 0; k < 3; ++k) jp6[3 * r + k] = jr[r].v[6 + k];
  }
}

// Marker-chain functor value + AutoDiff Jacobian (8 x 18, columns camera|time|marker; unused blocks 0).
void oracle_marker_residual_jacobian(const double* camera /*or NULL*/, const double* time, const double* marker /*or NULL*/,
                                     double marker_side, const double* intr4, const double* obs8, double* r8, double* j8x18) {
  typedef Jet<18> J18;
  Intrinsics K{intr4[0], intr4[1], intr4[2], intr4[3]};
  J18 cam[6], tim[6], mar[6], jr[8];
  for (int k = 0; k < 6; ++k) {
    if (camera) cam[k] = J18(camera[k], k);
    tim[k] = J18(time[k], 6 + k);
    if (marker) mar[k] = J18(marker[k], 12 + k);
  }
  MarkerChainReprojectionError<J18>(camera ? cam : nullptr, tim, marker ? mar : nullptr, marker_side / 2, K, obs8, jr);
  for (int r = 0; r < 8; ++r) { r8[r] = jr[r].a; for (int k = 0; k < 18; ++k) j8x18[18 * r + k] = jr[r].v[k]; }
}

int oracle_solve_marker_chain(int variant, int T, int C, int M, int N, const int* time_idx, const int* camera_idx,
                              const int* marker_idx, const double* observations, double* params /* in-out, 6(C+T+M) */,
                              const double* intrinsics4 /* 4 per camera */, double marker_side, const OracleOptions* oopt,
                              OracleSummary* out, double* iter_log, int max_log) {
  MarkerChainProblem p;
  p.variant = variant; p.T = T; p.C = C; p.M = M; p.N = N;
  p.time_idx = time_idx; p.camera_idx = camera_idx; p.marker_idx = marker_idx; p.observations = observations;
  std::vector<Intrinsics> K(C);
  for (int c = 0; c < C; ++c) K[c] = Intrinsics{intrinsics4[4 * c], intrinsics4[4 * c + 1], intrinsics4[4 * c + 2], intrinsics4[4 * c + 3]};
  p.intrinsics = K.data(); p.marker_side = marker_side;
  // variant + 16: the block-sparse model with the time blocks eliminated (MarkerChainSparseModel) — the same problem, the same
  // minimiser, at sizes the dense Jacobian cannot hold
  Summary s;
  std::vector<double> full(6 * p.num_blocks());
  double sec = 0;
  if (variant >= 16) {
    p.variant = variant - 16;
    MarkerChainSparseModel model(p, params, oopt && oopt->num_threads > 0 ? oopt->num_threads : 1);
    std::vector<double> x(model.num_parameters());
    model.GetActive(x.data());
    const auto t0 = std::chrono::steady_clock::now();
    TrustRegionMinimize(model, ToOptions(oopt), x.data(), &s);
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    model.Scatter(x.data(), full.data());
  } else {
    MarkerChainModel model(p, params);
    std::vector<double> x(model.num_parameters());
    model.GetActive(x.data());
    const auto t0 = std::chrono::steady_clock::now();
    TrustRegionMinimize(model, ToOptions(oopt), x.data(), &s);
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    model.Scatter(x.data(), full.data());
  }
  std::memcpy(params, full.data(), sizeof(double) * full.size());
  FillSummary(s, sec, out, iter_log, max_log);
  return s.termination;
}

// oracle_solve_marker_chain (dense model) with some blocks held constant: mask over the C + T + M blocks.
int oracle_solve_marker_chain_constant(int variant, int T, int C, int M, int N, const int* time_idx, const int* camera_idx,
                                       const int* marker_idx, const double* observations, double* params, const double* intrinsics4,
                                       double marker_side, const unsigned char* constant_block, const OracleOptions* oopt,
                                       OracleSummary* out, double* iter_log, int max_log) {
  MarkerChainProblem p;
  p.variant = variant; p.T = T; p.C = C; p.M = M; p.N = N;
  p.time_idx = time_idx; p.camera_idx = camera_idx; p.marker_idx = marker_idx; p.observations = observations;
  std::vector<Intrinsics> K(C);
  for (int c = 0; c < C; ++c) K[c] = Intrinsics{intrinsics4[4 * c], intrinsics4[4 * c + 1], intrinsics4[4 * c + 2], intrinsics4[4 * c + 3]};
  p.intrinsics = K.data(); p.marker_side = marker_side; p.constant_block = constant_block;
  Summary s;
  std::vector<double> full(6 * p.num_blocks());
  MarkerChainModel model(p, params);
  std::vector<double> x(model.num_parameters());
  model.GetActive(x.data());
  const auto t0 = std::chrono::steady_clock::now();
  TrustRegionMinimize(model, ToOptions(oopt), x.data(), &s);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  model.Scatter(x.data(), full.data());
  std::memcpy(params, full.data(), sizeof(double) * full.size());
  FillSummary(s, sec, out, iter_log, max_log);
  return s.termination;
}

double oracle_marker_chain_cost(int variant, int T, int C, int M, int N, const int* time_idx, const int* camera_idx,
                                const int* marker_idx, const double* observations, const double* params,
                                const double* intrinsics4, double marker_side) {
  MarkerChainProblem p;
  p.variant = variant; p.T = T; p.C = C; p.M = M; p.N = N;
  p.time_idx = time_idx; p.camera_idx = camera_idx; p.marker_idx = marker_idx; p.observations = observations;
  std::vector<Intrinsics> K(C);
  for (int c = 0; c < C; ++c) K[c] = Intrinsics{intrinsics4[4 * c], intrinsics4[4 * c + 1], intrinsics4[4 * c + 2], intrinsics4[4 * c + 3]};
  p.intrinsics = K.data(); p.marker_side = marker_side;
  MarkerChainModel model(p, params);
  std::vector<double> x(model.num_parameters());
  model.GetActive(x.data());
  double cost = 0;
  model.Evaluate(x.data(), &cost, false);
  return cost;
}

void oracle_marker_corners3d(int variant, int T, int C, int M, int N, const int* time_idx, const int* camera_idx,
                             const int* marker_idx, const double* params, double marker_side, double* out12) {
  MarkerChainProblem p;
  p.variant = variant; p.T = T; p.C = C; p.M = M; p.N = N;
  p.time_idx = time_idx; p.camera_idx = camera_idx; p.marker_idx = marker_idx; p.marker_side = marker_side;
  MarkerCorners3d(p, params, out12);
}

static PointProblem MakePoint(int C, int P, int64_t N, const int* camera_idx, const int* point_idx, const double* observations,
                              const double* intrinsics4, std::vector<Intrinsics>& K) {
  K.resize(C);
  for (int c = 0; c < C; ++c) K[c] = Intrinsics{intrinsics4[4 * c], intrinsics4[4 * c + 1], intrinsics4[4 * c + 2], intrinsics4[4 * c + 3]};
  PointProblem p; p.C = C; p.P = P; p.N = N; p.camera_idx = camera_idx; p.point_idx = point_idx;
  p.observations = observations; p.intrinsics = K.data();
  return p;
}

int oracle_solve_points(int C, int P, int64_t N, const int* camera_idx, const int* point_idx, const double* observations,
                        double* params /* in-out 6C+3P */, const double* intrinsics4, const OracleOptions* oopt,
                        OracleSummary* out, double* iter_log, int max_log) {
  std::vector<Intrinsics> K;
  PointProblem p = MakePoint(C, P, N, camera_idx, point_idx, observations, intrinsics4, K);
  Options opt = ToOptions(oopt);
  PointSchurModel model(p, opt);
  Summary s;
  const auto t0 = std::chrono::steady_clock::now();
  TrustRegionMinimize(model, opt, params, &s);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  FillSummary(s, sec, out, iter_log, max_log);
  return s.termination;
}

// oracle_solve_points with some camera blocks held constant (Problem::SetParameterBlockConstant).
int oracle_solve_points_constant(int C, int P, int64_t N, const int* camera_idx, const int* point_idx, const double* observations,
                                 double* params, const double* intrinsics4, const unsigned char* constant_camera,
                                 const OracleOptions* oopt, OracleSummary* out, double* iter_log, int max_log) {
  std::vector<Intrinsics> K;
  PointProblem p = MakePoint(C, P, N, camera_idx, point_idx, observations, intrinsics4, K);
  Options opt = ToOptions(oopt);
  std::vector<unsigned char> cp((size_t)6 * C + (size_t)3 * P, 0);
  for (int c = 0; c < C; ++c) if (constant_camera[c]) for (int k = 0; k < 6; ++k) cp[6 * c + k] = 1;
  opt.constant_camera = constant_camera; opt.constant_parameter = cp.data();
  PointSchurModel model(p, opt);
  Summary s;
  const auto t0 = std::chrono::steady_clock::now();
  TrustRegionMinimize(model, opt, params, &s);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  FillSummary(s, sec, out, iter_log, max_log);
  return s.termination;
}

// ... camera AND point blocks (either mask may be null).
int oracle_solve_points_constant_blocks(int C, int P, int64_t N, const int* camera_idx, const int* point_idx, const double* observations,
                                        double* params, const double* intrinsics4, const unsigned char* constant_camera,
                                        const unsigned char* constant_point, const OracleOptions* oopt, OracleSummary* out, double* iter_log, int max_log) {
  std::vector<Intrinsics> K;
  PointProblem p = MakePoint(C, P, N, camera_idx, point_idx, observations, intrinsics4, K);
  Options opt = ToOptions(oopt);
  std::vector<unsigned char> cp((size_t)6 * C + (size_t)3 * P, 0);
  if (constant_camera) for (int c = 0; c < C; ++c) if (constant_camera[c]) for (int k = 0; k < 6; ++k) cp[6 * c + k] = 1;
  if (constant_point) for (int j = 0; j < P; ++j) if (constant_point[j]) for (int k = 0; k < 3; ++k) cp[(size_t)6 * C + 3 * (size_t)j + k] = 1;
  opt.constant_camera = constant_camera; opt.constant_point = constant_point; opt.constant_parameter = cp.data();
  PointSchurModel model(p, opt);
  Summary s;
  const auto t0 = std::chrono::steady_clock::now();
  TrustRegionMinimize(model, opt, params, &s);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  FillSummary(s, sec, out, iter_log, max_log);
  return s.termination;
}

// Cost 1/2 sum rho(|r|^2) at params (operator()<double>), and sum of squared raw residuals for the RMS metric.
void oracle_points_cost(int C, int P, int64_t N, const int* camera_idx, const int* point_idx, const double* observations,
                        const double* params, const double* intrinsics4, double huber_delta, int num_threads,
                        double* cost, double* sum_sq) {
  std::vector<Intrinsics> K;
  PointProblem p = MakePoint(C, P, N, camera_idx, point_idx, observations, intrinsics4, K);
  double total = 0, ss = 0;
  const double* pts = params + 6 * C;
#pragma omp parallel for reduction(+ : total, ss) num_threads(num_threads > 0 ? num_threads : 1)
  for (int64_t i = 0; i < N; ++i) {
    double r[2];
    PointReprojectionError<double>(params + 6 * camera_idx[i], pts + 3 * point_idx[i], K[camera_idx[i]], observations[2 * i], observations[2 * i + 1], r);
    const double s = r[0] * r[0] + r[1] * r[1];
    double rho[3] = {s, 1, 0};
    if (huber_delta > 0) HuberEvaluate(huber_delta, s, rho);
    else if (huber_delta < 0) CauchyEvaluate(-huber_delta, s, rho);
    total += rho[0]; ss += s;
  }
  *cost = 0.5 * total; *sum_sq = ss;
}

// One linearisation at params with a given radius and (optional) Jacobi scale: the reduced camera
// system exactly as PointSchurModel::Solve would build it, plus the full LM step.  Stage-level checker
// for the HIP kernels.  scale may be NULL: then it is computed as 1/(1+||J_k||) (iteration-0 rule).
// Outputs (any may be NULL): S (6C)^2, rhs 6C, step (6C+3P, already negated and unscaled: delta),
// scale_out (6C+3P), gradient (6C+3P), scalars[4] = {cost, model_cost_change, gradient_max_norm, solve_ok}.
int oracle_points_linearize_and_step(int C, int P, int64_t N, const int* camera_idx, const int* point_idx,
                                     const double* observations, const double* params, const double* intrinsics4,
                                     const OracleOptions* oopt, double radius, const double* scale_in, double* S,
                                     double* rhs, double* delta, double* scale_out, double* gradient, double* scalars) {
  std::vector<Intrinsics> K;
  PointProblem p = MakePoint(C, P, N, camera_idx, point_idx, observations, intrinsics4, K);
  Options opt = ToOptions(oopt);
  PointSchurModel model(p, opt);
  const int n = model.num_parameters(), nc = 6 * C;
  double cost = 0;
  if (!model.Evaluate(params, &cost, true)) return -1;
  std::vector<double> scale(n, 1.0), diag(n), lmd(n), y(n);
  if (gradient) std::memcpy(gradient, model.gradient(), sizeof(double) * n);
  double gmax = 0; for (int i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(model.gradient()[i]));
  if (opt.jacobi_scaling) {
    if (scale_in) std::memcpy(scale.data(), scale_in, sizeof(double) * n);
    else { model.SquaredColumnNorm(scale.data()); for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(scale[i])); }
    model.ScaleColumns(scale.data());
  }
  if (scale_out) std::memcpy(scale_out, scale.data(), sizeof(double) * n);
  model.SquaredColumnNorm(diag.data());
  for (int i = 0; i < n; ++i) { diag[i] = std::min(std::max(diag[i], opt.min_lm_diagonal), opt.max_lm_diagonal); lmd[i] = std::sqrt(diag[i] / radius); }
  if (S || rhs) {
    std::vector<double> St((size_t)nc * nc), rt(nc);
    model.BuildReducedSystem(lmd.data(), St.data(), rt.data(), nullptr);
    if (S) std::memcpy(S, St.data(), sizeof(double) * St.size());
    if (rhs) std::memcpy(rhs, rt.data(), sizeof(double) * nc);
  }
  const bool ok = model.Solve(lmd.data(), y.data());
  double mcc = 0;
  if (ok) { for (int i = 0; i < n; ++i) y[i] = -y[i]; mcc = model.ModelCostChange(y.data()); }
  if (delta) for (int i = 0; i < n; ++i) delta[i] = y[i] * scale[i];
  if (scalars) { scalars[0] = cost; scalars[1] = mcc; scalars[2] = gmax; scalars[3] = ok ? 1.0 : 0.0; }
  return 0;
}

// Same linear system through the dense normal equations (no Schur): cross-check of the eliminator.
int oracle_points_dense_step(int C, int P, int64_t N, const int* camera_idx, const int* point_idx,
                             const double* observations, const double* params, const double* intrinsics4,
                             const OracleOptions* oopt, double radius, double* delta) {
  std::vector<Intrinsics> K;
  PointProblem p = MakePoint(C, P, N, camera_idx, point_idx, observations, intrinsics4, K);
  Options opt = ToOptions(oopt);
  PointSchurModel model(p, opt);
  const int n = model.num_parameters();
  double cost = 0;
  if (!model.Evaluate(params, &cost, true)) return -1;
  std::vector<double> scale(n, 1.0), diag(n), lmd(n), y(n);
  if (opt.jacobi_scaling) { model.SquaredColumnNorm(scale.data()); for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(scale[i])); model.ScaleColumns(scale.data()); }
  model.SquaredColumnNorm(diag.data());
  for (int i = 0; i < n; ++i) { diag[i] = std::min(std::max(diag[i], opt.min_lm_diagonal), opt.max_lm_diagonal); lmd[i] = std::sqrt(diag[i] / radius); }
  if (!PointDenseSolve(p, model.Jc(), model.Jp(), model.residuals(), lmd.data(), y.data())) return -2;
  for (int i = 0; i < n; ++i) delta[i] = -y[i] * scale[i];
  return 0;
}

int oracle_num_procs() {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

}  // extern "C"
