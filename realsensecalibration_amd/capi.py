"""ctypes binding of librsba.so (include/rsba.h).  Plumbing for tests and bench.py; the product is the
shared library.  Loading fails loudly when the library has not been built: there is no fallback."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RSBA_LIB: another build of the library (tools/ build diagnostic variants to a scratch path and point this at them — the packaged
# library is never overwritten)
LIB_PATH = os.environ.get("RSBA_LIB") or os.path.join(HERE, "librsba.so")

OK, ERR_IO, ERR_FORMAT, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_COMM, ERR_UNSUPPORTED = range(8)
MODEL_POINTS, MODEL_MARKER_CHAIN, MODEL_MARKER_CHAIN_TEST2 = 0, 1, 2
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2

EXPORTS = [
    "rsba_version", "rsba_device_count", "rsba_error_string", "rsba_problem_create_points", "rsba_problem_create_marker_chain",
    "rsba_problem_load_points_file",
    "rsba_problem_load_correspondence", "rsba_problem_free", "rsba_problem_model", "rsba_problem_num_cameras",
    "rsba_problem_num_points", "rsba_problem_num_times", "rsba_problem_num_markers", "rsba_problem_num_observations",
    "rsba_problem_num_parameters", "rsba_problem_num_observations_per_time_camera", "rsba_problem_observations",
    "rsba_problem_parameters", "rsba_problem_camera_idx", "rsba_problem_point_idx", "rsba_problem_time_idx",
    "rsba_problem_marker_idx", "rsba_problem_camera_parameters", "rsba_problem_marker_transform",
    "rsba_problem_point3d_coordinates", "rsba_options_default", "rsba_solve", "rsba_solver_create", "rsba_solver_run",
    "rsba_solver_download", "rsba_solver_iterations", "rsba_solver_kernel_stats", "rsba_solver_final_costs",
    "rsba_solver_destroy", "rsba_points_linearize_and_step", "rsba_points_linearize_payload", "rsba_comm_unique_id", "rsba_comm_loopback_id", "rsba_read_intrinsics_xml",
    "rsba_write_outputs", "rsba_reprojection_error", "rsba_reprojection_check_files",
    "rsba_base_pose_from_marker_detection", "rsba_marker_pose_in_camera", "rsba_marker_corners_in_camera", "rsba_solve_pnp_epnp",
    "rsba_problem_initial_camera_poses", "rsba_problem_set_camera_constant", "rsba_problem_set_point_constant", "rsba_problem_set_parameter_block_constant", "rsba_solver_full_report", "rsba_solver_configure_run",
    "rsba_solver_comm_nranks", "rsba_solver_schedule_info", "rsba_comm_shm_id", "rsba_comm_finalize",
]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("jacobi_scaling", C.c_int32), ("minimizer_progress_to_stdout", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double), ("huber_delta", C.c_double),
                ("device", C.c_int32), ("schur_impl", C.c_int32), ("profile_kernels", C.c_int32), ("rank", C.c_int32),
                ("world_size", C.c_int32), ("loss_type", C.c_int32), ("comm_unique_id", C.c_void_p), ("stream", C.c_void_p),
                ("max_solver_time_in_seconds", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("stop_reason", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("num_iterations", C.c_int32), ("reserved", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("minimizer_seconds", C.c_double),
                ("setup_seconds", C.c_double)]


class Iteration(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32),
                ("linear_solver_iterations", C.c_int32), ("cost", C.c_double), ("cost_change", C.c_double),
                ("gradient_max_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double), ("iteration_time_in_seconds", C.c_double),
                ("cumulative_time_in_seconds", C.c_double)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


class ScheduleInfo(C.Structure):
    _fields_ = [("schedule", C.c_int32), ("stalls", C.c_int32), ("fallbacks", C.c_int32), ("comm_nranks", C.c_int32),
                ("chol_workgroups", C.c_int32), ("schur_impl", C.c_int32), ("comm_kind", C.c_char * 16)]


class RsbaError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s: %s (code %d)" % (what, error_string(code), code))


_LIB = None


def load():
    """dlopen librsba.so and check that every symbol of include/rsba.h is exported."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("librsba.so is not built: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                          "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise ImportError("librsba.so lacks symbols declared in include/rsba.h: %s" % missing)
    lib.rsba_error_string.restype = C.c_char_p
    lib.rsba_problem_parameters.restype = C.POINTER(C.c_double)
    lib.rsba_problem_observations.restype = C.POINTER(C.c_double)
    lib.rsba_problem_camera_parameters.restype = C.POINTER(C.c_double)
    lib.rsba_problem_marker_transform.restype = C.POINTER(C.c_double)
    lib.rsba_problem_num_observations.restype = C.c_int64
    lib.rsba_problem_num_parameters.restype = C.c_int64
    lib.rsba_problem_create_points.argtypes = [C.c_int32, C.c_int32, C.c_int64] + [C.c_void_p] * 6
    lib.rsba_problem_create_marker_chain.argtypes = [C.c_int32] * 4 + [C.c_int64] + [C.c_void_p] * 6 + [C.c_double, C.c_void_p]
    lib.rsba_problem_load_points_file.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    lib.rsba_problem_load_correspondence.argtypes = [C.c_char_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]
    lib.rsba_problem_free.argtypes = [C.c_void_p]
    for f in ("rsba_problem_model", "rsba_problem_num_cameras", "rsba_problem_num_points", "rsba_problem_num_times",
              "rsba_problem_num_markers", "rsba_problem_num_observations", "rsba_problem_num_parameters",
              "rsba_problem_parameters", "rsba_problem_observations"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.rsba_problem_num_observations_per_time_camera.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    for f in ("rsba_problem_camera_idx", "rsba_problem_point_idx", "rsba_problem_time_idx", "rsba_problem_marker_idx"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_int64]
    lib.rsba_problem_camera_parameters.argtypes = [C.c_void_p, C.c_int32]
    lib.rsba_problem_marker_transform.argtypes = [C.c_void_p, C.c_int32]
    lib.rsba_problem_point3d_coordinates.argtypes = [C.c_void_p, C.c_void_p]
    lib.rsba_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rsba_solver_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rsba_solver_run.argtypes = [C.c_void_p, C.c_void_p]
    lib.rsba_solver_download.argtypes = [C.c_void_p]
    lib.rsba_solver_iterations.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.rsba_solver_kernel_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.rsba_solver_full_report.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    lib.rsba_solver_configure_run.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.rsba_solver_final_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rsba_solver_destroy.argtypes = [C.c_void_p]
    lib.rsba_solver_comm_nranks.argtypes = [C.c_void_p]
    lib.rsba_solver_schedule_info.argtypes = [C.c_void_p, C.c_void_p]
    lib.rsba_comm_shm_id.argtypes = [C.c_char_p, C.c_void_p]
    lib.rsba_points_linearize_and_step.argtypes = [C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 4
    lib.rsba_points_linearize_payload.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int64, C.c_void_p]
    lib.rsba_comm_unique_id.argtypes = [C.c_void_p]
    lib.rsba_comm_loopback_id.argtypes = [C.c_void_p]
    lib.rsba_read_intrinsics_xml.argtypes = [C.c_char_p, C.c_void_p]
    lib.rsba_write_outputs.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.rsba_reprojection_error.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rsba_reprojection_check_files.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rsba_base_pose_from_marker_detection.argtypes = [C.c_void_p] * 3
    lib.rsba_marker_pose_in_camera.argtypes = [C.c_void_p] * 3
    lib.rsba_marker_corners_in_camera.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    lib.rsba_solve_pnp_epnp.argtypes = [C.c_int32] + [C.c_void_p] * 4
    lib.rsba_problem_initial_camera_poses.argtypes = [C.c_void_p]
    lib.rsba_problem_set_camera_constant.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.rsba_problem_set_point_constant.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.rsba_problem_set_parameter_block_constant.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    _LIB = lib
    return lib


def error_string(code):
    return load().rsba_error_string(code).decode()


def _chk(code, what):
    if code != OK:
        raise RsbaError(code, what)


def default_options(**kw):
    o = Options()
    load().rsba_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class Problem:
    """Owns an rsba_problem*.  `params` is a numpy view of the problem's own parameter array."""

    def __init__(self, handle):
        self.h = handle
        lib = load()
        n = lib.rsba_problem_num_parameters(self.h)
        self.params = np.ctypeslib.as_array(lib.rsba_problem_parameters(self.h), shape=(n,))

    @classmethod
    def points(cls, prob):
        h = C.c_void_p()
        cam = np.ascontiguousarray(prob["cam_idx"], np.int32)
        pt = np.ascontiguousarray(prob["pt_idx"], np.int32)
        obs = np.ascontiguousarray(prob["obs"], np.float64)
        par = np.ascontiguousarray(prob["params"], np.float64)
        intr = np.ascontiguousarray(prob["intr"], np.float64)
        _chk(load().rsba_problem_create_points(prob["C"], prob["P"], prob["N"], _vp(cam), _vp(pt), _vp(obs), _vp(par), _vp(intr),
                                               C.byref(h)), "rsba_problem_create_points")
        return cls(h)

    @classmethod
    def marker_chain(cls, prob, model=MODEL_MARKER_CHAIN):
        """prob: dict with T, C, M, N, t, c, m (int32 per row), obs (N x 8), params (6 (C + T + M)), intr (C x 4), marker_side."""
        h = C.c_void_p()
        t, c, m = (np.ascontiguousarray(prob[k], np.int32) for k in ("t", "c", "m"))
        obs = np.ascontiguousarray(prob["obs"], np.float64)
        par = np.ascontiguousarray(prob["params"], np.float64)
        intr = np.ascontiguousarray(prob["intr"], np.float64)
        _chk(load().rsba_problem_create_marker_chain(model, prob["C"], prob["T"], prob["M"], prob["N"], _vp(t), _vp(c), _vp(m), _vp(obs),
                                                     _vp(par), _vp(intr), C.c_double(prob["marker_side"]), C.byref(h)),
             "rsba_problem_create_marker_chain")
        return cls(h)

    @classmethod
    def points_file(cls, path, intrinsics4):
        h = C.c_void_p()
        k = np.ascontiguousarray(intrinsics4, np.float64)
        _chk(load().rsba_problem_load_points_file(path.encode(), _vp(k), C.byref(h)), "rsba_problem_load_points_file")
        return cls(h)

    @classmethod
    def correspondence(cls, path, model, marker_side, intrinsics):
        h = C.c_void_p()
        k = np.ascontiguousarray(intrinsics, np.float64)
        _chk(load().rsba_problem_load_correspondence(path.encode(), model, marker_side, _vp(k), C.byref(h)),
             "rsba_problem_load_correspondence")
        return cls(h)

    def __getattr__(self, name):
        f = {"model": "rsba_problem_model", "num_cameras": "rsba_problem_num_cameras", "num_points": "rsba_problem_num_points",
             "num_times": "rsba_problem_num_times", "num_markers": "rsba_problem_num_markers",
             "num_observations": "rsba_problem_num_observations", "num_parameters": "rsba_problem_num_parameters"}.get(name)
        if f is None:
            raise AttributeError(name)
        return getattr(load(), f)(self.h)

    def num_observations_per_time_camera(self, t, c):
        return load().rsba_problem_num_observations_per_time_camera(self.h, t, c)

    def point3d(self):
        out = np.zeros((4 * self.num_observations, 3))
        _chk(load().rsba_problem_point3d_coordinates(self.h, _vp(out)), "rsba_problem_point3d_coordinates")
        return out

    def write_outputs(self, xml=None, extrinsics_dir=None, point3d=None):
        e = lambda s: s.encode() if s else None  # noqa: E731
        _chk(load().rsba_write_outputs(self.h, e(xml), e(extrinsics_dir), e(point3d)), "rsba_write_outputs")

    def reprojection_error(self, opts=None):
        err, rms = C.c_double(), C.c_double()
        o = opts or default_options()
        _chk(load().rsba_reprojection_error(self.h, C.byref(o), C.byref(err), C.byref(rms)), "rsba_reprojection_error")
        return err.value, rms.value

    def set_camera_constant(self, camera_idx, constant=True):
        _chk(load().rsba_problem_set_camera_constant(self.h, camera_idx, 1 if constant else 0), "rsba_problem_set_camera_constant")

    def set_point_constant(self, point_idx, constant=True):
        _chk(load().rsba_problem_set_point_constant(self.h, point_idx, 1 if constant else 0), "rsba_problem_set_point_constant")

    def set_parameter_block_constant(self, parameter_offset, constant=True):
        _chk(load().rsba_problem_set_parameter_block_constant(self.h, parameter_offset, 1 if constant else 0), "rsba_problem_set_parameter_block_constant")

    def initial_camera_poses(self):
        _chk(load().rsba_problem_initial_camera_poses(self.h), "rsba_problem_initial_camera_poses")

    def solve(self, opts=None):
        s = Summary()
        o = opts or default_options()
        _chk(load().rsba_solve(self.h, C.byref(o), C.byref(s)), "rsba_solve")
        return s

    def close(self):
        if self.h:
            self.params = None
            load().rsba_problem_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Solver:
    def __init__(self, problem, opts=None):
        self.problem = problem
        self.opts = opts or default_options()
        self.h = C.c_void_p()
        _chk(load().rsba_solver_create(problem.h, C.byref(self.opts), C.byref(self.h)), "rsba_solver_create")

    def run(self):
        s = Summary()
        _chk(load().rsba_solver_run(self.h, C.byref(s)), "rsba_solver_run")
        return s

    def configure_run(self, max_num_iterations, profile_kernels=0):
        """Options of the next run() of this solver (ceres::Solve takes them per call); drops the kernel statistics."""
        _chk(load().rsba_solver_configure_run(self.h, int(max_num_iterations), int(profile_kernels)), "rsba_solver_configure_run")

    def download(self):
        _chk(load().rsba_solver_download(self.h), "rsba_solver_download")

    def iterations(self, cap=256):
        arr = (Iteration * cap)()
        n = load().rsba_solver_iterations(self.h, arr, cap)
        return np.array([[a.iteration, a.cost, a.cost_change, a.gradient_max_norm, a.step_norm, a.relative_decrease,
                          a.trust_region_radius, a.step_is_valid + 2 * a.step_is_successful] for a in arr[:n]])

    def iteration_times(self, cap=256):
        """iteration_time_in_seconds of the latest run's rows (row 0: the evaluation at the start), as the host observed them."""
        arr = (Iteration * cap)()
        n = load().rsba_solver_iterations(self.h, arr, cap)
        return np.array([a.iteration_time_in_seconds for a in arr[:n]])

    def kernel_stats(self, cap=32):
        arr = (KernelStat * cap)()
        n = load().rsba_solver_kernel_stats(self.h, arr, cap)
        return {a.name.decode(): (a.launches, a.total_ms) for a in arr[:n]}

    def comm_nranks(self):
        """ncclCommCount of the solver's communicator (1 without one)."""
        return int(load().rsba_solver_comm_nranks(self.h))

    def schedule_info(self):
        """Which schedule this solver runs and what stalled so far (rsba_schedule_info) as a dict."""
        i = ScheduleInfo()
        _chk(load().rsba_solver_schedule_info(self.h, C.byref(i)), "rsba_solver_schedule_info")
        return {"schedule": ("sequential", "pipelined", "pipelined_mg")[i.schedule], "stalls": int(i.stalls), "fallbacks": int(i.fallbacks),
                "comm_nranks": int(i.comm_nranks), "chol_workgroups": int(i.chol_workgroups), "schur_impl": int(i.schur_impl),
                "comm_kind": i.comm_kind.decode()}

    def full_report(self):
        n = load().rsba_solver_full_report(self.h, None, 0)
        if n < 0:
            raise RsbaError(3, "rsba_solver_full_report")
        buf = C.create_string_buffer(n + 1)
        load().rsba_solver_full_report(self.h, buf, n + 1)
        return buf.value.decode()

    def final_costs(self):
        c, ss = C.c_double(), C.c_double()
        _chk(load().rsba_solver_final_costs(self.h, C.byref(c), C.byref(ss)), "rsba_solver_final_costs")
        return c.value, ss.value

    def close(self):
        if self.h:
            load().rsba_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_points(prob, opts=None):
    """dict from synthetic.make_problem -> (parameters, Summary, iteration log)."""
    p = Problem.points(prob)
    o = opts or default_options(huber_delta=prob.get("huber_delta", 0.0))
    sv = Solver(p, o)
    try:
        s = sv.run()
        sv.download()
        log = sv.iterations()
        return p.params.copy(), s, log
    finally:
        sv.close()
        p.close()


def points_linearize_and_step(prob, radius, opts=None):
    p = Problem.points(prob)
    nc, n = 6 * prob["C"], 6 * prob["C"] + 3 * prob["P"]
    S, rhs, delta, scal = np.zeros((nc, nc)), np.zeros(nc), np.zeros(n), np.zeros(8)
    o = opts or default_options()
    try:
        _chk(load().rsba_points_linearize_and_step(p.h, C.byref(o), radius, _vp(S), _vp(rhs), _vp(delta), _vp(scal)),
             "rsba_points_linearize_and_step")
    finally:
        p.close()
    return dict(S=S, rhs=rhs, delta=delta, cost=scal[0], model_cost_change=scal[1], gradient_max_norm=scal[2],
                solve_ok=bool(scal[3]), cost_candidate=scal[4], step_norm=scal[5], x_norm=scal[6])


def points_linearize_payload(prob, radius, opts=None):
    """The all-reduce payload of one linearisation (see rsba.h); returns (payload without the last value, max |g_p|)."""
    problem = Problem.points(prob)
    o = opts or default_options()
    n = C.c_int64()
    _chk(load().rsba_points_linearize_payload(problem.h, C.byref(o), C.c_double(radius), None, 0, C.byref(n)), "rsba_points_linearize_payload")
    buf = np.zeros(n.value)
    _chk(load().rsba_points_linearize_payload(problem.h, C.byref(o), C.c_double(radius), _vp(buf), n.value, C.byref(n)), "rsba_points_linearize_payload")
    problem.close()
    return buf[:-1], buf[-1]


def comm_unique_id():
    buf = (C.c_char * 128)()
    _chk(load().rsba_comm_unique_id(buf), "rsba_comm_unique_id")
    return bytes(buf)


def comm_loopback_id():
    """Id of a loopback group: the ranks are solvers of this process on one GPU (rsba.h)."""
    buf = (C.c_char * 128)()
    _chk(load().rsba_comm_loopback_id(buf), "rsba_comm_loopback_id")
    return bytes(buf)


def comm_shm_id(name):
    """128-byte id of a SHARED-MEMORY group: the ranks are processes on one host (one GPU or several); `name` must be the same
    string on every rank and unique per group (rsba_comm_shm_id)."""
    buf = (C.c_char * 128)()
    _chk(load().rsba_comm_shm_id(name.encode(), buf), "rsba_comm_shm_id")
    return bytes(buf)


def solve_points_sharded_loopback(shards, opts_kw=None, max_iterations=None):
    """The multi-rank schedule on ONE GPU: shard r of `shards` (dicts from synthetic.make_problem(..., point_range=...), all
    with the same cameras) is rank r of a loopback group, created and run by its own host thread, as one process per GPU
    would.  Returns a list of (parameters, Summary, iteration log, comm_nranks) per rank; raises if a rank failed."""
    import threading
    world = len(shards)
    uid = C.create_string_buffer(comm_loopback_id(), 128)
    out, err = [None] * world, [None] * world

    def rank_main(r):
        p = sv = None
        try:
            o = default_options(rank=r, world_size=world, **(opts_kw or {}))
            o.comm_unique_id = C.cast(uid, C.c_void_p)
            p = Problem.points(shards[r])
            sv = Solver(p, o)
            if max_iterations is not None:
                sv.configure_run(max_iterations)
            nr = sv.comm_nranks()
            s = sv.run()
            sv.download()
            out[r] = (p.params.copy(), s, sv.iterations(), nr)
        except Exception as e:  # noqa: BLE001 (reported by the caller's thread)
            err[r] = e
        finally:
            if sv is not None:
                sv.close()
            if p is not None:
                p.close()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r, e in enumerate(err):
        if e is not None:
            raise RuntimeError("loopback rank %d failed: %s" % (r, e))
    return out


def read_intrinsics_xml(path):
    out = np.zeros(4)
    _chk(load().rsba_read_intrinsics_xml(path.encode(), _vp(out)), "rsba_read_intrinsics_xml")
    return out


def reprojection_check_files(correspondence_txt, point3d_txt, camera_transform_xml, intrinsics):
    """ReprojectionCheck::Reproject from its files (reprojection_check.cpp:5-101); returns (error, rms)."""
    intr = np.ascontiguousarray(intrinsics, dtype=np.float64)
    err, rms = C.c_double(), C.c_double()
    _chk(load().rsba_reprojection_check_files(str(correspondence_txt).encode(), str(point3d_txt).encode(),
                                               str(camera_transform_xml).encode(), _vp(intr), C.byref(err), C.byref(rms)),
         "rsba_reprojection_check_files")
    return err.value, rms.value


def _pose_op(fn, a, b):
    a, b, out = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64), np.zeros(6)
    _chk(fn(_vp(a), _vp(b), _vp(out)), "pose composition")
    return out


def base_pose_from_marker_detection(marker_from_camera, marker_from_base):
    return _pose_op(load().rsba_base_pose_from_marker_detection, marker_from_camera, marker_from_base)


def marker_pose_in_camera(base_from_camera, marker_from_base):
    return _pose_op(load().rsba_marker_pose_in_camera, base_from_camera, marker_from_base)


def marker_corners_in_camera(pose, marker_side):
    pose, out = np.ascontiguousarray(pose, np.float64), np.zeros((4, 3))
    _chk(load().rsba_marker_corners_in_camera(_vp(pose), C.c_double(marker_side), _vp(out)), "rsba_marker_corners_in_camera")
    return out


def solve_pnp_epnp(object_points, image_points, intrinsics4):
    obj = np.ascontiguousarray(object_points, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(image_points, np.float64).reshape(-1, 2)
    k, out = np.ascontiguousarray(intrinsics4, np.float64), np.zeros(6)
    _chk(load().rsba_solve_pnp_epnp(len(obj), _vp(obj), _vp(img), _vp(k), _vp(out)), "rsba_solve_pnp_epnp")
    return out
