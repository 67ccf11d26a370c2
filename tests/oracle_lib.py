"""ctypes binding of oracle/liboracle.so and parsers for the committed fixtures.

Test infrastructure only: the product package never imports this module.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")

# my_const.h:15 (Main_Calibration) and Test2_BundleAdjustment/main.cpp serial order
SERIALS_MAIN = ["821312061029", "816612062327", "821212062536", "821212061326"]
SERIALS_TEST2 = ["819612072493", "825312072048"]
MARKER_SIDE_MAIN = 0.0148  # my_const.h:9
MARKER_SIDE_TEST2 = 0.048  # inferred from the corner spacing of test2/point3d.txt


class OracleOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("jacobi_scaling", C.c_int), ("num_threads", C.c_int),
                ("reserved", C.c_int), ("initial_trust_region_radius", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("huber_delta", C.c_double)]


class OracleSummary(C.Structure):
    _fields_ = [("termination", C.c_int), ("stop_reason", C.c_int), ("num_successful_steps", C.c_int),
                ("num_unsuccessful_steps", C.c_int), ("num_iterations", C.c_int), ("reserved", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("minimizer_seconds", C.c_double)]


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def dp(a):
    return _ptr(a, C.c_double)


def ip(a):
    return _ptr(a, C.c_int)


class Oracle:
    def __init__(self, path):
        self.lib = C.CDLL(path)
        L = self.lib
        L.oracle_marker_chain_cost.restype = C.c_double
        L.oracle_num_procs.restype = C.c_int

    def options(self, **kw):
        o = OracleOptions()
        self.lib.oracle_options_default(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def rotate(self, aa, pt):
        out = np.zeros(3)
        self.lib.oracle_angle_axis_rotate_point(dp(np.ascontiguousarray(aa, float)), dp(np.ascontiguousarray(pt, float)), dp(out))
        return out

    def rodrigues(self, rvec):
        R = np.zeros(9)
        self.lib.oracle_rodrigues(dp(np.ascontiguousarray(rvec, float)), dp(R))
        return R.reshape(3, 3)

    def point_residual_jacobian(self, cam, pt, intr, uv):
        r, jc, jp = np.zeros(2), np.zeros(12), np.zeros(6)
        self.lib.oracle_point_residual_jacobian(dp(np.ascontiguousarray(cam, float)), dp(np.ascontiguousarray(pt, float)),
                                                dp(np.ascontiguousarray(intr, float)), dp(np.ascontiguousarray(uv, float)), dp(r), dp(jc), dp(jp))
        return r, jc.reshape(2, 6), jp.reshape(2, 3)

    def marker_residual_jacobian(self, cam, tim, mar, side, intr, obs8):
        r, j = np.zeros(8), np.zeros(8 * 18)
        c = dp(np.ascontiguousarray(cam, float)) if cam is not None else None
        m = dp(np.ascontiguousarray(mar, float)) if mar is not None else None
        self.lib.oracle_marker_residual_jacobian(c, dp(np.ascontiguousarray(tim, float)), m, C.c_double(side),
                                                 dp(np.ascontiguousarray(intr, float)), dp(np.ascontiguousarray(obs8, float)), dp(r), dp(j))
        return r, j.reshape(8, 18)

    def solve_marker_chain(self, prob, variant, marker_side, intr, opts=None, max_log=64):
        params = prob["params"].copy()
        s = OracleSummary()
        log = np.zeros((max_log, 8))
        o = opts or self.options()
        self.lib.oracle_solve_marker_chain(variant, prob["T"], prob["C"], prob["M"], prob["N"], ip(prob["t"]), ip(prob["c"]),
                                           ip(prob["m"]), dp(prob["obs"]), dp(params), dp(np.ascontiguousarray(intr, float)),
                                           C.c_double(marker_side), C.byref(o), C.byref(s), dp(log), max_log)
        return params, s, log[: s.num_iterations + 1]

    def solve_marker_chain_constant(self, prob, variant, marker_side, intr, constant_blocks, opts=None, max_log=64):
        """constant_blocks: indices into the [C cameras | T times | M markers] block array."""
        params = prob["params"].copy()
        s = OracleSummary()
        log = np.zeros((max_log, 8))
        o = opts or self.options()
        mask = np.zeros(prob["C"] + prob["T"] + prob["M"], np.uint8)
        mask[list(constant_blocks)] = 1
        self.lib.oracle_solve_marker_chain_constant(variant, prob["T"], prob["C"], prob["M"], prob["N"], ip(prob["t"]), ip(prob["c"]),
                                                    ip(prob["m"]), dp(prob["obs"]), dp(params), dp(np.ascontiguousarray(intr, float)),
                                                    C.c_double(marker_side), mask.ctypes.data_as(C.c_void_p), C.byref(o), C.byref(s), dp(log), max_log)
        return params, s, log[: s.num_iterations + 1]

    def marker_chain_cost(self, prob, variant, marker_side, intr, params):
        return self.lib.oracle_marker_chain_cost(variant, prob["T"], prob["C"], prob["M"], prob["N"], ip(prob["t"]), ip(prob["c"]),
                                                 ip(prob["m"]), dp(prob["obs"]), dp(np.ascontiguousarray(params, float)),
                                                 dp(np.ascontiguousarray(intr, float)), C.c_double(marker_side))

    def marker_corners3d(self, prob, variant, marker_side, params):
        out = np.zeros(12 * prob["N"])
        self.lib.oracle_marker_corners3d(variant, prob["T"], prob["C"], prob["M"], prob["N"], ip(prob["t"]), ip(prob["c"]),
                                         ip(prob["m"]), dp(np.ascontiguousarray(params, float)), C.c_double(marker_side), dp(out))
        return out.reshape(-1, 3)

    def solve_points(self, prob, opts=None, max_log=64):
        params = prob["params"].copy()
        s = OracleSummary()
        log = np.zeros((max_log, 8))
        o = opts or self.options()
        self.lib.oracle_solve_points(prob["C"], prob["P"], C.c_int64(prob["N"]), ip(prob["cam_idx"]), ip(prob["pt_idx"]), dp(prob["obs"]),
                                     dp(params), dp(prob["intr"]), C.byref(o), C.byref(s), dp(log), max_log)
        return params, s, log[: s.num_iterations + 1]

    def solve_points_constant(self, prob, constant_cameras, opts=None, max_log=64):
        params = prob["params"].copy()
        s = OracleSummary()
        log = np.zeros((max_log, 8))
        o = opts or self.options()
        mask = np.zeros(prob["C"], np.uint8)
        mask[list(constant_cameras)] = 1
        self.lib.oracle_solve_points_constant(prob["C"], prob["P"], C.c_int64(prob["N"]), ip(prob["cam_idx"]), ip(prob["pt_idx"]), dp(prob["obs"]),
                                              dp(params), dp(prob["intr"]), mask.ctypes.data_as(C.c_void_p), C.byref(o), C.byref(s), dp(log), max_log)
        return params, s, log[: s.num_iterations + 1]

    def solve_points_constant_blocks(self, prob, constant_cameras=(), constant_points=(), opts=None, max_log=64):
        """Problem::SetParameterBlockConstant on camera and / or point blocks."""
        params = prob["params"].copy()
        s = OracleSummary()
        log = np.zeros((max_log, 8))
        o = opts or self.options()
        cm = np.zeros(prob["C"], np.uint8); cm[list(constant_cameras)] = 1
        pm = np.zeros(prob["P"], np.uint8); pm[list(constant_points)] = 1
        self.lib.oracle_solve_points_constant_blocks(prob["C"], prob["P"], C.c_int64(prob["N"]), ip(prob["cam_idx"]), ip(prob["pt_idx"]), dp(prob["obs"]),
                                                     dp(params), dp(prob["intr"]), cm.ctypes.data_as(C.c_void_p), pm.ctypes.data_as(C.c_void_p),
                                                     C.byref(o), C.byref(s), dp(log), max_log)
        return params, s, log[: s.num_iterations + 1]

    def points_cost(self, prob, params, huber_delta=0.0, num_threads=1):
        cost, ss = C.c_double(), C.c_double()
        self.lib.oracle_points_cost(prob["C"], prob["P"], C.c_int64(prob["N"]), ip(prob["cam_idx"]), ip(prob["pt_idx"]), dp(prob["obs"]),
                                    dp(np.ascontiguousarray(params, float)), dp(prob["intr"]), C.c_double(huber_delta), num_threads,
                                    C.byref(cost), C.byref(ss))
        return cost.value, ss.value

    def points_linearize_and_step(self, prob, params, radius, opts=None, scale_in=None, want_S=True):
        Cn, P = prob["C"], prob["P"]
        n, nc = 6 * Cn + 3 * P, 6 * Cn
        S = np.zeros((nc, nc)) if want_S else None
        rhs = np.zeros(nc)
        delta, scale, grad, scal = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(4)
        o = opts or self.options()
        rc = self.lib.oracle_points_linearize_and_step(Cn, P, C.c_int64(prob["N"]), ip(prob["cam_idx"]), ip(prob["pt_idx"]), dp(prob["obs"]),
                                                       dp(np.ascontiguousarray(params, float)), dp(prob["intr"]), C.byref(o), C.c_double(radius),
                                                       dp(scale_in), dp(S), dp(rhs), dp(delta), dp(scale), dp(grad), dp(scal))
        assert rc == 0
        return dict(S=S, rhs=rhs, delta=delta, scale=scale, gradient=grad, cost=scal[0], model_cost_change=scal[1],
                    gradient_max_norm=scal[2], solve_ok=bool(scal[3]))

    def points_dense_step(self, prob, params, radius, opts=None):
        n = 6 * prob["C"] + 3 * prob["P"]
        delta = np.zeros(n)
        o = opts or self.options()
        rc = self.lib.oracle_points_dense_step(prob["C"], prob["P"], C.c_int64(prob["N"]), ip(prob["cam_idx"]), ip(prob["pt_idx"]), dp(prob["obs"]),
                                               dp(np.ascontiguousarray(params, float)), dp(prob["intr"]), C.byref(o), C.c_double(radius), dp(delta))
        assert rc == 0
        return delta


_ORACLE = None
_VARIANT = None


def load_nocontract(build=True):
    """The oracle built with -ffp-contract=off (oracle/Makefile): same algorithm, different roundings."""
    global _VARIANT
    if _VARIANT is None:
        so = os.path.join(ORACLE_DIR, "liboracle_nocontract.so")
        if build:
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle_nocontract.so"])
        _VARIANT = Oracle(so)
    return _VARIANT


_WIDE = None


def load_wide(build=True):
    """THE REFEREE (oracle/Makefile: liboracle_wide.so, -DRSBA_ORACLE_WIDE): the point model's linear solve in long double.  Not the
    reference's arithmetic; used only by oracle_spread.referee."""
    global _WIDE
    if _WIDE is None:
        so = os.path.join(ORACLE_DIR, "liboracle_wide.so")
        if build:
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle_wide.so"])
        _WIDE = Oracle(so)
    return _WIDE


def load(build=True):
    global _ORACLE
    if _ORACLE is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if build:
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
        _ORACLE = Oracle(so)
    return _ORACLE


# ------------------------------------------------------------------ fixture parsers (plain text / OpenCV XML)
def read_correspondence(path):
    """correspondence.txt as BALProblem::loadFile reads it (bundle_adjustment.cpp:132-187)."""
    tok = open(path).read().split()
    it = iter(tok)
    T, Cn, M, N = (int(next(it)) for _ in range(4))
    counts = np.zeros((T, Cn), np.int32)
    for t in range(T):
        next(it)  # leading time id is discarded
        for c in range(Cn):
            counts[t, c] = int(next(it))
    ti, ci, mi = np.zeros(N, np.int32), np.zeros(N, np.int32), np.zeros(N, np.int32)
    obs = np.zeros(8 * N)
    for i in range(N):
        ti[i], ci[i], mi[i] = int(next(it)), int(next(it)), int(next(it))
        for j in range(8):
            obs[8 * i + j] = float(next(it))
    npar = 6 * (Cn + T + M)
    params = np.array([float(next(it)) for _ in range(npar)])
    return dict(T=T, C=Cn, M=M, N=N, counts=counts, t=ti, c=ci, m=mi, obs=obs, params=params)


def read_opencv_xml(path):
    """Minimal OpenCV FileStorage reader: {name: ndarray} for opencv-matrix nodes of dt 'd'."""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"<(\w+) type_id=\"opencv-matrix\">\s*<rows>(\d+)</rows>\s*<cols>(\d+)</cols>\s*<dt>(\w+)</dt>\s*<data>(.*?)</data>", txt, re.S):
        name, r, c = m.group(1), int(m.group(2)), int(m.group(3))
        out[name] = np.array([float(x) for x in m.group(5).split()]).reshape(r, c)
    return out


def read_intrinsics(serials):
    intr = np.zeros((len(serials), 4))
    for k, sn in enumerate(serials):
        K = read_opencv_xml(os.path.join(GOLDEN, "intrinsics", sn + ".xml"))["intrinsics"]
        intr[k] = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]  # bundle_adjustment.h:66-69
    return intr


def read_point3d(path):
    tok = open(path).read().split()
    n, T, Cn = int(tok[0]), int(tok[1]), int(tok[2])
    counts = np.array(tok[3:3 + T * (Cn + 1)], float).reshape(T, Cn + 1)[:, 1:].astype(int)
    pts = np.array(tok[3 + T * (Cn + 1):], float).reshape(n, 3)
    return n, counts, pts


def read_two_cam_data(path):
    """Test1 file: 'C P', P rows 'cam pt u v', then 6C + 3P parameters (bundle_adjustmenter.cpp:55-85)."""
    tok = open(path).read().split()
    Cn, P = int(tok[0]), int(tok[1])
    N = P
    rows = np.array(tok[2:2 + 4 * N], float).reshape(N, 4)
    params = np.array(tok[2 + 4 * N:2 + 4 * N + 6 * Cn + 3 * P], float)
    return dict(C=Cn, P=P, N=N, cam_idx=rows[:, 0].astype(np.int32).copy(), pt_idx=rows[:, 1].astype(np.int32).copy(),
                obs=rows[:, 2:4].reshape(-1).copy(), params=params)
