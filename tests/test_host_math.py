"""The product's analytic residual/Jacobian arithmetic (csrc/ba_math.hpp, host build) against the
oracle's AutoDiff-equivalent dual numbers.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm():
    so = os.path.join(HERE, "harness", "libmath_harness.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "harness", "math_harness.cpp")])
    lib = C.CDLL(so)
    lib.h_loss.restype = C.c_double
    lib.h_loss.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
    lib.h_point_block_inverse.argtypes = [C.POINTER(C.c_double)] * 2 + [C.c_double] * 3 + [C.POINTER(C.c_double)]
    return lib


def _rj(hm, cam, intr, X, uv):
    r, jc, jp = np.zeros(2), np.zeros(12), np.zeros(6)
    hm.h_residual_jacobian(ol.dp(cam), ol.dp(intr), ol.dp(X), ol.dp(uv), ol.dp(r), ol.dp(jc), ol.dp(jp))
    return r, jc.reshape(2, 6), jp.reshape(2, 3)


def test_matches_autodiff_on_random_cameras(hm, oracle):
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(500):
        ang = rng.uniform(1e-3, 3.1)
        ax = rng.normal(size=3)
        cam = np.concatenate([ax / np.linalg.norm(ax) * ang, rng.normal(0, 1, 3) + [0, 0, 3.0]])
        X = rng.uniform(-0.5, 0.5, 3)
        intr = np.array([rng.uniform(380, 640), rng.uniform(380, 640), rng.uniform(300, 330), rng.uniform(230, 245)])
        uv = rng.uniform(0, 640, 2)
        r0, jc0, jp0 = oracle.point_residual_jacobian(cam, X, intr, uv)
        r1, jc1, jp1 = _rj(hm, cam, intr, X, uv)
        scale = max(1.0, np.abs(jc0).max(), np.abs(jp0).max())
        worst = max(worst, np.abs(r0 - r1).max() / max(1.0, np.abs(r0).max()), np.abs(jc0 - jc1).max() / scale, np.abs(jp0 - jp1).max() / scale)
    assert worst < 1e-12, worst


@pytest.mark.parametrize("theta", [0.0, 1e-12, 1e-9, 1.2e-8, 2e-8, 1e-7, 1e-5, 1e-3])
def test_small_rotations_both_branches(hm, oracle, theta):
    """Across the theta^2 = DBL_EPSILON switch the analytic blocks must follow AutoDiff's branch."""
    ax = np.array([0.3, -0.5, 0.8])
    ax /= np.linalg.norm(ax)
    cam = np.concatenate([ax * theta, [0.05, -0.02, 2.5]])
    X = np.array([0.2, 0.1, -0.3])
    intr = np.array([630.0, 625.0, 316.0, 240.0])
    uv = np.array([310.0, 250.0])
    r0, jc0, jp0 = oracle.point_residual_jacobian(cam, X, intr, uv)
    r1, jc1, jp1 = _rj(hm, cam, intr, X, uv)
    assert np.abs(r0 - r1).max() < 1e-10
    assert np.abs(jc0 - jc1).max() < 1e-9 * np.abs(jc0).max()
    assert np.abs(jp0 - jp1).max() < 1e-12 * np.abs(jp0).max()


def test_point_block_inverse(hm):
    rng = np.random.default_rng(1)
    for _ in range(100):
        A = rng.normal(size=(5, 3))
        V = A.T @ A
        s = 1.0 / (1.0 + np.sqrt(np.diag(V)))
        radius = 10 ** rng.uniform(-2, 6)
        V6 = np.array([V[0, 0], V[0, 1], V[0, 2], V[1, 1], V[1, 2], V[2, 2]])
        out = np.zeros(6)
        assert hm.h_point_block_inverse(ol.dp(V6), ol.dp(s), 1e-6, 1e32, radius, ol.dp(out)) == 1
        Vs = np.diag(s) @ V @ np.diag(s)
        M = Vs + np.diag(np.clip(np.diag(Vs), 1e-6, 1e32) / radius)
        ref = np.diag(s) @ np.linalg.inv(M) @ np.diag(s)
        got = np.array([[out[0], out[1], out[2]], [out[1], out[3], out[4]], [out[2], out[4], out[5]]])
        assert np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()
    bad = np.array([1.0, 2.0, 0.0, 1.0, 0.0, -1.0])
    assert hm.h_point_block_inverse(ol.dp(bad), ol.dp(np.ones(3)), 0.0, 1e32, 1e4, ol.dp(out)) == 0


def test_huber_matches_ceres_definition(hm):
    sq = C.c_double()
    assert hm.h_loss(1.0, 0.25, C.byref(sq)) == 0.25 and sq.value == 1.0
    rho = hm.h_loss(1.0, 9.0, C.byref(sq))
    assert abs(rho - (2 * 3.0 - 1.0)) < 1e-15 and abs(sq.value - np.sqrt(1.0 / 3.0)) < 1e-15
    assert hm.h_loss(0.0, 9.0, C.byref(sq)) == 9.0 and sq.value == 1.0


# ------------------------------------------------------------------ marker-chain functors, analytic (ba_math.hpp MarkerCornerResidualJacobian)
def _marker_rj(hm, cam, tim, mar, side, intr, obs8):
    r, j = np.zeros(8), np.zeros(8 * 18)
    hm.h_marker_residual_jacobian.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    cam = None if cam is None else np.ascontiguousarray(cam, float)
    mar = None if mar is None else np.ascontiguousarray(mar, float)
    tim, intr, obs8 = (np.ascontiguousarray(x, float) for x in (tim, intr, obs8))
    hm.h_marker_residual_jacobian(vp(cam), vp(tim), vp(mar), C.c_double(side / 2), vp(intr), vp(obs8), vp(r), vp(j))
    return r, j.reshape(8, 18)


@pytest.mark.parametrize("with_cam,with_marker", [(True, True), (True, False), (False, True), (False, False)])
def test_marker_chain_analytic_matches_autodiff(hm, oracle, with_cam, with_marker):
    """The four functors of Main_Calibration/bundle_adjustment.h:56-343 (camera and / or marker transform absent): residuals
    and the 8 x 18 Jacobian of the analytic routine against the oracle's dual numbers, random poses."""
    rng = np.random.default_rng(5 + 2 * with_cam + with_marker)
    worst = 0.0
    for _ in range(300):
        def pose(z):
            ang = rng.uniform(1e-3, 3.0)
            ax = rng.normal(size=3)
            return np.concatenate([ax / np.linalg.norm(ax) * ang, rng.normal(0, 0.2, 2), [z]])
        cam = pose(0.3) if with_cam else None
        tim, mar = pose(1.5), (pose(0.1) if with_marker else None)
        intr = np.array([rng.uniform(380, 640), rng.uniform(380, 640), rng.uniform(300, 330), rng.uniform(230, 245)])
        obs8 = rng.uniform(0, 640, 8)
        r0, j0 = oracle.marker_residual_jacobian(cam, tim, mar, 0.0148, intr, obs8)
        if not np.all(np.isfinite(j0)) or np.abs(j0).max() > 1e7:
            continue   # a corner (almost) in the camera plane
        r1, j1 = _marker_rj(hm, cam, tim, mar, 0.0148, intr, obs8)
        worst = max(worst, np.abs(r0 - r1).max() / max(1.0, np.abs(r0).max()), np.abs(j0 - j1).max() / max(1.0, np.abs(j0).max()))
    assert worst < 1e-11, worst


@pytest.mark.parametrize("theta", [0.0, 1e-12, 1e-9, 1.2e-8, 2e-8, 1e-7, 1e-4])
def test_marker_chain_analytic_small_rotations(hm, oracle, theta):
    """Each of the three rotations across AngleAxisRotatePoint's theta^2 = DBL_EPSILON switch (the test2 fixture has all
    marker rvecs exactly zero): the analytic blocks follow the branch AutoDiff differentiates."""
    ax = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    base = [np.array([0.4, -0.2, 0.3, 0.05, -0.02, 0.4]), np.array([-0.3, 0.5, 0.2, 0.1, 0.05, 1.2]), np.array([0.2, 0.1, -0.4, 0.02, 0.03, 0.1])]
    intr, obs8 = np.array([630.0, 625.0, 316.0, 240.0]), np.linspace(200, 400, 8)
    for which in range(3):
        poses = [b.copy() for b in base]
        poses[which][:3] = ax * theta
        r0, j0 = oracle.marker_residual_jacobian(poses[0], poses[1], poses[2], 0.048, intr, obs8)
        r1, j1 = _marker_rj(hm, poses[0], poses[1], poses[2], 0.048, intr, obs8)
        assert np.abs(r0 - r1).max() < 1e-9
        assert np.abs(j0 - j1).max() < 1e-8 * np.abs(j0).max(), which


def test_cube_is_the_correctly_rounded_power(hm):
    """rsba::Cube — the (2 rho - 1)^3 of the radius update, the same exact sequence on host (MinimizeLoop) and device
    (DecideStep) — against t^3 computed exactly (fractions) and rounded once: the same bits for every value.  Ceres and the
    oracle call pow(t, 3) (trust_region_minimizer.cc; oracle/ba_oracle.hpp TrustRegionMinimize): glibc's pow is that same
    correctly rounded value in all but ~6 of 10 000 cases and one ulp off in those, so Cube is pow to within pow's own error;
    (t * t) * t, which round 2 used, was an ulp off in one case of four."""
    from fractions import Fraction
    hm.h_cube.restype = C.c_double
    hm.h_cube.argtypes = [C.c_double]
    libm = C.CDLL("libm.so.6")
    libm.pow.restype = C.c_double
    libm.pow.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(0)
    # the range rho takes on accepted steps, the neighbourhood of rho = 1 where the update is clamped, the tail towards t = 1
    rho = np.concatenate([rng.uniform(1e-3, 1.0, 30000), rng.uniform(0.9, 1.1, 10000), 1.0 - np.logspace(-16, -1, 500)])
    t = 2.0 * rho - 1.0
    exact = np.array([float(Fraction(float(x)) ** 3) for x in t])
    cube = np.array([hm.h_cube(float(x)) for x in t])
    glibc = np.array([libm.pow(float(x), 3.0) for x in t])
    assert np.array_equal(cube, exact)
    off = glibc != exact
    assert off.mean() < 5e-3 and np.all(np.abs(glibc - exact)[off] <= np.spacing(np.abs(exact[off])))
    assert (((t * t) * t) != exact).mean() > 0.1
