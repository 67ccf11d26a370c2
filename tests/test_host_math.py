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
