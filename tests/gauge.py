"""Comparison of two bundle-adjustment solutions modulo the gauge of the point model.

With every camera and every point free (Test1_BundleAdjustment/main.cpp:76-79 adds no constant block) the cost is
invariant under a similarity of the world, X -> s Q X + b, cameras (R, t) -> (R Q', s t - R Q' b): seven directions in
which the damped system's eigenvalues are only the LM diagonal.  On a long run with a large trust-region radius two
correct solvers drift apart ALONG that orbit (rounding differences are amplified by ~radius there) while cost,
accept/reject decisions and reprojection RMS agree to the last digits.  `align` removes the orbit: it fits the
similarity on the points (Umeyama) and applies it to cameras and points.
"""
import numpy as np


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-300:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _angle_axis(R):
    th = np.arccos(np.clip((np.trace(R) - 1) / 2, -1.0, 1.0))
    if th < 1e-12:
        return np.zeros(3)
    return th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def apply(params, C, s, Q, b):
    """The similarity X -> s Q X + b applied to a parameter vector (C cameras of rvec3 + tvec3, then the points)."""
    out = np.array(params, dtype=np.float64, copy=True)
    pts = out[6 * C:].reshape(-1, 3)
    out[6 * C:] = (s * (Q @ pts.T).T + b).reshape(-1)
    for i in range(C):
        Rn = _rodrigues(params[6 * i:6 * i + 3]) @ Q.T
        out[6 * i:6 * i + 3] = _angle_axis(Rn)
        out[6 * i + 3:6 * i + 6] = s * params[6 * i + 3:6 * i + 6] - Rn @ b
    return out


def align(got, ref, C):
    """`got` moved along the gauge orbit onto `ref` (least squares over the points); returns (aligned, s, Q, b)."""
    gp, rp = got[6 * C:].reshape(-1, 3), ref[6 * C:].reshape(-1, 3)
    mg, mr = gp.mean(0), rp.mean(0)
    G, R = gp - mg, rp - mr
    U, S, Vt = np.linalg.svd(R.T @ G)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    Q = U @ D @ Vt
    s = np.trace(np.diag(S) @ D) / (G ** 2).sum()
    b = mr - s * Q @ mg
    return apply(got, C, s, Q, b), s, Q, b
