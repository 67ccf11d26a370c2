#!/usr/bin/env python3
"""A second, independent opinion on the POINT model (BASELINE configs 2-5): numpy only, complex-step Jacobians, dense normal
equations — no Schur complement, no dual numbers, no line of oracle/ or of the product.

Why: the reference commits no output for its only point-model input (Common/Correspondence/two_cam_data.txt;
Test1_BundleAdjustment/main.cpp:56-87 writes none), so oracle/ba_oracle.hpp's point model and losses were pinned only
transitively (same LM driver as the marker-chain model that IS pinned by the reference's committed XML).  This script is
SURVEY.md Appendix B's replay — the one that reproduced hongo/Camera_Transform.xml to 7e-16 — extended to the point functor:

  residual     Test1_BundleAdjustment/bundle_adjustmenter.cpp:122-141: p = AngleAxisRotatePoint(cam[0:3], X) + cam[3:6],
               r = (fx p0/p2 + ppx - u, fy p1/p2 + ppy - v); AngleAxisRotatePoint as ceres/rotation.h (both branches, the test on
               the REAL part of theta^2 so that the complex step differentiates the branch that is taken)
  Jacobian     complex step, h = 1e-30: J[:, k] = Im r(x + i h e_k) / h  (exact to rounding for an analytic r)
  loss         HuberLoss(a) / CauchyLoss(b) with Ceres' corrector for rho'' <= 0: r, J scaled by sqrt(rho') per 2-residual block
  minimiser    SURVEY.md Appendix A.2 verbatim: Jacobi scaling fixed at iteration 0, D^2 = clamp(diag J'J) / radius, dense
               numpy.linalg.solve on J'J + D^2, model cost change, the three tolerances in Ceres' order, rho > 1e-3,
               radius / max(1/3, 1 - (2 rho - 1)^3), radius / 2, 4, 8 ...

Output: tests/golden/point_model_<case>.json — the inputs (so that the fixture is self-contained), every iterate's cost /
gradient / step norm / radius / accepted flag, the termination, and the final parameters.  tests/test_point_model_replay.py
holds the oracle (CPU) and the HIP path (-m gpu) to them.  Regenerate: `python tools/replay_point_model.py` (seconds).
"""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EPS = np.finfo(float).eps


def rotate(w, X):
    """ceres::AngleAxisRotatePoint on rows of complex w (n,3), X (n,3)."""
    th2 = np.sum(w * w, axis=1)
    big = th2.real > EPS
    th = np.sqrt(np.where(big, th2, 1.0))
    c, s = np.cos(th), np.sin(th)
    k = w / th[:, None]
    kxX = np.cross(k, X)
    kdX = np.sum(k * X, axis=1)
    rod = X * c[:, None] + kxX * s[:, None] + k * (kdX * (1.0 - c))[:, None]
    small = X + np.cross(w, X)
    return np.where(big[:, None], rod, small)


def residuals(x, prob):
    C, P = prob["C"], prob["P"]
    cams = x[:6 * C].reshape(C, 6)
    pts = x[6 * C:].reshape(P, 3)
    ci, pi = prob["cam_idx"], prob["pt_idx"]
    K = prob["intr"].reshape(C, 4)
    p = rotate(cams[ci, :3], pts[pi]) + cams[ci, 3:]
    u = K[ci, 0] * p[:, 0] / p[:, 2] + K[ci, 2] - prob["obs"][0::2]
    v = K[ci, 1] * p[:, 1] / p[:, 2] + K[ci, 3] - prob["obs"][1::2]
    return np.stack([u, v], axis=1).reshape(-1)


def jacobian(x, prob):
    """Complex step, column by column — but only the rows a parameter block touches are non-zero, so per camera / per point
    columns are evaluated on the whole residual vector at once for all blocks of the same kind (their rows are disjoint... not for
    cameras and points together: 6 + 3 evaluations with every block of a kind perturbed at the same coordinate)."""
    C, P, N = prob["C"], prob["P"], prob["N"]
    h = 1e-30
    J = np.zeros((2 * N, 6 * C + 3 * P))
    rows_c = np.repeat(prob["cam_idx"], 2)
    rows_p = np.repeat(prob["pt_idx"], 2)
    r_idx = np.arange(2 * N)
    for d in range(6):
        xx = x.astype(complex)
        xx[d:6 * C:6] += 1j * h
        J[r_idx, 6 * rows_c + d] = residuals(xx, prob).imag / h
    for d in range(3):
        xx = x.astype(complex)
        xx[6 * C + d::3] += 1j * h
        J[r_idx, 6 * C + 3 * rows_p + d] = residuals(xx, prob).imag / h
    return J


def loss(prob, s):
    """rho(s), rho'(s) per residual block (s = squared norm); Ceres loss_function.cc."""
    a = prob.get("loss_scale", 0.0)
    kind = prob.get("loss", "none")
    if kind == "none" or a <= 0:
        return s, np.ones_like(s)
    if kind == "huber":
        b = a * a
        out = s > b
        r = np.sqrt(np.where(out, s, 1.0))
        return np.where(out, 2 * a * r - b, s), np.where(out, np.maximum(np.finfo(float).tiny, a / r), 1.0)
    if kind == "cauchy":
        b = a * a
        t = 1.0 + s / b
        return b * np.log(t), np.maximum(np.finfo(float).tiny, 1.0 / t)
    raise ValueError(kind)


def evaluate(x, prob, with_jacobian):
    r = residuals(x.astype(complex), prob).real
    s = r[0::2] ** 2 + r[1::2] ** 2
    rho, rho1 = loss(prob, s)
    cost = 0.5 * np.sum(rho)
    if not with_jacobian:
        return cost, None, None, float(np.sum(s))
    sq = np.repeat(np.sqrt(rho1), 2)   # corrector with rho'' <= 0 (Huber, Cauchy): alpha = 0, both scaled by sqrt(rho')
    J = jacobian(x, prob) * sq[:, None]
    return cost, r * sq, J, float(np.sum(s))


def minimise(prob, max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8,
             initial_radius=1e4, max_radius=1e16, min_radius=1e-32, min_relative_decrease=1e-3, min_lm_diagonal=1e-6,
             max_lm_diagonal=1e32, max_invalid=5):
    x = prob["params"].astype(float).copy()
    rows = []
    cost, r, J, sumsq = evaluate(x, prob, True)
    g = J.T @ r
    scale = 1.0 / (1.0 + np.sqrt(np.sum(J * J, axis=0)))
    Js = J * scale
    rows.append(dict(iteration=0, cost=cost, cost_change=0.0, gradient_max_norm=float(np.abs(g).max()), step_norm=0.0, relative_decrease=0.0,
                     trust_region_radius=initial_radius, valid=0, successful=0))
    out = dict(initial_cost=cost)
    if np.abs(g).max() <= gradient_tolerance:
        return x, dict(out, termination="CONVERGENCE", reason="gradient", final_cost=cost, final_sumsq=sumsq), rows
    radius, dec, invalid = initial_radius, 2.0, 0
    it = 0
    while True:
        if it >= max_num_iterations:
            return x, dict(out, termination="NO_CONVERGENCE", reason="max_iterations", final_cost=cost, final_sumsq=sumsq), rows
        if np.abs(g).max() <= gradient_tolerance:
            return x, dict(out, termination="CONVERGENCE", reason="gradient", final_cost=cost, final_sumsq=sumsq), rows
        if radius < min_radius:
            return x, dict(out, termination="CONVERGENCE", reason="min_radius", final_cost=cost, final_sumsq=sumsq), rows
        it += 1
        row = dict(iteration=it, cost=cost, cost_change=0.0, gradient_max_norm=float(np.abs(g).max()), step_norm=0.0, relative_decrease=0.0,
                   trust_region_radius=radius, valid=0, successful=0)
        H = Js.T @ Js
        D2 = np.clip(np.diag(H), min_lm_diagonal, max_lm_diagonal) / radius
        ok = True
        try:
            L = np.linalg.cholesky(H + np.diag(D2))
            y = np.linalg.solve(L.T, np.linalg.solve(L, Js.T @ r))
        except np.linalg.LinAlgError:
            ok = False
        if ok:
            step = -y
            Jd = Js @ step
            mcc = -float(Jd @ (r + 0.5 * Jd))
            ok = np.all(np.isfinite(step)) and mcc > 0.0
        if not ok:
            invalid += 1
            radius /= dec
            dec *= 2.0
            row["trust_region_radius"] = radius
            rows.append(row)
            if invalid >= max_invalid:
                return x, dict(out, termination="FAILURE", reason="invalid_steps", final_cost=cost, final_sumsq=sumsq), rows
            continue
        invalid = 0
        row["valid"] = 1
        delta = step * scale
        xc = x + delta
        cand, _, _, cand_sumsq = evaluate(xc, prob, False)
        row["step_norm"] = float(np.linalg.norm(delta))
        if row["step_norm"] <= parameter_tolerance * (np.linalg.norm(x) + parameter_tolerance):
            rows.append(row)
            return x, dict(out, termination="CONVERGENCE", reason="parameter", final_cost=cost, final_sumsq=sumsq), rows
        row["cost_change"] = cost - cand
        if abs(cost - cand) <= function_tolerance * cost:
            rows.append(row)
            return x, dict(out, termination="CONVERGENCE", reason="function", final_cost=cost, final_sumsq=sumsq), rows
        rho = (cost - cand) / mcc
        row["relative_decrease"] = rho
        if np.isfinite(cand) and rho > min_relative_decrease:
            x = xc
            cost, r, J, sumsq = evaluate(x, prob, True)
            g = J.T @ r
            Js = J * scale
            radius = min(max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            dec = 2.0
            row.update(successful=1, cost=cost, gradient_max_norm=float(np.abs(g).max()))
        else:
            radius /= dec
            dec *= 2.0
        row["trust_region_radius"] = radius
        rows.append(row)


# ------------------------------------------------------------------ cases
def read_intrinsics(serial):
    txt = open(os.path.join(GOLDEN, "intrinsics", "%s.xml" % serial)).read()
    m = re.search(r"<intrinsics[^>]*>.*?<data>(.*?)</data>", txt, re.S)
    K = np.array(m.group(1).split(), float).reshape(3, 3)
    return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])


def case_two_cam():
    """The reference's own file, as Test1_BundleAdjustment reads it (bundle_adjustmenter.cpp:55-85: header 'C P', one observation
    per point), every block with serial_numbers[1]'s intrinsics (main.cpp:73-74)."""
    tok = open(os.path.join(GOLDEN, "two_cam_data.txt")).read().split()
    C, P = int(tok[0]), int(tok[1])
    rows = np.array(tok[2:2 + 4 * P], float).reshape(P, 4)
    params = np.array(tok[2 + 4 * P:2 + 4 * P + 6 * C + 3 * P], float)
    K = read_intrinsics("825312072048")
    return dict(name="two_cam", C=C, P=P, N=P, cam_idx=rows[:, 0].astype(int), pt_idx=rows[:, 1].astype(int), obs=rows[:, 2:4].reshape(-1).copy(),
                intr=np.tile(K, C), params=params)


def synthetic(name, C, P, k, seed, outlier_frac=0.0, loss_kind="none", loss_scale=0.0):
    """Cameras on a ring of radius 3 m looking at the origin, points in the unit cube, k views per point, 0.5 px noise — SURVEY 8(d)'s
    recipe at a size a dense Jacobian holds, from this script's own generator (numpy default_rng; nothing of synthetic.py)."""
    rng = np.random.default_rng([seed, 0x5EED])
    ang = 2 * np.pi * (np.arange(C) + 0.3 * rng.random(C)) / C
    centre = 3.0 * np.stack([np.cos(ang), np.sin(ang), 0.4 * rng.standard_normal(C)], 1)
    fwd = -centre / np.linalg.norm(centre, axis=1, keepdims=True)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 1)
    t = -np.einsum("cij,cj->ci", R, centre)
    # rotation matrix -> angle-axis
    th = np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1))
    ax = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1) / (2 * np.sin(th))[:, None]
    cams = np.hstack([ax * th[:, None], t])
    f = rng.uniform(620, 640, C)
    intr = np.stack([f, f, rng.uniform(305, 325, C), rng.uniform(230, 245, C)], 1)
    X = rng.uniform(-0.5, 0.5, (P, 3))
    vis = np.sort(np.argsort(rng.random((P, C)), axis=1)[:, :k], axis=1)
    ci, pi = vis.reshape(-1), np.repeat(np.arange(P), k)
    prob = dict(name=name, C=C, P=P, N=P * k, cam_idx=ci, pt_idx=pi, intr=intr.reshape(-1), obs=np.zeros(2 * P * k),
                params=np.concatenate([cams.reshape(-1), X.reshape(-1)]))
    uv = residuals(prob["params"].astype(complex), prob).real   # obs = 0: the projections
    uv += rng.normal(0, 0.5, uv.shape)
    out = np.repeat(rng.random(P * k) < outlier_frac, 2)
    uv += np.where(out, rng.uniform(-50, 50, uv.shape), 0.0)
    prob["obs"] = uv
    start = prob["params"].copy()
    start[:6 * C] += np.hstack([rng.normal(0, 0.01, (C, 3)), rng.normal(0, 0.01, (C, 3))]).reshape(-1)
    start[6 * C:] += rng.normal(0, 0.02, 3 * P)
    prob["params"] = start
    prob["loss"], prob["loss_scale"] = loss_kind, loss_scale
    return prob


CASES = [
    lambda: case_two_cam(),
    lambda: synthetic("cfg2_small", 8, 60, 8, 2),                                        # BASELINE config 2's shape: every camera sees every point
    lambda: synthetic("sparse_views", 12, 90, 5, 3),                                     # config 3's shape: a subset of the cameras per point
    lambda: synthetic("huber_outliers", 12, 90, 6, 5, 0.05, "huber", 1.0),               # config 5's: 5 % outliers, Huber delta = 1 px
    lambda: synthetic("cauchy_outliers", 10, 70, 6, 6, 0.05, "cauchy", 2.0),
]


def main():
    for make in CASES:
        prob = make()
        x, summary, rows = minimise(prob)
        fx = dict(name=prob["name"], C=int(prob["C"]), P=int(prob["P"]), N=int(prob["N"]), loss=prob.get("loss", "none"), loss_scale=float(prob.get("loss_scale", 0.0)),
                  cam_idx=[int(v) for v in prob["cam_idx"]], pt_idx=[int(v) for v in prob["pt_idx"]], obs=[float(v) for v in prob["obs"]],
                  intr=[float(v) for v in prob["intr"]], params=[float(v) for v in prob["params"]],
                  expected=dict(summary=summary, iterations=rows, final_params=[float(v) for v in x]),
                  generator="tools/replay_point_model.py (numpy %s): complex-step Jacobians, dense normal equations, SURVEY.md Appendix A.2" % np.__version__)
        path = os.path.join(GOLDEN, "point_model_%s.json" % prob["name"])
        json.dump(fx, open(path, "w"))
        print("%-16s C=%d P=%d N=%d: %d iterations, %s (%s), cost %.9e -> %.9e  -> %s" % (
            prob["name"], prob["C"], prob["P"], prob["N"], len(rows) - 1, summary["termination"], summary["reason"], summary["initial_cost"],
            summary["final_cost"], os.path.relpath(path, ROOT)))


if __name__ == "__main__":
    sys.exit(main())
