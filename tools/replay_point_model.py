#!/usr/bin/env python3
"""A second, independent opinion on the POINT model (BASELINE configs 2-5): numpy only, complex-step Jacobians, dense normal
equations — no Schur complement, no dual numbers, no line of oracle/ or of the product.

Why: the reference commits no output for its only point-model input (Common/Correspondence/two_cam_data.txt;
Test1_BundleAdjustment/main.cpp:56-87 writes none), so oracle/ba_oracle.hpp's point model and losses were pinned only
transitively (same LM driver as the marker-chain model that IS pinned by the reference's committed XML).  This script is
SURVEY.md Appendix B's replay — the one that reproduces hongo/Camera_Transform.xml to 7e-16: mc_solve below, checked by
tests/test_point_model_replay.py::test_replay_reproduces_the_references_xml since round 6 — extended to the point functor:

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
holds the oracle (CPU) and the HIP path (-m gpu) to them.  Regenerate: `python tools/replay_point_model.py [case names]` (a minute: the
two 40- / 72-camera cases of round 6 hold dense 4800 x 1140 Jacobians).
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
    if prob.get("model") == "marker_chain":
        return mc_evaluate(x, prob, with_jacobian)
    r = residuals(x.astype(complex), prob).real
    s = r[0::2] ** 2 + r[1::2] ** 2
    rho, rho1 = loss(prob, s)
    cost = 0.5 * np.sum(rho)
    if not with_jacobian:
        return cost, None, None, float(np.sum(s))
    sq = np.repeat(np.sqrt(rho1), 2)   # corrector with rho'' <= 0 (Huber, Cauchy): alpha = 0, both scaled by sqrt(rho')
    J = jacobian(x, prob) * sq[:, None]
    if prob.get("const_params") is not None:
        J[:, prob["const_params"]] = 0.0   # Problem::SetParameterBlockConstant: the block has no columns (Ceres removes it from the program)
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
        xn = np.linalg.norm(x if prob.get("const_params") is None else np.delete(x, prob["const_params"]))   # (|x| over the blocks in the program)
        if row["step_norm"] <= parameter_tolerance * (xn + parameter_tolerance):
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


# ------------------------------------------------------------------ the marker-chain model (round 6)
# Main_Calibration/bundle_adjustment.h:56-343 — the four functors are ONE chain with blocks left out:
#   corner (-h,+h,0) (+h,+h,0) (+h,-h,0) (-h,-h,0)                          :77-89
#   p = R(marker) corner + t_marker      skipped by the *BaseMarker functors    :97-100   (marker_idx == 0; Test2's variant always applies it)
#   p = R(time) p + t_time                                                       :103-106
#   p = R(camera) p + t_camera           skipped by the BaseCamera* functors     :109-112  (camera_idx == 0)
#   r = (fx p0 / p2 + ppx - u, fy p1 / p2 + ppy - v) per corner                  :114-121
# wired per observation by camera_idx == 0 / marker_idx == 0 (bundle_adjustment_manager.cpp:26-87); the blocks no residual names
# (camera 0, and marker 0 in Main's wiring) are not in the problem.  Parameters [C cameras | T times | M markers] x 6 as
# BALProblem lays them out (bundle_adjustment.cpp:64-87).  This is SURVEY.md Appendix B's replay, committed: the sentence "the same
# replay reproduces the reference's XML" is now a test (tests/test_point_model_replay.py::test_replay_reproduces_the_references_xml).
def mc_parse(path):
    """correspondence.txt as BALProblem::loadFile reads it (bundle_adjustment.cpp:132-187): every token whitespace-separated, the
    count rows' leading time id discarded."""
    tok = open(path).read().split()
    T, C, M, N = (int(v) for v in tok[:4])
    q = 4 + T * (1 + C)
    rows = np.array(tok[q:q + 11 * N], float).reshape(N, 11)
    q += 11 * N
    params = np.array(tok[q:q + 6 * (C + T + M)], float)
    return dict(T=T, C=C, M=M, N=N, t=rows[:, 0].astype(int), c=rows[:, 1].astype(int), m=rows[:, 2].astype(int), obs=rows[:, 3:].copy(), params=params)


def mc_problem(path, serials, marker_side, marker0_is_free, constant_blocks=()):
    """marker0_is_free: Test2_BundleAdjustment's variant (bundle_adjustmenter.cpp:217-366) — two functors only, the marker transform
    always applied, marker 0 a block like any other."""
    mc = mc_parse(path)
    C, T, M = mc["C"], mc["T"], mc["M"]
    used = np.zeros(C + T + M, bool)
    used[mc["c"][mc["c"] != 0]] = True
    used[C + mc["t"]] = True
    mm = mc["m"] if marker0_is_free else mc["m"][mc["m"] != 0]
    used[C + T + mm] = True
    used[list(constant_blocks)] = False   # Problem::SetParameterBlockConstant: applied in the residuals, not in the program
    free = np.flatnonzero(np.repeat(used, 6))
    intr = np.stack([read_intrinsics(sn) for sn in serials])
    return dict(mc, model="marker_chain", intr=intr, h=marker_side / 2.0, marker0_is_free=marker0_is_free, free=free, full=mc["params"].copy(),
                params=mc["params"][free].copy())


def mc_residuals(xfree, prob):
    C, T, N, h = prob["C"], prob["T"], prob["N"], prob["h"]
    full = prob["full"].astype(complex)
    full[prob["free"]] = xfree
    blk = full.reshape(-1, 6)
    corners = np.array([[-h, h, 0.0], [h, h, 0.0], [h, -h, 0.0], [-h, -h, 0.0]], complex)
    p = np.tile(corners, (N, 1))                                   # (4 N, 3), observation-major, corner-major
    ci, ti, mi = (np.repeat(prob[k], 4) for k in ("c", "t", "m"))
    mar, tim, cam = blk[C + T + mi], blk[C + ti], blk[ci]
    with_marker = np.ones(4 * N, bool) if prob["marker0_is_free"] else mi != 0
    p = np.where(with_marker[:, None], rotate(mar[:, :3], p) + mar[:, 3:], p)
    p = rotate(tim[:, :3], p) + tim[:, 3:]
    p = np.where((ci != 0)[:, None], rotate(cam[:, :3], p) + cam[:, 3:], p)
    K = prob["intr"][ci]
    obs = prob["obs"].reshape(-1, 2)
    u = K[:, 0] * p[:, 0] / p[:, 2] + K[:, 2] - obs[:, 0]
    v = K[:, 1] * p[:, 1] / p[:, 2] + K[:, 3] - obs[:, 1]
    return np.stack([u, v], axis=1).reshape(-1)


def mc_evaluate(x, prob, with_jacobian):
    r = mc_residuals(x.astype(complex), prob).real
    cost = 0.5 * float(r @ r)
    if not with_jacobian:
        return cost, None, None, float(r @ r)
    hstep = 1e-30
    J = np.zeros((r.size, x.size))
    for k in range(x.size):
        xx = x.astype(complex)
        xx[k] += 1j * hstep
        J[:, k] = mc_residuals(xx, prob).imag / hstep
    return cost, r, J, float(r @ r)


def rodrigues(w):
    """cv::Rodrigues, rvec -> R (bundle_adjustment_manager.cpp:118-121)."""
    th = np.linalg.norm(w)
    if th < EPS:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def mc_solve(which, constant_blocks=()):
    """'hongo' (Main_Calibration: my_const.h:9,15) or 'test2' (Test2_BundleAdjustment; marker side inferred from test2/point3d.txt's corner
    spacing) -> (all C + T + M blocks after the solve, summary, iteration rows, the problem)."""
    if which == "hongo":
        prob = mc_problem(os.path.join(GOLDEN, "hongo", "correspondence.txt"), ["821312061029", "816612062327", "821212062536", "821212061326"], 0.0148, False, constant_blocks)
    else:
        prob = mc_problem(os.path.join(GOLDEN, "test2", "correspondence_test.txt"), ["819612072493", "825312072048"], 0.048, True, constant_blocks)
    x, summary, rows = minimise(prob)
    full = prob["full"].copy()
    full[prob["free"]] = x
    return full.reshape(-1, 6), summary, rows, prob


def read_xml_matrices(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"<(\w+) type_id=\"opencv-matrix\">\s*<rows>(\d+)</rows>\s*<cols>(\d+)</cols>\s*<dt>\w+</dt>\s*<data>(.*?)</data>", txt, re.S):
        out[m.group(1)] = np.array(m.group(4).split(), float).reshape(int(m.group(2)), int(m.group(3)))
    return out


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


def with_constant_blocks(prob, cameras, points):
    C = prob["C"]
    idx = [6 * c + k for c in cameras for k in range(6)] + [6 * C + 3 * j + k for j in points for k in range(3)]
    prob["const_params"] = np.array(sorted(idx), int)
    prob["constant_cameras"], prob["constant_points"] = list(cameras), list(points)
    return prob


CASES = [
    lambda: case_two_cam(),
    lambda: synthetic("cfg2_small", 8, 60, 8, 2),                                        # BASELINE config 2's shape: every camera sees every point
    lambda: synthetic("sparse_views", 12, 90, 5, 3),                                     # config 3's shape: a subset of the cameras per point
    lambda: synthetic("huber_outliers", 12, 90, 6, 5, 0.05, "huber", 1.0),               # config 5's: 5 % outliers, Huber delta = 1 px
    lambda: synthetic("cauchy_outliers", 10, 70, 6, 6, 0.05, "cauchy", 2.0),
    # round 6: the code paths the benchmark runs, not only the one-workgroup ones — 33 .. 64 cameras: the pipelined schedule, three camera
    # groups, the last one as the factorisation's border; more than 64: sparse pair segments and the tiled factorisation, with Huber
    lambda: with_constant_blocks(synthetic("constant_blocks", 12, 90, 5, 9), cameras=[0, 7], points=[3, 17, 40, 89]),   # round 6: SetParameterBlockConstant
    lambda: synthetic("border_40cams", 40, 300, 8, 7),
    lambda: synthetic("tiles_72cams_huber", 72, 200, 9, 8, 0.05, "huber", 1.0),
]


def main():
    for which in ("hongo", "test2"):
        blocks, summary, rows, prob = mc_solve(which)
        xml = read_xml_matrices(os.path.join(GOLDEN, which, "Camera_Transform.xml"))
        err = 0.0
        for c in range(prob["C"]):
            R = xml["R%d" % c]
            err = max(err, np.abs((rodrigues(blocks[c, :3]) if R.shape == (3, 3) else blocks[c, :3].reshape(3, 1)) - R).max(), np.abs(blocks[c, 3:] - xml["t%d" % c][:, 0]).max())
        print("%-16s marker-chain model: %d iterations, %s (%s), cost %.9e -> %.9e; the reference's committed Camera_Transform.xml reproduced to %.1e" % (
            which, len(rows) - 1, summary["termination"], summary["reason"], summary["initial_cost"], summary["final_cost"], err))
    only = sys.argv[1:]
    for make in CASES:
        prob = make()
        if only and prob["name"] not in only:
            continue
        x, summary, rows = minimise(prob)
        fx = dict(name=prob["name"], C=int(prob["C"]), P=int(prob["P"]), N=int(prob["N"]), loss=prob.get("loss", "none"), loss_scale=float(prob.get("loss_scale", 0.0)),
                  cam_idx=[int(v) for v in prob["cam_idx"]], pt_idx=[int(v) for v in prob["pt_idx"]], obs=[float(v) for v in prob["obs"]],
                  intr=[float(v) for v in prob["intr"]], params=[float(v) for v in prob["params"]],
                  constant_cameras=[int(v) for v in prob.get("constant_cameras", [])], constant_points=[int(v) for v in prob.get("constant_points", [])],
                  expected=dict(summary=summary, iterations=rows, final_params=[float(v) for v in x]),
                  generator="tools/replay_point_model.py (numpy %s): complex-step Jacobians, dense normal equations, SURVEY.md Appendix A.2" % np.__version__)
        path = os.path.join(GOLDEN, "point_model_%s.json" % prob["name"])
        json.dump(fx, open(path, "w"))
        print("%-16s C=%d P=%d N=%d: %d iterations, %s (%s), cost %.9e -> %.9e  -> %s" % (
            prob["name"], prob["C"], prob["P"], prob["N"], len(rows) - 1, summary["termination"], summary["reason"], summary["initial_cost"],
            summary["final_cost"], os.path.relpath(path, ROOT)))


if __name__ == "__main__":
    sys.exit(main())
