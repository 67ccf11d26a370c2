"""Synthetic point-model bundle-adjustment problems (BASELINE.json configs 2-5).

The reference ships one 16-point file for this model (Common/Correspondence/two_cam_data.txt,
written by Test1_ReprojectionError/main.cpp:162-183); larger problems of the same shape
(ReprojectionError<2,6,3>, Test1_BundleAdjustment/bundle_adjustmenter.cpp:106-148) are generated here.

Every quantity that belongs to a point (position, visibility, noise, initial guess) is drawn from a
generator seeded by (seed, point_block), so any rank can build exactly its own shard of a larger
problem without generating the rest: `point_range=(lo, hi)` returns points lo..hi-1 re-indexed from 0.
"""
import numpy as np

BLOCK = 4096  # points per independently seeded block

CONFIGS = {
    # name: (cameras, points, views per point, seed, outlier fraction, huber delta)
    "cfg2": (8, 10_000, 8, 2, 0.0, 0.0),
    "cfg3": (64, 100_000, 20, 3, 0.0, 0.0),
    "cfg4": (64, 1_000_000, 20, 4, 0.0, 0.0),
    "cfg5": (256, 500_000, 20, 5, 0.05, 1.0),
}


def _rotvec_from_matrix(R):
    """log map SO(3) -> angle-axis, batch of 3x3 (angles well inside (0, pi) here)."""
    R = np.asarray(R, float)
    tr = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(tr)
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    s = 2.0 * np.sin(th)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(s[..., None] > 1e-12, v / s[..., None], 0.0)
    return k * th[..., None]


def _matrix_from_rotvec(w):
    w = np.asarray(w, float)
    th = np.linalg.norm(w, axis=-1)
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -w[..., 2], w[..., 1]
    K[..., 1, 0], K[..., 1, 2] = w[..., 2], -w[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -w[..., 1], w[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(th > 1e-12, np.sin(th) / th, 1.0)[..., None, None]
        b = np.where(th > 1e-12, (1 - np.cos(th)) / th ** 2, 0.5)[..., None, None]
    return np.eye(3) + a * K + b * (K @ K)


def make_cameras(C, seed):
    """C cameras on a band of the 3 m sphere looking at the origin; returns truth (C,6), intrinsics (C,4)."""
    rng = np.random.default_rng([seed, 0xC0FFEE])
    i = np.arange(C) + 0.5
    z = 0.6 * (1 - 2 * i / C)  # |z| < 0.6: a band, not the poles
    phi = np.pi * (1 + 5 ** 0.5) * i
    r = np.sqrt(1 - z * z)
    centre = 3.0 * np.stack([r * np.cos(phi), r * np.sin(phi), z], -1)
    fwd = -centre / np.linalg.norm(centre, axis=1, keepdims=True)  # optical axis -> origin
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    R0 = np.stack([right, down, fwd], 1)  # rows: camera axes in world
    jitter = _matrix_from_rotvec(rng.normal(0, 0.05, (C, 3)))
    R = jitter @ R0
    t = -np.einsum("cij,cj->ci", R, centre)
    cams = np.hstack([_rotvec_from_matrix(R), t])
    f = rng.uniform(620, 640, C)
    intr = np.stack([f, f, rng.uniform(305, 325, C), rng.uniform(230, 245, C)], -1)
    return cams, intr


def make_problem(C, P, k, seed, point_range=None, outlier_frac=0.0, noise_px=0.5,
                 sigma_rvec=0.01, sigma_tvec=0.01, sigma_point=0.02):
    """Returns a dict with the arrays `rsba_problem_create_points` takes (observations sorted by point,
    then camera) plus the ground truth.  Parameter layout [C x (rvec, tvec) | P x xyz]."""
    k = min(k, C)
    lo, hi = (0, P) if point_range is None else point_range
    cams_true, intr = make_cameras(C, seed)
    rng_c = np.random.default_rng([seed, 0xCA11])
    cams0 = cams_true + np.hstack([rng_c.normal(0, sigma_rvec, (C, 3)), rng_c.normal(0, sigma_tvec, (C, 3))])
    Rm = _matrix_from_rotvec(cams_true[:, :3])
    pts_true, pts0, cam_idx, pt_idx, obs = [], [], [], [], []
    for b in range(lo // BLOCK, (hi + BLOCK - 1) // BLOCK):
        rng = np.random.default_rng([seed, 1, b])
        nb = min(BLOCK, P - b * BLOCK)
        X = rng.uniform(-0.5, 0.5, (nb, 3))
        vis = np.sort(np.argsort(rng.random((nb, C)), axis=1)[:, :k], axis=1).astype(np.int32)
        noise = rng.normal(0, noise_px, (nb, k, 2))
        out_mask = rng.random((nb, k)) < outlier_frac
        out_off = rng.uniform(-50, 50, (nb, k, 2))
        X0 = X + rng.normal(0, sigma_point, (nb, 3))
        s, e = max(lo, b * BLOCK) - b * BLOCK, min(hi, (b + 1) * BLOCK) - b * BLOCK
        X, vis, noise, out_mask, out_off, X0 = X[s:e], vis[s:e], noise[s:e], out_mask[s:e], out_off[s:e], X0[s:e]
        pc = np.einsum("nkij,nj->nki", Rm[vis], X) + cams_true[vis, 3:]
        uv = np.stack([intr[vis, 0] * pc[..., 0] / pc[..., 2] + intr[vis, 2],
                       intr[vis, 1] * pc[..., 1] / pc[..., 2] + intr[vis, 3]], -1)
        uv = uv + noise + np.where(out_mask[..., None], out_off, 0.0)
        base = b * BLOCK + s - lo
        pts_true.append(X)
        pts0.append(X0)
        cam_idx.append(vis.reshape(-1))
        pt_idx.append(np.repeat(np.arange(base, base + (e - s), dtype=np.int32), k))
        obs.append(uv.reshape(-1))
    pts_true = np.concatenate(pts_true)
    pts0 = np.concatenate(pts0)
    n_local = hi - lo
    return dict(C=C, P=n_local, N=int(n_local * k), k=k,
                cam_idx=np.ascontiguousarray(np.concatenate(cam_idx), np.int32),
                pt_idx=np.ascontiguousarray(np.concatenate(pt_idx), np.int32),
                obs=np.ascontiguousarray(np.concatenate(obs), np.float64),
                intr=np.ascontiguousarray(intr.reshape(-1), np.float64),
                params=np.ascontiguousarray(np.concatenate([cams0.reshape(-1), pts0.reshape(-1)]), np.float64),
                truth=np.concatenate([cams_true.reshape(-1), pts_true.reshape(-1)]))


def make_config(name, point_range=None, points=None):
    C, P, k, seed, outl, huber = CONFIGS[name]
    if points is not None:
        P = points
    prob = make_problem(C, P, k, seed, point_range=point_range, outlier_frac=outl)
    prob["huber_delta"] = huber
    prob["name"] = name
    return prob


def algorithmic_bytes_per_iteration(C, P, N):
    """every observation record (24 B) read twice, every point (24 B) read twice and
    written once, cameras + intrinsics (80 B) read twice, the reduced system written once."""
    return 2 * 24 * N + 3 * 24 * P + 2 * 80 * C + 8 * ((6 * C) ** 2 + 6 * C)


def schur_flops_per_iteration(views_per_point_counts, full_diagonal_blocks=False):
    """Algorithmic flops of the point elimination: per point with k views, k(k-1)/2 off-diagonal blocks of (6x3)(3x6)
    products (108 FMA each), k diagonal blocks of which only the 21 unique entries exist (63 FMA each) and k 6x3 mat-vecs
    for the right-hand side.  full_diagonal_blocks=True: the diagonal blocks at the full 108 FMA (what rounds 1-4 reported)."""
    k = np.asarray(views_per_point_counts, np.float64)
    diag = 216.0 if full_diagonal_blocks else 126.0
    return float(np.sum(k * (k - 1) / 2 * 216 + k * diag + k * 36))


# ------------------------------------------------------------------------------------------------
# Marker-chain model (BASELINE config 1's model at larger sizes): C cameras rigidly mounted around a
# main camera, a board carrying M ArUco markers, T shots of the board.  Layout of the reference's
# BALProblem (Main_Calibration/bundle_adjustment.h:18-54): parameters [C | T | M] x (rvec, tvec), one
# residual block per detected marker = row (time, camera, marker, 4 corners x (u, v)).
def _rotate(rvec, p):
    return np.einsum("...ij,...j->...i", _matrix_from_rotvec(rvec), p)


def make_marker_chain(C, T, M, seed, marker_side=0.08, keep=0.9, noise_px=0.3, sigma_rvec=0.02, sigma_tvec=0.02):
    """Returns the arrays `rsba_problem_create_marker_chain` takes (rows sorted by time, camera, marker) and the truth.
    Camera 0 / marker 0 are the base blocks (identity, not part of the problem)."""
    rng = np.random.default_rng([seed, 0xA2C0])
    f = rng.uniform(600, 640, C)
    intr = np.stack([f, f, rng.uniform(305, 335, C), rng.uniform(225, 255, C)], -1)
    cams = np.zeros((C, 6))
    cams[1:, :3] = rng.normal(0, 0.12, (C - 1, 3))
    cams[1:, 3:5] = rng.uniform(-0.4, 0.4, (C - 1, 2))
    cams[1:, 5] = rng.normal(0, 0.08, C - 1)
    side = int(np.ceil(np.sqrt(M)))
    marks = np.zeros((M, 6))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    cells = np.stack([gx.ravel(), gy.ravel()], -1)[:M].astype(float)
    marks[:, 3:5] = (cells - cells[0]) * (1.6 * marker_side)
    marks[1:, 5] = rng.normal(0, 0.01, M - 1)
    marks[1:, :3] = rng.normal(0, 0.05, (M - 1, 3))
    centre = marks[:, 3:].mean(0)
    times = np.zeros((T, 6))
    times[:, :3] = rng.normal(0, 0.25, (T, 3))
    times[:, 3] = rng.uniform(-0.3, 0.3, T)
    times[:, 4] = rng.uniform(-0.2, 0.2, T)
    times[:, 5] = rng.uniform(1.2, 2.4, T)
    times[:, 3:] -= _rotate(times[:, :3], np.broadcast_to(centre, (T, 3)))  # the board's middle, not marker 0, sits there
    h = marker_side / 2
    corners = np.array([[-h, h, 0.0], [h, h, 0.0], [h, -h, 0.0], [-h, -h, 0.0]])  # bundle_adjustment.h:92-101
    # corners of every marker in the board frame (M, 4, 3), then in the base camera's frame at every time (T, M, 4, 3)
    in_board = _rotate(marks[:, None, :3], corners[None]) + marks[:, None, 3:]
    in_base = _rotate(times[:, None, None, :3], in_board[None]) + times[:, None, None, 3:]
    rows, obs = [], []
    for c in range(C):
        pc = _rotate(cams[c, :3], in_base) + cams[c, 3:]
        u = intr[c, 0] * pc[..., 0] / pc[..., 2] + intr[c, 2]
        v = intr[c, 1] * pc[..., 1] / pc[..., 2] + intr[c, 3]
        ok = (pc[..., 2] > 0.3).all(-1) & (u > 0).all(-1) & (u < 640).all(-1) & (v > 0).all(-1) & (v < 480).all(-1)
        ok &= rng.random((T, M)) < keep
        tt, mm = np.nonzero(ok)
        rows.append(np.stack([tt, np.full_like(tt, c), mm], -1))
        obs.append(np.stack([u[tt, mm], v[tt, mm]], -1).reshape(-1, 8) + rng.normal(0, noise_px, (len(tt), 8)))
    rows, obs = np.concatenate(rows), np.concatenate(obs)
    order = np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))   # by time, camera, marker: the reference file's order
    rows, obs = rows[order], obs[order]
    rows_t, rows_c, rows_m = rows[:, 0], rows[:, 1], rows[:, 2]
    N = len(rows_t)
    truth = np.concatenate([cams.ravel(), times.ravel(), marks.ravel()])
    start = truth.reshape(-1, 6).copy()
    jitter = np.hstack([rng.normal(0, sigma_rvec, (C + T + M, 3)), rng.normal(0, sigma_tvec, (C + T + M, 3))])
    jitter[0] = 0.0          # base camera
    jitter[C + T] = 0.0      # base marker
    start += jitter
    return {"T": T, "C": C, "M": M, "N": N, "t": np.array(rows_t, np.int32), "c": np.array(rows_c, np.int32),
            "m": np.array(rows_m, np.int32), "obs": np.array(obs, float).reshape(N, 8), "params": start.ravel(), "truth": truth,
            "intr": intr, "marker_side": marker_side}
