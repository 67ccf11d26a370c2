"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference front end's initial-guess math, the checker for
realsensecalibration_amd/csrc/ba_initial_guess.cpp.  Only tests may import this file.

Follows /root/reference/Main_Calibration/correspondencer.cpp:
  compose / invert of marker poses        :119-127, :137-147
  GetCornersInCameraWorld                 :5-39
  solvePnP(..., SOLVEPNP_EPNP)            :192-195 -> OpenCV calib3d epnp.cpp (third-party, not vendored, no version pin
                                          in the reference): EPnP of Lepetit, Moreno-Noguer & Fua (IJCV 2009) as that
                                          file runs it.  Pinned on the reference's own output: the camera rows of the
                                          committed Common/Correspondence/hongo/correspondence.txt are EPnP results
                                          (tests/test_initial_guess.py::test_epnp_reproduces_the_committed_initial_guesses).
Linear algebra is numpy's LAPACK (svd / lstsq / eigh), deliberately not the Jacobi routines of the product — except the
3 x 3 SVD that fixes the SIGNS of the control-point axes, which restates OpenCV 4.0.1's own Jacobi routine (see
opencv_jacobi_svd_ut): with it the committed camera rows are reproduced to their six printed digits.
"""
import numpy as np


def rodrigues(rvec):
    rvec = np.asarray(rvec, float)
    th = np.linalg.norm(rvec)
    if th < 1e-300:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def rvec_from_matrix(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v)
    return np.zeros(3) if s < 1e-300 else v / s * th


def base_pose_from_marker_detection(marker_from_camera, marker_from_base):
    Rc, Rb = rodrigues(marker_from_camera[:3]), rodrigues(marker_from_base[:3])
    R = Rc @ Rb.T
    return np.concatenate([rvec_from_matrix(R), R @ (-np.asarray(marker_from_base[3:])) + marker_from_camera[3:]])


def marker_pose_in_camera(base_from_camera, marker_from_base):
    Rb, Rm = rodrigues(base_from_camera[:3]), rodrigues(marker_from_base[:3])
    return np.concatenate([rvec_from_matrix(Rb @ Rm), Rb @ np.asarray(marker_from_base[3:]) + base_from_camera[3:]])


def marker_corners_in_camera(pose, side):
    R, h = rodrigues(pose[:3]), side / 2
    E, F, t = R[:, 0] * h, R[:, 1] * h, np.asarray(pose[3:], float)
    return np.stack([t - E + F, t + E + F, t + E - F, t - E - F])


def opencv_jacobi_svd_ut(A):
    """U' (rows) and the singular values of a small square matrix as OpenCV 4.0.1's cvSVD(..., CV_SVD_U_T) returns them:
    JacobiSVDImpl_<double> of modules/core/src/lapack.cpp (one-sided Hestenes Jacobi on the rows of A', cyclic pairs, the
    a < b branch of the rotation, selection sort of the singular values).  Only the SIGNS of the vectors are the point of
    restating it: EPnP's control points inherit them, and on noisy data the pose depends on them."""
    A = np.asarray(A, float)
    n = A.shape[0]
    At = A.T.copy()
    w2 = (At * At).sum(1)
    eps = np.finfo(float).eps * 10
    for _ in range(max(n, 30)):
        changed = False
        for i in range(n - 1):
            for j in range(i + 1, n):
                a, b, p = w2[i], w2[j], float(At[i] @ At[j])
                if abs(p) <= eps * np.sqrt(a * b):
                    continue
                p *= 2
                beta = a - b
                gamma = np.hypot(p, beta)
                if beta < 0:
                    s = np.sqrt((gamma - beta) * 0.5 / gamma)
                    c = p / (gamma * s * 2)
                else:
                    c = np.sqrt((gamma + beta) / (gamma * 2))
                    s = p / (gamma * c * 2)
                t0, t1 = c * At[i] + s * At[j], -s * At[i] + c * At[j]
                At[i], At[j] = t0, t1
                w2[i], w2[j] = t0 @ t0, t1 @ t1
                changed = True
        if not changed:
            break
    W = np.sqrt((At * At).sum(1))
    for i in range(n - 1):
        j = i + int(np.argmax(W[i:]))   # first maximum, as the strict comparison of the selection sort picks it
        if i != j:
            W[[i, j]] = W[[j, i]]
            At[[i, j]] = At[[j, i]]
    return At / np.where(W > 0, W, 1.0)[:, None], W


def epnp(obj, img, k4):
    obj, img = np.asarray(obj, float).reshape(-1, 3), np.asarray(img, float).reshape(-1, 2)
    n = len(obj)
    fu, fv, uc, vc = k4
    c0 = obj.mean(0)
    # principal axes and their signs as OpenCV's epnp.cpp gets them (cvSVD of PW0'PW0 with CV_SVD_U_T)
    axes, w = opencv_jacobi_svd_ut((obj - c0).T @ (obj - c0))
    cws = np.vstack([c0] + [c0 + np.sqrt(max(w[i], 0) / n) * axes[i] for i in range(3)])
    CC = (cws[1:] - cws[0]).T
    al = np.linalg.solve(CC, (obj - c0).T).T
    alphas = np.hstack([1 - al.sum(1, keepdims=True), al])
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[:, j] * fu
        M[0::2, 3 * j + 2] = alphas[:, j] * (uc - img[:, 0])
        M[1::2, 3 * j + 1] = alphas[:, j] * fv
        M[1::2, 3 * j + 2] = alphas[:, j] * (vc - img[:, 1])
    ew, ev = np.linalg.eigh(M.T @ M)
    v = [ev[:, i] for i in range(4)]  # ascending: v[0] belongs to the smallest eigenvalue
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    dv = np.array([[vi[3 * a:3 * a + 3] - vi[3 * b:3 * b + 3] for a, b in pairs] for vi in v])  # (4, 6, 3)
    d = lambda i, j: np.einsum("pk,pk->p", dv[i], dv[j])  # noqa: E731
    L = np.stack([d(0, 0), 2 * d(0, 1), d(1, 1), 2 * d(0, 2), 2 * d(1, 2), d(2, 2), 2 * d(0, 3), 2 * d(1, 3), 2 * d(2, 3), d(3, 3)], 1)
    rho = np.array([np.sum((cws[a] - cws[b]) ** 2) for a, b in pairs])

    def gauss_newton(b):
        b = b.copy()
        for _ in range(5):
            A = np.stack([2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3],
                          L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                          L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3],
                          L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]], 1)
            bb = np.array([b[0] * b[0], b[0] * b[1], b[1] * b[1], b[0] * b[2], b[1] * b[2], b[2] * b[2], b[0] * b[3], b[1] * b[3],
                           b[2] * b[3], b[3] * b[3]])
            b += np.linalg.lstsq(A, rho - L @ bb, rcond=None)[0]
        return b

    def pose(b):
        ccs = sum(b[i] * v[i].reshape(4, 3) for i in range(4))
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:
            pcs = -pcs
        pc0, pw0 = pcs.mean(0), obj.mean(0)
        U, _, Vt = np.linalg.svd((pcs - pc0).T @ (obj - pw0))
        R = U @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        pc = obj @ R.T + t
        err = np.mean(np.hypot(uc + fu * pc[:, 0] / pc[:, 2] - img[:, 0], vc + fv * pc[:, 1] / pc[:, 2] - img[:, 1]))
        return err, R, t

    cands = []
    x = np.linalg.lstsq(L[:, [0, 1, 3, 6]], rho, rcond=None)[0]
    sg = -1.0 if x[0] < 0 else 1.0
    b0 = np.sqrt(sg * x[0])
    cands.append(np.array([b0, sg * x[1] / b0, sg * x[2] / b0, sg * x[3] / b0]))
    x = np.linalg.lstsq(L[:, :3], rho, rcond=None)[0]
    b = np.zeros(4)
    if x[0] < 0:
        b[0], b[1] = np.sqrt(-x[0]), (np.sqrt(-x[2]) if x[2] < 0 else 0.0)
    else:
        b[0], b[1] = np.sqrt(x[0]), (np.sqrt(x[2]) if x[2] > 0 else 0.0)
    if x[1] < 0:
        b[0] = -b[0]
    cands.append(b)
    x = np.linalg.lstsq(L[:, :5], rho, rcond=None)[0]
    b = np.zeros(4)
    if x[0] < 0:
        b[0], b[1] = np.sqrt(-x[0]), (np.sqrt(-x[2]) if x[2] < 0 else 0.0)
    else:
        b[0], b[1] = np.sqrt(x[0]), (np.sqrt(x[2]) if x[2] > 0 else 0.0)
    if x[1] < 0:
        b[0] = -b[0]
    b[2] = x[3] / b[0]
    cands.append(b)
    best = None
    for b in cands:
        r = pose(gauss_newton(b))
        if best is None or r[0] < best[0]:
            best = r
    return np.concatenate([rvec_from_matrix(best[1]), best[2]]), best[0]
