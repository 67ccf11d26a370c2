"""Pin the oracle against every known-answer pair the reference commits.

hongo/correspondence.txt  -> hongo/Camera_Transform.xml, hongo/point3d.txt, Extrinsics/mat{0..3}.txt
test2/correspondence_test.txt -> test2/Camera_Transform.xml, test2/point3d.txt
These are outputs of the reference's own Ceres 1.14 run; tolerance 1e-12 on the 17-digit XML,
text-rounding tolerance on the 6-digit files.
"""
import os

import numpy as np

import oracle_lib as ol

G = ol.GOLDEN


def _hongo(oracle):
    prob = ol.read_correspondence(os.path.join(G, "hongo", "correspondence.txt"))
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    params, s, log = oracle.solve_marker_chain(prob, 0, ol.MARKER_SIDE_MAIN, intr)
    return prob, intr, params, s, log


def test_hongo_problem_shape():
    prob = ol.read_correspondence(os.path.join(G, "hongo", "correspondence.txt"))
    assert (prob["T"], prob["C"], prob["M"], prob["N"]) == (6, 4, 11, 68)
    kinds = {(c != 0, m != 0) for c, m in zip(prob["c"], prob["m"])}
    n_target = sum(1 for c, m in zip(prob["c"], prob["m"]) if c != 0 and m != 0)
    n_base = sum(1 for c, m in zip(prob["c"], prob["m"]) if c == 0 and m != 0)
    n_tbm = sum(1 for c, m in zip(prob["c"], prob["m"]) if c != 0 and m == 0)
    n_bbm = sum(1 for c, m in zip(prob["c"], prob["m"]) if c == 0 and m == 0)
    assert (n_target, n_base, n_tbm, n_bbm) == (48, 14, 6, 0)
    assert prob["counts"].sum() == 68 and len(kinds) == 3


def test_hongo_trajectory(oracle):
    prob, intr, params, s, log = _hongo(oracle)
    # regression anchors of BASELINE.md §2 (replay of Ceres-1.14 semantics on the committed input)
    assert abs(s.initial_cost - 138796.696054) < 1e-5
    assert s.termination == 0 and s.stop_reason == 3  # CONVERGENCE by function tolerance
    assert s.num_iterations == 7 and s.num_successful_steps == 6 and s.num_unsuccessful_steps == 0
    expect = [1.044261172e5, 1.619009258e4, 1.140581758e4, 5.783243409e2, 1.441617308e2, 1.436293889e2]
    for k, e in enumerate(expect):
        assert abs(log[k + 1, 1] - e) / e < 1e-9
    assert abs(s.final_cost - 143.629388852) < 1e-8
    rms = np.sqrt(2 * s.final_cost / (2 * 4 * prob["N"]))
    assert abs(rms - 0.726669955) < 1e-8


def test_hongo_camera_transform_xml(oracle):
    prob, intr, params, s, log = _hongo(oracle)
    xml = ol.read_opencv_xml(os.path.join(G, "hongo", "Camera_Transform.xml"))
    worst = 0.0
    for i in range(prob["C"]):
        R = oracle.rodrigues(params[6 * i:6 * i + 3])  # BAManager::Write converts rvec -> R
        worst = max(worst, np.abs(R - xml["R%d" % i]).max(), np.abs(params[6 * i + 3:6 * i + 6] - xml["t%d" % i].ravel()).max())
    assert worst < 1e-12, worst
    # untouched blocks: camera 0 and marker 0 are never handed to the solver
    assert np.all(params[:6] == 0) and np.all(params[6 * (4 + 6):6 * (4 + 6) + 6] == 0)


def test_hongo_point3d_and_extrinsics(oracle):
    prob, intr, params, s, log = _hongo(oracle)
    n, counts, pts = ol.read_point3d(os.path.join(G, "hongo", "point3d.txt"))
    assert n == 4 * prob["N"] and np.array_equal(counts, 4 * prob["counts"])
    mine = oracle.marker_corners3d(prob, 0, ol.MARKER_SIDE_MAIN, params)
    assert np.abs(mine - pts).max() < 6e-7  # 6 significant digits in the file
    for i in range(4):
        ref = np.loadtxt(os.path.join(G, "extrinsics", "mat%d.txt" % i)).reshape(3, 4)
        R = oracle.rodrigues(params[6 * i:6 * i + 3])
        t = params[6 * i + 3:6 * i + 6]
        inv = np.hstack([R.T, (-R.T @ t)[:, None]])
        assert np.abs(inv - ref).max() < 6e-7


def test_hongo_reference_text_path_rms(oracle):
    """reprojection_check.cpp:76-101 applied to the committed 6-digit point3d.txt + XML."""
    prob = ol.read_correspondence(os.path.join(G, "hongo", "correspondence.txt"))
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    xml = ol.read_opencv_xml(os.path.join(G, "hongo", "Camera_Transform.xml"))
    n, counts, pts = ol.read_point3d(os.path.join(G, "hongo", "point3d.txt"))
    err = 0.0
    for i in range(prob["N"]):
        c = prob["c"][i]
        R, t = xml["R%d" % c], xml["t%d" % c].ravel()
        fx, fy, px, py = intr[c]
        for j in range(4):
            p = R @ pts[4 * i + j] + t
            u, v = fx * p[0] / p[2] + px, fy * p[1] / p[2] + py
            ou, ov = np.float32(prob["obs"][8 * i + 2 * j]), np.float32(prob["obs"][8 * i + 2 * j + 1])
            err += ((float(ou) - u) ** 2 + (float(ov) - v) ** 2) / 2
    rms = np.sqrt(err * 2.0 / (n * 2.0))
    assert abs(err - 143.639831820) < 1e-6 and abs(rms - 0.726696372) < 1e-8


def test_test2_fixture(oracle):
    prob = ol.read_correspondence(os.path.join(G, "test2", "correspondence_test.txt"))
    assert (prob["T"], prob["C"], prob["M"], prob["N"]) == (4, 2, 4, 20)
    intr = ol.read_intrinsics(ol.SERIALS_TEST2)
    params, s, log = oracle.solve_marker_chain(prob, 1, ol.MARKER_SIDE_TEST2, intr)
    xml = ol.read_opencv_xml(os.path.join(G, "test2", "Camera_Transform.xml"))
    assert s.termination == 0 and s.stop_reason == 3
    assert s.num_iterations == 4 and s.num_successful_steps == 3
    assert abs(s.final_cost - 13.301709) < 1e-5
    # Test2 writes the rvec itself under R{i} (Test2_BundleAdjustment/main.cpp:128)
    assert np.abs(params[6:9] - xml["R1"].ravel()).max() < 1e-12
    assert np.abs(params[9:12] - xml["t1"].ravel()).max() < 1e-12
    assert np.all(params[:6] == 0)
    n, counts, pts = ol.read_point3d(os.path.join(G, "test2", "point3d.txt"))
    mine = oracle.marker_corners3d(prob, 1, ol.MARKER_SIDE_TEST2, params)
    assert n == 80 and np.abs(mine - pts).max() < 6e-7


def test_termination_rule_is_decisive(oracle):
    """Applying the discarded candidate would move the answer outside the 1e-6 parity budget."""
    prob, intr, params, s, log = _hongo(oracle)
    o = oracle.options(function_tolerance=0.0, max_num_iterations=8)
    params8, s8, _ = oracle.solve_marker_chain(prob, 0, ol.MARKER_SIDE_MAIN, intr, o)
    assert np.abs(params8 - params).max() > 1e-6


# ------------------------------------------------------------------ the block-sparse marker-chain oracle (time blocks eliminated)
def _sparse_against_dense_and_xml(oracle, prob, variant, side, intr):
    dense, s_d, log_d = oracle.solve_marker_chain(prob, variant, side, intr)
    sparse, s_s, log_s = oracle.solve_marker_chain(prob, variant + 16, side, intr)
    assert (s_s.termination, s_s.stop_reason, s_s.num_iterations, s_s.num_successful_steps) == (s_d.termination, s_d.stop_reason, s_d.num_iterations, s_d.num_successful_steps)
    assert np.array_equal(log_s[:, 7], log_d[:, 7])
    assert np.abs(log_s[:, 1] - log_d[:, 1]).max() <= 1e-12 * log_d[:, 1].max()          # every iterate's cost
    assert np.abs(log_s[:, 6] - log_d[:, 6]).max() <= 1e-9 * log_d[:, 6].max()           # radius
    assert np.abs(sparse - dense).max() < 1e-11, np.abs(sparse - dense).max()
    return sparse


def test_sparse_marker_chain_oracle_on_the_reference_inputs(oracle):
    """oracle/ba_oracle.hpp MarkerChainSparseModel (round 5: block-sparse Jacobian, the time blocks eliminated — what lets the
    oracle run the marker-chain model at the size the product's path is benchmarked on) against the dense model on both inputs
    the reference commits, AND against the reference's own committed output (hongo/Camera_Transform.xml, 1e-12; test2's rvec)."""
    prob = ol.read_correspondence(os.path.join(G, "hongo", "correspondence.txt"))
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    params = _sparse_against_dense_and_xml(oracle, prob, 0, ol.MARKER_SIDE_MAIN, intr)
    xml = ol.read_opencv_xml(os.path.join(G, "hongo", "Camera_Transform.xml"))
    for i in range(prob["C"]):
        R = oracle.rodrigues(params[6 * i:6 * i + 3])
        assert np.abs(R - xml["R%d" % i]).max() < 1e-12 and np.abs(params[6 * i + 3:6 * i + 6] - xml["t%d" % i].ravel()).max() < 1e-12
    prob2 = ol.read_correspondence(os.path.join(G, "test2", "correspondence_test.txt"))
    intr2 = ol.read_intrinsics(ol.SERIALS_TEST2)
    params2 = _sparse_against_dense_and_xml(oracle, prob2, 1, ol.MARKER_SIDE_TEST2, intr2)
    xml2 = ol.read_opencv_xml(os.path.join(G, "test2", "Camera_Transform.xml"))
    assert np.abs(params2[6:9] - xml2["R1"].ravel()).max() < 1e-12 and np.abs(params2[9:12] - xml2["t1"].ravel()).max() < 1e-12


def test_sparse_marker_chain_oracle_on_a_synthetic_rig(oracle):
    """... and on a synthetic rig at the largest size the dense model still holds (6 cameras x 60 shots x 9 markers)."""
    from realsensecalibration_amd import synthetic as syn
    prob = syn.make_marker_chain(6, 60, 9, seed=4)
    intr = prob["intr"].reshape(-1, 4)
    _sparse_against_dense_and_xml(oracle, prob, 0, prob["marker_side"], intr)


def test_constant_blocks_of_the_marker_chain_model_match_the_numpy_replay(oracle):
    """Problem::SetParameterBlockConstant on blocks of the marker-chain model (round 6): camera 2, time 3 and marker 5 of the
    committed hongo input keep their file values — applied in every residual that names them, no columns, not in the norms.  The
    oracle (MarkerChainProblem::constant_block) against the independent numpy replay (tools/replay_point_model.py: complex-step
    Jacobians, dense normal equations; the same replay reproduces the reference's XML with every block free): same iteration count,
    final cost to 1e-10, parameters to 1e-9."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("replay_point_model", os.path.join(ol.ROOT, "tools", "replay_point_model.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    prob = ol.read_correspondence(os.path.join(G, "hongo", "correspondence.txt"))
    intr = ol.read_intrinsics(ol.SERIALS_MAIN)
    C, T = prob["C"], prob["T"]
    const = [2, C + 3, C + T + 5]
    params, s, log = oracle.solve_marker_chain_constant(prob, 0, ol.MARKER_SIDE_MAIN, intr, const)
    blocks, summary, rows, _ = rp.mc_solve("hongo", constant_blocks=const)
    assert s.num_iterations == len(rows) - 1 and summary["termination"] == "CONVERGENCE"
    assert abs(s.final_cost - summary["final_cost"]) < 1e-10 * summary["final_cost"]
    assert np.abs(params.reshape(-1, 6) - blocks).max() < 1e-9
    for b in const:
        assert np.array_equal(params[6 * b:6 * b + 6], prob["params"][6 * b:6 * b + 6])
    free, s_free, _ = oracle.solve_marker_chain(prob, 0, ol.MARKER_SIDE_MAIN, intr)
    assert s.final_cost > s_free.final_cost * (1 + 1e-6), "holding blocks at their initial guess must cost something"
