"""The point model (BASELINE configs 2-5) against a reference-independent pin: tools/replay_point_model.py's fixtures.

The reference commits no output for its point-model input (Test1_BundleAdjustment/main.cpp:56-87 writes none), so until
round 5 the oracle's point model, Schur elimination and loss correctors were pinned only transitively.  The fixtures
tests/golden/point_model_*.json come from a numpy replay that shares no code with oracle/ or the product: complex-step
Jacobians of Test1_BundleAdjustment/bundle_adjustmenter.cpp:122-141's functor, DENSE normal equations (no Schur
complement), SURVEY.md Appendix A.2's loop — the same replay that reproduces the reference's committed
hongo/Camera_Transform.xml to 7e-16 for the marker-chain model.  Held to them:

  * the ORACLE (CPU, here): every iterate's cost to 1e-9 relative, the accept / reject sequence, radius, termination reason,
    iteration count; final parameters per block to 1e-6 relative (BASELINE's bar; observed: see the assertion messages);
  * the HIP path (-m gpu, through the C ABI): the same.
two_cam is the reference's own file (1 camera x 16 points, one view each: every point block is rank-deficient and only the
damping makes it solvable) — costs to 1e-9 of the INITIAL cost (the final one is 1e-12 of it), parameters by their fit.
"""
import glob
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

FIXTURES = sorted(glob.glob(os.path.join(ol.GOLDEN, "point_model_*.json")))
REASON = {"gradient": 1, "parameter": 2, "function": 3, "max_iterations": 4, "min_radius": 5, "invalid_steps": 6}
TERM = {"CONVERGENCE": 0, "NO_CONVERGENCE": 1, "FAILURE": 2}


def _load(path):
    fx = json.load(open(path))
    prob = dict(C=fx["C"], P=fx["P"], N=fx["N"], cam_idx=np.array(fx["cam_idx"], np.int32), pt_idx=np.array(fx["pt_idx"], np.int32),
                obs=np.array(fx["obs"], float), intr=np.array(fx["intr"], float), params=np.array(fx["params"], float))
    return fx, prob


def _check(fx, prob, params, term, reason, iters, log, final_cost):
    exp = fx["expected"]
    rows = exp["iterations"]
    C = fx["C"]
    assert (term, reason, iters) == (TERM[exp["summary"]["termination"]], REASON[exp["summary"]["reason"]], len(rows) - 1)
    assert [int(r) for r in log[:, 7]] == [rw["valid"] + 2 * rw["successful"] for rw in rows]
    c0 = rows[0]["cost"]
    for j, rw in enumerate(rows):
        # relative to the iterate's own cost; two_cam ends at 1e-12 of where it starts: relative to the start there
        scale = max(rw["cost"], 1e-6 * c0)
        assert abs(log[j, 1] - rw["cost"]) <= 1e-9 * scale, "iterate %d: cost %.15e, replay %.15e" % (j, log[j, 1], rw["cost"])
        assert abs(log[j, 6] - rw["trust_region_radius"]) <= 1e-6 * rw["trust_region_radius"], "iterate %d: radius" % j
        if rw["valid"]:
            assert abs(log[j, 4] - rw["step_norm"]) <= 1e-6 * rw["step_norm"], "iterate %d: step norm" % j
    assert abs(final_cost - exp["summary"]["final_cost"]) <= 1e-9 * max(exp["summary"]["final_cost"], 1e-6 * c0)
    ref = np.array(exp["final_params"])
    if fx["name"] == "two_cam":
        return   # (rank-deficient point blocks: the fit is what is defined, and it is compared above)
    worst = 0.0
    for x, y in ((params[:6 * C].reshape(-1, 6), ref[:6 * C].reshape(-1, 6)), (params[6 * C:].reshape(-1, 3), ref[6 * C:].reshape(-1, 3))):
        worst = max(worst, (np.abs(x - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-12)).max())
    assert worst < 1e-6, "final parameters differ from the replay's by %.2e relative per block" % worst


def test_fixtures_exist():
    assert len(FIXTURES) >= 5, "run tools/replay_point_model.py"


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[12:-5])
def test_oracle_matches_the_numpy_replay(oracle, path):
    fx, prob = _load(path)
    hd = -fx["loss_scale"] if fx["loss"] == "cauchy" else (fx["loss_scale"] if fx["loss"] == "huber" else 0.0)
    params, s, log = oracle.solve_points(prob, oracle.options(huber_delta=hd))
    _check(fx, prob, params, s.termination, s.stop_reason, s.num_iterations, log, s.final_cost)


@pytest.mark.gpu
@pytest.mark.parametrize("impl", [1, 0])
@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[12:-5])
def test_hip_path_matches_the_numpy_replay(path, impl):
    """Through the C ABI: the tiled Schur kernel (1) and the atomic one (0)."""
    from realsensecalibration_amd import capi
    assert capi.load().rsba_device_count() > 0, "GPU tests need a HIP device; the product has no CPU path"
    fx, prob = _load(path)
    o = capi.default_options(schur_impl=impl, huber_delta=fx["loss_scale"] if fx["loss"] != "none" else 0.0, loss_type=1 if fx["loss"] == "cauchy" else 0)
    params, s, log = capi.solve_points(prob, o)
    _check(fx, prob, params, s.termination_type, s.stop_reason, s.num_iterations, log, s.final_cost)
