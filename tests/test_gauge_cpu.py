"""tests/gauge.py: the similarity orbit of the point model — the oracle's cost does not see it, `align` undoes it."""
import numpy as np

import gauge
from realsensecalibration_amd import synthetic as syn


def _random_similarity(rng):
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    return 1.0 + 0.05 * rng.normal(), Q, 0.3 * rng.normal(size=3)


def test_cost_is_invariant_along_the_orbit(oracle):
    prob = syn.make_problem(6, 200, 4, seed=5)
    rng = np.random.default_rng(1)
    s, Q, b = _random_similarity(rng)
    moved = gauge.apply(prob["params"], prob["C"], s, Q, b)
    c0, ss0 = oracle.points_cost(prob, prob["params"])
    c1, ss1 = oracle.points_cost(prob, moved)
    assert abs(c1 - c0) < 1e-9 * c0 and abs(ss1 - ss0) < 1e-9 * ss0
    assert np.abs(moved - prob["params"]).max() > 1e-2   # and it is a different parameter vector


def test_align_recovers_the_similarity():
    prob = syn.make_problem(5, 300, 4, seed=8)
    rng = np.random.default_rng(2)
    s, Q, b = _random_similarity(rng)
    moved = gauge.apply(prob["params"], prob["C"], s, Q, b)
    back, s2, Q2, b2 = gauge.align(moved, prob["params"], prob["C"])
    assert np.abs(back - prob["params"]).max() < 1e-10
    assert abs(s2 * s - 1.0) < 1e-12
