"""How far the oracle ends from ITSELF when only the roundings change (CPU; tests/oracle_spread.py) — the evidence behind the
parameter tolerances of the GPU parity tests.

VERDICT r02 asked: does the Ceres-1.14 restatement differ from itself by more than 1e-6 on the suite's long robust run under
a change of summation order or contraction?  It does not (< 1e-7, first test), so tests/test_gpu_parity.py compares RAW
parameters at 1e-6 there, and the round-2 gauge alignment is gone.  On other problems it does — the second and third test pin
one of each kind from the randomised sweep (tests/fuzz_cases.py) — and there the sweep's bar is ten times the oracle's own
spread instead of a number the reference algorithm itself does not meet.
"""
import numpy as np

import fuzz_cases
import oracle_lib
import oracle_spread
from realsensecalibration_amd import synthetic as syn


def _fuzz_problem(index):
    c = fuzz_cases.cases(210, 1)[index]
    prob = syn.make_problem(c["C"], c["P"], c["k"], seed=c["seed"], outlier_frac=0.0 if c["loss"] == "none" else 0.05)
    return c, prob, (-c["scale"] if c["loss"] == "cauchy" else c["scale"])


def test_oracle_agrees_with_itself_on_the_long_huber_run(oracle):
    """test_huber_and_rejected_steps' problem: 17 iterations, radius up to 4e11, every block free.  Three executions of the
    restatement (as built; -ffp-contract=off; points summed in reverse order) end within 5e-7 of each other in the RAW
    parameters (measured 8e-8 .. 9e-8) and 1e-10 after alignment along the gauge orbit: the 1e-6 bar is defined here."""
    prob = syn.make_problem(8, 1500, 6, seed=9, outlier_frac=0.05)
    sp = oracle_spread.spread(oracle, prob, dict(huber_delta=1.0))
    assert sp["same_trajectory"] and sp["part"] < 0
    assert sp["raw"] < 5e-7 and sp["aligned"] < 1e-9 and sp["final_cost"] < 1e-12
    assert oracle_spread.bars(sp, prob["N"])["raw"] <= 5e-6


def test_oracle_drifts_along_the_gauge_orbit_at_large_radius(oracle):
    """Sweep case 41 (31 cameras, 4 views per point, Huber 2.5, 26 iterations, radius up to 8e15): same trajectory, same costs
    to 1e-9 at every iterate — and the oracle's own runs end 1e-3 apart in the raw parameters, 3e-7 after alignment.  1e-6
    relative on the final poses is not defined for the reference algorithm on this problem."""
    c, prob, hd = _fuzz_problem(41)
    assert (c["C"], c["k"], c["loss"]) == (31, 4, "huber")
    sp = oracle_spread.spread(oracle, prob, dict(huber_delta=hd))
    assert sp["same_trajectory"] and sp["part"] < 0 and sp["final_cost"] < 1e-9
    assert sp["raw"] > 1e-5, "the oracle used to part from itself by 1e-3 here"
    assert sp["aligned"] < 1e-5 and sp["aligned"] < 0.05 * sp["raw"]


def test_oracle_runs_part_on_a_two_view_robust_problem(oracle):
    """Sweep case 21 (128 cameras, TWO views per point, Huber, ends at the iteration limit): the oracle's own runs part at
    iteration 8 and end 0.2 apart."""
    c, prob, hd = _fuzz_problem(21)
    assert (c["C"], c["k"], c["loss"]) == (128, 2, "huber")
    sp = oracle_spread.spread(oracle, prob, dict(huber_delta=hd))
    assert 3 < sp["part"] < 20 and sp["raw"] > 1e-3
    b = oracle_spread.bars(sp, prob["N"])
    assert b["agree_until"] == max(3, sp["part"] - 3) and b["final_cost"] >= 1e-2


def test_reversed_points_is_the_same_problem(oracle):
    prob = syn.make_problem(5, 200, 3, seed=4)
    q = oracle_spread.reversed_points(prob)
    a, _ = oracle.points_cost(prob, prob["params"])
    b, _ = oracle.points_cost(q, q["params"])
    assert abs(a - b) < 1e-12 * a
    assert np.array_equal(oracle_spread.unreverse(q["params"], 5), prob["params"])


def test_the_referee_decides_the_case_whose_bar_was_below_the_oracles_own_noise(oracle):
    """Sweep 400 / seed 777, case 224 (32 cameras, every point seen by TWO of them, Huber 2.5): the oracle's three double-precision
    executions are 1.7e-10 apart on the first three iterates' costs — above the first-three-iterates ceiling of 1e-10, which
    therefore could neither pass nor indict an implementation (round 5 recorded the HIP path at 1.6e-10, left standing).  THE REFEREE
    RULE (oracle_spread.py): against the same trust-region loop with the linear solve in long double (liboracle_wide.so), the oracle's
    main execution is 1.4e-10 off and its two variants ~3e-11 — the spread is the main execution's own rounding error, and the bar for
    an implementation is three times that distance."""
    c = fuzz_cases.cases(400, 777)[224]
    assert (c["C"], c["k"], c["loss"]) == (32, 2, "huber")
    prob = syn.make_problem(c["C"], c["P"], c["k"], seed=c["seed"], outlier_frac=0.05)
    hd = c["scale"]
    sp = oracle_spread.spread(oracle, prob, dict(huber_delta=hd))
    assert 1e-10 < sp["first3"] < 1e-9 and oracle_spread.needs_referee(sp)
    ds = [oracle_spread.referee(oracle, prob, dict(huber_delta=hd), log, sp["logs"]) for log in sp["logs"]]
    assert all(n == 3 for _, _, n in ds), "the referee takes the same three decisions as every execution"
    d_or = ds[0][1]
    assert 5e-11 < d_or < 5e-10 and abs(max(d for d, _, _ in ds) - d_or) < 1e-15
    assert min(d for d, _, _ in ds) < 0.5 * d_or, "two of the three executions are much closer to the long-double solve than the third"
    # the long-double solve is the same algorithm: every iterate it shares with the oracle agrees to 1e-9
    wide = oracle_lib.load_wide()
    _, _, lw = wide.solve_points(prob, wide.options(huber_delta=hd))
    assert np.abs(lw[:4, 1] - sp["logs"][0][:4, 1]).max() < 1e-9 * lw[0, 1]


def test_the_referee_build_is_the_oracle_on_a_well_conditioned_problem(oracle):
    """liboracle_wide.so differs from the oracle in the precision of the linear solve only: on a problem with eight views per point
    both take the same decisions and every iterate's cost agrees to 1e-12."""
    prob = syn.make_problem(8, 400, 8, seed=21)
    wide = oracle_lib.load_wide()
    a, sa, la = oracle.solve_points(prob, oracle.options())
    b, sb, lb = wide.solve_points(prob, wide.options())
    assert sa.num_iterations == sb.num_iterations and np.array_equal(la[:, 7], lb[:, 7])
    assert np.abs(la[:, 1] - lb[:, 1]).max() < 1e-12 * la[0, 1]
    assert oracle_spread.block_rel(a, b, prob["C"]) < 1e-8
