"""How far the ORACLE ends from itself on a problem when nothing but the roundings change (test infrastructure).

BASELINE.json asks for final poses / points within 1e-6 relative of the reference path's.  That bar presumes the reference
path's result is DEFINED to 1e-6 — and with every block free (Test1_BundleAdjustment/main.cpp:76-79) it is not on every
problem: the cost has a 7-dof gauge orbit, along which the damped system's eigenvalues are only diag / radius, so at a
trust-region radius of 1e13 rounding errors of the gradient are amplified 1e13-fold along the orbit; and a robust loss on a
point with two or three views makes runs that go to the iteration limit with the radius at its cap, on which any two
summation orders part after ~20 iterations.  Two faithful executions of the Ceres algorithm differ there, so a third
implementation may differ from either by as much.

`spread` measures that: the restatement run twice more with different roundings and the SAME arithmetic —
  * built with -ffp-contract=off (oracle/Makefile: liboracle_nocontract.so), and
  * with the points renamed in reverse order (the Schur sums over the points run the other way round) —
and returns how far the three runs end apart.  `bars` turns that into the tolerances a parity test may use: BASELINE's
own (1e-6 raw per block, 1e-9 final cost, 1e-4 px) wherever the oracle agrees with itself ten times better than that, and
ten times the oracle's own spread where it does not.  On the suite's fixed cases the spread is < 1e-7 and the bars are
BASELINE's; tests/test_oracle_self_sensitivity.py pins both situations on the CPU.
"""
import numpy as np

import gauge
import oracle_lib


def block_rel(a, b, C):
    """max over 6-/3-blocks of |da| / max(|b|, 1e-12)."""
    worst = 0.0
    for x, y in ((a[:6 * C].reshape(-1, 6), b[:6 * C].reshape(-1, 6)), (a[6 * C:].reshape(-1, 3), b[6 * C:].reshape(-1, 3))):
        worst = max(worst, (np.abs(x - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-12)).max())
    return worst


def cost_tolerance(cost, n_obs, rel=1e-12):
    """Two evaluations of the same cost: `rel` relative, or what a rounding error of `rel` px in every residual explains
    (sum |r| dr <= sqrt(2 N cost) dr) — the second term matters when the fit is exact and the cost is ~0."""
    return rel * cost + rel * np.sqrt(2.0 * n_obs * cost)


def reversed_points(prob):
    """The same problem with point j renamed P - 1 - j (observations re-sorted by point, then camera)."""
    C, P = prob["C"], prob["P"]
    q = dict(prob)
    newpt = (P - 1 - prob["pt_idx"]).astype(np.int32)
    order = np.lexsort((prob["cam_idx"], newpt))
    q["pt_idx"] = np.ascontiguousarray(newpt[order])
    q["cam_idx"] = np.ascontiguousarray(prob["cam_idx"][order])
    q["obs"] = np.ascontiguousarray(prob["obs"].reshape(-1, 2)[order].reshape(-1))
    par = prob["params"].copy()
    par[6 * C:] = prob["params"][6 * C:].reshape(-1, 3)[::-1].reshape(-1)
    q["params"] = par
    return q


def unreverse(x, C):
    y = x.copy()
    y[6 * C:] = x[6 * C:].reshape(-1, 3)[::-1].reshape(-1)
    return y


def first_parting(log_a, log_b, n_obs, rel):
    """First iterate whose costs differ by more than cost_tolerance(rel); -1: never."""
    m = min(len(log_a), len(log_b))
    for j in range(m):
        if abs(log_a[j, 1] - log_b[j, 1]) > cost_tolerance(abs(log_b[j, 1]), n_obs, rel):
            return j
    return -1 if len(log_a) == len(log_b) else m


def spread(oracle, prob, optkw, ref=None):
    """ref = (params, summary, log) of `oracle` on `prob` if the caller has it already."""
    C, N = prob["C"], prob["N"]
    params, s, log = ref if ref is not None else oracle.solve_points(prob, oracle.options(**optkw))
    variant = oracle_lib.load_nocontract()
    runs = [variant.solve_points(prob, variant.options(**optkw))]
    a, sa, la = oracle.solve_points(reversed_points(prob), oracle.options(**optkw))
    runs.append((unreverse(a, C), sa, la))
    hd = optkw.get("huber_delta", 0.0)
    _, ss_ref = oracle.points_cost(prob, params, huber_delta=hd)
    # first iterate whose trust-region radius exceeds RADIUS_CAP (column 6 of the log: the radius after the step); -1: none
    big = np.nonzero(log[:, 6] > RADIUS_CAP)[0]
    out = dict(same_trajectory=True, raw=0.0, aligned=0.0, final_cost=0.0, rms=0.0, part=-1, radius_iter=int(big[0]) if len(big) else -1, first3=0.0,
               logs=[log] + [la for _, _, la in runs])   # (the three executions' iteration logs: oracle_spread.referee)
    for a, sa, la in runs:
        same = (sa.termination == s.termination and sa.stop_reason == s.stop_reason and sa.num_iterations == s.num_iterations and
                np.array_equal(la[:, 7], log[:, 7]))
        out["same_trajectory"] = out["same_trajectory"] and bool(same)
        out["raw"] = max(out["raw"], block_rel(a, params, C))
        out["aligned"] = max(out["aligned"], block_rel(gauge.align(a, params, C)[0], params, C))
        out["final_cost"] = max(out["final_cost"], abs(sa.final_cost - s.final_cost) / max(s.final_cost, 1e-300))
        _, ss = oracle.points_cost(prob, a, huber_delta=hd)
        out["rms"] = max(out["rms"], abs(np.sqrt(ss / (2 * N)) - np.sqrt(ss_ref / (2 * N))))
        # how far the first three iterates' costs of the oracle's own executions lie apart (in units of cost_tolerance(., ., 1))
        for j in range(1, min(len(la), len(log), 4)):
            out["first3"] = max(out["first3"], abs(la[j, 1] - log[j, 1]) / cost_tolerance(abs(log[j, 1]), N, 1.0))
        p = first_parting(la, log, N, 1e-10)
        if p >= 0:
            out["part"] = p if out["part"] < 0 else min(out["part"], p)
    return out


# THE MARGIN RULE for runs on which the oracle parts from itself.  The iterates' costs of the implementation under test have to
# agree with the oracle's (1e-9) up to and including iterate
#     agree_until = max(3, min(P - 3, R))
# where P is the first iterate at which the oracle's own three executions part (1e-10) and R the first iterate whose trust-region
# radius exceeds RADIUS_CAP = 1e13 (no such iterate: R = infinity); beyond it nothing is asked of single iterates, only of the end
# state.  Why R: the damped system's eigenvalues along the 7-dof gauge orbit are diag / radius, so from 1e13 on a rounding error
# of one ulp in the gradient moves the iterate along the orbit by more than the 1e-9 bar resolves, and the accept / reject
# decision of a step whose cost change is at the rounding level (rho ~ 0 / 0) is decided by the roundings — two samples of the
# oracle's own parting point (P) do not bound where a third execution parts, the radius does.  (Seed-31 sweep case 038, 2 cameras
# x 69 points x 2 views, Cauchy: radius 1.2e13 at iterate 19, the oracle's runs part at 25, the HIP path at 22 with a final cost
# equal to 1.6e-13 — pinned in tests/test_gpu_fuzz.py.)
RADIUS_CAP = 1e13
FIRST3_CEILING = 1e-10   # no first-three-iterates bar above this, whatever the oracle's own spread

# THE REFEREE RULE (round 6) for the first-three-iterates bar where the ceiling above is BELOW the checker's own noise.  When the
# oracle's own three executions differ on the first three iterates' costs by more than a tenth of the ceiling (`first3 > 1e-11`:
# points seen by two cameras, nearly singular 3 x 3 point blocks whose inverse amplifies the last bits of the very first solve),
# the comparison "implementation against oracle" measures the sum of two rounding errors of the same size and can neither pass nor
# indict anybody (round 5: sweep 400 / seed 777, case 224 — 1.6e-10 from the oracle, whose own executions are 1.67e-10 apart).  Such a
# case goes to the referee: the same trust-region loop with the LINEAR SOLVE in long double (oracle/ba_oracle.hpp, RSBA_ORACLE_WIDE:
# point-block inverses, Schur sums, dense LLT, back-substitution; 11 more bits).  Measured against IT,
#     d_impl   = the implementation's distance on the first three iterates' costs,
#     d_oracle = the largest distance of the oracle's three double-precision executions,
# the implementation passes iff  d_impl <= max(1e-12, REFEREE_FACTOR x d_oracle): it solves the step as exactly as the double-precision
# restatement of the reference does, which is all a double-precision implementation can be asked for.  No ceiling is needed: both
# sides are measured against the same, better answer.  Only iterates the referee shares with every execution count (same accept /
# reject decisions so far).  The class is counted, and tests/test_gpu_fuzz.py asserts it stays rare.
REFEREE_TRIGGER = 0.1 * FIRST3_CEILING
REFEREE_FACTOR = 3.0


def needs_referee(sp):
    return sp.get("first3", 0.0) > REFEREE_TRIGGER


def referee(oracle, prob, optkw, log_impl, logs_oracle):
    """(d_impl, d_oracle, iterates compared): distances on the first three iterates' costs to the long-double referee's, in units
    of cost_tolerance(., ., 1) like `first3`.  logs_oracle: the iteration logs of the oracle's double-precision executions."""
    wide = oracle_lib.load_wide()
    _, sw, lw = wide.solve_points(prob, wide.options(**optkw))
    N = prob["N"]

    def dist(log):
        d, n = 0.0, 0
        for j in range(1, min(len(log), len(lw), 4)):
            if not np.array_equal(log[:j + 1, 7], lw[:j + 1, 7]):
                break
            d = max(d, abs(log[j, 1] - lw[j, 1]) / cost_tolerance(abs(lw[j, 1]), N, 1.0))
            n = j
        return d, n
    d_impl, n_impl = dist(log_impl)
    ds = [dist(l) for l in logs_oracle]
    return d_impl, max(d for d, _ in ds), min([n_impl] + [n for _, n in ds])


def bars(sp, n_obs):
    """The tolerances a parity test may use on this problem (see the module docstring and THE MARGIN RULE above)."""
    parted = sp["part"] >= 0 or not sp["same_trajectory"]
    # Runs whose iterates part (the oracle's own!) are chaotic from there on: two samples of the end state do not bound a third.
    # Round 5: the end state of such a run is no parity bar any more (rounds 3-4: cost within 1 %, RMS within 0.1 px) — fuzz_cases.run
    # re-runs both solvers with the iteration limit at agree_until and compares THAT state at BASELINE's bars; final_cost / rms / raw
    # below are what a caller without that second run may still use.
    # the first three iterates' costs: 1e-12 (a defect shows at once, rounding shows late) — or ten times what the oracle's own
    # executions differ by there, on the problems whose FIRST solves already amplify the roundings (points seen by two cameras:
    # nearly singular point blocks; seed-123 sweep, cases 080 / 218 / 289: the oracle's runs 0.8 - 5.7e-12 apart at iterates 1 - 3, the HIP path 1.7 - 6.0e-12 from the oracle)
    # ... with an ABSOLUTE ceiling of 1e-10 (the relaxed bar was derived from the cases a build had failed; beyond
    # 1e-10 on the first three iterates nothing the oracle does to itself is an excuse), and tests/test_gpu_fuzz.py asserts that the
    # relaxed bar stays the rare exception
    return dict(first3=min(FIRST3_CEILING, max(1e-12, 10.0 * sp.get("first3", 0.0))), raw=max(1e-6, 10.0 * sp["raw"]), final_cost=max(1e-2 if parted else 1e-9, 10.0 * sp["final_cost"]),
                rms=max(1e-1 if parted else 1e-4, 10.0 * sp["rms"]), same_trajectory=sp["same_trajectory"],
                # iterates whose costs have to agree to 1e-9: all of them, or — when the oracle's own runs part — the ones up to
                # three iterations before they do
                agree_until=None if sp["part"] < 0 else max(3, min(sp["part"] - 3, sp["radius_iter"] if sp.get("radius_iter", -1) >= 0 else sp["part"])))
