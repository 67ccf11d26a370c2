"""The randomised parity sweep, in the suite: 160 seeded random shapes of the point model through the C ABI against the oracle.

Cases, criterion and its rationale: tests/fuzz_cases.py, tests/oracle_spread.py.  Every block is free, as the reference
leaves them (Test1_BundleAdjustment/main.cpp:76-79; functor bundle_adjustmenter.cpp:106-148).  A case on which three
executions of the oracle agree with each other to a tenth of BASELINE.json's bars — seven in eight — is held to exactly
those bars (raw parameters 1e-6 per block, final cost 1e-9, RMS 1e-4 px, same trajectory, every iterate's cost 1e-9); a case
on which the oracle's own executions PART is compared where they still agree: its iterates' costs up to the last common iterate
(THE MARGIN RULE), and — round 5 — both solvers are run again with the iteration limit there and parameters, cost and RMS compared
at BASELINE's bars at that state (rounds 3-4 held the chaotic end state to flat 1 % / 0.1 px bars instead); the first three
iterates' costs to 1e-12 (never above 1e-10) in every case.  tools/fuzz_parity.py runs the same sweep with other seeds.
"""
import numpy as np
import pytest

import fuzz_cases
from realsensecalibration_amd import capi

pytestmark = pytest.mark.gpu
CASES = fuzz_cases.cases(160, 1)
SEEN = {}


@pytest.fixture(scope="module", autouse=True)
def _lib():
    lib = capi.load()
    assert lib.rsba_device_count() > 0, "GPU tests need a HIP device; the product has no CPU path"
    return lib


@pytest.mark.parametrize("case", CASES, ids=fuzz_cases.label)
def test_random_shape_matches_oracle(oracle, case):
    r = fuzz_cases.run(oracle, capi, case)
    SEEN[case["index"]] = r
    bad = fuzz_cases.verdict(r)
    assert not bad, "; ".join(bad) + " | raw %.1e, oracle against itself %.1e" % (r["raw"], r["spread"]["raw"])


@pytest.mark.parametrize("case", fuzz_cases.pinned(), ids=lambda c: "seed%d-%s" % (c["sweep"], fuzz_cases.label(c)))
def test_pinned_sensitive_case_passes_the_margin_rule(oracle, case):
    """Sensitive cases of other sweeps, by explicit seed (fuzz_cases.pinned): judged by the same criterion, whose margin is a
    rule (oracle_spread.py, THE MARGIN RULE) — the suite decides, not a note in DESIGN.md."""
    r = fuzz_cases.run(oracle, capi, case)
    assert fuzz_cases.sensitive(r), "the case is pinned because the oracle parts from itself on it"
    bad = fuzz_cases.verdict(r)
    assert not bad, "; ".join(bad) + " | raw %.1e, oracle against itself %.1e" % (r["raw"], r["spread"]["raw"])
    assert r["first3"] <= r["bars"]["first3"]


def test_sweep_is_mostly_held_to_the_baseline_bars():
    """The looser bars must stay the exception, and the strict class must really have been strict."""
    if len(SEEN) < len(CASES):
        pytest.skip("needs the whole sweep (run the module, not a selection)")
    sens = [i for i, r in SEEN.items() if fuzz_cases.sensitive(r)]
    strict = [r for i, r in SEEN.items() if i not in sens]
    assert len(sens) <= len(CASES) // 5, sens
    assert all(r["raw"] < 1e-6 and r["final_cost"] <= max(1e-9, r["final_cost_tol"]) and r["rms"] <= 1e-4 and r["same_trajectory"] and r["part"] < 0 for r in strict)
    # the first-three-iterates bar is the flat 1e-12 in all but a handful of cases (at most one in twenty), never above 1e-10
    relaxed = [i for i, r in SEEN.items() if r["bars"]["first3"] > 1e-12]
    assert len(relaxed) <= len(CASES) // 20, relaxed
    assert all(r["bars"]["first3"] <= 1e-10 for r in SEEN.values())
    # ... and THE REFEREE RULE (oracle_spread.py) decides at most one case in fifty
    refereed = [i for i, r in SEEN.items() if r.get("referee") is not None]
    assert len(refereed) <= max(1, len(CASES) // 50), refereed
    # no sensitive case without a robust loss and none with more than six views per point: the phenomenon is the one described
    for i in sens:
        c = CASES[i]
        assert c["loss"] != "none" or c["k"] <= 2, fuzz_cases.label(c)
        assert c["k"] <= 6, fuzz_cases.label(c)
    print("%d of %d cases on which the oracle parts from itself; worst raw difference of the others %.1e" % (
        len(sens), len(CASES), max(r["raw"] for r in strict)))
