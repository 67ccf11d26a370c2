"""Randomised parity sweep of the HIP path against the oracle, beyond the seeds the suite pins (tests/test_gpu_fuzz.py runs
the same cases function with fixed seeds; criterion and case generator: tests/fuzz_cases.py).  Test infrastructure.
usage: python tools/fuzz_parity.py [cases] [seed]   — one line per case, exits non-zero when a case violates the criterion."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases, oracle_lib
from realsensecalibration_amd import capi

oracle = oracle_lib.load()
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nbad = nsens = nref = 0
for c in fuzz_cases.cases(ncase, seed):
    r = fuzz_cases.run(oracle, capi, c)
    bad = fuzz_cases.verdict(r)
    nbad += 1 if bad else 0
    sens = fuzz_cases.sensitive(r)
    nsens += 1 if sens else 0
    sp = r["spread"]
    rf = r.get("referee")
    nref += 1 if rf else 0
    print("%-34s iters %2d radius max %.0e | raw %.1e final cost %.1e rms %.0e first three %.0e part at %d | oracle vs itself: raw %.1e cost %.1e part at %d %s | %s" % (
        fuzz_cases.label(c), r["iterations"], r["radius_max"], r["raw"], r["final_cost"], r["rms"], r["first3"], r["part"],
        sp["raw"], sp["final_cost"], sp["part"], ("SENSITIVE, at the last common iterate %d: raw %.1e cost %.1e rms %.0e" % (r["trunc"]["k"], r["trunc"]["raw"], r["trunc"]["final_cost"], r["trunc"]["rms"])) if r.get("trunc") else ("SENSITIVE" if sens else ""),
        ("ok" if not bad else "MISMATCH: " + "; ".join(bad)) + ((" | REFEREED: %.1e from the long-double solve, the oracle's executions %.1e" % (rf["d_impl"], rf["d_oracle"])) if rf else "")), flush=True)
print("mismatches:", nbad, "of", ncase, "; cases on which the oracle parts from itself:", nsens, "; first three iterates decided by the referee:", nref)
sys.exit(1 if nbad else 0)
