"""Child process of tests/test_gpu_switches.py: solves a few problems through the C ABI with whatever RSBA_* switches the
environment carries (they are read once per process) and prints, as one JSON line, how each solve compares with the oracle.
Test infrastructure: the oracle is the checker."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle_lib  # noqa: E402
from oracle_spread import block_rel  # noqa: E402
from realsensecalibration_amd import capi, synthetic as syn  # noqa: E402

CASES = {"c40": (40, 3000, 9, 340, 0.0), "c64_huber": (64, 4000, 12, 364, 1.0), "c70_huber": (70, 1500, 10, 5, 1.0), "c130": (130, 1500, 12, 430, 0.0),
         "c240": (240, 1200, 14, 77, 0.0),
         # dense visibility (every camera sees every point: a lane of a pair tile has 512 hits per chunk, the listed pair segments
         # cut their blocks down to 64 points) and a problem of several chunks per segment
         "c8_dense": (8, 6000, 8, 11, 0.0), "c33_long": (33, 60000, 6, 12, 0.0), "c64_dense_huber": (64, 2500, 40, 13, 1.0),
         "c64_long": (64, 20000, 10, 7, 0.0), "c130_long": (130, 6000, 12, 8, 0.0)}   # 240 cameras: 276 tiles of the reduced system on 256 CUs (the tiles' placement, TileOrder, matters)


def dump(path, names):
    """--dump FILE case ...: the raw results of the cases' solves (no oracle), for bit-for-bit comparisons between two settings."""
    res = {}
    for name in names:
        C, P, k, seed, huber = CASES[name]
        prob = syn.make_problem(C, P, k, seed=seed, outlier_frac=0.05 if huber else 0.0)
        kw = {}
        if os.environ.get("SWITCH_FORCED_STEPS"):   # a long run of forced LM steps (tolerances off), as tools/stress_pipeline.py does
            kw = dict(max_num_iterations=int(os.environ["SWITCH_FORCED_STEPS"]), function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0,
                      max_num_consecutive_invalid_steps=1 << 30, min_trust_region_radius=0.0)
        got, s, log = capi.solve_points(prob, capi.default_options(schur_impl=1, huber_delta=huber, **kw))
        res[name] = got
        res[name + "_log"] = log
    np.savez(path, **res)


def main():
    if sys.argv[1] == "--dump":
        return dump(sys.argv[2], sys.argv[3:])
    oracle = oracle_lib.load()
    out = {}
    for name in sys.argv[1:]:
        C, P, k, seed, huber = CASES[name]
        prob = syn.make_problem(C, P, k, seed=seed, outlier_frac=0.05 if huber else 0.0)
        ref, s_ref, log_ref = oracle.solve_points(prob, oracle.options(huber_delta=huber))
        got, s, log = capi.solve_points(prob, capi.default_options(schur_impl=1, huber_delta=huber))
        again, s2, _ = capi.solve_points(prob, capi.default_options(schur_impl=1, huber_delta=huber))
        out[name] = dict(iterations=int(s.num_iterations), iterations_ref=int(s_ref.num_iterations),
                         same_decisions=bool(len(log) == len(log_ref) and np.array_equal(log[:, 7], log_ref[:, 7])),
                         stop=int(s.stop_reason), stop_ref=int(s_ref.stop_reason), raw=float(block_rel(got, ref, C)),
                         final_cost_rel=float(abs(s.final_cost - s_ref.final_cost) / s_ref.final_cost),
                         iterate_costs=float(np.abs(log[:, 1] - log_ref[:, 1]).max() / log_ref[:, 1].max()) if len(log) == len(log_ref) else 1.0,
                         reproducible=bool(np.array_equal(got, again)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
