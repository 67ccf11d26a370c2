"""Raw / gauge-aligned parameter differences of the HIP path against the oracle on the suite's long robust runs
(test infrastructure: uses tests/ helpers and oracle/).  usage: python tools/parity_probe.py  (env switches select kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, gauge
from realsensecalibration_amd import capi, synthetic as syn

def block_rel(a, b, C):
    worst = 0.0
    for x, y in ((a[:6 * C].reshape(-1, 6), b[:6 * C].reshape(-1, 6)), (a[6 * C:].reshape(-1, 3), b[6 * C:].reshape(-1, 3))):
        worst = max(worst, (np.abs(x - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-12)).max())
    return worst

oracle = oracle_lib.load()
cases = [(8, 1500, 6, 9, 0.05, 1.0)] + [(C, P, k, 300 + C, 0.05 if h else 0.0, h) for C, P, k, h in
         [(40, 2500, 30, 0.0), (40, 2500, 30, 1.5), (96, 2000, 24, 0.0), (96, 2000, 24, 1.5), (128, 1500, 14, 0.0)]]
for C, P, k, seed, outl, huber in cases:
    prob = syn.make_problem(C, P, k, seed=seed, outlier_frac=outl)
    ref, s_ref, log_ref = oracle.solve_points(prob, oracle.options(huber_delta=huber))
    for impl in (0, 1):
        got, s, log = capi.solve_points(prob, capi.default_options(schur_impl=impl, huber_delta=huber))
        m = min(len(log), len(log_ref))
        dc = np.abs(log[:m, 1] - log_ref[:m, 1]) / np.abs(log_ref[:m, 1])
        raw = block_rel(got, ref, C)
        al = block_rel(gauge.align(got, ref, C)[0], ref, C)
        print("C=%3d P=%4d k=%2d huber=%.1f impl=%d iters %2d/%2d raw %.2e aligned %.2e cost %.1e iter-cost max %.1e radius max %.1e" % (
            C, P, k, huber, impl, s.num_iterations, s_ref.num_iterations, raw, al, abs(s.final_cost - s_ref.final_cost) / s_ref.final_cost, dc.max(), log_ref[:, 6].max()), flush=True)
