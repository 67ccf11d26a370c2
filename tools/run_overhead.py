"""Where a run's fixed time goes: per-iteration host times of the log, minimizer time, wall time of run()."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes
from realsensecalibration_amd import capi, synthetic as syn
prob = syn.make_config("cfg3")
problem = capi.Problem.points(prob)
fixed = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0, max_num_consecutive_invalid_steps=1 << 30, min_trust_region_radius=0.0)
sv = capi.Solver(problem, capi.default_options(max_num_iterations=3, **fixed))
sv.run()
for K in (25, 50):
    sv.configure_run(K, 0)
    t0 = time.perf_counter(); s = sv.run(); wall = time.perf_counter() - t0
    arr = (capi.Iteration * 256)()
    n = capi.load().rsba_solver_iterations(sv.h, arr, 256)
    its = [arr[i].iteration_time_in_seconds * 1e6 for i in range(n)]
    print("K=%d wall %.1f us, minimizer %.1f us, rows %d" % (K, wall * 1e6, s.minimizer_seconds * 1e6 if hasattr(s, 'minimizer_seconds') else -1, n))
    print("  first rows (us):", " ".join("%.0f" % v for v in its[:8]), "... last:", " ".join("%.0f" % v for v in its[-4:]), " sum %.0f" % sum(its))
sv.close()
