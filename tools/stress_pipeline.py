"""Stress the pipelined schedule: many solver lifetimes (stream / event / CU-mask churn) and a long run of LM steps.
Prints iterations per second per phase; any stall prints the library's fallback message on stderr."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realsensecalibration_amd import capi, synthetic as syn

# cameras from the command line: below 32 runs the one-workgroup Cholesky, 32 .. 64 the four-workgroup one
C = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prob = syn.make_problem(C, 20000, 10, seed=7)
problem = capi.Problem.points(prob)
ref = None
t0 = time.time()
for rep in range(60):
    sv = capi.Solver(problem, capi.default_options())
    s = sv.run()
    sv.close()
    if ref is None:
        ref = (s.num_iterations, s.final_cost)
    assert (s.num_iterations, s.final_cost) == ref, (rep, s.num_iterations, s.final_cost, ref)
print("%d cameras:" % C, "60 solver lifetimes ok, %.2f s, iterations %d final cost %.6f" % (time.time() - t0, ref[0], ref[1]))
fixed = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0, max_num_consecutive_invalid_steps=1 << 30,
             min_trust_region_radius=0.0)
sv = capi.Solver(problem, capi.default_options(max_num_iterations=3000, **fixed))
t0 = time.time()
s = sv.run()
dt = time.time() - t0
sv.close()
print("3000 forced steps: %d done, %.1f steps/s, final cost %.6f" % (s.num_iterations, s.num_iterations / dt, s.final_cost))
