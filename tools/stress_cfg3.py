import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realsensecalibration_amd import capi, synthetic as syn
prob = syn.make_problem(64, 100000, 20, seed=3)
problem = capi.Problem.points(prob)
fixed = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0, max_num_consecutive_invalid_steps=1 << 30, min_trust_region_radius=0.0)
sv = capi.Solver(problem, capi.default_options(max_num_iterations=20000, **fixed))
t0 = time.time(); s = sv.run(); dt = time.time() - t0; sv.close()
print("20000 forced steps at cfg3: %d done, %.1f steps/s, final cost %.6f" % (s.num_iterations, s.num_iterations / dt, s.final_cost))
