"""Stress the pipelined schedule: many solver lifetimes (stream / event / CU-mask churn) and a long run of LM steps.
Prints iterations per second per phase; any stall prints the library's fallback message on stderr.
usage: python tools/stress_pipeline.py [cameras] [shards]   — shards > 1: the multi-GPU schedule on a loopback group of that many ranks
(the point shards of the same problem; every lifetime and the long run must end bit-identical on all ranks and equal to the first)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realsensecalibration_amd import capi, synthetic as syn

# cameras from the command line: below 32 runs the one-workgroup Cholesky, 32 .. 64 the four-workgroup one
C = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SHARDS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fixed = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0, max_num_consecutive_invalid_steps=1 << 30,
             min_trust_region_radius=0.0)
if SHARDS > 1:
    from realsensecalibration_amd import distributed as rd
    shards = [syn.make_problem(C, 20000, 10, 7, point_range=rd.shard_range(20000, r, SHARDS)) for r in range(SHARDS)]
    ref = None
    t0 = time.time()
    for rep in range(20):
        ranks = capi.solve_points_sharded_loopback(shards)
        cams = [p[:6 * C].copy() for p, _, _, _ in ranks]
        assert all(np.array_equal(cams[0], c) for c in cams[1:]), ("ranks differ", rep)
        key = (ranks[0][1].num_iterations, ranks[0][1].final_cost)
        if ref is None:
            ref = key
        assert key == ref, (rep, key, ref)
    print("%d cameras, %d loopback ranks:" % (C, SHARDS), "20 group lifetimes ok, %.2f s, iterations %d final cost %.6f" % (time.time() - t0, ref[0], ref[1]))
    t0 = time.time()
    ranks = capi.solve_points_sharded_loopback(shards, dict(max_num_iterations=600, **fixed))
    cams = [p[:6 * C].copy() for p, _, _, _ in ranks]
    assert all(np.array_equal(cams[0], c) for c in cams[1:]), "ranks differ after the long run"
    print("600 forced steps on %d ranks: %d done, %.1f s, final cost %.6f" % (SHARDS, ranks[0][1].num_iterations, time.time() - t0, ranks[0][1].final_cost))
    sys.exit(0)
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
sv = capi.Solver(problem, capi.default_options(max_num_iterations=3000, **fixed))
t0 = time.time()
s = sv.run()
dt = time.time() - t0
sv.close()
print("3000 forced steps: %d done, %.1f steps/s, final cost %.6f" % (s.num_iterations, s.num_iterations / dt, s.final_cost))
