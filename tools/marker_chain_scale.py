"""Marker-chain model at scale on the GPU: per-kernel times of the time-elimination path.

    python tools/marker_chain_scale.py [C T M]        default 8 5000 16
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from realsensecalibration_amd import capi, synthetic  # noqa: E402

C_, T_, M_ = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 5000, 16)
t0 = time.time()
prob = synthetic.make_marker_chain(C_, T_, M_, seed=11)
gen = time.time() - t0
p = capi.Problem.marker_chain(prob)
out = {}
for mode in (0, 1):
    s = capi.Solver(p, capi.default_options(profile_kernels=mode))
    s.run()            # warm-up: code objects, allocations
    sm = s.run()
    if mode == 0:
        out.update({"cameras": C_, "times": T_, "markers": M_, "residual_blocks": int(prob["N"]), "generate_s": round(gen, 2),
                    "iterations": sm.num_iterations, "initial_cost": sm.initial_cost, "final_cost": sm.final_cost,
                    "minimizer_ms": 1e3 * sm.minimizer_seconds,
                    "ms_per_iteration": 1e3 * sm.minimizer_seconds / max(1, sm.num_iterations + 1)})
    else:
        out["kernels_us"] = {k: round(1e3 * ms / n, 1) for k, (n, ms) in s.kernel_stats().items()}
    s.download()
    s.close()
err = np.abs(p.params - prob["truth"]).reshape(-1, 6)
out["max_abs_error_vs_truth"] = float(err.max())
print(json.dumps(out))
