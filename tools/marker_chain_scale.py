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
    for _ in range(int(os.environ.get("MC_RUNS", "3")) - 1):   # (the kernel statistics include the warm-up's first launches: amortised)
        s.run()
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
# roofline entry of the elimination of the time blocks: ALGORITHMIC work per LM iteration.  Per residual block (8 residuals, a camera,
# a time and a marker block of 6: J is 8 x 18) the normal-equation products J'J / J'r restricted to what the elimination needs —
# V += Jt'Jt (8 x 21 unique), g_t (8 x 6), W_c = Jt'Jc and W_m = Jt'Jm (8 x 36 each), U_cc, U_mm (8 x 21 each), U_cm (8 x 36),
# g_c, g_m (8 x 6 each): 1512 multiply-adds where all three blocks are parameters (camera 0 / marker 0 are not: those rows drop
# their terms); per time block with f touched camera / marker blocks the Schur update W'V^-1 W: (6f)^2/2 x 6 + the 6 x 6 solve.
# Bytes: a residual block's record (8 pixels + 3 indices: 76 B) once, every time block's parameters read and its step written (96 B).
t_arr, c_arr, m_arr = prob["t"], prob["c"], prob["m"]
fma = 0.0
for has_c, has_m in ((True, True), (True, False), (False, True), (False, False)):
    n = int(np.sum(((c_arr != 0) == has_c) & ((m_arr != 0) == has_m)))
    fma += n * 8 * (21 + 6 + (36 + 21 + 6 if has_c else 0) + (36 + 21 + 6 if has_m else 0) + (36 if has_c and has_m else 0))
key = t_arr.astype(np.int64) * 4096
fc = np.unique(key[c_arr != 0] + c_arr[c_arr != 0]); fm = np.unique(key[m_arr != 0] + 2048 + m_arr[m_arr != 0])
f_per_time = np.bincount((np.concatenate([fc, fm]) // 4096).astype(np.int64), minlength=T_).astype(float)
fma += float(np.sum((6 * f_per_time) ** 2 / 2 * 6 + 6 ** 3 / 3))
flops = 2.0 * fma
nbytes = 76.0 * prob["N"] + 96.0 * T_
# the elimination of the time blocks: round 6's split kernels (csrc/ba_marker_split.hpp; timed one after the other here — in a run
# that is not profiled the three product kernels share the chip on three streams), or round 4's k_time_eliminate (RSBA_MT_SPLIT=0)
SPLIT = ["k_mc_slot_products", "k_mc_time_products", "k_mc_cross", "k_mc_accumulate"]
ku = out.get("kernels_us", {})
parts = {k: ku[k] for k in SPLIT if k in ku} or ({"k_time_eliminate": ku["k_time_eliminate"]} if "k_time_eliminate" in ku else {})
if parts:
    us = sum(parts.values())
    # the split kernels form a residual block's Jacobian rows 2.7 times instead of once: NOT counted — algorithmic work only
    out["roofline"] = {"kernel": "+".join(parts), "kernels_us": parts, "avg_launch_us": round(us, 1), "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes,
                       "flops_per_residual_block": flops / prob["N"], "bytes_per_residual_block": nbytes / prob["N"],
                       "achieved_TFLOPs": flops / (us * 1e-6) / 1e12, "frac_of_fp64_peak_78.6": flops / (us * 1e-6) / 1e12 / 78.6,
                       "achieved_GBps": nbytes / (us * 1e-6) / 1e9, "frac_of_hbm_8000": nbytes / (us * 1e-6) / 1e9 / 8000.0,
                       "bound": ("fp64 VALU issue of threads that own their sums (one wavefront a SIMD at 334 registers), then two barriers a time in the accumulation" if len(parts) > 1 else
                                 "neither roof: a latency chain of five barriers per 32-block tile inside one workgroup per chunk of shots (HISTORY.md round 4)")}
err = np.abs(p.params - prob["truth"]).reshape(-1, 6)
out["max_abs_error_vs_truth"] = float(err.max())
print(json.dumps(out))
