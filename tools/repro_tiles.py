"""Repeat solves above 64 cameras and count runs that differ from the first (a stall of a persistent kernel falls back to
another factorisation, which changes the rounding): python tools/repro_tiles.py [cameras ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realsensecalibration_amd import capi, synthetic as syn
for C in [int(a) for a in sys.argv[1:]] or [65, 100, 70]:
    prob = syn.make_problem(C, 20 * C, 9, seed=900 + C)
    ref = None; bad = 0
    for rep in range(20):
        got, s, log = capi.solve_points(prob)
        if ref is None: ref = got
        elif not np.array_equal(ref, got): bad += 1
    print(C, "cameras: runs that differ from the first:", bad, "of 19", flush=True)
