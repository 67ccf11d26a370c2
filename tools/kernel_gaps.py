#!/usr/bin/env python3
"""Step timeline from a rocprofv3 --kernel-trace CSV: where the time between two Schur kernels goes.

usage: kernel_gaps.py <dir of the --kernel-trace run>
For every LM step of the pipelined schedule (one k_schur_tiles launch each) prints the median, in microseconds, of
  previous back-substitution's end -> k_point_damp start -> its end -> k_schur_tiles start -> its end,
  the factorisation's end and the back-substitution's end relative to the Schur kernel's start, and the step period.
(The back-substitution is launched ahead and waits inside the kernel: its START is not meaningful, its end is.)"""
import csv
import glob
import statistics
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
schur = [i for i, r in enumerate(rows) if "k_schur_tiles" in r[2]]


def find(i0, i1, key):
    for i in range(i0, i1):
        if key in rows[i][2]:
            return rows[i]
    return None


rec = []
for a, b in zip(schur[:-1], schur[1:]):
    s0, s1 = rows[a], rows[b]
    # kernels launched between the two Schur kernels (the factorisation of step b precedes its Schur kernel)
    bs = find(a, b, "k_backsub_candidate")
    damp = find(a, b, "k_point_damp")
    solve = find(a - 3 if a >= 3 else 0, a + 1, "k_reduced_system_solve")
    if bs is None or solve is None:
        continue
    e = {"schur": (s0[1] - s0[0]) * 1e-3, "solve_end": (solve[1] - s0[0]) * 1e-3, "backsub_end": (bs[1] - s0[0]) * 1e-3,
         "period": (s1[0] - s0[0]) * 1e-3, "backsub_end_to_next_schur": (s1[0] - bs[1]) * 1e-3}
    if damp is not None:
        e["backsub_end_to_damp"] = (damp[0] - bs[1]) * 1e-3
        e["damp"] = (damp[1] - damp[0]) * 1e-3
        e["damp_end_to_schur"] = (s1[0] - damp[1]) * 1e-3
    rec.append(e)
print("steps", len(rec))
for k in ("period", "schur", "solve_end", "backsub_end", "backsub_end_to_next_schur", "backsub_end_to_damp", "damp", "damp_end_to_schur"):
    v = [e[k] for e in rec if k in e]
    if v:
        print("%-28s median %8.1f us   p10 %8.1f   p90 %8.1f   (n=%d)" % (k, statistics.median(v), sorted(v)[len(v) // 10], sorted(v)[(9 * len(v)) // 10], len(v)))
