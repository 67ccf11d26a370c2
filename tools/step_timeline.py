#!/usr/bin/env python3
"""One LM step as rocprofv3 --kernel-trace saw it: every kernel between two k_schur_tiles launches, start / end in microseconds
after the Schur kernel's start (median step of the run).  usage: step_timeline.py <dir of the --kernel-trace run> [step index]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
schur = [i for i, r in enumerate(rows) if "k_schur_tiles" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(schur) // 2
a, b = schur[k], schur[k + 1]
t0 = rows[a][0]
print("step %d of %d, period %.1f us" % (k, len(schur), (rows[b][0] - t0) * 1e-3))
for s, e, n in rows[max(0, a - 4):b + 1]:
    print("%9.1f %9.1f  %8.1f  %s" % ((s - t0) * 1e-3, (e - t0) * 1e-3, (e - s) * 1e-3, n[:110]))
