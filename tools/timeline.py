"""Print the start/end of every kernel of the last LM iteration in a rocprofv3 kernel trace (us, relative)."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rsba::", "")[:40]))
rows.sort()
# last occurrence of k_point_pass starts the last iteration
idx = [i for i, r in enumerate(rows) if "k_point_pass" in r[2]]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
i1 = idx[-1]
t0 = rows[i0][0]
for s, e, n in rows[i0:i1 + 1]:
    print("%9.1f %9.1f  %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
