"""Summary of the point back-substitution's per-workgroup stamps (RSBA_TRACE=4, RSBA_TRACE_FILE): us after the first workgroup
saw the solve's flag — past the flag, tables staged, pass at x done, block sums out."""
import sys

import numpy as np

a = np.loadtxt(sys.argv[1])
names = ("past the flag", "tables staged", "pass at x done", "block sums out")
print("%d workgroups" % len(a))
for k, name in enumerate(names):
    v = a[:, 1 + k]
    print("%-16s min %5.1f  median %5.1f  p90 %5.1f  max %5.1f" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
half = len(a) // 2
for lo, hi, what in ((0, half, "first workgroup of a CU (dispatched first)"), (half, len(a), "second workgroup of a CU")):
    d = a[lo:hi, 3] - a[lo:hi, 2]
    r = a[lo:hi, 4] - a[lo:hi, 3]
    print("%-44s pass at x %5.1f us, candidate pass + sums %5.1f us, out at %5.1f" % (what, d.mean(), r.mean(), a[lo:hi, 4].max()))
print("(the two workgroups of a CU share its four fp64 pipes; the older wavefront is served first: the kernel ends when the")
print(" pipes have issued both workgroups' work, ~2 x 20 slots x 2 passes per SIMD)")
