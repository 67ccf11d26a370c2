"""Per-panel chain of the diagonal-workgroup Cholesky (ba_cholesky_diag.hpp) from an RSBA_MC_TRACE=1 log (stderr of a run).
Stamps of workgroup 0, us: 0 panel start, 1 factorisation done (wave 0), 2 update waves have issued their loads, 5 their small
loads are there, 6 X(p+1, p-1) formed (first meeting), 3 wave 1's unit done, 4 barrier [A], 7 end of the panel.  Row workgroups:
0 start, 2 strip_ready seen, 4 update done (block handed over right after), 5 T(p) loaded, 6 solved, 7 end.
usage: python tools/mc_chain.py log.txt"""
import re
import sys

rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r'rsba\[mc\] wg (\d) panel\s+(\d+):(.*)', ln)
    if m:
        rows[(int(m.group(1)), int(m.group(2)))] = [float(x) for x in m.group(3).split()]
G = 1 + max(w for w, _ in rows)
NP = 1 + max(p for w, p in rows if w == 0)
prev = None
for p in range(NP):
    r = rows[(0, p)]
    s = "p%2d start %7.2f | factor +%.1f | loads issued +%.1f data +%.1f X +%.1f unit +%.1f | [A] +%.1f | end +%.1f" % (
        p, r[0], r[1] - r[0], r[2] - r[0] if r[2] > 0 else 0, r[5] - r[0] if r[5] > 0 else 0, r[6] - r[0] if r[6] > 0 else 0,
        r[3] - r[0] if r[3] > 0 else 0, r[4] - r[0], r[7] - r[0])
    if prev is not None:
        s += "  [period %.1f]" % (r[0] - prev)
    prev = r[0]
    print(s)
for w in range(1, G):
    print("row workgroup %d: panel start -> flag / update done / T / solved / end" % w)
    for p in range(NP):
        r = rows.get((w, p))
        if r and r[4] > 0:
            print("  p%2d %7.2f  +%.1f +%.1f %s %s +%.1f" % (p, r[0], r[2] - r[0] if r[2] > 0 else 0, r[4] - r[0], ("+%.1f" % (r[5] - r[0])) if r[5] > 0 else "  - ", ("+%.1f" % (r[6] - r[0])) if r[6] > 0 else "  - ", r[7] - r[0] if r[7] > 0 else 0))
