"""Per-panel chain of the multi-workgroup Cholesky from an RSBA_MC_TRACE=1 log (stderr of a run): owner's factorisation,
publication of T, and what the next owner was doing meanwhile.  usage: python tools/mc_chain.py log.txt"""
import re
import sys

rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r'rsba\[mc\] wg (\d) panel\s+(\d+):(.*)', ln)
    if m:
        rows[(int(m.group(1)), int(m.group(2)))] = [float(x) for x in m.group(3).split()]
G = 1 + max(w for w, _ in rows)
NP = 1 + max(p for _, p in rows)
prev = None
for p in range(NP):
    o = p % G
    r = rows[(o, p)]
    rn = rows.get(((p + 1) % G, p))
    s = "p%2d owner wg%d: factor %.1f->%.1f (%.1f) T pub %.1f" % (p, o, r[0], r[4], r[4] - r[0], r[5])
    if prev is not None:
        s += "  [period %.1f]" % (r[5] - prev)
    prev = r[5]
    if rn and rn[4] > 0:
        s += " | next owner: start %.1f strip %.1f/%.1f upd done %.1f T %.1f X %.1f tail %.1f" % (rn[0], rn[2], rn[3], rn[4], rn[5], rn[6], rn[7])
    print(s)
