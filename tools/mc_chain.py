"""Per-panel chain of the diagonal-workgroup Cholesky (ba_cholesky_diag.hpp) from an RSBA_MC_TRACE=1 log (stderr of a run).
Stamps of workgroup 0, us: 0 panel start, 1 factorisation done (wave 0), 2 update waves have issued their loads, 5 their small
loads are there, 6 X(p+1, p-1) formed (first meeting), 3 wave 1's unit done, 4 barrier [A], 7 end of the panel.  Row workgroups:
0 start, 2 strip_ready seen, 4 update done (block handed over right after), 5 T(p) loaded, 6 solved, 7 end.
The border's workgroup (ba_cholesky_border.hpp, line "rsba[mc] border:"): 1 + 4 g group g begins (its product with the groups before
is in registers), 2 + 4 g tile (g, B) seen, 3 + 4 g X_g solved, 4 + 4 g its sums / stores done; 14 y_A in LDS, 15 last tile seen, 16 L_C
factored, 17 x_C, 13 the back-substitution's right-hand side, 19 back-substituted, 18 end; 20 .. 25: inside the last group (T / L blocks
in LDS, panels 0 and 1 solved, first slice of X X', last slice) and the product formed after the group before it (24 -> 25); 26 C' formed, 27 .. 29
the three 32 x 32 factorisations of the border's block done.
usage: python tools/mc_chain.py log.txt"""
import re
import sys

rows = {}
border = None
for ln in open(sys.argv[1]):
    if ln.startswith("rsba[mc] border:"):
        border = {int(k): float(v) for k, v in (kv.split(":") for kv in ln.split()[2:])}
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

if border:
    B = max(g for g in range(3) if border.get(4 + 4 * g, -1) > 0) + 1   # (at most three leading groups: stamps 13 .. 19 are the end of the solve)
    print("border workgroup (%d leading groups):" % B)
    for g in range(B):
        print("  group %d: begins %7.2f | tile +%.1f | X solved +%.1f | sums, stores +%.1f" % (g, border[1 + 4 * g], border[2 + 4 * g] - border[1 + 4 * g], border[3 + 4 * g] - border[1 + 4 * g], border[4 + 4 * g] - border[1 + 4 * g]))
    print("  last tile seen %7.2f | factored +%.1f | x_C +%.1f | right-hand side of the back-substitution +%.1f | back-substituted +%.1f | end +%.1f" % (
        border[15], border[16] - border[15], border[17] - border[15], border[13] - border[15], border[19] - border[15], border[18] - border[15]))
