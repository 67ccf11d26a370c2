"""Summary of a block timeline of the Schur kernel (one LM step).
  RSBA_TRACE=2 RSBA_TRACE_FILE=gpurun_out/wgtrace.txt python bench.py --no-cpu-baseline --steps 8 --warmup 2
  python tools/schur_timeline_summary.py gpurun_out/wgtrace.txt > profiles/r01_schur_block_timeline.txt
Columns of the trace: list position, segment, tile, type (0 pair, 1 self, 2 / 3 reducer of a pair / self tile), stage, words,
start, last partial sum stored, end (us after the first block)."""
import sys
import numpy as np

d = np.loadtxt(sys.argv[1])
st, ce, en = d[:, 6], d[:, 7], d[:, 8]
print("entries %d, span %.1f us, slot time %.0f workgroup-us" % (len(d), en.max() - st.min(), (en - st).sum()))
names = {0: "pair", 1: "self", 2: "reducer (pair tile)", 3: "reducer (self tile)"}
for ty in range(4):
    m = d[:, 3] == ty
    if m.any():
        print("%-20s n %4d  duration mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f   until the sums are stored: mean %.1f" % (
            names[ty], m.sum(), (en - st)[m].mean(), *np.percentile((en - st)[m], [10, 50, 90, 100]), (ce - st)[m].mean()))
print("pair entries by tile (diagonal tiles hold 120 pairs in both halves of the workgroup):")
for t in np.unique(d[d[:, 3] == 0, 2]):
    m = (d[:, 3] == 0) & (d[:, 2] == t)
    print("  tile %2d stage %d  n %3d  duration mean %.1f" % (t, d[m, 4][0], m.sum(), (en - st)[m].mean()))
print("stage: first start .. last start | last end")
for s in np.unique(d[:, 4]):
    m = d[:, 4] == s
    print("  %d: %.1f .. %.1f | %.1f" % (s, st[m].min(), st[m].max(), en[m].max()))
# the tail of a stage: from the moment its last compute entry has stored its partial sums (group sums, reducers, tile
# finishers and the stage's publication are what follows) to the end of its last entry (= the publication)
print("stage tails (last compute entry's sums stored -> last entry of the stage ends):")
for s in np.unique(d[:, 4]):
    m = d[:, 4] == s
    comp = m & (d[:, 3] < 2)
    print("  %d: sums stored %.1f -> published %.1f  = %.1f us" % (s, ce[comp].max(), en[m].max(), en[m].max() - ce[comp].max()))
# tile by tile: the last compute entry's sums stored -> the tile's reducers have the last group added (the latest of them) -> the
# tile's last entry ends (its blocks of S are written, it has arrived at the stage counter)
print("tiles: stage, type | last sums stored -> reducers' last group added -> last entry ends")
for t in np.unique(d[:, 2]):
    m = d[:, 2] == t
    comp, red = m & (d[:, 3] < 2), m & (d[:, 3] >= 2)
    if red.any():
        print("  tile %2d stage %d %-4s | %.1f -> %.1f (+%.1f) -> %.1f (+%.1f)" % (t, d[m, 4][0], "self" if (d[m, 3] == 1).any() else "pair", ce[comp].max(),
              ce[red].max(), ce[red].max() - ce[comp].max(), en[red].max(), en[red].max() - ce[red].max()))
print("entries running at t (us):", " ".join("%d:%d" % (t, ((st <= t) & (en > t)).sum()) for t in range(20, int(en.max()), 40)))
order_s, order_e = np.sort(st), np.sort(en)
nslots = int(((st <= 20) & (en > 20)).sum())
gap = order_s[nslots:] - order_e[:len(order_s) - nslots]
print("slots in use %d; end of an entry -> start of the next on the freed slot: mean %.1f us, p10 %.1f, p90 %.1f" % (
    nslots, gap.mean(), *np.percentile(gap, [10, 90])))
