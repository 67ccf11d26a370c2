"""Summarise a block timeline written with RSBA_TRACE=2 (see ba_solver.hip): per tile durations, stage ends, slot use."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1])
blk, seg, tile, self_, stage, words, t0, tc, t1 = a.T
print("blocks", len(a), "kernel span %.1f us" % (t1.max() - t0.min()))
for t in np.unique(tile):
    m = tile == t
    d = t1[m] - t0[m]
    c = tc[m] - t0[m]
    print("tile %2d self %d stage %d segs %4d words %5.1f | start %6.1f..%6.1f end %6.1f..%6.1f | dur mean %6.1f p50 %6.1f max %6.1f | compute mean %6.1f | us/word %5.2f" % (
        t, self_[m][0], stage[m][0], m.sum(), words[m].mean(), t0[m].min(), t0[m].max(), t1[m].min(), t1[m].max(), d.mean(), np.median(d), d.max(), c.mean(), (c / words[m]).mean()))
for g in np.unique(stage):
    m = stage == g
    print("stage %d: blocks %4d, last compute end %.1f, last end %.1f, slot-us %.0f" % (g, m.sum(), tc[m].max(), t1[m].max(), (t1[m] - t0[m]).sum()))
# occupancy over time
ts = np.arange(0, t1.max(), 20.0)
occ = [((t0 <= x) & (t1 > x)).sum() for x in ts]
print("resident blocks every 20 us:", occ)
