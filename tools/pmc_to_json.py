#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/rNN_pmc_<workload>.json.

usage: pmc_to_json.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [workload label] [schedule the passes ran]
                      [dir of the SQ pass (SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU ...)] [dir of the GRBM pass (GRBM_GUI_ACTIVE ...)]
With the two SQ / GRBM directories (round 6) every kernel also carries the issue-side ratios the roofline line quotes beside `traffic`:
  lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)      (active lanes per issued vector instruction)
  valu_busy        = SQ_ACTIVE_INST_VALU / (32 GRBM_GUI_ACTIVE)            (= 4 cycles an instruction x instructions / (1024 SIMDs x the launch's
                                                                              cycles; GRBM_GUI_ACTIVE is summed over the eight XCDs))
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads, so reads are counted twice (8-byte accesses are uncalibrated).  bench.py reads
the result for the `traffic` field of its roofline objects."""
import collections
import csv
import glob
import json
import sys


def avg(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("rsba::", "")
                agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
label = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
schedule = sys.argv[5] if len(sys.argv) > 5 else "sequential schedule (counter collection serialises kernels)"
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
              "note": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced reads; 8-B accesses are uncalibrated) + "
                      "WRITE_SIZE, separate --pmc passes, %s, %s" % (label, schedule)}
if len(sys.argv) > 7:
    act, thr = avg(sys.argv[6], "SQ_ACTIVE_INST_VALU"), avg(sys.argv[6], "SQ_THREAD_CYCLES_VALU")
    gui, mfma = avg(sys.argv[7], "GRBM_GUI_ACTIVE"), avg(sys.argv[7], "SQ_INSTS_VALU_MFMA_MOPS_F64")
    for k in out:
        a, t, g = act.get(k), thr.get(k), gui.get(k)
        if not a or not g:
            continue
        out[k].update({"SQ_ACTIVE_INST_VALU": a, "SQ_THREAD_CYCLES_VALU": t, "GRBM_GUI_ACTIVE": g, "SQ_INSTS_VALU_MFMA_MOPS_F64": mfma.get(k, 0.0),
                       "lane_utilisation": (t / (64.0 * a)) if t else None, "valu_busy": a / (32.0 * g)})
out["__schedule__"] = schedule
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], "kernels:", ", ".join(out))
