#!/usr/bin/env python3
"""Average PMC counters per kernel from rocprofv3 --pmc CSV output directories."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if "rocclr" in k:
                continue
            print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
