#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (dev tool; run on the GPU box)."""
import collections
import csv
import glob
import re

for kind in ("fetch", "write"):
    for f in glob.glob("gpurun_out/prof_%s/**/*counter_collection.csv" % kind, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))
            k = k.replace("ark355::", "")[:70]
            agg[(k, row.get("Counter_Name"))][0] += 1
            agg[(k, row.get("Counter_Name"))][1] += float(row.get("Counter_Value", 0))
        for (k, c), (n, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:30]:
            print("%s %-70s dispatches=%d sum_KiB=%.1f avg_KiB=%.1f" % (c, k, n, v, v / n))
