#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (dev tool; run on the GPU box).

usage: pmc_summary.py [--json OUT.json] [--workload TEXT] [--recorded TEXT]
Reads <dir>/prof_fetch and <dir>/prof_write (two separate passes: FETCH_SIZE, WRITE_SIZE; --dir, default gpurun_out).
With --merge the record is added to the "workloads" map of an existing JSON (one record per workload; bench.py's pmc_traffic).
Counter unit: KiB.  With --json also writes the per-launch figures of the dominant kernel
(msm_accumulate_kernel) that bench.py reports as `roofline.traffic`."""
import collections
import csv
import glob
import json
import re
import sys

BASE = sys.argv[sys.argv.index("--dir") + 1] if "--dir" in sys.argv else "gpurun_out"
agg = collections.defaultdict(lambda: [0, 0.0])
for kind in ("fetch", "write"):
    for f in glob.glob("%s/prof_%s/**/*counter_collection.csv" % (BASE, kind), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))
            k = k.replace("ark355::", "")[:70]
            agg[(k, row.get("Counter_Name"))][0] += 1
            agg[(k, row.get("Counter_Name"))][1] += float(row.get("Counter_Value", 0))
for (k, c), (n, v) in sorted(agg.items(), key=lambda x: (x[0][1], -x[1][1])):
    print("%s %-70s dispatches=%d sum_KiB=%.1f avg_KiB=%.1f" % (c, k, n, v, v / n))

if "--json" in sys.argv:
    out = sys.argv[sys.argv.index("--json") + 1]
    wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else ""
    recorded = sys.argv[sys.argv.index("--recorded") + 1] if "--recorded" in sys.argv else ""
    rec = {"workload": wl, "recorded": recorded, "unit": "KiB", "kernels": {}}
    for (k, c), (n, v) in agg.items():
        rec["kernels"].setdefault(k, {})[c] = {"dispatches": n, "sum_kib": v}
    acc = {c: [0, 0.0] for c in ("FETCH_SIZE", "WRITE_SIZE")}
    for (k, c), (n, v) in agg.items():
        if "msm_accumulate" in k and c in acc:
            acc[c][0] += n
            acc[c][1] += v
    rec["msm_accumulate_kernel"] = {
        "launches": acc["FETCH_SIZE"][0],
        "fetch_bytes_per_launch": acc["FETCH_SIZE"][1] * 1024 / max(1, acc["FETCH_SIZE"][0]),
        "write_bytes_per_launch": acc["WRITE_SIZE"][1] * 1024 / max(1, acc["WRITE_SIZE"][0]),
        "note": "FETCH_SIZE/WRITE_SIZE as counted (KiB->B); random row gathers (BLS12-381: 128-B G1 rows / 2x128-B G2 halves, one word per limb; BN254: 64-B packed G1 rows / 2x64-B halves): no gfx950 "
                "doubling applied (calibration in DESIGN.md section 3)",
    }
    if "--merge" in sys.argv:
        import os
        top = json.load(open(out)) if os.path.exists(out) else {}
        if "workloads" not in top:
            top = {"workloads": ({top["workload"]: top} if top.get("workload") else {})}
        rec.pop("kernels", None)          # the per-kernel table of every workload would be megabytes; the accumulation rows are what bench.py reads
        top["workloads"][wl] = rec
        json.dump(top, open(out, "w"), indent=1)
    else:
        json.dump(rec, open(out, "w"), indent=1)
