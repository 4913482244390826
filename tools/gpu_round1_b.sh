#!/bin/bash
# second GPU contact: parity tests, bench (with C-oracle cpu_baseline), rocprof stats (CSV) + PMC passes
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1
timeout 900 python bench.py > gpurun_out/bench20.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $R/gpurun_out/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $R/gpurun_out/rocprof_write.log 2>&1
cd $R
ls -R gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/ls.log 2>&1
# keep the merged payload small: drop per-dispatch traces except the PMC ones we aggregate
python - <<'PY' > gpurun_out/pmc_summary.log 2>&1
import csv, glob, collections, re
for kind in ("fetch", "write"):
    for f in glob.glob("gpurun_out/prof_%s/**/*counter_collection.csv" % kind, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))[:80]
            agg[(k, row.get("Counter_Name"))][0] += 1
            agg[(k, row.get("Counter_Name"))][1] += float(row.get("Counter_Value", 0))
        for (k, c), (n, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:25]:
            print(kind, c, "%-80s" % k, "dispatches=%d" % n, "sum=%.1f" % v, "avg=%.1f" % (v / n))
PY
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out -name "*counter_collection.csv" -size +20M -delete
exit 0
