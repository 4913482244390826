#!/bin/bash
# round 2, run S: where a stand-alone 2^24-term G1 MSM spends the 52 ms that are not bucket accumulation
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2s_prof -o s -- python $R/tools/msm_bench.py --min-log 24 --max-log 24 --reps 3 --groups 1 --dists uniform > $O/r2s_msm.txt 2>&1
grep "n=2" $O/r2s_msm.txt
f=$(find $O/r2s_prof -name "*kernel_stats.csv" | head -n 1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:40]:
    n=r["Name"]
    if any(k in n for k in ("precomp","fixed_base","table_to28","batch_to_affine")): continue
    print("%-64s calls %4s  total %9.3f ms  avg %9.3f ms" % (n.split("(")[0].replace("ark355::","")[:64], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
find $O/r2s_prof -name "*kernel_trace.csv" -delete
exit 0
