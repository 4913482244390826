#!/bin/bash
# round 2, run K: per-kernel times of the batch-affine G2 MSM at 2^20 (rocprofv3 --kernel-trace --stats)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARK355_G2_BATCH_AFFINE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2k_prof -o ba -- python $R/tools/msm_bench.py --min-log 20 --max-log 20 --reps 3 --groups 2 --dists uniform > $O/r2k_msm.txt 2>&1
tail -n 3 $O/r2k_msm.txt
f=$(find $O/r2k_prof -name "*kernel_stats.csv" | head -n 1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows:
    n=r["Name"]
    if "ba_" in n or "accumulate" in n or "merge" in n or "reduce" in n or "scan" in n:
        print("%-60s calls %4s  total %9.3f ms  avg %9.3f ms" % (n.split("(")[0].replace("ark355::","")[:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
cp $f $O/r2k_kernel_stats.csv
exit 0
