#!/bin/bash
# round 5, GPU run D: hardware counters of ONE stand-alone 2^20-term G1 MSM (tools/msm_bench.py), the accumulation kernel's
# rows: wave / issue / wait cycles and the instruction cache, separate --pmc passes (no tracing options beside them).
R=$PWD; O=$R/gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
CMD="python $R/tools/msm_bench.py --min-log 20 --max-log 20 --groups 1 --dists uniform --reps 2 --no-check"
pass() {  # tag counters...
  tag=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -o p -- $CMD > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?" >> $O/status.txt
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH
IC=$(grep -o -E "SQC_ICACHE_[A-Z_]+" $O/counters_avail.txt | sort -u | head -6 | tr '\n' ' ')
echo "icache counters: $IC" >> $O/status.txt
if [ -n "$IC" ]; then pass ic $IC; fi
python - <<PY > $O/pmc_summary.txt 2>&1
import csv, glob, collections
for tag in ("sq1", "sq2", "ic"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "msm_accumulate28" not in k:
                continue
            agg[row.get("Counter_Name")][0] += 1
            agg[row.get("Counter_Name")][1] += float(row.get("Counter_Value", 0))
    for c, (n, v) in sorted(agg.items()):
        print("%-4s %-34s dispatches=%d  per dispatch=%.4g" % (tag, c, n, v / max(1, n)))
PY
rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_ic
exit 0
