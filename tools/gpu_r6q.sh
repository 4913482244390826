#!/bin/bash
# round 6, GPU run Q: hardware counters of stand-alone 2^20-term MSMs (tools/msm_bench.py), the rows of the two accumulation kernels:
# waves / cycles / VALU instructions / waits, and what the LDS accumulator of the G2 kernel costs (LDS instructions, bank conflicts,
# LDS waits) next to its memory instructions.  Separate --pmc passes, no tracing options beside them.
R=$PWD; O=$R/gpurun_out/r6q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
for grp in 1 2; do
CMD="python $R/tools/msm_bench.py --min-log 20 --max-log 20 --groups $grp --dists uniform --reps 2 --no-check"
pass() { tag=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_g${grp}_$tag -o p -- $CMD > $O/pmc_g${grp}_$tag.log 2>&1
  echo "pmc g$grp $tag rc=$?" >> $O/status.txt
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
python - <<PY > $O/pmc_summary.txt 2>&1
import csv, glob, collections
for grp in (1, 2):
    for tag in ("sq1", "sq2"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob("$O/pmc_g%d_%s/**/*counter_collection.csv" % (grp, tag), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "msm_accumulate" not in k:
                    continue
                agg[row.get("Counter_Name")][0] += 1
                agg[row.get("Counter_Name")][1] += float(row.get("Counter_Value", 0))
        for c, (n, v) in sorted(agg.items()):
            print("G%d %-4s %-28s dispatches=%d  per dispatch=%.5g" % (grp, tag, c, n, v / max(1, n)))
PY
rm -rf $O/pmc_g*_sq1 $O/pmc_g*_sq2
cat $O/pmc_summary.txt; cat $O/status.txt
exit 0
