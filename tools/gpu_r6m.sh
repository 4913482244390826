#!/bin/bash
# round 6, GPU run M (one box): where the fixed cost of 2^19 buckets per MSM goes -- serial kernel trace of 2^20 proofs at c = 17 and c = 20.
R=$PWD; O=$R/gpurun_out/r6m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 17 20; do
  ARK355_MSM_C=$c ARK355_SERIAL=1 ARK355_SIDE_WM=0 ARK355_SIDE_G2_TAILS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_c$c -o serial -- python $R/bench.py --profile-run --inflight 1 --steps 5 --warmup 2 > $O/serial_c$c.log 2>&1; echo "serial c=$c rc=$?" >> $O/status.txt
  find $O/serial_c$c -name "*kernel_trace.csv" -delete; find $O/serial_c$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/serial_c${c}_kernel_stats.csv
done
cat $O/status.txt
exit 0
