#!/bin/bash
# round 5, GPU run M: kernel trace of the per-rank path of a sharded 2^22 proof (rank 0 of 8, distributed witness map).
R=$PWD; O=$R/gpurun_out/r5m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o shard -- python $R/tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm dist --steps 8 > $O/shard.json 2> $O/shard.log
echo "rc=$?" >> $O/status.txt
ls -R $O/prof | head -30 >> $O/status.txt
# keep only the stats csv (the trace itself is large)
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
exit 0
