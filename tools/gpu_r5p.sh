#!/bin/bash
# round 5, GPU run P: kernel trace of a lone 2^20 proof (latency path).
R=$PWD; O=$R/gpurun_out/r5p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARK355_BENCH_WATCHDOG=250 timeout 280 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof -o lone -- python $R/bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e --inflight 1 --steps 6 --warmup 3 > $O/lone.json 2> $O/lone.log
echo "rc=$?" >> $O/status.txt
exit 0
