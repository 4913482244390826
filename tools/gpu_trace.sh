#!/bin/bash
# kernel timeline of the in-flight bench (rocprofv3 --kernel-trace); the CSV is small enough to bring back
R=$PWD; rm -rf $R/gpurun_out/trace; mkdir -p $R/gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -o t -- python $R/bench.py --no-cpu-baseline --inflight ${1:-3} --steps 9 --warmup 3 > $R/gpurun_out/trace/bench.log 2>&1
cd $R; find gpurun_out/trace -name "*.csv" | head; du -sh gpurun_out/trace
exit 0
