#!/bin/bash
# isolated per-kernel durations: serial prover (ARK355_SERIAL=1), one proof in flight, rocprofv3 kernel trace
R=$PWD; mkdir -p $R/gpurun_out/serial
cd /tmp && export TMPDIR=/tmp
ARK355_SERIAL=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/serial -o serial -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > $R/gpurun_out/serial/bench.log 2>&1
cd $R; ls gpurun_out/serial | head; rm -f gpurun_out/serial/*kernel_trace.csv gpurun_out/serial/*/*kernel_trace.csv
exit 0
