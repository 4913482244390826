#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_g.log 2>&1
ARK355_MSM_C=15 timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench20_c15.log 2>&1
ARK355_MSM_C=20 timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench20_c20.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --log-n 18 > gpurun_out/bench18.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --tight > gpurun_out/bench20_tight.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1
cd $R
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
exit 0
