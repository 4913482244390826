#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_f.log 2>&1
for c in 15 17 18 19 20; do ARK355_MSM_C=$c timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench20_c$c.log 2>&1; done
R=$PWD
cd /tmp && export TMPDIR=/tmp
ARK355_MSM_C=20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c20 -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_c20.log 2>&1
cd $R
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
exit 0
