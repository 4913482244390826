#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_streams.log 2>&1
ARK355_SERIAL=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_serial.log 2>&1
ARK355_G2_INLINE=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_g2ni.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $R/gpurun_out/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $R/gpurun_out/rocprof_write.log 2>&1
cd $R
python tools/pmc_summary.py > gpurun_out/pmc_summary.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out -name "*counter_collection.csv" -size +20M -delete
exit 0
