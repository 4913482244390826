#!/bin/bash
# full evidence run: parity tests, smoke, default bench (with cpu_baseline), rocprof stats + PMC passes -> profiles
mkdir -p gpurun_out
rocm-smi --showmaxpower --showperflevel --showclocks > gpurun_out/smi_caps.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 > gpurun_out/bench_inflight1.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --curve bn254 > gpurun_out/bench_bn254.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --tight > gpurun_out/bench_tight.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --log-n 18 > gpurun_out/bench_n18.log 2>&1
timeout 300 tools/ubench.bin > gpurun_out/ubench.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > $R/gpurun_out/rocprof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $R/gpurun_out/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $R/gpurun_out/rocprof_write.log 2>&1
ARK355_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/serial -o serial -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > $R/gpurun_out/serial_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -o t -- python $R/bench.py --no-cpu-baseline --steps 12 --warmup 4 > $R/gpurun_out/trace_bench.log 2>&1
cd $R
python tools/trace_analyze.py gpurun_out/trace/t_kernel_trace.csv 12 > gpurun_out/timeline_summary.txt 2>&1
python tools/pmc_summary.py --json gpurun_out/pmc_latest.json --workload "bls12_381:n=1048576" > gpurun_out/pmc_summary.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out -name "*counter_collection.csv" -size +20M -delete
exit 0
