#!/bin/bash
mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showperflevel > gpurun_out/smi_before.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > gpurun_out/bench20_a.log 2>&1
rocm-smi --showclocks --showpower > gpurun_out/smi_mid.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 12 --warmup 3 > gpurun_out/bench20_b.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 3 --steps 12 --warmup 3 > gpurun_out/bench20_c.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > gpurun_out/bench20_d.log 2>&1
rocm-smi --showclocks --showpower > gpurun_out/smi_after.log 2>&1
exit 0
