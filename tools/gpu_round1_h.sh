#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 > gpurun_out/bench20_if1.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 2 --steps 10 > gpurun_out/bench20_if2.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 3 --steps 12 > gpurun_out/bench20_if3.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_default.log 2>&1
exit 0
