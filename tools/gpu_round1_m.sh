#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/ubench.bin > gpurun_out/ubench_chunk.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > gpurun_out/bench20_a.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 3 > gpurun_out/bench20_c.log 2>&1
exit 0
