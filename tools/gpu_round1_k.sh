#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
for k in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --inflight $k --steps 12 > gpurun_out/bench20_if$k.log 2>&1; done
exit 0
