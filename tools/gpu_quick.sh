#!/bin/bash
# quick loop: GPU parity tests + default bench (+ optional extra commands passed as arguments)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_quick.log 2>&1
for cmd in "$@"; do bash -c "$cmd"; done
exit 0
