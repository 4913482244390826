#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/ubench.bin > gpurun_out/ubench_dual.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_dual.log 2>&1
ARK355_MSM_C=18 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_c18.log 2>&1
ARK355_MSM_C=20 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_c20.log 2>&1
ARK355_MSM_C=14 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20_c14.log 2>&1
exit 0
