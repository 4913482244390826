#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --inflight 1 > gpurun_out/bench20_seg32.log 2>&1
ARK355_LIB=$PWD/variants/libark355_seg64.so timeout 600 python bench.py --no-cpu-baseline --inflight 1 > gpurun_out/bench20_seg64.log 2>&1
ARK355_LIB=$PWD/variants/libark355_seg128.so timeout 600 python bench.py --no-cpu-baseline --inflight 1 > gpurun_out/bench20_seg128.log 2>&1
ARK355_LIB=$PWD/variants/libark355_seg64.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or prove" > gpurun_out/pytest_seg64.log 2>&1
exit 0
