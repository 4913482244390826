#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 > gpurun_out/bench20_g2l.log 2>&1
ARK355_G2_WHOLE=1 timeout 600 python bench.py --no-cpu-baseline --inflight 1 > gpurun_out/bench20_g2whole.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 --tight > gpurun_out/bench20_tight.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 --log-n 18 > gpurun_out/bench18.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --curve bn254 --inflight 1 > gpurun_out/bench_bn254.log 2>&1
exit 0
