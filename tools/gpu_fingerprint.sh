#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/ubench.bin > gpurun_out/fp_ubench.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > gpurun_out/fp_bench_g2l.log 2>&1
ARK355_G2_WHOLE=1 timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > gpurun_out/fp_bench_whole.log 2>&1
exit 0
