#!/bin/bash
mkdir -p gpurun_out
for c in 19 20 22; do
ARK355_MSM_C=$c timeout 400 python bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > gpurun_out/exp_c$c.log 2>&1
done
exit 0
