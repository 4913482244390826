#!/bin/bash
mkdir -p gpurun_out
for k in 2 4 6; do
  timeout 400 python bench.py --no-cpu-baseline --inflight $k --steps 12 --warmup 4 > gpurun_out/ab_inflight${k}_3.log 2>&1
done
exit 0
