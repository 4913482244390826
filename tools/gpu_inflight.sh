#!/bin/bash
# throughput vs number of proofs in flight (same box)
mkdir -p gpurun_out
for k in "$@"; do
  timeout 400 python bench.py --no-cpu-baseline --inflight $k --steps 24 --warmup $k > gpurun_out/ab_inflight${k}_3.log 2>&1
done
exit 0
