#!/bin/bash
# A/B on ONE box: tools/gpu_ab.sh <tag>=<lib.so> ...   (single-proof and 3-in-flight bench per variant)
mkdir -p gpurun_out
for kv in "$@"; do
  tag=${kv%%=*}; lib=${kv#*=}
  ARK355_LIB=$PWD/$lib timeout 400 python bench.py --no-cpu-baseline --inflight 1 --steps 6 --warmup 2 > gpurun_out/ab_${tag}_1.log 2>&1
  ARK355_LIB=$PWD/$lib timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/ab_${tag}_3.log 2>&1
done
exit 0
