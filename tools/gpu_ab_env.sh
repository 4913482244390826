#!/bin/bash
# A/B on ONE box with the in-tree library: tools/gpu_ab_env.sh <tag>:<ENV=VAL,ENV=VAL|-> ...
mkdir -p gpurun_out
for kv in "$@"; do
  tag=${kv%%:*}; envs=${kv#*:}
  [ "$envs" = "-" ] && envs=""
  envs=${envs//,/ }
  env $envs timeout 400 python bench.py --no-cpu-baseline --inflight 1 --steps 6 --warmup 2 > gpurun_out/ab_${tag}_1.log 2>&1
  env $envs timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/ab_${tag}_3.log 2>&1
done
exit 0
