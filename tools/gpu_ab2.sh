#!/bin/bash
# A/B builds on the same box while sampling clocks/power
mkdir -p gpurun_out
rocm-smi --showmaxpower --showperflevel --showclocks --showmemuse > gpurun_out/smi_caps.log 2>&1
( for i in $(seq 1 200); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Graphics Package Power" | tr '\n' ' '; echo; sleep 0.5; done > gpurun_out/smi_samples.log ) &
SAMPLER=$!
for v in prev chunk; do
  ARK355_LIB=$PWD/variants/libark355_$v.so timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 8 --warmup 2 > gpurun_out/ab_${v}_1.log 2>&1
  echo "=== after $v" >> gpurun_out/smi_samples.log
done
kill $SAMPLER 2>/dev/null
exit 0
