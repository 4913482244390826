#!/bin/bash
# A/B two builds of the library on the SAME box, alternating
mkdir -p gpurun_out
for i in 1 2; do
  for v in prev chunk; do
    ARK355_LIB=$PWD/variants/libark355_$v.so timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > gpurun_out/ab_${v}_$i.log 2>&1
  done
done
exit 0
