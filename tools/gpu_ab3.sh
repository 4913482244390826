#!/bin/bash
mkdir -p gpurun_out
for v in prev chunk nopf; do
  ARK355_LIB=$PWD/variants/libark355_$v.so timeout 600 python bench.py --no-cpu-baseline --inflight 1 --steps 6 --warmup 2 > gpurun_out/ab_${v}_1.log 2>&1
done
exit 0
