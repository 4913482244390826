#!/bin/bash
# round 3, run C (diagnostic): where does bench.py stop?  Stage markers on stderr + a watchdog that dumps every thread's
# Python stack.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ARK355_BENCH_WATCHDOG=150 timeout 200 python bench.py --no-cpu-baseline --steps 8 --warmup 2 > $O/r3c_default.log 2> $O/r3c_default.err
echo "rc=$?"; grep -v amdgpu.ids $O/r3c_default.err | tail -60; cut -c1-600 $O/r3c_default.log
exit 0
