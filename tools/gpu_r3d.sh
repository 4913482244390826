#!/bin/bash
# round 3, run D (diagnostic): what leaves the box in a state where the next process hangs?
R=$PWD; O=$R/gpurun_out; mkdir -p $O
state() { echo "--- $1"; ps -eo pid,stat,etime,cmd | grep -v "ps -eo" | grep -E "python|pytest" | head; rocm-smi --showmemuse 2>/dev/null | grep -E "GPU\[|%" | head -4; ls /dev/shm | head; }
timeout 120 python tools/rccl_inproc_probe.py > $O/r3d_probe.log 2>&1; echo "probe rc=$?"; tail -2 $O/r3d_probe.log
state "after in-process RCCL probe"
ARK355_BENCH_WATCHDOG=100 timeout 150 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/r3d_b1.log 2> $O/r3d_b1.err; echo "bench1 rc=$?"; grep "bench " $O/r3d_b1.err | tail -3
ARK355_BENCH_WATCHDOG=200 timeout 260 python bench.py --no-cpu-baseline --log-n 22 --inflight 1 --steps 2 --warmup 1 > $O/r3d_b22.log 2> $O/r3d_b22.err; echo "bench22 rc=$?"; grep "bench " $O/r3d_b22.err | tail -4
state "after 2^22 bench"
ARK355_BENCH_WATCHDOG=100 timeout 150 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/r3d_b2.log 2> $O/r3d_b2.err; echo "bench2 rc=$?"; grep "bench " $O/r3d_b2.err | tail -3
exit 0
