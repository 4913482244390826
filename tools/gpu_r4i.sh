#!/bin/bash
# round 4, GPU run I: the whole -m gpu suite, smoke and the driver's bench command on the tree the round ends with.
R=$PWD; O=$R/gpurun_out/r4i; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo "smoke rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
exit 0
