#!/bin/bash
# round 5, GPU run J (one box): the tree the round ends with -- the whole -m gpu suite, smoke, the driver's bench command.
R=$PWD; O=$R/gpurun_out/r5j; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo "smoke rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=400 timeout 420 python bench.py --no-cpu-baseline --no-micro --no-ab --no-e2e --inflight 1 --steps 8 --warmup 2 > $O/bench_if1.json 2> $O/bench_if1.log
echo "bench if1 rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=400 timeout 420 python bench.py --no-cpu-baseline --no-micro --no-ab --no-e2e --curve bn254 --steps 16 --warmup 4 > $O/bench_bn254.json 2> $O/bench_bn254.log
echo "bench bn254 rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
