#!/bin/bash
# round 5, GPU run U (one box): the tree the round ends with (one wave per workgroup in the LDS-free accumulation kernels) --
# the whole -m gpu suite, smoke, the driver's bench command, the sharded rank path.
R=$PWD; O=$R/gpurun_out/r5u; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
t0=$(date +%s)
timeout 1000 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" >> $O/status.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo "smoke rc=$?" >> $O/status.txt
t0=$(date +%s)
ARK355_BENCH_WATCHDOG=400 timeout 450 python bench.py > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$? ($(( $(date +%s) - t0 )) s)" >> $O/status.txt
timeout 200 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --wm dist --steps 8 > $O/shard_rank_22.json 2> $O/shard_rank_22.log
echo "shard rank rc=$?" >> $O/status.txt
exit 0
