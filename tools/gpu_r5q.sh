#!/bin/bash
# round 5, GPU run Q (one box): the tree the round ends with (window tables always 28-bit rows, H tails on the sort stream) --
# the whole -m gpu suite, smoke, the driver's bench command (timed), the sharded rank path, kernel stats of the driver's command.
R=$PWD; O=$R/gpurun_out/r5q; mkdir -p $O $O/inflight
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
t0=$(date +%s)
timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" >> $O/status.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo "smoke rc=$?" >> $O/status.txt
t0=$(date +%s)
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$? ($(( $(date +%s) - t0 )) s)" >> $O/status.txt
timeout 300 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --whole --wm dist --steps 8 > $O/shard_rank_22.json 2> $O/shard_rank_22.log
echo "shard rank rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inflight -o inflight -- python $R/bench.py --profile-run --no-check --steps 16 --warmup 4 > $O/inflight/bench.log 2>&1
echo "rocprof inflight rc=$?" >> $O/status.txt
find $O/inflight -name '*kernel_trace.csv' -delete
cd $R
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
