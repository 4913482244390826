#!/bin/bash
# round 5, GPU run N: the tails of the H MSM on the sort stream (policy SIDE_H_TAILS) against the reduction stream, same box,
# interleaved: per-rank path of the sharded 2^22 proof, a lone 2^20 proof on the five-stream pipeline; kernel trace of the rank.
R=$PWD; O=$R/gpurun_out/r5n; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
rank() { tag=$1; shift; env "$@" timeout 300 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --wm dist --steps 8 > $O/shard_$tag.json 2> $O/shard_$tag.log; echo "$tag rc=$?" >> $O/status.txt; }
lone() { tag=$1; shift; env "$@" ARK355_SCHED=1 ARK355_BENCH_WATCHDOG=200 timeout 220 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e --inflight 1 --steps 10 --warmup 3 > $O/lone_$tag.json 2> $O/lone_$tag.log; echo "$tag rc=$?" >> $O/status.txt; }
for pass in 1 2; do
  rank off_$pass ARK355_SIDE_H_TAILS=0
  rank on_$pass ARK355_SIDE_H_TAILS=1
done
lone off ARK355_SIDE_H_TAILS=0
lone on ARK355_SIDE_H_TAILS=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof -o shard -- python $R/tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm dist --steps 8 > $O/shard_traced.json 2> $O/shard_traced.log
echo "trace rc=$?" >> $O/status.txt
exit 0
