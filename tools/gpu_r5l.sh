#!/bin/bash
# round 5, GPU run L: per-rank path of a sharded 2^22 proof (rank 0 of 8, distributed witness map) under three table layouts for
# the rank's 2^19-term vectors: the planner's c = 16 (16 windows), c = 17 with negated high scalars (15 windows), the same with
# bit-packed rows.
R=$PWD; O=$R/gpurun_out/r5l; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
run() { tag=$1; shift; env "$@" timeout 300 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm dist --steps 8 > $O/shard_$tag.json 2> $O/shard_$tag.log; echo "$tag rc=$?" >> $O/status.txt; }
run c16 ARK355_X=0
run c17 ARK355_MSM_C=17
run c17_packed ARK355_MSM_C=17 ARK355_PACK_ROWS=1
run c16_packed ARK355_PACK_ROWS=1
run c16_again ARK355_X=0
exit 0
