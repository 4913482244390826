#!/bin/bash
# round 5, GPU run O: per-rank path of the sharded 2^22 proof (rank 0 of 8) against the segment length of its one-round
# accumulation launches and the two-level bucket reduction at 2^15 buckets.
R=$PWD; O=$R/gpurun_out/r5o; mkdir -p $O
rank() { tag=$1; shift; env "$@" timeout 300 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm dist --steps 8 > $O/shard_$tag.json 2> $O/shard_$tag.log; echo "$tag rc=$?" >> $O/status.txt; }
rank default ARK355_X=0
rank seg32 ARK355_MSM_SEG=32
rank seg48 ARK355_MSM_SEG=48
rank seg96 ARK355_MSM_SEG=96
rank seg128 ARK355_MSM_SEG=128
rank twolevel ARK355_MSM_TWO_LEVEL_MIN=16384
rank default2 ARK355_X=0
exit 0
