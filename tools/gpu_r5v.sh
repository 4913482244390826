#!/bin/bash
# round 5, GPU run V: the sharded rank path under ACC_THREADS 64 (default) / 256, same box, interleaved.
R=$PWD; O=$R/gpurun_out/r5v; mkdir -p $O
rank() { tag=$1; shift; env "$@" timeout 100 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm dist --steps 8 > $O/shard_$tag.json 2> $O/shard_$tag.log; echo "$tag rc=$?" >> $O/status.txt; }
rank t64_1 ARK355_ACC_THREADS=64
rank t256_1 ARK355_ACC_THREADS=256
rank t64_2 ARK355_ACC_THREADS=64
rank t256_2 ARK355_ACC_THREADS=256
exit 0
