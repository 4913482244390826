#!/bin/bash
# round 5, GPU run W: the per-call choice of the accumulation workgroup size (policy ACC_THREADS = 0): a lone proof, the rank path,
# the driver's timed region (short), parity subset.
R=$PWD; O=$R/gpurun_out/r5w; mkdir -p $O
timeout 60 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm dist --steps 8 > $O/shard_auto.json 2> $O/shard_auto.log; echo "shard rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=100 timeout 110 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e --steps 12 --warmup 4 > $O/if4_auto.json 2> $O/if4_auto.log; echo "if4 rc=$?" >> $O/status.txt
timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_stream_schedule or one_stream_tail or last_msm_tails" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/status.txt
exit 0
