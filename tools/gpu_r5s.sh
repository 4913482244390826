#!/bin/bash
# round 5, GPU run S: the tails of A, B1, L' of a lone one-stream proof on the reduction stream under the H accumulation (policy
# SIDE_G1_TAILS) against the batch of four at the end, same box, interleaved; the tail-variant tests.
# (The policy existed in an experiment tree only: it measured +1 ms and was not kept -- DESIGN.md 11.6, profiles/r05_runS_*.)
R=$PWD; O=$R/gpurun_out/r5s; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_stream_tail or last_msm_tails" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
lone() { tag=$1; shift; env "$@" ARK355_SCHED=0 ARK355_BENCH_WATCHDOG=200 timeout 220 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e --inflight 1 --steps 12 --warmup 3 > $O/lone_$tag.json 2> $O/lone_$tag.log; echo "$tag rc=$?" >> $O/status.txt; }
for pass in 1 2 3; do
  lone off_$pass ARK355_SIDE_G1_TAILS=0
  lone on_$pass ARK355_SIDE_G1_TAILS=1
done
exit 0
