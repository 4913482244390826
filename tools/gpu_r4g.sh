#!/bin/bash
# round 4, GPU run G (one box, short): the driver's command and the schedule tests on the last library (one stream as the static
# default of a lone proof too, 5 % margin in both classes; stdout of bench.py = the JSON line only); where the last millisecond of a lone proof goes (host wall-clock phases of prove_run),
# and two sweeps on the final library: entries per accumulation lane at c = 17, proofs in flight.
R=$PWD; O=$R/gpurun_out/r4g; mkdir -p $O
run() {
  tag=$1; shift
  ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "schedule or stream or tail or batch or sharded" > $O/pytest_sched.txt 2>&1
echo "pytest sched rc=$?" >> $O/status.txt
ARK355_TRACE_HOST=1 run trace_if1 --inflight 1 --steps 6 --warmup 2
for seg in 48 60 72 96; do
  ARK355_MSM_SEG=$seg run seg${seg}_if4 --steps 16 --warmup 4
done
for k in 3 4 5 6 8; do
  run inflight$k --inflight $k --steps 24 --warmup 8
done
exit 0
