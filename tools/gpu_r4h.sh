#!/bin/bash
# round 4, GPU run H (one box, short): the tuner's last repair (a proof's wall time over the proofs that shared the device with
# it) -- the driver's command and 5 / 6 / 8 proofs in flight: does the latched schedule agree with the in-run A/B every time?
R=$PWD; O=$R/gpurun_out/r4h; mkdir -p $O
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
for k in 6 5 8 6; do
  ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --inflight $k --steps 24 --warmup 8 > $O/inflight${k}_$RANDOM.json 2>> $O/inflight.log
  echo "inflight $k rc=$?" >> $O/status.txt
done
exit 0
