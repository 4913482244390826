#!/bin/bash
# round 5, GPU run H (one box): two cheap sweeps on the final library -- entries per accumulation lane (policy MSM_SEG: the
# relay of section 11.3 has a shorter last leg with more, shorter rounds) and proofs in flight.
R=$PWD; O=$R/gpurun_out/r5h; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
run() {   # tag extra-args...   (environment of the caller applies)
  tag=$1; shift
  ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
for pass in 1 2; do
  run seg_default_$pass --steps 16 --warmup 4
  for seg in 20 30 40; do ARK355_MSM_SEG=$seg run seg_${seg}_$pass --steps 16 --warmup 4; done
done
ARK355_MSM_SEG=30 run seg_30_if1 --steps 8 --warmup 2 --inflight 1
run seg_default_if1 --steps 8 --warmup 2 --inflight 1
ARK355_MSM_SEG=30 run seg_30_bn --steps 16 --warmup 4 --curve bn254
run seg_default_bn --steps 16 --warmup 4 --curve bn254
for k in 3 5 6; do run inflight_$k --steps 24 --warmup 6 --inflight $k; done
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
