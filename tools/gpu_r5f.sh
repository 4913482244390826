#!/bin/bash
# round 5, GPU run F (one box): which segment walk ships?  kara (run A: unpacked 128-byte rows, one-ahead row prefetch),
# park2c (packed rows for both groups, two-deep index prefetch, first finished run parked in LDS), x (packed G1 rows + two-deep
# prefetch without parking; G2 on unpacked halves with the plain loop).  Interleaved, three passes; BN254; 2^18 x 8; the e2e
# reading of bench.py (synthesis in the loop) on its own.
R=$PWD; O=$R/gpurun_out/r5f; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
run() {   # tag lib extra-args...
  tag=$1; lib=$2; shift 2
  ARK355_BENCH_WATCHDOG=280 ARK355_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
KARA=$R/variants/lib_kara.so; P2=$R/variants/lib_park2c.so; X=$R/variants/lib_x.so
for pass in 1 2 3; do
  run ab_kara_if4_$pass $KARA --steps 16 --warmup 4
  run ab_park2c_if4_$pass $P2 --steps 16 --warmup 4
  run ab_x_if4_$pass $X --steps 16 --warmup 4
done
run ab_kara_bn $KARA --steps 16 --warmup 4 --curve bn254
run ab_park2c_bn $P2 --steps 16 --warmup 4 --curve bn254
run ab_x_bn $X --steps 16 --warmup 4 --curve bn254
run ab_park2c_n18 $P2 --steps 32 --warmup 8 --log-n 18 --inflight 8
run ab_x_n18 $X --steps 32 --warmup 8 --log-n 18 --inflight 8
run ab_park2c_if1 $P2 --steps 8 --warmup 2 --inflight 1
run ab_x_if1 $X --steps 8 --warmup 2 --inflight 1
ARK355_BENCH_WATCHDOG=500 ARK355_LIB=$P2 timeout 600 python bench.py --no-cpu-baseline --no-micro --no-ab --steps 12 --warmup 3 > $O/bench_e2e.json 2> $O/bench_e2e.log
echo "bench e2e rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
