#!/bin/bash
# round 4, GPU run B (one box): the dispatch-count work (one fill per proof, batched G1 tails, batched NTT passes, phase-based
# schedule tuner) against the library of the round's first commit (variants/lib_prev.so), interleaved on ONE box:
# 2^20 x 4 (the driver's command), 2^20 x 1, 2^18 x 8, BN254; isolated kernel stats of the new library (rocprofv3); then the
# whole -m gpu suite with durations.
R=$PWD; O=$R/gpurun_out/r4b; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
run() {   # tag lib extra-args...
  tag=$1; lib=$2; shift 2
  ARK355_BENCH_WATCHDOG=280 ARK355_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
NEW=$R/snark_amd/libark355.so; PREV=$R/variants/lib_prev.so
for pass in 1 2; do
  for v in new:$NEW prev:$PREV; do
    IFS=: read nm lib <<< "$v"
    run ab_${nm}_n20_if4_$pass $lib --steps 16 --warmup 4 --no-ab
    run ab_${nm}_n18_if8_$pass $lib --steps 32 --warmup 8 --log-n 18 --inflight 8 --no-ab
  done
done
for v in new:$NEW prev:$PREV; do
  IFS=: read nm lib <<< "$v"
  run ab_${nm}_n20_if1 $lib --steps 8 --warmup 2 --inflight 1 --no-ab
  run ab_${nm}_bn254_if4 $lib --steps 16 --warmup 4 --curve bn254 --no-ab
done
ARK355_BATCH_TAILS=0 run ab_new_nobatchtails_n18_if8 $NEW --steps 32 --warmup 8 --log-n 18 --inflight 8 --no-ab
# isolated per-kernel durations of the new library: one proof at a time on one stream
mkdir -p $O/serial; cd /tmp && export TMPDIR=/tmp
ARK355_SCHED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --no-cpu-baseline --no-ab --no-micro --no-telemetry --inflight 1 --steps 5 --warmup 2 > $O/serial/bench.log 2>&1
echo "serial prof rc=$?" >> $O/status.txt
cd $R; find $O/serial -name "*kernel_trace.csv" -delete
# the whole GPU suite on the new tree
timeout 1100 python -m pytest tests -m gpu -x -q --durations=70 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
