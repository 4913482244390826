#!/bin/bash
# round 4, GPU run D (one box): the final library (tuner: default phase first and last, 5 % margin; slice-first H2D of sharded
# proofs; per-curve infinity rule of the decoders) -- the driver's bench command twice (does the measured schedule agree with the
# in-run A/B?), the lazy-flush build at c = 17 / c = 16 against it (two interleaved passes), the per-rank path of a sharded
# 2^22 proof again, the GPU tests of the files that changed since run C.
R=$PWD; O=$R/gpurun_out/r4d; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
for i in 1 2; do
  ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_$i.json 2> $O/bench_default_$i.log
  echo "bench default $i rc=$?" >> $O/status.txt
done
run() {   # tag lib c extra-args...
  tag=$1; lib=$2; c=$3; shift 3
  ARK355_BENCH_WATCHDOG=280 ARK355_LIB=$lib ARK355_MSM_C=$c timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
NEW=$R/snark_amd/libark355.so; LAZY=$R/variants/lib_lazy.so
for pass in 1 2; do
  run ab_base_c16_$pass $NEW 0 --steps 16 --warmup 4
  run ab_lazy_c17_$pass $LAZY 17 --steps 16 --warmup 4
  run ab_base_c17_$pass $NEW 17 --steps 16 --warmup 4
  run ab_lazy_c16_$pass $LAZY 0 --steps 16 --warmup 4
done
run ab_lazy_c17_n18 $LAZY 17 --steps 32 --warmup 8 --log-n 18 --inflight 8
run ab_base_c16_n18 $NEW 0 --steps 32 --warmup 8 --log-n 18 --inflight 8
timeout 500 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm both > $O/shard_rank_22.json 2> $O/shard_rank_22.log
echo "shard rank rc=$?" >> $O/status.txt
timeout 600 python -m pytest tests/test_gpu_wire.py tests/test_gpu_parity.py -m gpu -x -q --durations=10 > $O/pytest_changed.txt 2>&1
echo "pytest changed rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
