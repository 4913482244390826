#!/bin/bash
# round 4, GPU run A (one box): telemetry + the driver's bench command on the new tree, the c = 20 / lazy-flush A/B of
# DESIGN 8.4 (same box, two interleaved passes), per-rank critical path of a sharded 2^22 proof.
R=$PWD; O=$R/gpurun_out/r4a; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
for pass in 1 2; do
  for v in base:0:0 base:20:0 lazy:20:0 lazy:19:0 lazy:0:20 lazypf:20:0; do
    IFS=: read lib c ch <<< "$v"
    libpath=$R/snark_amd/libark355.so; [ "$lib" != base ] && libpath=$R/variants/lib_$lib.so
    ARK355_BENCH_WATCHDOG=280 ARK355_LIB=$libpath ARK355_MSM_C=$c ARK355_MSM_C_H=$ch timeout 300 python bench.py --steps 16 --warmup 4 \
      --no-cpu-baseline --no-micro --no-ab --no-telemetry > $O/ab_${lib}_c${c}_h${ch}_$pass.json 2> $O/ab_${lib}_c${c}_h${ch}_$pass.log
    echo "ab $v pass $pass rc=$?" >> $O/status.txt
  done
done
timeout 500 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --whole > $O/shard_rank_22.json 2> $O/shard_rank_22.log
echo "shard rank rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
