#!/bin/bash
# round 5, GPU run T: workgroup size of the LDS-free accumulation kernels (policy ACC_THREADS 256 / 128 / 64), same box,
# interleaved: the driver's command (four in flight) and a lone proof; parity on the small size first.
R=$PWD; O=$R/gpurun_out/r5t; mkdir -p $O
ARK355_ACC_THREADS=64 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident_msm or prove_small or one_stream_schedule or both_row_formats" > $O/pytest64.txt 2>&1
echo "pytest64 rc=$?" >> $O/status.txt
run() { tag=$1; shift; env "$@" ARK355_BENCH_WATCHDOG=200 timeout 220 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e --steps 16 --warmup 4 > $O/if4_$tag.json 2> $O/if4_$tag.log; echo "$tag rc=$?" >> $O/status.txt; }
for pass in 1 2; do
  run t256_$pass ARK355_ACC_THREADS=256
  run t64_$pass ARK355_ACC_THREADS=64
  run t128_$pass ARK355_ACC_THREADS=128
done
exit 0
