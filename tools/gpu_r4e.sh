#!/bin/bash
# round 4, GPU run E (one box): the library with c = 17 planned for large BLS12-381 tables, the latency-scored phases of the
# tuner and the G2 tails of a lone one-stream proof on a side stream -- against its own switches, interleaved:
# MSM_C=16 (the window of rounds 1-3), SIDE_G2_TAILS=0; then the O3 parity tests that now run at c = 17.
R=$PWD; O=$R/gpurun_out/r4e; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
run() {   # tag extra-args...   (environment of the caller applies)
  tag=$1; shift
  ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
for pass in 1 2; do
  run if4_default_$pass --steps 20 --warmup 5
  ARK355_MSM_C=16 run if4_c16_$pass --steps 20 --warmup 5 --no-ab
  run if1_side_$pass --steps 8 --warmup 2 --inflight 1 --no-ab
  ARK355_SIDE_G2_TAILS=0 run if1_noside_$pass --steps 8 --warmup 2 --inflight 1 --no-ab
done
run n18_if8 --steps 32 --warmup 8 --log-n 18 --inflight 8 --no-ab
run bn254_if4 --steps 16 --warmup 4 --curve bn254 --no-ab
timeout 600 python -m pytest "tests/test_gpu_o3_large.py::test_s2_2p20_bls12_381_vs_o3" "tests/test_gpu_o3_large.py::test_s2_2p20_tight_bls12_381_vs_o3" \
  "tests/test_gpu_o3_large.py::test_resident_msm_vs_o3" tests/test_gpu_parity.py -m gpu -x -q --durations=8 > $O/pytest_c17.txt 2>&1
echo "pytest c17 rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
