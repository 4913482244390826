#!/bin/bash
# round 5, GPU run I (one box): a lone one-stream proof with its witness map and the sort of h on a side stream (policy SIDE_WM)
# against SIDE_WM=0, interleaved: one proof at a time (throughput and the 8d latency), then the in-flight default (unaffected by
# the switch) and the proving tests.
R=$PWD; O=$R/gpurun_out/r5i; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
run() {   # tag extra-args...
  tag=$1; shift
  ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
for pass in 1 2 3; do
  ARK355_SIDE_WM=1 run if1_side_$pass --steps 10 --warmup 3 --inflight 1
  ARK355_SIDE_WM=0 run if1_noside_$pass --steps 10 --warmup 3 --inflight 1
done
ARK355_SIDE_WM=1 run bn_if1_side --steps 10 --warmup 3 --inflight 1 --curve bn254
ARK355_SIDE_WM=0 run bn_if1_noside --steps 10 --warmup 3 --inflight 1 --curve bn254
ARK355_SIDE_WM=1 run n18_if1_side --steps 16 --warmup 4 --inflight 1 --log-n 18
ARK355_SIDE_WM=0 run n18_if1_noside --steps 16 --warmup 4 --inflight 1 --log-n 18
run if4_default --steps 16 --warmup 4
timeout 900 python -m pytest tests/test_gpu_parity.py "tests/test_gpu_o3_large.py::test_s2_2p20_bls12_381_vs_o3" "tests/test_gpu_o3_large.py::test_s2_2p20_tight_bls12_381_vs_o3" \
  "tests/test_gpu_o3_large.py::test_s2_2p20_bn254_vs_o3" "tests/test_gpu_o3_large.py::test_batch_2p18_vs_o3" -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
