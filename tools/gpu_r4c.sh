#!/bin/bash
# round 4, GPU run C (one box): the new tests first (distributed witness map, 2^20 key streams), per-rank critical path of a
# sharded 2^22 proof with the replicated and with the distributed witness map, rocprofv3 kernel stats (one proof at a time
# on one stream = isolated kernels; four in flight = the bench's timed region) and the two PMC passes, the whole -m gpu
# suite with durations, then the driver's bench command.
R=$PWD; O=$R/gpurun_out/r4c; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_wire_large.py "tests/test_gpu_o3_large.py::test_distributed_witness_map_vs_o3_in_full" \
  "tests/test_gpu_o3_large.py::test_witness_map_large_vs_o3_in_full" -x -q -s --durations=20 > $O/pytest_new.txt 2>&1
echo "pytest new rc=$?" >> $O/status.txt
timeout 700 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --whole --wm both > $O/shard_rank_22.json 2> $O/shard_rank_22.log
echo "shard rank rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
mkdir -p $O/serial $O/inflight $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
ARK355_SCHED=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 6 --warmup 2 > $O/serial/bench.log 2>&1
echo "serial prof rc=$?" >> $O/status.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inflight -o inflight -- python $R/bench.py --profile-run --no-check --steps 16 --warmup 4 > $O/inflight/bench.log 2>&1
echo "inflight prof rc=$?" >> $O/status.txt
rm -rf $R/gpurun_out/prof_fetch/* $R/gpurun_out/prof_write/*
ARK355_SCHED=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o f -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?" >> $O/status.txt
ARK355_SCHED=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o w -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
echo "pmc write rc=$?" >> $O/status.txt
cd $R
python tools/pmc_summary.py --json $O/pmc_latest.json --workload "bls12_381:n=1048576" --recorded "round 4 run C" > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
rm -rf $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
timeout 1100 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-micro --no-ab --log-n 18 --inflight 8 --steps 32 --warmup 8 > $O/bench_n18_if8.json 2> $O/bench_n18_if8.log
ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-micro --no-ab --curve bn254 --steps 16 --warmup 4 > $O/bench_bn254.json 2> $O/bench_bn254.log
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
