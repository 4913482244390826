#!/bin/bash
# round 5, GPU run G (one box): the evidence of the FINAL library -- the whole -m gpu suite with durations, smoke, the driver's
# bench command, the other bench lines (2^18 x 8, BN254, one proof at a time, 2^22 replica, BLS12-381 with PACKED rows, 2^23
# with packed rows), the per-rank path of a sharded 2^22 proof, stand-alone MSMs, rocprofv3 kernel stats (one proof at a time
# on one stream; four in flight), the PMC passes of three workloads.
R=$PWD; O=$R/gpurun_out/r5g; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
timeout 1100 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo "smoke rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
run() {
  tag=$1; shift
  ARK355_BENCH_WATCHDOG=400 timeout 420 python bench.py --no-cpu-baseline --no-micro --no-ab --no-e2e "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
run bench_n18_if8 --log-n 18 --inflight 8 --steps 32 --warmup 8
run bench_bn254 --curve bn254 --steps 16 --warmup 4
run bench_if1 --inflight 1 --steps 8 --warmup 2
run bench_n22_if2 --log-n 22 --inflight 2 --steps 6 --warmup 2
ARK355_PACK_ROWS=1 run bench_packed_rows --steps 16 --warmup 4
ARK355_PACK_ROWS=1 run bench_n23_if1_packed --log-n 23 --inflight 1 --steps 3 --warmup 1
timeout 500 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --whole --wm both > $O/shard_rank_22.json 2> $O/shard_rank_22.log
echo "shard rank rc=$?" >> $O/status.txt
timeout 400 python tools/msm_bench.py --min-log 16 --max-log 22 --step 2 --reps 5 --json $O/msm_microbench.json > $O/msm_microbench.txt 2>&1
echo "msm bench rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
mkdir -p $O/serial $O/inflight
ARK355_SCHED=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 6 --warmup 2 > $O/serial/bench.log 2>&1
echo "serial prof rc=$?" >> $O/status.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inflight -o inflight -- python $R/bench.py --profile-run --no-check --steps 16 --warmup 4 > $O/inflight/bench.log 2>&1
echo "inflight prof rc=$?" >> $O/status.txt
pmc() {   # workload-key bench-args...
  key=$1; shift
  rm -rf $O/pmc/prof_fetch $O/pmc/prof_write; mkdir -p $O/pmc/prof_fetch $O/pmc/prof_write
  ARK355_SCHED=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/prof_fetch -o f -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 1 --warmup 0 "$@" > $O/pmc_fetch.log 2>&1
  echo "pmc fetch $key rc=$?" >> $O/status.txt
  ARK355_SCHED=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/prof_write -o w -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 1 --warmup 0 "$@" > $O/pmc_write.log 2>&1
  echo "pmc write $key rc=$?" >> $O/status.txt
  (cd $R && python tools/pmc_summary.py --dir $O/pmc --json $O/pmc_latest.json --merge --workload "$key" --recorded "round 5 run G, final library" >> $O/pmc_summary.txt 2>&1)
}
pmc "bls12_381:n=1048576"
pmc "bn254:n=1048576" --curve bn254
pmc "bls12_381:n=262144" --log-n 18
cd $R
find $O -name "*kernel_trace.csv" -delete
rm -rf $O/pmc
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
