#!/bin/bash
# round 6, GPU run A: first run of the 28-bit tails (tails28_impl.cuh): parity subset, the driver's command (short), a serial
# kernel trace (per-kernel durations of one proof alone on one stream), the rank path, the 2^18 batch.
R=$PWD; O=$R/gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_tables.py -m gpu -x -q > $O/pytest_parity.txt 2>&1; echo "pytest rc=$?" >> $O/status.txt
tail -5 $O/pytest_parity.txt
ARK355_BENCH_WATCHDOG=200 timeout 220 python bench.py --no-cpu-baseline --no-e2e --steps 12 --warmup 4 > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?" >> $O/status.txt
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r6a/bench_default.json"))
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "latency", {k:v for k,v in (d.get("latency") or {}).items() if k.endswith("_ms")})
    print("isolated", {k:d["isolated"][k] for k in ("accumulate_ms_per_proof","total_ms","witness_map_ms")})
    print("micro", {k:v["ms"] for k,v in d["micro"]["msm"].items()})
except Exception as e: print("bench parse failed", e)
PY
cd /tmp && export TMPDIR=/tmp
ARK355_SERIAL=1 ARK355_SIDE_WM=0 ARK355_SIDE_G2_TAILS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --profile-run --inflight 1 --steps 5 --warmup 2 > $O/serial_bench.log 2>&1; echo "serial rc=$?" >> $O/status.txt
cd $R; find $O/serial -name "*kernel_trace.csv" -delete; find $O/serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/serial_kernel_stats.csv
head -30 $O/serial_kernel_stats.csv | cut -c1-160
timeout 100 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --wm dist --steps 8 > $O/shard_rank_22.json 2> $O/shard_rank.log; echo "shard rc=$?" >> $O/status.txt
cat $O/shard_rank_22.json | head -c 1500
ARK355_BENCH_WATCHDOG=100 timeout 110 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e --log-n 18 --inflight 8 --steps 48 --warmup 8 > $O/bench_n18_if8.json 2> $O/bench_n18.log; echo "n18 rc=$?" >> $O/status.txt
python -c "
import json; d=json.load(open('gpurun_out/r6a/bench_n18_if8.json')); print('n18x8 ms_per_step', d['ms_per_step'], 'value', d['value'])"
exit 0
