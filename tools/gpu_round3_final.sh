#!/bin/bash
# round-3 evidence run (lean: 20 GPU-minutes were left): bench lines (driver-style default with cpu_baseline, 2^18 x 8,
# shard mode at its default 2^22 with both exchanges, 2^22 replica, BN254, a 2^23-constraint key), rocprofv3 stats (pipelined
# single stream + serial), PMC passes, then the new / changed GPU tests, smoke and a last bench (the box must still answer).
R=$PWD; O=$R/gpurun_out; mkdir -p $O
line() { python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r03f_bench_$1.log") if l.startswith("{")][0])
    lat=d.get("latency") or {}
    print("$1", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cores %.2f" % d["host_cpu_cores"], "solo %.2f" % d["phases_ms"]["total_ms"], "lat pinned %s" % lat.get("host_pinned_z_ms"), "prep %.1fs" % d["prep_s"], d["parity"], (d.get("cpu_baseline") or {}).get("sample","")[:160])
except Exception as e:
    print("$1 FAILED", e); print(open("$O/r03f_bench_$1.err").read()[-600:])
PY
}
b() { tag=$1; tmo=$2; shift; shift; ARK355_BENCH_WATCHDOG=$tmo timeout $((tmo+20)) python bench.py "$@" > $O/r03f_bench_$tag.log 2> $O/r03f_bench_$tag.err; line $tag; }
b default 240
b n18_if8 120 --no-cpu-baseline --log-n 18 --inflight 8 --steps 64 --warmup 8
b shard22_window 200 --no-cpu-baseline --mode shard --steps 6 --warmup 2
b shard22_ring 200 --no-cpu-baseline --mode shard --shard-exchange ring --steps 6 --warmup 2
b n22_if2 200 --no-cpu-baseline --log-n 22 --inflight 2 --steps 8 --warmup 2
b bn254 120 --no-cpu-baseline --curve bn254
ARK355_TRACE_HOST=1 ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --log-n 23 --inflight 1 --steps 3 --warmup 1 > $O/r03f_bench_n23_if1.log 2> $O/r03f_bench_n23_if1.err; line n23_if1; grep "window table" $O/r03f_bench_n23_if1.err | head -5
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03f_prof_stats -o r3 -- python $R/bench.py --no-cpu-baseline --no-check --inflight 1 --steps 5 --warmup 2 > $O/r03f_rocprof_stats.log 2>&1
ARK355_SERIAL=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03f_serial -o serial -- python $R/bench.py --no-cpu-baseline --no-check --inflight 1 --steps 5 --warmup 2 > $O/r03f_serial_bench.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o r3 -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $O/r03f_rocprof_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o r3 -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $O/r03f_rocprof_write.log 2>&1
cd $R
python tools/pmc_summary.py --json $O/r03f_pmc_latest.json --workload "bls12_381:n=1048576" --recorded "round 3, final tree, $(date -u +%Y-%m-%dT%H:%MZ)" > $O/r03f_pmc_summary.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
timeout 480 python -m pytest tests/test_gpu_wire.py tests/test_gpu_o3_large.py -x -q -m gpu --durations=8 -k "wire or point_codecs or key_stream or validation or ntt_large or witness_map_large or (2p22 and tight)" > $O/r03f_pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $O/r03f_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03f_smoke.log 2>&1; tail -1 $O/r03f_smoke.log
b after_tests 100 --no-cpu-baseline --steps 8 --warmup 2
exit 0
