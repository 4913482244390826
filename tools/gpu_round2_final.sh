#!/bin/bash
# round-2 evidence run: parity tests, smoke, bench lines (default with cpu_baseline, single stream, BN254, tight, 2^18,
# shard modes), rocprofv3 stats + PMC passes + serial + timeline -> gpurun_out/r02f_*; copy into profiles/ afterwards
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $O/r02f_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/r02f_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02f_smoke.log 2>&1; tail -2 $O/r02f_smoke.log
timeout 900 python bench.py > $O/r02f_bench_default.log 2> $O/r02f_bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --inflight 1 > $O/r02f_bench_inflight1.log 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --curve bn254 > $O/r02f_bench_bn254.log 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --tight > $O/r02f_bench_tight.log 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --log-n 18 > $O/r02f_bench_n18.log 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --mode shard --log-n 20 > $O/r02f_bench_shard_window.log 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --mode shard --shard-exchange ring --log-n 20 > $O/r02f_bench_shard_ring.log 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02f_prof_stats -o r2 -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > $O/r02f_rocprof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o r2 -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $O/r02f_rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o r2 -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $O/r02f_rocprof_write.log 2>&1
ARK355_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02f_serial -o serial -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > $O/r02f_serial_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/r02f_trace -o t -- python $R/bench.py --no-cpu-baseline --steps 12 --warmup 4 > $O/r02f_trace_bench.log 2>&1
cd $R
python tools/trace_analyze.py $(find $O/r02f_trace -name "*kernel_trace.csv" | head -1) 12 > $O/r02f_timeline_summary.txt 2>&1
python tools/pmc_summary.py --json $O/r02f_pmc_latest.json --workload "bls12_381:n=1048576" > $O/r02f_pmc_summary.log 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +20M -delete
for f in default inflight1 bn254 tight n18 shard_window shard_ring; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r02f_bench_$f.log") if l.startswith("{")][0])
    print("$f", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "prep %.1fs" % d["prep_s"], d["parity"], (d.get("cpu_baseline") or {}).get("sample","")[:200])
except Exception as e:
    print("$f FAILED", e)
PY
done
tail -5 $O/r02f_timeline_summary.txt
exit 0
