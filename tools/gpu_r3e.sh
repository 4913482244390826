#!/bin/bash
# round 3, run E: A/B of the adaptive wait (host cores) at 2^20 x 4 and 2^18 x 8 in flight; NTT pair stores (serial kernel
# stats + WRITE_SIZE); timeline of 2^18 x 8 in flight.  Every step under its own timeout.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; shift; ARK355_BENCH_WATCHDOG=150 timeout 170 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3e_$tag.log 2> $O/r3e_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3e_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cores %.2f" % d["host_cpu_cores"], d["host_cpu_threads"], "solo %.2f" % d["phases_ms"]["total_ms"], "lat pinned %.2f dev %.2f" % (d["latency"]["host_pinned_z_ms"], d["latency"]["device_z_ms"]), d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3e_$tag.err").read()[-800:])
PY
}
EXTRA="--steps 24 --warmup 4"
b adapt1_a A=1
b adapt0_a ARK355_WAIT_ADAPT=0
b adapt1_b A=1
EXTRA="--log-n 18 --inflight 8 --steps 64 --warmup 8"
b n18_adapt1 A=1
b n18_adapt0 ARK355_WAIT_ADAPT=0
cd /tmp && export TMPDIR=/tmp
for ps in 1 0; do
  ARK355_SERIAL=1 ARK355_NTT_PAIR_STORE=$ps timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3e_serial_ps$ps -o s -- python $R/bench.py --no-cpu-baseline --no-check --inflight 1 --steps 4 --warmup 1 > $O/r3e_serial_ps$ps.log 2>&1
  f=$(find $O/r3e_serial_ps$ps -name "*kernel_stats.csv" | head -1); echo "pair_store=$ps"; grep -E "ntt_|qap_|spmv|accumulate" $f | cut -d, -f1-4 | cut -c1-150 | head -12
  ARK355_NTT_PAIR_STORE=$ps timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r3e_write_ps$ps -o w -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $O/r3e_write_ps$ps.log 2>&1
  python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$O/r3e_write_ps$ps/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row.get("Kernel_Name", "")).replace("ark355::", "")[:60]
        if "ntt" in k:
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
for k, (n, v) in sorted(agg.items()):
    print("  WRITE_SIZE pair_store=$ps %-60s dispatches=%d avg_KiB=%.0f" % (k, n, v / n))
PY
done
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/r3e_trace18 -o t -- python $R/bench.py --no-cpu-baseline --no-check --log-n 18 --inflight 8 --steps 32 --warmup 8 > $O/r3e_trace18.log 2>&1
cd $R
python tools/trace_analyze.py $(find $O/r3e_trace18 -name "*kernel_trace.csv" | head -1) 32 > $O/r3e_timeline18.txt 2>&1; cat $O/r3e_timeline18.txt | head -30
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
exit 0
