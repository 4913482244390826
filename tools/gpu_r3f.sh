#!/bin/bash
# round 3, run F: host cost of the HIP calls a proof is made of; host time to queue one proof (ARK355_TRACE_HOST);
# NTT pass kernels sized for two waves per SIMD (variants/lib_ntt2.so: no scratch) against three (default).
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 120 tools/host_api_cost.bin 3000 > $O/r3f_host_api.txt 2>&1; cat $O/r3f_host_api.txt
for inf in 1 4; do
  ARK355_TRACE_HOST=1 timeout 120 python bench.py --no-cpu-baseline --no-check --inflight $inf --steps 8 --warmup 4 > $O/r3f_trace_if$inf.log 2> $O/r3f_trace_if$inf.err
  echo "inflight $inf:"; grep "prove host wall" $O/r3f_trace_if$inf.err | tail -6 | cut -c1-220
done
b() { tag=$1; shift; ARK355_BENCH_WATCHDOG=150 timeout 170 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3f_$tag.log 2> $O/r3f_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3f_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cores %.2f" % d["host_cpu_cores"], "solo %.2f wm %.2f" % (d["phases_ms"]["total_ms"], d["phases_ms"]["witness_map_ms"]), d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3f_$tag.err").read()[-800:])
PY
}
EXTRA="--steps 24 --warmup 4"
b ntt3_a A=1
b ntt2_a ARK355_LIB=$R/variants/lib_ntt2.so
b ntt3_b A=1
b ntt2_b ARK355_LIB=$R/variants/lib_ntt2.so
cd /tmp && export TMPDIR=/tmp
ARK355_SERIAL=1 ARK355_LIB=$R/variants/lib_ntt2.so timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3f_serial_ntt2 -o s -- python $R/bench.py --no-cpu-baseline --no-check --inflight 1 --steps 4 --warmup 1 > $O/r3f_serial_ntt2.log 2>&1
f=$(find $O/r3f_serial_ntt2 -name "*kernel_stats.csv" | head -1); echo "ntt2 (two waves per SIMD), serial:"; grep -E "ntt_" $f | cut -d, -f1-4 | sed 's/ark355:://g' | cut -c1-120 | head -6
find $O -name "*kernel_trace.csv" -delete
exit 0
