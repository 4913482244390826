#!/bin/bash
# round 6, GPU run L (one box): the planner's c = 20 for tables of 2^22 terms and more -- parity of everything that reaches that size
# (both 2^22-size keys whole and sharded incl. RCCL_SELF, the 2^22 resident MSMs, the 2^23-point NTT / witness map), then the bench
# lines at 2^22 (two in flight, one at a time, --mode shard at world size 1) and 2^23 (the reference's own benchmark size).
R=$PWD; O=$R/gpurun_out/r6l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_o3_large.py -m gpu -x -q -k "2p22 or resident_msm or witness_map_large" > $O/pytest_large.txt 2>&1; echo "pytest rc=$?" >> $O/status.txt
tail -n 3 $O/pytest_large.txt
run() { tag=$1; shift; ARK355_BENCH_WATCHDOG=380 timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab "$@" > $O/$tag.json 2> $O/$tag.log; echo "$tag rc=$?" >> $O/status.txt
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); b = d["box"]; i = d.get("isolated") or {}; l = d.get("latency") or {}
    print("%-14s %8.3f ms/step %6.2f M/s  cyc/constraint %.2f  acc %.2f  alone %.2f  lat host %.2f dev %.2f  tables %.1f GB c=%s %s" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, b.get("gfx_cycles_per_constraint", 0), i.get("accumulate_ms_per_proof", 0), i.get("total_ms", 0), l.get("host_pinned_z_ms", 0), l.get("device_z_ms", 0), d["key_tables"]["table_bytes"] / 1e9, d["key_tables"]["window_bits"], d["parity"][:12]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run n22_if2 --log-n 22 --inflight 2 --steps 6 --warmup 2
run n22_if1 --log-n 22 --inflight 1 --steps 4 --warmup 1
run shard22_w1 --mode shard --steps 4 --warmup 1
run n21_if3 --log-n 21 --inflight 3 --steps 9 --warmup 3
run n23_if1 --log-n 23 --inflight 1 --steps 3 --warmup 1
cat $O/status.txt
exit 0
