#!/bin/bash
# round 6, GPU run K (one box): larger windows for LARGE keys (2^22 constraints on one GPU: 2^22 / 2^23-term MSMs), where the fixed
# cost of 8x the buckets (run B: +2.3 ms per proof at c = 20) is small against the 13 % of additions it saves.
R=$PWD; O=$R/gpurun_out/r6k; mkdir -p $O
for c in 17 20 19 17 20; do
  ARK355_MSM_C=$c ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab --log-n 22 --inflight 2 --steps 6 --warmup 2 > $O/n22_c${c}_$SECONDS.json 2> $O/n22_c$c.log
  python - $O/n22_c${c}_$SECONDS.json $c <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); b = d["box"]; i = d["isolated"]; l = d["latency"]
    print("2^22 c = %s: %8.3f ms/step %6.2f M/s  cyc/constraint %.2f  acc %.2f  alone %.2f  lat host %.2f dev %.2f  tables %.1f GB parity %s" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, b.get("gfx_cycles_per_constraint", 0), i["accumulate_ms_per_proof"], i["total_ms"], l["host_pinned_z_ms"], l["device_z_ms"], d["key_tables"]["table_bytes"] / 1e9, d["parity"][:12]))
except Exception as e: print("c =", sys.argv[2], "FAILED", e)
PY
done
exit 0
