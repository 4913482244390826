#!/bin/bash
# round 2, run G: cost of the latency-bound tails in throughput mode: two-level bucket reduction at c = 16, longer segments
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r2g_$tag.log 2> $O/r2g_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2g_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f finalize %.2f" % (d["phases_ms"]["total_ms"], d["phases_ms"]["finalize_ms"]), d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2g_$tag.err").read()[-1500:])
PY
}
EXTRA="--steps 16 --warmup 4" b base_a A=1
EXTRA="--steps 16 --warmup 4" b twolevel_a ARK355_MSM_TWO_LEVEL_MIN=1024
EXTRA="--steps 16 --warmup 4" b seg128_a ARK355_MSM_SEG=128
EXTRA="--steps 16 --warmup 4" b both_a ARK355_MSM_TWO_LEVEL_MIN=1024 ARK355_MSM_SEG=128
EXTRA="--steps 16 --warmup 4" b base_b A=1
EXTRA="--steps 16 --warmup 4" b both_b ARK355_MSM_TWO_LEVEL_MIN=1024 ARK355_MSM_SEG=128
EXTRA="--log-n 18 --inflight 6 --steps 32 --warmup 8" b n18_base A=1
EXTRA="--log-n 18 --inflight 6 --steps 32 --warmup 8" b n18_both ARK355_MSM_TWO_LEVEL_MIN=1024 ARK355_MSM_SEG=128
