#!/bin/bash
# round 6, GPU run I (one box): window size 17 where the planner still picks 16 -- vectors below 2^20 terms (2^18 proofs; the
# 2^19-term shards of a rank of the sharded 2^22 proof) and BN254 (254-bit scalars: 15 windows at c = 17 without the negation
# trick) -- now that a flush is stores only and the tails are cheap.  Same box, interleaved.
R=$PWD; O=$R/gpurun_out/r6i; mkdir -p $O
run() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6i/%s.json" % tag))
    lat = {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}
    box = d.get("box") or {}; iso = d.get("isolated") or {}
    print("%-16s %7.3f ms/step %6.2f M/s cyc/constraint %.2f  acc %.2f iso_total %.2f  lat %s  c=%s" % (tag, d["ms_per_step"], d["value"] / 1e6, box.get("gfx_cycles_per_constraint", 0), iso.get("accumulate_ms_per_proof", 0), iso.get("total_ms", 0), lat, (d.get("key_tables") or {}).get("window_bits")))
except Exception as e: print(tag, "FAILED", e)
PY
}
for rep in 1 2; do
run n18_c16_$rep ARK355_MSM_C=16 -- --log-n 18 --inflight 8 --steps 48 --warmup 8
run n18_c17_$rep ARK355_MSM_C=17 -- --log-n 18 --inflight 8 --steps 48 --warmup 8
done
for rep in 1 2; do
run bn_c16_$rep ARK355_MSM_C=16 -- --curve bn254
run bn_c17_$rep ARK355_MSM_C=17 -- --curve bn254
done
run n19_c16 ARK355_MSM_C=16 -- --log-n 19 --inflight 6 --steps 24 --warmup 6
run n19_c17 ARK355_MSM_C=17 -- --log-n 19 --inflight 6 --steps 24 --warmup 6
for c in 16 17 16 17; do
ARK355_MSM_C=$c timeout 100 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0 --wm dist --steps 8 > $O/shard_c$c.json 2>> $O/shard.log
python - $c <<'PY'
import json, sys
for l in open("gpurun_out/r6i/shard_c%s.json" % sys.argv[1]):
    try:
        d = json.loads(l); print("rank path c =", sys.argv[1], "median", d["ms_median"], "acc", d["accumulate_ms"], "window", d["tables"]["window_bits"])
    except Exception: pass
PY
done
exit 0
