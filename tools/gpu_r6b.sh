#!/bin/bash
# round 6, GPU run B (one box, same-box A/B): lanes per output of the 28-bit tail sums (TAIL_LPO 64 = one wave per output, 0 = by
# launch size), and larger windows now that a flush is stores only and the tails are plain sums: MSM_C / MSM_C_H = 19, 20.
R=$PWD; O=$R/gpurun_out/r6b; mkdir -p $O
run() { # tag envs... -- bench args
  tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-telemetry --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6b/%s.json" % tag))
    lat = {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}
    iso = d.get("isolated") or {}
    print("%-22s %7.3f ms/step  acc/proof %.2f  iso_total %.2f  lat %s  parity %s" % (tag, d["ms_per_step"], iso.get("accumulate_ms_per_proof", 0), iso.get("total_ms", 0), lat, str(d.get("parity"))[:10]))
except Exception as e:
    print(tag, "FAILED", e)
PY
}
for rep in 1 2; do
run lpo0_if4_$rep ARK355_TAIL_LPO=0 -- --steps 12 --warmup 4
run lpo64_if4_$rep ARK355_TAIL_LPO=64 -- --steps 12 --warmup 4
done
run lpo0_n18 ARK355_TAIL_LPO=0 -- --log-n 18 --inflight 8 --steps 48 --warmup 8
run lpo64_n18 ARK355_TAIL_LPO=64 -- --log-n 18 --inflight 8 --steps 48 --warmup 8
run lpo32_n18 ARK355_TAIL_LPO=32 -- --log-n 18 --inflight 8 --steps 48 --warmup 8
run c19_if4 ARK355_MSM_C=19 -- --steps 12 --warmup 4
run c20_if4 ARK355_MSM_C=20 -- --steps 12 --warmup 4
run ch20_if4 ARK355_MSM_C_H=20 -- --steps 12 --warmup 4
run c18_if4 ARK355_MSM_C=18 -- --steps 12 --warmup 4
run c17_if4 ARK355_MSM_C=17 -- --steps 12 --warmup 4
run c20_if4_seg48 ARK355_MSM_C=20 ARK355_MSM_SEG=48 -- --steps 12 --warmup 4
run c20_if4_seg96 ARK355_MSM_C=20 ARK355_MSM_SEG=96 -- --steps 12 --warmup 4
exit 0
