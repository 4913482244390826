#!/bin/bash
# round 2, run F: single-round segment lengths at mid sizes; window size at n = 2^18
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r2f_$tag.log 2> $O/r2f_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2f_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], {k: round(v,2) for k,v in d["phases_ms"].items()}, d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2f_$tag.err").read()[-1500:])
PY
}
EXTRA="--steps 16 --warmup 4" b n20_if4 A=1
EXTRA="--log-n 18 --inflight 4 --steps 32 --warmup 8" b n18_if4 A=1
EXTRA="--log-n 18 --inflight 8 --steps 32 --warmup 8" b n18_if8 A=1
EXTRA="--log-n 18 --inflight 1 --steps 16 --warmup 4" b n18_if1 A=1
EXTRA="--log-n 18 --inflight 4 --steps 32 --warmup 8" b n18_c16_if4 ARK355_MSM_C=16
EXTRA="--log-n 18 --inflight 8 --steps 32 --warmup 8" b n18_c16_if8 ARK355_MSM_C=16
EXTRA="--log-n 18 --inflight 1 --steps 16 --warmup 4" b n18_c16_if1 ARK355_MSM_C=16
EXTRA="--log-n 16 --inflight 8 --steps 64 --warmup 16" b n16_if8 A=1
EXTRA="--log-n 16 --inflight 8 --steps 64 --warmup 16" b n16_c16_if8 ARK355_MSM_C=16
EXTRA="--log-n 22 --inflight 2 --steps 6 --warmup 2" b n22_if2 A=1
