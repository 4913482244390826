#!/bin/bash
# round 2, run E: A/B of (bucket clears off the accumulation stream + 128-workgroup heavy merge) against the previous
# build on the same box; proofs in flight at n = 2^18
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --steps 16 --warmup 4 ${EXTRA} > $O/r2e_$tag.log 2> $O/r2e_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2e_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2e_$tag.err").read()[-1500:])
PY
}
EXTRA="" b new_if4_a A=1
EXTRA="" b prev_if4_a ARK355_LIB=$R/variants/libark355_prev.so
EXTRA="" b new_if4_b A=1
EXTRA="" b prev_if4_b ARK355_LIB=$R/variants/libark355_prev.so
EXTRA="--inflight 1" b new_if1 A=1
EXTRA="--inflight 1" b prev_if1 ARK355_LIB=$R/variants/libark355_prev.so
EXTRA="--inflight 6" b new_if6 A=1
EXTRA="--log-n 18 --inflight 4 --steps 32 --warmup 8" b n18_if4 A=1
EXTRA="--log-n 18 --inflight 8 --steps 32 --warmup 8" b n18_if8 A=1
EXTRA="--log-n 18 --inflight 12 --steps 48 --warmup 12" b n18_if12 A=1
