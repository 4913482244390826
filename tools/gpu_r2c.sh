#!/bin/bash
# round 2, run C: window-size A/B with the two-level bucket reduction and round-filling segments; end-to-end pipeline;
# MSM micro-benchmark 2^16..2^24
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --steps 12 --warmup 3 ${EXTRA} > $O/r2c_$tag.log 2> $O/r2c_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2c_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], {k: round(v,2) for k,v in d["phases_ms"].items()}, d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2c_$tag.err").read()[-1500:])
PY
}
EXTRA="--inflight 1" b c16_if1 A=1
EXTRA="--inflight 1" b c20_if1 ARK355_MSM_C=20
EXTRA="--inflight 1" b c19_if1 ARK355_MSM_C=19
EXTRA="--inflight 1" b c22_if1 ARK355_MSM_C=22
EXTRA="" b c16_if4 A=1
EXTRA="" b c20_if4 ARK355_MSM_C=20
timeout 400 tests/cpp/test_host_mirror --e2e bls12_381 1048576 24 8 4 > $O/r2c_e2e.log 2>&1; echo "e2e rc=$?"; cat $O/r2c_e2e.log
timeout 600 python tools/msm_bench.py --json $O/r2c_msm.json > $O/r2c_msm.log 2>&1; echo "msm rc=$?"; grep -v amdgpu.ids $O/r2c_msm.log | tail -32
