#!/bin/bash
# round 2, run I: (1) end-to-end pipeline (synthesis threads -> in-flight proofs) with persistent worker contexts and
# recycled page-locked buffers, 48 proofs of 2^20 constraints, next to the device-only figures (pageable / page-locked
# assignments); (2) window size for 2^18-constraint proofs (BASELINE configs[4]) with 8 in flight
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ARK355_E2E_SWEEP="4,8" timeout 600 tests/cpp/test_host_mirror --e2e bls12_381 1048576 48 6 4 > $O/r2i_e2e.txt 2> $O/r2i_e2e.err
cat $O/r2i_e2e.txt; tail -n 5 $O/r2i_e2e.err
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r2i_$tag.log 2> $O/r2i_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2i_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2i_$tag.err").read()[-1500:])
PY
}
EXTRA="--log-n 18 --inflight 8 --steps 48 --warmup 8"
b n18_default A=1
b n18_c15 ARK355_MSM_C=15
b n18_c16 ARK355_MSM_C=16
b n18_c13 ARK355_MSM_C=13
b n18_default_b A=1
exit 0
