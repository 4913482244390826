#!/bin/bash
# round 3, run A (prepared at the end of round 2, not yet run): same-box A/B of the window size for resident keys,
# c = 16 (default) against ARK355_MSM_C=17 (MsmPlan::negate_high: 15 windows for 255-bit scalars; BN254's 254-bit scalars
# need no negation at c = 17), interleaved a/b/a/b, four proofs in flight and one, BLS12-381 / BN254 / 2^18.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "window_17 or msm_vs_naive or exceptional" > $O/r3a_pytest.txt 2>&1; tail -n 2 $O/r3a_pytest.txt
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3a_$tag.log 2> $O/r3a_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3a_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], "host cores %.2f" % d["host_cpu_cores"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3a_$tag.err").read()[-1500:])
PY
}
for rep in a b c; do
  EXTRA="--steps 32 --warmup 6"
  b c16_if4_$rep A=1
  b c17_if4_$rep ARK355_MSM_C=17
done
EXTRA="--inflight 1 --steps 12 --warmup 3"
b c16_if1 A=1
b c17_if1 ARK355_MSM_C=17
EXTRA="--curve bn254 --steps 32 --warmup 6"
b c16_bn A=1
b c17_bn ARK355_MSM_C=17
EXTRA="--log-n 18 --inflight 8 --steps 64 --warmup 8"
b c16_n18 A=1
b c17_n18 ARK355_MSM_C=17
exit 0
