#!/bin/bash
# round 2, run N: lane-pair G2 tail kernels (merge / reduce / combine on Fp2L) against ARK355_G2_PAIR_TAILS=0, same library
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm or prove_small or dummy or batch_in_flight" > $O/r2n_pytest.txt 2>&1; tail -n 2 $O/r2n_pytest.txt
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r2n_$tag.log 2> $O/r2n_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2n_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2n_$tag.err").read()[-1500:])
PY
}
EXTRA="--steps 16 --warmup 4"
b old_4_a ARK355_G2_PAIR_TAILS=0
b new_4_a A=1
b old_4_b ARK355_G2_PAIR_TAILS=0
b new_4_b A=1
EXTRA="--inflight 1 --steps 8 --warmup 2"
b old_1 ARK355_G2_PAIR_TAILS=0
b new_1 A=1
EXTRA="--log-n 18 --inflight 8 --steps 48 --warmup 8"
b old_n18 ARK355_G2_PAIR_TAILS=0
b new_n18 A=1
for v in old new; do
  e="A=1"; [ $v = old ] && e="ARK355_G2_PAIR_TAILS=0"
  env $e timeout 600 python tools/msm_bench.py --min-log 16 --max-log 20 --step 2 --reps 5 --groups 2 --dists uniform,equal > $O/r2n_msm_$v.txt 2>&1
  grep -h "n=2" $O/r2n_msm_$v.txt | sed "s/^/$v /"
done
exit 0
