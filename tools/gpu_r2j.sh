#!/bin/bash
# round 2, run J: batch-affine bucket accumulation (msm_ba_impl.cuh): parity first, then the stand-alone MSM latency at 2^20
# (G2 and G1, default XYZZ kernels vs batch-affine tree levels), then the proof throughput with the G2 MSM on it
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_affine" > $O/r2j_pytest.txt 2>&1; tail -n 3 $O/r2j_pytest.txt
m() { tag=$1; shift; timeout 600 env "$@" python tools/msm_bench.py --min-log 20 --max-log 20 --reps 5 ${EXTRA} > $O/r2j_msm_$tag.txt 2>&1; grep -h "n=2" $O/r2j_msm_$tag.txt | sed "s/^/$tag /" ; tail -n 2 $O/r2j_msm_$tag.txt | grep -i "error\|Traceback" ; }
EXTRA="--groups 2 --dists uniform,boolean"
m g2_default A=1
m g2_ba5 ARK355_G2_BATCH_AFFINE=1
m g2_ba3 ARK355_G2_BATCH_AFFINE=1 ARK355_BA_LEVELS=3
m g2_ba7 ARK355_G2_BATCH_AFFINE=1 ARK355_BA_LEVELS=7
EXTRA="--groups 1 --dists uniform"
m g1_default A=1
m g1_ba5 ARK355_G1_BATCH_AFFINE=1
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r2j_$tag.log 2> $O/r2j_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2j_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2j_$tag.err").read()[-1500:])
PY
}
EXTRA="--steps 16 --warmup 4"
b base_a A=1
b g2ba_a ARK355_G2_BATCH_AFFINE=1
b base_b A=1
b g2ba_b ARK355_G2_BATCH_AFFINE=1
EXTRA="--inflight 1 --steps 8 --warmup 2"
b base_1 A=1
b g2ba_1 ARK355_G2_BATCH_AFFINE=1
exit 0
