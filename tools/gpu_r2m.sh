#!/bin/bash
# round 2, run M: out-of-line G1 group operations with inlined multiplications (merge / bucket reduction / combination
# tails) and host-side normalisation of stand-alone MSM results: new build vs variants/lib_base.so on one box
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; lib=$2; shift 2; timeout 300 env ARK355_LIB=$lib python bench.py --no-cpu-baseline "$@" > $O/r2m_$tag.log 2> $O/r2m_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2m_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2m_$tag.err").read()[-1500:])
PY
}
NEW=$R/snark_amd/libark355.so; OLD=$R/variants/lib_base.so
for rep in a b; do
  b old_4_$rep $OLD --steps 16 --warmup 4
  b new_4_$rep $NEW --steps 16 --warmup 4
done
b old_1 $OLD --inflight 1 --steps 8 --warmup 2
b new_1 $NEW --inflight 1 --steps 8 --warmup 2
b old_n18 $OLD --log-n 18 --inflight 8 --steps 48 --warmup 8
b new_n18 $NEW --log-n 18 --inflight 8 --steps 48 --warmup 8
b old_bn $OLD --curve bn254 --steps 16 --warmup 4
b new_bn $NEW --curve bn254 --steps 16 --warmup 4
for v in old new; do
  lib=$OLD; [ $v = new ] && lib=$NEW
  ARK355_LIB=$lib timeout 600 python tools/msm_bench.py --min-log 16 --max-log 20 --step 2 --reps 5 --groups 1,2 --dists uniform > $O/r2m_msm_$v.txt 2>&1
  grep -h "n=2" $O/r2m_msm_$v.txt | sed "s/^/$v /"
done
exit 0
