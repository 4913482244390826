#!/bin/bash
# round 2, run H: bucket-accumulation knobs as library variants (tools/build_variant.sh):
#   base  no key/index prefetch, conversion inside the flush          pf1/pf2  ARK_G2L28_PREFETCH = 1 / 2 (+ G1 key prefetch)
#   w3  lz1 with the G1 kernel sized for three waves per SIMD (168 VGPRs)
#   lz1/lz2  the same with ARK_LAZY_FLUSH (raw 28-bit runs, converted once in front of the merge)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
# correctness of the lazy flush on the device first (MSM edge cases, small proofs on both curves, a 2^16 proof)
ARK355_LIB=$R/variants/lib_lz2.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "msm_vs_naive or exceptional or alternate_limb or known_dlog or prove_small or dummy or batch_in_flight" > $O/r2h_pytest_lz2.txt 2>&1
tail -n 3 $O/r2h_pytest_lz2.txt
b() { tag=$1; lib=$2; shift 2; timeout 300 env ARK355_LIB=$R/variants/lib_$lib.so python bench.py --no-cpu-baseline "$@" > $O/r2h_$tag.log 2> $O/r2h_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2h_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2h_$tag.err").read()[-1500:])
PY
}
for rep in a b; do
  for v in base pf1 pf2 lz1 lz2 w3; do
    b ${v}_4_$rep $v --steps 16 --warmup 4
  done
done
for v in base pf1 pf2 lz1 lz2 w3; do
  b ${v}_1 $v --inflight 1 --steps 8 --warmup 2
done
exit 0
