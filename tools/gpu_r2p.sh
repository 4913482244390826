#!/bin/bash
# round 2, run P: the proving thread polls its last event and sleeps in between (default) instead of spinning inside the
# HIP runtime (ARK355_WAIT_SPIN=1): host CPU cores burnt during the bench, and the end-to-end pipeline with synthesis in the loop
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; shift; timeout 300 env ${ENVX} python bench.py --no-cpu-baseline "$@" > $O/r2p_$tag.log 2> $O/r2p_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2p_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cpu cores %.2f" % d["host_cpu_cores"], "solo total %.2f" % d["phases_ms"]["total_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2p_$tag.err").read()[-1500:])
PY
}
for mode in spin poll; do
  ENVX="A=1"; [ $mode = spin ] && ENVX="ARK355_WAIT_SPIN=1"
  b ${mode}_if4_a --steps 40 --warmup 4
  b ${mode}_if4_b --steps 40 --warmup 4
  b ${mode}_if1 --inflight 1 --steps 20 --warmup 4
  b ${mode}_n18 --log-n 18 --inflight 8 --steps 64 --warmup 8
  env ${ENVX} ARK355_E2E_SWEEP="4,8,12" timeout 600 tests/cpp/test_host_mirror --e2e bls12_381 1048576 48 6 4 > $O/r2p_e2e_$mode.txt 2> $O/r2p_e2e_$mode.err
  echo "--- e2e $mode"; cat $O/r2p_e2e_$mode.txt; tail -n 3 $O/r2p_e2e_$mode.err
done
exit 0
