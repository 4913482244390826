#!/bin/bash
# round 6, GPU run F (one box): the new GPU tests (CHECK_SATISFIED, box diagnostics, the G1 tails of a lone proof aside), the same-box
# A/B of policy SIDE_G1_TAILS on a lone proof (tails of A, B1, L' as a batch of three under the H accumulation), and the driver's
# command with the per-XCD cycle counter (box-independent cycles per constraint: to be compared with the other boxes of the round).
R=$PWD; O=$R/gpurun_out/r6f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "check_satisfied or box_diagnostics or tail_variants or one_stream" > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/status.txt
tail -n 3 $O/pytest_new.txt
run() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6f/%s.json" % tag))
    lat = {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}
    box = d.get("box") or {}
    iso = d.get("isolated") or {}
    print("%-14s %7.3f ms/step  cyc/constraint %.2f  clk on chip %.0f smi %s  at_ref %.3f  iso_total %.2f  lat %s" % (tag, d["ms_per_step"], box.get("gfx_cycles_per_constraint", 0), box.get("gfxclk_mhz_mean_on_chip", 0), box.get("gfxclk_mhz_mean_smi"), box.get("ms_per_step_at_ref_clock", 0), iso.get("total_ms", 0), lat))
except Exception as e: print(tag, "FAILED", e)
PY
}
for rep in 1 2 3; do
run g1side0_$rep ARK355_SIDE_G1_TAILS=0 -- --inflight 1 --steps 8 --warmup 2
run g1side1_$rep ARK355_SIDE_G1_TAILS=1 -- --inflight 1 --steps 8 --warmup 2
done
run n18_side0 ARK355_SIDE_G1_TAILS=0 -- --log-n 18 --inflight 1 --steps 16 --warmup 4
run n18_side1 ARK355_SIDE_G1_TAILS=1 -- --log-n 18 --inflight 1 --steps 16 --warmup 4
run default_a ARK355_SIDE_G1_TAILS=0 --
run bn254 ARK355_SIDE_G1_TAILS=0 -- --curve bn254
run default_b ARK355_SIDE_G1_TAILS=0 --
ARK355_BENCH_WATCHDOG=400 timeout 420 python bench.py > $O/bench_driver.json 2> $O/bench_driver.log; echo "driver bench rc=$?" >> $O/status.txt
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6f/bench_driver.json"))
    print("driver cmd: ms_per_step %.3f value %.0f" % (d["ms_per_step"], d["value"]), "box", json.dumps(d.get("box"))[:700])
except Exception as e: print("driver bench parse failed", e)
PY
cat $O/status.txt
exit 0
