#!/bin/bash
# round 6, GPU run G: the driver's command and two short default runs on whatever box comes -- one more sample of the
# box-independent figures (gfx cycles per constraint from the per-XCD probe) to set against run F's box.
R=$PWD; O=$R/gpurun_out/r6h; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
for t in a b; do
ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab > $O/default_$t.json 2> $O/default_$t.log
done
ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab --inflight 1 > $O/if1.json 2> $O/if1.log
ARK355_BENCH_WATCHDOG=400 timeout 420 python bench.py > $O/bench_driver.json 2> $O/bench_driver.log; echo "driver bench rc=$?" >> $O/status.txt
python - <<'PY'
import json
for t in ("default_a", "default_b", "if1", "bench_driver"):
    try:
        d = json.load(open("gpurun_out/r6h/%s.json" % t)); b = d["box"]
        print("%-13s %7.3f ms/step  cyc/constraint %.2f  clk on chip %.0f smi %s  at_ref %.3f  lat %s" % (t, d["ms_per_step"], b["gfx_cycles_per_constraint"], b["gfxclk_mhz_mean_on_chip"], b.get("gfxclk_mhz_mean_smi"), b["ms_per_step_at_ref_clock"], {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}))
    except Exception as e: print(t, "FAILED", e)
PY
grep -m3 -i "bdf\|serial\|uuid" $O/telemetry.txt
exit 0
