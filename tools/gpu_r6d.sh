#!/bin/bash
# round 6, GPU run D (one box): the lane-pair G2 accumulator in LDS.  Same-box A/B of how many of its coordinates live there
# (ARK_G2L28_LDS_VALUES: 0 = registers as in rounds 2-5 / 2 = zz, zzz / 3 = + x / 4 = + y, the in-tree library), parity of the
# in-tree library, one serial kernel trace per variant (G2 launch duration), then the driver's command with the new bench line
# (in-run multiplier peak, box-normalised figures, oracle cross-check in the cpu_baseline leg, e2e = S2 only).
R=$PWD; O=$R/gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_tables.py -m gpu -x -q > $O/pytest_parity.txt 2>&1; echo "pytest rc=$?" >> $O/status.txt
tail -n 3 $O/pytest_parity.txt
run() { # tag lib -- bench args
  tag=$1; lib=$2; shift; shift; shift
  ARK355_LIB=$lib ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-telemetry --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6d/%s.json" % tag))
    lat = {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}
    iso = d.get("isolated") or {}
    box = d.get("box") or {}
    print("%-12s %7.3f ms/step (normalised %.3f, peak %s)  acc/proof %.2f  iso_total %.2f  lat %s  parity %s" % (tag, d["ms_per_step"], box.get("ms_per_step_normalised", 0), box.get("mad_peak_t_measured"), iso.get("accumulate_ms_per_proof", 0), iso.get("total_ms", 0), lat, str(d.get("parity"))[:10]))
except Exception as e:
    print(tag, "FAILED", e)
PY
}
for rep in 1 2; do
for v in 4:$R/snark_amd/libark355.so 0:$R/variants/lib_g2lds0.so 3:$R/variants/lib_g2lds3.so 2:$R/variants/lib_g2lds2.so; do
  run lds${v%%:*}_$rep ${v#*:} -- --steps 12 --warmup 4
done
done
run lds4_n18 $R/snark_amd/libark355.so -- --log-n 18 --inflight 8 --steps 48 --warmup 8
run lds0_n18 $R/variants/lib_g2lds0.so -- --log-n 18 --inflight 8 --steps 48 --warmup 8
cd /tmp && export TMPDIR=/tmp
for v in 4:$R/snark_amd/libark355.so 0:$R/variants/lib_g2lds0.so 3:$R/variants/lib_g2lds3.so; do
  tag=lds${v%%:*}; lib=${v#*:}
  ARK355_LIB=$lib ARK355_SERIAL=1 ARK355_SIDE_WM=0 ARK355_SIDE_G2_TAILS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$tag -o serial -- python $R/bench.py --profile-run --inflight 1 --steps 5 --warmup 2 > $O/serial_$tag.log 2>&1; echo "serial $tag rc=$?" >> $O/status.txt
  find $O/serial_$tag -name "*kernel_trace.csv" -delete; find $O/serial_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/serial_${tag}_kernel_stats.csv
  grep -i "accumulate" $O/serial_${tag}_kernel_stats.csv | sed 's/void ark355:://; s/<ark355::BlsFqParams[^"]*"/"/' | cut -c1-150
done
cd $R
ARK355_BENCH_WATCHDOG=400 timeout 420 python bench.py > $O/bench_driver.json 2> $O/bench_driver.log; echo "driver bench rc=$?" >> $O/status.txt
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6d/bench_driver.json"))
    print("driver cmd: ms_per_step", d["ms_per_step"], "value", d["value"], "box", json.dumps(d.get("box"))[:600])
    print("alu", json.dumps(d["roofline"]["alu"])[:700])
    print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:700])
    print("e2e", json.dumps(d.get("e2e"))[:500])
except Exception as e: print("driver bench parse failed", e)
PY
exit 0
