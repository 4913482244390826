#!/bin/bash
# round 6, GPU run O (one box): EXPERIMENT -- the G1 accumulation at three waves per SIMD (x, zz, zzz of the accumulator in LDS, no
# row prefetch; variants/lib_g1w3.so, -DARK_G1_W3=1) against the shipped two-wave kernel.  Same box, interleaved.
R=$PWD; O=$R/gpurun_out/r6o; mkdir -p $O
ARK355_LIB=$R/variants/lib_g1w3.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or prove or resident" > $O/pytest_w3.txt 2>&1; echo "pytest w3 rc=$?" >> $O/status.txt
tail -n 2 $O/pytest_w3.txt
run() { tag=$1; lib=$2; shift; shift; shift
  ARK355_LIB=$lib ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6o/%s.json" % tag))
    lat = {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}
    box = d.get("box") or {}; iso = d.get("isolated") or {}
    print("%-10s %7.3f ms/step  cyc/constraint %.2f  acc/proof %.2f (H launch %.2f)  iso_total %.2f  lat %s  %s" % (tag, d["ms_per_step"], box.get("gfx_cycles_per_constraint", 0), iso.get("accumulate_ms_per_proof", 0), iso.get("accumulate_h_launch_ms", 0), iso.get("total_ms", 0), lat, str(d.get("parity"))[:10]))
except Exception as e: print(tag, "FAILED", e)
PY
}
for rep in 1 2; do
run w2_$rep $R/snark_amd/libark355.so -- --steps 12 --warmup 4
run w3_$rep $R/variants/lib_g1w3.so -- --steps 12 --warmup 4
done
run w2_if1 $R/snark_amd/libark355.so -- --inflight 1 --steps 6 --warmup 2
run w3_if1 $R/variants/lib_g1w3.so -- --inflight 1 --steps 6 --warmup 2
cd /tmp && export TMPDIR=/tmp
for v in w2:$R/snark_amd/libark355.so w3:$R/variants/lib_g1w3.so; do
  tag=${v%%:*}; lib=${v#*:}
  ARK355_LIB=$lib ARK355_SERIAL=1 ARK355_SIDE_WM=0 ARK355_SIDE_G2_TAILS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$tag -o serial -- python $R/bench.py --profile-run --inflight 1 --steps 5 --warmup 2 > $O/serial_$tag.log 2>&1; echo "serial $tag rc=$?" >> $O/status.txt
  find $O/serial_$tag -name "*kernel_trace.csv" -delete; find $O/serial_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/serial_${tag}_kernel_stats.csv
  grep -i "accumulate" $O/serial_${tag}_kernel_stats.csv | sed 's/void ark355:://; s/<ark355::BlsFqParams[^"]*"/"/' | cut -c1-150
done
exit 0
