#!/bin/bash
# round 6, GPU run C (one box): the six-transform witness map (c leaves after its inverse transform) -- parity at N = 2^21..2^23 incl.
# the distributed map, then the same-box A/B of the NTT kernels' register budget: ARK_NTT_WAVES = 3 (168 VGPRs, ~40 spilled; shipped)
# against 2 (256 VGPRs, no spills; variants/lib_nttw2.so), and a serial kernel trace of both.
R=$PWD; O=$R/gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_o3_large.py tests/test_gpu_parity.py -m gpu -x -q -k "witness or ntt or prove or s2_2p20" > $O/pytest_wm.txt 2>&1; echo "pytest rc=$?" >> $O/status.txt
tail -n 4 $O/pytest_wm.txt
run() { # tag lib -- bench args
  tag=$1; lib=$2; shift; shift; shift
  ARK355_LIB=$lib ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-telemetry --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6c/%s.json" % tag))
    lat = {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}
    iso = d.get("isolated") or {}
    print("%-16s %7.3f ms/step  acc/proof %.2f  iso_total %.2f  wm %.3f  lat %s  parity %s" % (tag, d["ms_per_step"], iso.get("accumulate_ms_per_proof", 0), iso.get("total_ms", 0), iso.get("witness_map_ms", 0), lat, str(d.get("parity"))[:10]))
except Exception as e:
    print(tag, "FAILED", e)
PY
}
for rep in 1 2; do
run w3_$rep $R/snark_amd/libark355.so -- --steps 12 --warmup 4
run w2_$rep $R/variants/lib_nttw2.so -- --steps 12 --warmup 4
done
run w3_n18 $R/snark_amd/libark355.so -- --log-n 18 --inflight 8 --steps 48 --warmup 8
run w2_n18 $R/variants/lib_nttw2.so -- --log-n 18 --inflight 8 --steps 48 --warmup 8
cd /tmp && export TMPDIR=/tmp
for v in w3:$R/snark_amd/libark355.so w2:$R/variants/lib_nttw2.so; do
  tag=${v%%:*}; lib=${v#*:}
  ARK355_LIB=$lib ARK355_SERIAL=1 ARK355_SIDE_WM=0 ARK355_SIDE_G2_TAILS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$tag -o serial -- python $R/bench.py --profile-run --inflight 1 --steps 5 --warmup 2 > $O/serial_$tag.log 2>&1; echo "serial $tag rc=$?" >> $O/status.txt
  find $O/serial_$tag -name "*kernel_trace.csv" -delete; find $O/serial_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/serial_${tag}_kernel_stats.csv
  grep -i "ntt\|qap\|spmv" $O/serial_${tag}_kernel_stats.csv | cut -c1-60,100-220
done
exit 0
