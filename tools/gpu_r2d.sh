#!/bin/bash
# round 2, run D: new NTT (register radix-8 groups, direct tables, fused seam): parity, bench, isolated kernel durations; e2e sweep
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; nproc; grep -c processor /proc/cpuinfo) > $O/r2d_cpu.log 2>&1; cat $O/r2d_cpu.log | tr '\n' ' '; echo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_o3_large.py -m gpu -x -q -k "not resident_msm_vs_o3 and not s3_bench and not s1_dummy and not tight" > $O/r2d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r2d_pytest.log
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --steps 12 --warmup 3 ${EXTRA} > $O/r2d_$tag.log 2> $O/r2d_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2d_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], {k: round(v,2) for k,v in d["phases_ms"].items()}, d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2d_$tag.err").read()[-1500:])
PY
}
EXTRA="--inflight 1" b if1 A=1
EXTRA="" b if4 A=1
EXTRA="--inflight 1" b if1_nofuse ARK355_NTT_NOFUSE=1
mkdir -p $O/r2d_serial; cd /tmp && export TMPDIR=/tmp
ARK355_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2d_serial -o serial -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 5 --warmup 2 > $O/r2d_serial/bench.log 2>&1
cd $R; find $O/r2d_serial -name "*kernel_trace.csv" -delete; f=$(find $O/r2d_serial -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -25 "$f" | cut -c1-150
ARK355_E2E_SWEEP="3,4,6,12,16" timeout 500 tests/cpp/test_host_mirror --e2e bls12_381 1048576 24 8 4 > $O/r2d_e2e.log 2>&1; echo "e2e rc=$?"; cat $O/r2d_e2e.log
