#!/bin/bash
# round 6, GPU run E (evidence of the tree it is run on; one box).  tools/gpu_r6e.sh [tests] [bench] [profiles] [sweeps] [rccl]
# (no argument: everything).  Outputs under gpurun_out/r6t/; the summaries that are to be judged are copied to profiles/ by hand.
R=$PWD; O=$R/gpurun_out/r6t; mkdir -p $O
want() { [ $# -eq 0 ] && return 0; for a in "${ARGS[@]}"; do [ "$a" = "$1" ] && return 0; done; [ ${#ARGS[@]} -eq 0 ]; }
ARGS=("$@")
python tools/gpu_telemetry.py > $O/telemetry_start.txt 2>&1
if want tests; then
  t0=$(date +%s)
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$? ($(( $(date +%s) - t0 )) s)" >> $O/status.txt
  tail -n 4 $O/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/status.txt
  tail -n 2 $O/smoke.txt
fi
if want bench; then
  t0=$(date +%s)
  ARK355_BENCH_WATCHDOG=400 timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "driver bench rc=$? ($(( $(date +%s) - t0 )) s)" >> $O/status.txt
  python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6t/bench_default.json"))
    print("driver cmd: ms_per_step %.3f value %.0f parity %s" % (d["ms_per_step"], d["value"], d["parity"]))
    print("box", json.dumps(d.get("box"))[:900])
    print("alu", json.dumps({k: v for k, v in d["roofline"]["alu"].items() if k in ("achieved", "peak", "frac", "mads_per_add")}))
    print("latency", {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")})
    print("isolated", {k: d["isolated"][k] for k in ("accumulate_ms_per_proof", "total_ms", "witness_map_ms")})
    print("micro", {k: v["ms"] for k, v in d["micro"]["msm"].items()})
    print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
    print("e2e", json.dumps(d.get("e2e"))[:400])
except Exception as e: print("driver bench parse failed", e)
PY
fi
cd /tmp && export TMPDIR=/tmp
if want profiles; then
  ARK355_SERIAL=1 ARK355_SIDE_WM=0 ARK355_SIDE_G2_TAILS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --profile-run --inflight 1 --steps 5 --warmup 2 > $O/serial_bench.log 2>&1; echo "serial rc=$?" >> $O/status.txt
  find $O/serial -name "*kernel_trace.csv" -delete; find $O/serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/serial_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inflight -o inflight -- python $R/bench.py --profile-run --no-check --steps 16 --warmup 4 > $O/inflight_bench.log 2>&1; echo "inflight prof rc=$?" >> $O/status.txt
  find $O/inflight -name "*kernel_trace.csv" -delete; find $O/inflight -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/inflight_kernel_stats.csv
  pmc() {   # workload-key bench-args...
    key=$1; shift
    rm -rf $O/pmc/prof_fetch $O/pmc/prof_write; mkdir -p $O/pmc/prof_fetch $O/pmc/prof_write
    ARK355_SCHED=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/prof_fetch -o f -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 1 --warmup 0 "$@" > $O/pmc_fetch.log 2>&1
    echo "pmc fetch $key rc=$?" >> $O/status.txt
    ARK355_SCHED=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/prof_write -o w -- python $R/bench.py --profile-run --no-check --inflight 1 --steps 1 --warmup 0 "$@" > $O/pmc_write.log 2>&1
    echo "pmc write $key rc=$?" >> $O/status.txt
    (cd $R && python tools/pmc_summary.py --dir $O/pmc --json $O/pmc_latest.json --merge --workload "$key" --recorded "round 6 run T, final library" >> $O/pmc_summary.txt 2>&1)
  }
  pmc "bls12_381:n=1048576"
  pmc "bn254:n=1048576" --curve bn254
  pmc "bls12_381:n=262144" --log-n 18
  rm -rf $O/pmc
fi
if want rccl; then
  # the kernels of the rank-with-itself exchange (policy RCCL_SELF) by name: RCCL's send / receive kernels next to the library's own
  ARK355_RCCL_SELF_BIG=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rccl_self -o rccl -- python $R/tests/rccl_single_rank.py > $O/rccl_self_prof.log 2>&1; echo "rccl self prof rc=$?" >> $O/status.txt
  find $O/rccl_self -name "*kernel_trace.csv" -delete; find $O/rccl_self -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rccl_self_kernel_stats.csv
  grep -i "nccl\|rccl\|dwm_\|slot28_add" $O/rccl_self_kernel_stats.csv | cut -c1-200
fi
cd $R
if want sweeps; then
  run() { tag=$1; shift; ARK355_BENCH_WATCHDOG=280 timeout 300 python bench.py --no-cpu-baseline --no-e2e "$@" > $O/$tag.json 2> $O/$tag.log; echo "$tag rc=$?" >> $O/status.txt
    python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6t/%s.json" % tag))
    lat = {k: round(v, 2) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms")}
    box = d.get("box") or {}
    print("%-18s %8.3f ms/step  %6.2f M/s  cycles/constraint %s  at_ref_clock %s ms  lat %s" % (tag, d["ms_per_step"], d["value"] / 1e6, box.get("gfx_cycles_per_constraint"), box.get("ms_per_step_at_ref_clock"), lat))
except Exception as e: print(tag, "FAILED", e)
PY
  }
  run bench_if1 --inflight 1 --no-micro --no-ab
  run bench_if8 --inflight 8 --no-micro --no-ab
  run bench_n18_if8 --log-n 18 --inflight 8 --steps 48 --warmup 8 --no-micro --no-ab
  run bench_n18_if1 --log-n 18 --inflight 1 --no-micro --no-ab
  run bench_bn254 --curve bn254 --no-micro --no-ab
  run bench_tight --tight --no-micro --no-ab
  run bench_n22_if2 --log-n 22 --inflight 2 --steps 6 --warmup 2 --no-micro --no-ab
  run bench_shard22_w1 --mode shard --steps 6 --warmup 2 --no-micro --no-ab
  timeout 100 python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 --wm dist --steps 8 > $O/shard_rank_22.json 2> $O/shard_rank.log; echo "shard rc=$?" >> $O/status.txt
  python - <<'PY'
import json
for l in open("gpurun_out/r6t/shard_rank_22.json"):
    try:
        d = json.loads(l); print("rank path shard", d["shard"], "median", d["ms_median"], "acc", d["accumulate_ms"])
    except Exception: pass
PY
  timeout 400 python tools/msm_bench.py --min-log 16 --max-log 22 > $O/msm_microbench.txt 2>&1; echo "msm_bench rc=$?" >> $O/status.txt
  tail -n 30 $O/msm_microbench.txt
fi
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
cat $O/status.txt
exit 0
