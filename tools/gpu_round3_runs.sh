#!/bin/bash
# Round-3 GPU runs, one parametrised script (the round-2 one-offs tools/gpu_r2?.sh were removed; they are in the git history
# up to commit 363ad0d): tools/gpu_round3_runs.sh <a|b|c|d|e|f|g>.  Evidence of the final tree: tools/gpu_round3_final.sh.
#   a  window size c = 16 vs 17, same box            -> profiles/r03_c17_ab.txt
#   b  first run of the configs[2] tests + large benches (the benches after pytest hung on that box; see c, d)
#   c  bench.py with stage markers + watchdog (diagnostic)
#   d  what leaves a box in a bad state? (in-process RCCL probe, 2^22 bench, bench again)
#   e  adaptive wait A/B, NTT pair stores A/B (serial stats + WRITE_SIZE), 2^18 x 8 timeline -> profiles/r03_ntt_pair_store_ab.txt
#   f  host cost of HIP calls, host time to queue a proof, NTT kernels at two waves per SIMD -> profiles/r03_host_cpu.txt
#   g  proof epilogue without stream synchronisations, GPU_MAX_HW_QUEUES                  -> profiles/r03_epilogue_ab.txt
#   h  validation of the final epilogue / wait policy (BN254, default)                    -> profiles/r03_epilogue_ab.txt
#   jkl  one stream per proof against the pipeline, several proofs in flight; final library -> profiles/r03_one_stream_ab.txt
stage=$1
run_a() {
  # round 3, run A (prepared at the end of round 2, not yet run): same-box A/B of the window size for resident keys,
  # c = 16 (default) against ARK355_MSM_C=17 (MsmPlan::negate_high: 15 windows for 255-bit scalars; BN254's 254-bit scalars
  # need no negation at c = 17), interleaved a/b/a/b, four proofs in flight and one, BLS12-381 / BN254 / 2^18.
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "window_17 or msm_vs_naive or exceptional" > $O/r3a_pytest.txt 2>&1; tail -n 2 $O/r3a_pytest.txt
  b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3a_$tag.log 2> $O/r3a_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3a_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], "host cores %.2f" % d["host_cpu_cores"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3a_$tag.err").read()[-1500:])
PY
  }
for rep in a b c; do
    EXTRA="--steps 32 --warmup 6"
    b c16_if4_$rep A=1
    b c17_if4_$rep ARK355_MSM_C=17
  done
  EXTRA="--inflight 1 --steps 12 --warmup 3"
  b c16_if1 A=1
  b c17_if1 ARK355_MSM_C=17
  EXTRA="--curve bn254 --steps 32 --warmup 6"
  b c16_bn A=1
  b c17_bn ARK355_MSM_C=17
  EXTRA="--log-n 18 --inflight 8 --steps 64 --warmup 8"
  b c16_n18 A=1
  b c17_n18 ARK355_MSM_C=17

}
run_b() {
  # round 3, run B: configs[2] parity tests (2^22 proofs whole + sharded, large NTT / witness map vs oracle/c), the
  # SURVEY-8d latency fields of the bench line, shard-mode bench at its default 2^22, 2^22 replica, a 2^23-constraint key
  # (the reference's own bench size) on one GPU, and the cost of window stride 2 at 2^20.
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  nproc > $O/r3b_host.txt; cat /sys/fs/cgroup/cpu.max >> $O/r3b_host.txt 2>/dev/null; free -g >> $O/r3b_host.txt
  timeout 1500 python -m pytest tests/test_gpu_o3_large.py -x -q -m gpu -s --durations=12 \
    -k "2p22 or ntt_large or witness_map_large or s2_2p20_bls12_381_vs_o3" > $O/r3b_pytest.txt 2>&1; tail -n 25 $O/r3b_pytest.txt
  b() { tag=$1; shift; timeout 900 env "$@" python bench.py ${EXTRA} > $O/r3b_$tag.log 2> $O/r3b_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3b_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "acc avg launch %.2f ms" % d["roofline"]["avg_launch_ms"], "solo total %.2f" % d["phases_ms"]["total_ms"], "host cores %.2f" % d["host_cpu_cores"], d["parity"], "prep %.1f s" % d["prep_s"])
    print("   threads", d.get("host_cpu_threads")); print("   latency", d.get("latency"))
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3b_$tag.err").read()[-1500:])
PY
  }
  EXTRA="--no-cpu-baseline --steps 20 --warmup 5"
  b default A=1
  EXTRA="--no-cpu-baseline --mode shard --steps 8 --warmup 2"
  b shard22_window A=1
  EXTRA="--no-cpu-baseline --mode shard --shard-exchange ring --steps 8 --warmup 2"
  b shard22_ring A=1
  EXTRA="--no-cpu-baseline --log-n 22 --inflight 2 --steps 8 --warmup 2"
  b n22_if2 A=1
  EXTRA="--no-cpu-baseline --log-n 23 --inflight 1 --steps 4 --warmup 1"
  b n23_if1 ARK355_TRACE_HOST=1
  grep "window table" $O/r3b_n23_if1.err | head -5
  EXTRA="--no-cpu-baseline --steps 20 --warmup 5"
  b stride2 ARK355_TABLE_STRIDE=2

}
run_c() {
  # round 3, run C (diagnostic): where does bench.py stop?  Stage markers on stderr + a watchdog that dumps every thread's
  # Python stack.
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  ARK355_BENCH_WATCHDOG=150 timeout 200 python bench.py --no-cpu-baseline --steps 8 --warmup 2 > $O/r3c_default.log 2> $O/r3c_default.err
  echo "rc=$?"; grep -v amdgpu.ids $O/r3c_default.err | tail -60; cut -c1-600 $O/r3c_default.log

}
run_d() {
  # round 3, run D (diagnostic): what leaves the box in a state where the next process hangs?
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  state() { echo "--- $1"; ps -eo pid,stat,etime,cmd | grep -v "ps -eo" | grep -E "python|pytest" | head; rocm-smi --showmemuse 2>/dev/null | grep -E "GPU\[|%" | head -4; ls /dev/shm | head; }
  timeout 120 python tools/rccl_inproc_probe.py > $O/r3d_probe.log 2>&1; echo "probe rc=$?"; tail -2 $O/r3d_probe.log
  state "after in-process RCCL probe"
  ARK355_BENCH_WATCHDOG=100 timeout 150 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/r3d_b1.log 2> $O/r3d_b1.err; echo "bench1 rc=$?"; grep "bench " $O/r3d_b1.err | tail -3
  ARK355_BENCH_WATCHDOG=200 timeout 260 python bench.py --no-cpu-baseline --log-n 22 --inflight 1 --steps 2 --warmup 1 > $O/r3d_b22.log 2> $O/r3d_b22.err; echo "bench22 rc=$?"; grep "bench " $O/r3d_b22.err | tail -4
  state "after 2^22 bench"
  ARK355_BENCH_WATCHDOG=100 timeout 150 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/r3d_b2.log 2> $O/r3d_b2.err; echo "bench2 rc=$?"; grep "bench " $O/r3d_b2.err | tail -3

}
run_e() {
  # round 3, run E: A/B of the adaptive wait (host cores) at 2^20 x 4 and 2^18 x 8 in flight; NTT pair stores (serial kernel
  # stats + WRITE_SIZE); timeline of 2^18 x 8 in flight.  Every step under its own timeout.
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  b() { tag=$1; shift; ARK355_BENCH_WATCHDOG=150 timeout 170 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3e_$tag.log 2> $O/r3e_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3e_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cores %.2f" % d["host_cpu_cores"], d["host_cpu_threads"], "solo %.2f" % d["phases_ms"]["total_ms"], "lat pinned %.2f dev %.2f" % (d["latency"]["host_pinned_z_ms"], d["latency"]["device_z_ms"]), d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3e_$tag.err").read()[-800:])
PY
  }
  EXTRA="--steps 24 --warmup 4"
  b adapt1_a A=1
  b adapt0_a ARK355_WAIT_ADAPT=0
  b adapt1_b A=1
  EXTRA="--log-n 18 --inflight 8 --steps 64 --warmup 8"
  b n18_adapt1 A=1
  b n18_adapt0 ARK355_WAIT_ADAPT=0
  cd /tmp && export TMPDIR=/tmp
for ps in 1 0; do
    ARK355_SERIAL=1 ARK355_NTT_PAIR_STORE=$ps timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3e_serial_ps$ps -o s -- python $R/bench.py --no-cpu-baseline --no-check --inflight 1 --steps 4 --warmup 1 > $O/r3e_serial_ps$ps.log 2>&1
    f=$(find $O/r3e_serial_ps$ps -name "*kernel_stats.csv" | head -1); echo "pair_store=$ps"; grep -E "ntt_|qap_|spmv|accumulate" $f | cut -d, -f1-4 | cut -c1-150 | head -12
    ARK355_NTT_PAIR_STORE=$ps timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r3e_write_ps$ps -o w -- python $R/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-check > $O/r3e_write_ps$ps.log 2>&1
    python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$O/r3e_write_ps$ps/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row.get("Kernel_Name", "")).replace("ark355::", "")[:60]
        if "ntt" in k:
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
for k, (n, v) in sorted(agg.items()):
    print("  WRITE_SIZE pair_store=$ps %-60s dispatches=%d avg_KiB=%.0f" % (k, n, v / n))
PY
  done
  timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/r3e_trace18 -o t -- python $R/bench.py --no-cpu-baseline --no-check --log-n 18 --inflight 8 --steps 32 --warmup 8 > $O/r3e_trace18.log 2>&1
  cd $R
  python tools/trace_analyze.py $(find $O/r3e_trace18 -name "*kernel_trace.csv" | head -1) 32 > $O/r3e_timeline18.txt 2>&1; cat $O/r3e_timeline18.txt | head -30
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +5M -delete

}
run_f() {
  # round 3, run F: host cost of the HIP calls a proof is made of; host time to queue one proof (ARK355_TRACE_HOST);
  # NTT pass kernels sized for two waves per SIMD (variants/lib_ntt2.so: no scratch) against three (default).
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  timeout 120 tools/host_api_cost.bin 3000 > $O/r3f_host_api.txt 2>&1; cat $O/r3f_host_api.txt
for inf in 1 4; do
    ARK355_TRACE_HOST=1 timeout 120 python bench.py --no-cpu-baseline --no-check --inflight $inf --steps 8 --warmup 4 > $O/r3f_trace_if$inf.log 2> $O/r3f_trace_if$inf.err
    echo "inflight $inf:"; grep "prove host wall" $O/r3f_trace_if$inf.err | tail -6 | cut -c1-220
  done
  b() { tag=$1; shift; ARK355_BENCH_WATCHDOG=150 timeout 170 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3f_$tag.log 2> $O/r3f_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3f_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cores %.2f" % d["host_cpu_cores"], "solo %.2f wm %.2f" % (d["phases_ms"]["total_ms"], d["phases_ms"]["witness_map_ms"]), d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3f_$tag.err").read()[-800:])
PY
  }
  EXTRA="--steps 24 --warmup 4"
  b ntt3_a A=1
  b ntt2_a ARK355_LIB=$R/variants/lib_ntt2.so
  b ntt3_b A=1
  b ntt2_b ARK355_LIB=$R/variants/lib_ntt2.so
  cd /tmp && export TMPDIR=/tmp
  ARK355_SERIAL=1 ARK355_LIB=$R/variants/lib_ntt2.so timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3f_serial_ntt2 -o s -- python $R/bench.py --no-cpu-baseline --no-check --inflight 1 --steps 4 --warmup 1 > $O/r3f_serial_ntt2.log 2>&1
  f=$(find $O/r3f_serial_ntt2 -name "*kernel_stats.csv" | head -1); echo "ntt2 (two waves per SIMD), serial:"; grep -E "ntt_" $f | cut -d, -f1-4 | sed 's/ark355:://g' | cut -c1-120 | head -6
  find $O -name "*kernel_trace.csv" -delete

}
run_g() {
  # round 3, run G: proof epilogue without stream synchronisations (new library) against the previous one
  # (variants/lib_ntt2.so: same prover with the synchronises), and the number of hardware queues the runtime spreads its
  # streams over (GPU_MAX_HW_QUEUES, default 4), at 2^20 x 4 and 2^18 x 8 in flight.
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  b() { tag=$1; shift; ARK355_BENCH_WATCHDOG=150 timeout 170 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3g_$tag.log 2> $O/r3g_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3g_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cores %.2f" % d["host_cpu_cores"], d["host_cpu_threads"], "solo %.2f" % d["phases_ms"]["total_ms"], "lat dev %.2f" % d["latency"]["device_z_ms"], d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3g_$tag.err").read()[-800:])
PY
  }
  EXTRA="--steps 24 --warmup 4"
  b new_a A=1
  b old_a ARK355_LIB=$R/variants/lib_ntt2.so
  b new_b A=1
  b old_b ARK355_LIB=$R/variants/lib_ntt2.so
  b new_q8 GPU_MAX_HW_QUEUES=8
  b new_q16 GPU_MAX_HW_QUEUES=16
  b new_q2 GPU_MAX_HW_QUEUES=2
  EXTRA="--log-n 18 --inflight 8 --steps 64 --warmup 8"
  b n18_new A=1
  b n18_old ARK355_LIB=$R/variants/lib_ntt2.so
  b n18_q8 GPU_MAX_HW_QUEUES=8
  b n18_q16 GPU_MAX_HW_QUEUES=16

}
run_h() {
R=$PWD; O=$R/gpurun_out; mkdir -p $O
b() { tag=$1; shift; ARK355_BENCH_WATCHDOG=120 timeout 140 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3i_$tag.log 2> $O/r3i_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r3i_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "value %.1fM" % (d["value"]/1e6), "host cores %.2f" % d["host_cpu_cores"], d["host_cpu_threads"], "solo %.2f" % d["phases_ms"]["total_ms"], "lat pinned %.2f dev %.2f" % (d["latency"]["host_pinned_z_ms"], d["latency"]["device_z_ms"]), d["parity"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r3i_$tag.err").read()[-800:])
PY
}
EXTRA="--curve bn254 --steps 24 --warmup 6"
b bn254_a A=1
EXTRA="--steps 20 --warmup 5"
b default A=1
EXTRA="--curve bn254 --steps 24 --warmup 6"
b bn254_b A=1
}
run_jkl() {
  # runs j, k, l: the whole proof on one stream (ARK355_SERIAL=1) against the five-stream pipeline, several proofs in flight;
  # l = the final library (schedule picked per proof), parity checks on
  R=$PWD; O=$R/gpurun_out; mkdir -p $O
  b() { tag=$1; shift; ARK355_BENCH_WATCHDOG=70 timeout 80 env "$@" python bench.py --no-cpu-baseline ${EXTRA} > $O/r3jkl_$tag.log 2> $O/r3jkl_$tag.err; grep -c "^{" $O/r3jkl_$tag.log; }
  EXTRA="--no-check --steps 32 --warmup 8 --inflight 4"; b n20_if4_pipe ARK355_SERIAL=0; b n20_if4_serial ARK355_SERIAL=1
  EXTRA="--no-check --steps 32 --warmup 8 --inflight 8"; b n20_if8_serial ARK355_SERIAL=1
  EXTRA="--no-check --curve bn254 --steps 32 --warmup 8 --inflight 6"; b bn_if6_serial ARK355_SERIAL=1
  EXTRA="--no-check --log-n 18 --inflight 8 --steps 64 --warmup 8"; b n18_if8_pipe ARK355_SERIAL=0; b n18_if8_serial ARK355_SERIAL=1
  EXTRA="--no-check --log-n 18 --inflight 12 --steps 96 --warmup 12"; b n18_if12_serial ARK355_SERIAL=1
  EXTRA="--steps 20 --warmup 5"; b default A=1
  EXTRA="--curve bn254 --steps 20 --warmup 5"; b bn254 A=1
  EXTRA="--log-n 18 --inflight 8 --steps 64 --warmup 8"; b n18_if8 A=1
}
case "$stage" in a|b|c|d|e|f|g|h|jkl) run_$stage ;; *) echo "usage: $0 <a|b|c|d|e|f|g|h|jkl>"; exit 2 ;; esac
