#!/bin/bash
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
exit 0
