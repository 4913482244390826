#!/bin/bash
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
exit 0
