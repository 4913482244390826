#!/bin/bash
# round 2, run Q: does the HIP runtime spin while a host thread waits for the GPU on this platform?  (process CPU time vs wall)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
probe() { env "$@" python - <<'PY'
import os, time, torch
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
torch.cuda.synchronize()
for mode in ("synchronize", "event"):
    c0, t0 = time.process_time(), time.perf_counter()
    for _ in range(200):
        y = x @ x
    if mode == "event":
        e = torch.cuda.Event(blocking=True); e.record(); e.synchronize()
    else:
        torch.cuda.synchronize()
    dt, dc = time.perf_counter() - t0, time.process_time() - c0
    print("  torch wait via %-12s wall %.3f s  cpu %.3f s  -> %.2f cores" % (mode, dt, dc, dc / dt))
PY
}
echo "default"; probe A=1
echo "HSA_ENABLE_INTERRUPT=1"; probe HSA_ENABLE_INTERRUPT=1
echo "AMD_DIRECT_DISPATCH=0"; probe AMD_DIRECT_DISPATCH=0
echo "ROC_ACTIVE_WAIT_TIMEOUT=1"; probe ROC_ACTIVE_WAIT_TIMEOUT=1
b() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --steps 24 --warmup 4 > $O/r2q_$tag.log 2> $O/r2q_$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r2q_$tag.log") if l.startswith("{")][0])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "host cpu cores %.2f" % d["host_cpu_cores"])
except Exception as e:
    print("$tag FAILED", e); print(open("$O/r2q_$tag.err").read()[-800:])
PY
}
b default A=1
b nodirect AMD_DIRECT_DISPATCH=0
exit 0
