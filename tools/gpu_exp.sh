#!/bin/bash
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/prio_range.log 2>&1
import torch
print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')
PY
ARK355_STREAM_PRIO=2 timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/ab_salow_3.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/ab_prio_3.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --inflight 4 --steps 12 --warmup 4 > gpurun_out/ab_prio4_3.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --inflight 2 --steps 12 --warmup 4 > gpurun_out/ab_prio2_3.log 2>&1
exit 0
