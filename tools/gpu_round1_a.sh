#!/bin/bash
# first GPU contact: microbench, parity tests, bench, rocprof
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.log 2>&1
nproc > gpurun_out/host.log; lscpu | grep -E "Model name|Socket|Thread|Core" >> gpurun_out/host.log
timeout 300 tools/ubench.bin > gpurun_out/ubench.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then
  ARK355_NO_TORCH=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt_vs_oracle or msm_vs_naive" > gpurun_out/pytest_gpu_notorch.log 2>&1
fi
timeout 600 python bench.py --log-n 16 --steps 3 --warmup 1 > gpurun_out/bench16.log 2>&1
timeout 900 python bench.py > gpurun_out/bench20.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
ls -R $R/gpurun_out/prof | head -30 >> $R/gpurun_out/rocprof.log
exit 0
