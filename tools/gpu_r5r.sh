#!/bin/bash
# round 5, GPU run R: the new pipeline-tails test and the default bench command with the S3 e2e reading over 12 proofs.
R=$PWD; O=$R/gpurun_out/r5r; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "last_msm_tails or one_stream_schedule or one_stream_tail" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/status.txt
t0=$(date +%s)
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$? ($(( $(date +%s) - t0 )) s)" >> $O/status.txt
exit 0
