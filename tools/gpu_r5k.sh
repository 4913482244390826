#!/bin/bash
# round 5, GPU run K (one box): last check of the final library -- the driver's bench command (incl. the stream map diagnostic,
# which now locks every context it probes), and the EXTENDED case of the suite (the domain-tight 2^22 - 100 key, whole + sharded).
R=$PWD; O=$R/gpurun_out/r5k; mkdir -p $O
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
ARK355_TEST_EXTENDED=1 timeout 900 python -m pytest "tests/test_gpu_o3_large.py::test_s2_2p22_bls12_381_vs_o3_whole_and_sharded" tests/test_gpu_zz_tables.py -m gpu -x -q --durations=5 > $O/pytest_extended.txt 2>&1
echo "pytest extended rc=$?" >> $O/status.txt
exit 0
