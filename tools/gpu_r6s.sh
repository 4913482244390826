#!/bin/bash
# round 6, GPU run S: the level-1 sort writes (key, value) pairs -- parity, then a serial kernel trace (sort kernel durations
# against run R's: sort_hi_scatter 286.8 us, sort_lo<true> 211.7 us per call) and the default bench.
R=$PWD; O=$R/gpurun_out/r6s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_tables.py -m gpu -x -q 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_o3_large.py -m gpu -x -q -k "resident_msm or s2_2p20 or batch_2p18" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
ARK355_SERIAL=1 ARK355_SIDE_WM=0 ARK355_SIDE_G2_TAILS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --profile-run --inflight 1 --steps 5 --warmup 2 > $O/serial.log 2>&1
find $O/serial -name "*kernel_trace.csv" -delete; find $O/serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/serial_kernel_stats.csv
grep -i "sort_\|scan_" $O/serial_kernel_stats.csv | sed 's/void ark355:://' | cut -c1-50,140-260
cd $R
for i in 1 2; do ARK355_BENCH_WATCHDOG=150 timeout 170 python bench.py --no-cpu-baseline --no-e2e --no-micro --no-ab 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']/1e6, d['box'].get('gfx_cycles_per_constraint'), d['latency']['host_pinned_z_ms'], d['isolated']['total_ms'], d['parity'])"; done
exit 0
