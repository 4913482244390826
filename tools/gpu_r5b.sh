#!/bin/bash
# round 5, GPU run B (one box): the accumulation kernels rebuilt around what run A showed (a wave alone gets half the
# multiplier): interior bucket flushes parked in LDS and flushed wave-uniformly at the end of the segment, index / key
# loads two entries ahead, bit-packed 96-byte table rows, DPP exchanges without destination moves -- against the
# Karatsuba-only library of run A (kara), same box, interleaved; v_bfi selects in the lane-pair Fq2 code (park2_bfi); one
# round of longer segments (MSM_SEG); tools/ubench5 with the added instruction rows; the parity tests on the new default
# library incl. the RCCL self-exchange; a rocprofv3 kernel trace of the self-exchange helper (RCCL kernels on the GPU).
R=$PWD; O=$R/gpurun_out/r5b; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
timeout 120 variants/ubench5_k1 > $O/ubench5_park.txt 2>&1; echo "ubench rc=$?" >> $O/status.txt
timeout 120 variants/ubench5_bfi > $O/ubench5_bfi.txt 2>&1; echo "ubench bfi rc=$?" >> $O/status.txt
run() {   # tag lib extra-args...
  tag=$1; lib=$2; shift 2
  ARK355_BENCH_WATCHDOG=280 ARK355_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
KARA=$R/variants/lib_kara.so; PARK=$R/variants/lib_park.so; P2=$R/variants/lib_park2.so; BFI=$R/variants/lib_park2_bfi.so
for pass in 1 2; do
  run ab_kara_if4_$pass $KARA --steps 16 --warmup 4
  run ab_park_if4_$pass $PARK --steps 16 --warmup 4
  run ab_park2_if4_$pass $P2 --steps 16 --warmup 4
  run ab_bfi_if4_$pass $BFI --steps 16 --warmup 4
  ARK355_MSM_SEG=122 run ab_park2_seg122_if4_$pass $P2 --steps 16 --warmup 4
done
run ab_kara_if1 $KARA --steps 8 --warmup 2 --inflight 1
run ab_park2_if1 $P2 --steps 8 --warmup 2 --inflight 1
run ab_bfi_if1 $BFI --steps 8 --warmup 2 --inflight 1
run ab_kara_bn $KARA --steps 16 --warmup 4 --curve bn254
run ab_park2_bn $P2 --steps 16 --warmup 4 --curve bn254
run ab_bfi_bn $BFI --steps 16 --warmup 4 --curve bn254
run ab_kara_n18 $KARA --steps 32 --warmup 8 --log-n 18 --inflight 8
run ab_park2_n18 $P2 --steps 32 --warmup 8 --log-n 18 --inflight 8
timeout 900 python -m pytest tests/test_gpu_parity.py "tests/test_gpu_o3_large.py::test_s2_2p20_bls12_381_vs_o3" \
  "tests/test_gpu_o3_large.py::test_resident_msm_vs_o3" tests/test_gpu_zz_tables.py tests/test_gpu_wire.py -m gpu -x -q --durations=8 > $O/pytest_park2.txt 2>&1
echo "pytest park2 rc=$?" >> $O/status.txt
cd /tmp && export TMPDIR=/tmp
ARK355_RCCL_SELF_BIG=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/rccl_self_prof -o rccl_self -- python $R/tests/rccl_single_rank.py > $O/rccl_self_prof.log 2>&1
echo "rccl self prof rc=$?" >> $O/status.txt
cd $R
find $O/rccl_self_prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $O/rccl_self_kernel_stats.csv
rm -rf $O/rccl_self_prof
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
