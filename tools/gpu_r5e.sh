#!/bin/bash
# round 5, GPU run E (one box): the final accumulation kernels (Karatsuba products, packed G1 rows, unpacked G2 halves, the
# plain segment walk of rounds 2-4, DPP exchanges without destination moves) against run A's (kara) and run B's (park2)
# libraries, interleaved; tools/ubench5; stand-alone MSMs; the e2e block; parity tests on the final library.
R=$PWD; O=$R/gpurun_out/r5e; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
timeout 120 variants/ubench5_k1 > $O/ubench5_final.txt 2>&1; echo "ubench rc=$?" >> $O/status.txt
run() {   # tag lib extra-args...
  tag=$1; lib=$2; shift 2
  ARK355_BENCH_WATCHDOG=280 ARK355_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab --no-e2e "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
KARA=$R/variants/lib_kara.so; P2=$R/variants/lib_park2.so; FIN=$R/variants/lib_final.so
for pass in 1 2 3; do
  run ab_kara_if4_$pass $KARA --steps 16 --warmup 4
  run ab_final_if4_$pass $FIN --steps 16 --warmup 4
  run ab_park2_if4_$pass $P2 --steps 16 --warmup 4
done
run ab_kara_if1 $KARA --steps 8 --warmup 2 --inflight 1
run ab_final_if1 $FIN --steps 8 --warmup 2 --inflight 1
run ab_kara_bn $KARA --steps 16 --warmup 4 --curve bn254
run ab_final_bn $FIN --steps 16 --warmup 4 --curve bn254
run ab_park2_bn $P2 --steps 16 --warmup 4 --curve bn254
run ab_kara_n18 $KARA --steps 32 --warmup 8 --log-n 18 --inflight 8
run ab_final_n18 $FIN --steps 32 --warmup 8 --log-n 18 --inflight 8
for lib in kara final; do
  ARK355_LIB=$R/variants/lib_$lib.so timeout 200 python tools/msm_bench.py --min-log 20 --max-log 20 --groups 1,2 --dists uniform --reps 5 > $O/mb_$lib.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_parity.py "tests/test_gpu_o3_large.py::test_s2_2p20_bls12_381_vs_o3" \
  "tests/test_gpu_o3_large.py::test_resident_msm_vs_o3" tests/test_gpu_zz_tables.py tests/test_gpu_wire.py -m gpu -x -q --durations=8 > $O/pytest_final.txt 2>&1
echo "pytest final rc=$?" >> $O/status.txt
ARK355_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 1 --steps 12 --warmup 3 > $O/bench_default.json 2> $O/bench_default.log
echo "bench default rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
