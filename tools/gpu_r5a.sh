#!/bin/bash
# round 5, GPU run A (one box): (1) tools/ubench5: cost per instruction at the occupancy of the accumulation kernels, the
# floating-point-multiplier go / no-go, Fp28 product and mixed additions with and without the Karatsuba level; (2) same-box,
# interleaved A/B of three libraries: schoolbook products (rounds 2-4), Karatsuba (this round's default), Karatsuba with the
# G2 kernel sized for one wave per SIMD; (3) the parity tests on the default (Karatsuba) library.
R=$PWD; O=$R/gpurun_out/r5a; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
timeout 120 variants/ubench5_k0 > $O/ubench5_school.txt 2>&1; echo "ubench k0 rc=$?" >> $O/status.txt
timeout 120 variants/ubench5_k1 > $O/ubench5_kara.txt 2>&1; echo "ubench k1 rc=$?" >> $O/status.txt
run() {   # tag lib extra-args...
  tag=$1; lib=$2; shift 2
  ARK355_BENCH_WATCHDOG=280 ARK355_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-micro --no-telemetry --no-ab "$@" > $O/$tag.json 2> $O/$tag.log
  echo "$tag rc=$?" >> $O/status.txt
}
SCHOOL=$R/variants/lib_school.so; KARA=$R/variants/lib_kara.so; KW1=$R/variants/lib_kara_w1.so
for pass in 1 2; do
  run ab_school_if4_$pass $SCHOOL --steps 16 --warmup 4
  run ab_kara_if4_$pass $KARA --steps 16 --warmup 4
  run ab_kw1_if4_$pass $KW1 --steps 16 --warmup 4
done
run ab_school_if1 $SCHOOL --steps 8 --warmup 2 --inflight 1
run ab_kara_if1 $KARA --steps 8 --warmup 2 --inflight 1
run ab_kw1_if1 $KW1 --steps 8 --warmup 2 --inflight 1
run ab_school_bn $SCHOOL --steps 16 --warmup 4 --curve bn254
run ab_kara_bn $KARA --steps 16 --warmup 4 --curve bn254
timeout 900 python -m pytest tests/test_gpu_parity.py "tests/test_gpu_o3_large.py::test_s2_2p20_bls12_381_vs_o3" \
  "tests/test_gpu_o3_large.py::test_resident_msm_vs_o3" -m gpu -x -q --durations=8 > $O/pytest_kara.txt 2>&1
echo "pytest kara rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
