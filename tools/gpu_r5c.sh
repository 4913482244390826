#!/bin/bash
# round 5, GPU run C (one box): why does a G1 accumulation launch take 2.75 ms when the register-resident loop of
# tools/ubench5 says 1.87 ms for the same 15.7 M additions?  Stand-alone 2^20-term G1 MSMs (tools/msm_bench.py prints the
# accumulation kernel's own time) over: the default library; EXPERIMENT builds that compute wrong sums on purpose -- no
# flush at all, no bucket boundaries at all, every gather inside the same 1024 rows; segment lengths 30 .. 160 on the
# default library; and the per-wave timestamps of one launch (tools/acc_trace.py on the tracing build).
R=$PWD; O=$R/gpurun_out/r5c; mkdir -p $O
python tools/gpu_telemetry.py > $O/telemetry.txt 2>&1
mb() {   # tag lib [env...]
  tag=$1; lib=$2; shift 2
  env ARK355_LIB=$lib "$@" timeout 200 python tools/msm_bench.py --min-log 20 --max-log 21 --step 1 --groups 1 --dists uniform --reps 5 --no-check > $O/mb_$tag.txt 2>&1
  echo "mb $tag rc=$?" >> $O/status.txt
}
P2=$R/variants/lib_park2.so
mb park2 $P2
mb kara $R/variants/lib_kara.so
mb noflush $R/variants/lib_exp_noflush.so
mb nobound $R/variants/lib_exp_nobound.so
mb smalltbl $R/variants/lib_exp_smalltbl.so
for seg in 30 45 61 90 122 160; do mb park2_seg$seg $P2 ARK355_MSM_SEG=$seg; done
mb park2_again $P2
ARK355_LIB=$R/variants/lib_exp_trace.so timeout 200 python tools/acc_trace.py --log-n 20 > $O/trace_n20.txt 2>&1; echo "trace20 rc=$?" >> $O/status.txt
ARK355_LIB=$R/variants/lib_exp_trace.so timeout 200 python tools/acc_trace.py --log-n 21 > $O/trace_n21.txt 2>&1; echo "trace21 rc=$?" >> $O/status.txt
ARK355_LIB=$R/variants/lib_exp_trace.so ARK355_MSM_SEG=122 timeout 200 python tools/acc_trace.py --log-n 20 > $O/trace_n20_seg122.txt 2>&1; echo "trace20 seg122 rc=$?" >> $O/status.txt
python tools/gpu_telemetry.py > $O/telemetry_end.txt 2>&1
exit 0
