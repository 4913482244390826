R=$PWD; O=$R/gpurun_out/r5c; mkdir -p $O
ARK355_TRACE_NPZ=$O/trace_n20.npz ARK355_LIB=$R/variants/lib_exp_trace.so timeout 200 python tools/acc_trace.py --log-n 20 > $O/trace2_n20.txt 2>&1
ARK355_LIB=$R/variants/lib_exp_trace.so ARK355_MSM_SEG=30 timeout 200 python tools/acc_trace.py --log-n 20 > $O/trace2_n20_seg30.txt 2>&1
exit 0
