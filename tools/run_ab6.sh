#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab6; mkdir -p $O; cd $R
L=$R/pyrayt_amd/csrc
python tools/ab.py --reps 3 "base:PRT_LIB=$L/libprt_hip_base.so" "v1_prev:PRT_LIB=$L/libprt_hip_v1.so" "v2_live:PRT_LIB=$L/libprt_hip_v2.so" "v3_live_carry:PRT_LIB=$L/libprt_hip_v3.so" "v4_waterfall_only:PRT_LIB=$L/libprt_hip_v4.so" "new_all:" > $O/ab.txt 2>&1
cat $O/ab.txt
