#!/bin/bash
# round 5, review item 1, step 1: A/B of the approximate-arithmetic experiment builds (csrc/Makefile libprt_hip_fast*.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/fast_ab; mkdir -p $out
cd $R
tools/ubench/rcp_accuracy > $out/rcp_accuracy.txt 2>&1
L=$R/pyrayt_amd/csrc
python3 tools/ab.py --reps 3 "exact::--streams 1 --side-steps 0" "fast:PRT_LIB=$L/libprt_hip_fast.so:--streams 1 --side-steps 0" \
  "fastdiv:PRT_LIB=$L/libprt_hip_fastdiv.so:--streams 1 --side-steps 0" "contract:PRT_LIB=$L/libprt_hip_contract.so:--streams 1 --side-steps 0" \
  "fast2:PRT_LIB=$L/libprt_hip_fast2.so:--streams 1 --side-steps 0" > $out/config2_one_stream.txt 2>&1
python3 tools/ab.py --reps 3 "exact::--side-steps 0" "fast:PRT_LIB=$L/libprt_hip_fast.so:--side-steps 0" \
  "fastdiv:PRT_LIB=$L/libprt_hip_fastdiv.so:--side-steps 0" "contract:PRT_LIB=$L/libprt_hip_contract.so:--side-steps 0" \
  "fast2:PRT_LIB=$L/libprt_hip_fast2.so:--side-steps 0" > $out/config2_overlap.txt 2>&1
C3="--workload config3 --rays 4000000 --steps 50 --warmup 5 --side-steps 0"
python3 tools/ab.py --reps 3 "exact::$C3" "fast:PRT_LIB=$L/libprt_hip_fast.so:$C3" \
  "fastdiv:PRT_LIB=$L/libprt_hip_fastdiv.so:$C3" "contract:PRT_LIB=$L/libprt_hip_contract.so:$C3" \
  "fast2:PRT_LIB=$L/libprt_hip_fast2.so:$C3" > $out/config3.txt 2>&1
cat $out/*.txt
