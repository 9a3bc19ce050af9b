#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/lean_ab4; mkdir -p $out
cd $R
L=$R/pyrayt_amd/csrc
PRT_LIB=$L/libprt_hip_lean2.so python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lean or stepwise or full_size" > $out/tests.txt 2>&1
C="--workload config4 --rays 8000000 --steps 40 --warmup 5 --side-steps 0 --reps 3"
python3 tools/ab.py --reps 4 "c4-three-rows::$C" "c4-one-row:PRT_LIB=$L/libprt_hip_lean2.so:$C" > $out/config4.txt 2>&1
python3 tools/ab.py --reps 5 "three-rows::--side-steps 0" "one-row:PRT_LIB=$L/libprt_hip_lean2.so:--side-steps 0" > $out/config2_overlap.txt 2>&1
C="--workload config3 --rays 4000000 --steps 40 --warmup 5 --side-steps 0 --reps 3"
python3 tools/ab.py --reps 3 "c3-three-rows::$C" "c3-one-row:PRT_LIB=$L/libprt_hip_lean2.so:$C" > $out/config3.txt 2>&1
grep -h "passed\|failed" $out/tests.txt; cat $out/config4.txt $out/config2_overlap.txt $out/config3.txt
