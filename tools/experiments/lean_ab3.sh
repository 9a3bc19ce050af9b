#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/lean_ab3; mkdir -p $out
cd $R
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lean or stepwise or full_size" > $out/tests.txt 2>&1
L=$R/pyrayt_amd/csrc
C="--workload config4 --rays 8000000 --steps 40 --warmup 5 --side-steps 0 --reps 3"
python3 tools/ab.py --reps 4 "c4-vector::$C" "c4-scalar:PRT_LIB=$L/libprt_hip_leans.so:$C" "c4-before:PRT_LIB=$L/libprt_hip_r5a.so:$C" > $out/config4.txt 2>&1
python3 tools/ab.py --reps 4 "vector::--side-steps 0" "scalar:PRT_LIB=$L/libprt_hip_leans.so:--side-steps 0" "before:PRT_LIB=$L/libprt_hip_r5a.so:--side-steps 0" > $out/config2_overlap.txt 2>&1
C="--workload config3 --rays 4000000 --steps 40 --warmup 5 --side-steps 0 --reps 3"
python3 tools/ab.py --reps 3 "c3-vector::$C" "c3-scalar:PRT_LIB=$L/libprt_hip_leans.so:$C" "c3-before:PRT_LIB=$L/libprt_hip_r5a.so:$C" > $out/config3.txt 2>&1
grep -h "passed\|failed" $out/tests.txt; cat $out/config4.txt $out/config2_overlap.txt $out/config3.txt
