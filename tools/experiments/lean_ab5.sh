#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/lean_ab5; mkdir -p $out
cd $R
L=$R/pyrayt_amd/csrc
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lean or stepwise or full_size" > $out/tests.txt 2>&1
{ for k in 1 2; do python3 tools/experiments/nonlean_timing.py 2>/dev/null; PRT_LIB=$L/libprt_hip_leanB.so python3 tools/experiments/nonlean_timing.py 2>/dev/null; PRT_LIB=$L/libprt_hip_r5a.so python3 tools/experiments/nonlean_timing.py 2>/dev/null; done; } > $out/nonlean.txt 2>&1
python3 tools/ab.py --reps 4 "head-first::--side-steps 0" "head-late:PRT_LIB=$L/libprt_hip_leanB.so:--side-steps 0" > $out/config2_overlap.txt 2>&1
C="--workload config4 --rays 8000000 --steps 40 --warmup 5 --side-steps 0 --reps 3"
python3 tools/ab.py --reps 3 "c4-head-first::$C" "c4-head-late:PRT_LIB=$L/libprt_hip_leanB.so:$C" > $out/config4.txt 2>&1
C="--workload config3 --rays 4000000 --steps 40 --warmup 5 --side-steps 0 --reps 3"
python3 tools/ab.py --reps 3 "c3-head-first::$C" "c3-head-late:PRT_LIB=$L/libprt_hip_leanB.so:$C" > $out/config3.txt 2>&1
grep -h "passed\|failed" $out/tests.txt; cat $out/nonlean.txt $out/config2_overlap.txt $out/config4.txt $out/config3.txt
