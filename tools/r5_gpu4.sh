#!/bin/bash
# clearance test of a lens chain's cylinder: parity, fuzz, timing A/B against the library without it
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r5_gpu4; mkdir -p $out
cd $R
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "rim or every_program_form or stepwise" > $out/clearance_tests.txt 2>&1
python3 -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
PRT_FUZZ_FIRST=2100000 PRT_FUZZ_SEEDS=8000 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 > $out/fuzz_soak.txt 2>&1
L=$R/pyrayt_amd/csrc
python3 tools/ab.py --reps 4 "clearance::--streams 1 --side-steps 0" "ssc_only:PRT_LIB=$L/libprt_hip_ssc.so:--streams 1 --side-steps 0" "without:PRT_LIB=$L/libprt_hip_noclear.so:--streams 1 --side-steps 0" > $out/ab_config2_one_stream.txt 2>&1
python3 tools/ab.py --reps 4 "clearance::--side-steps 0" "without:PRT_LIB=$L/libprt_hip_noclear.so:--side-steps 0" > $out/ab_config2_overlap.txt 2>&1
C3="--workload config3 --rays 4000000 --steps 50 --warmup 5 --side-steps 0"
python3 tools/ab.py --reps 4 "clearance::$C3" "ssc_only:PRT_LIB=$L/libprt_hip_ssc.so:$C3" "without:PRT_LIB=$L/libprt_hip_noclear.so:$C3" > $out/ab_config3.txt 2>&1
grep -h "passed\|failed" $out/clearance_tests.txt $out/gpu_suite.txt $out/fuzz_soak.txt; cat $out/ab_*.txt
