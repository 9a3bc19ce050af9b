#!/bin/bash
# lean state segments (three redundant rows of a wave as three numbers) against the library before them
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/lean_ab; mkdir -p $out
cd $R
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lean or stepwise or full_size or state_rows or every_program_form" > $out/tests.txt 2>&1
python3 -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
L=$R/pyrayt_amd/csrc
python3 tools/ab.py --reps 5 "lean::--side-steps 0" "before:PRT_LIB=$L/libprt_hip_r5a.so:--side-steps 0" > $out/config2_overlap.txt 2>&1
python3 tools/ab.py --reps 4 "lean::--side-steps 0 --streams 1" "before:PRT_LIB=$L/libprt_hip_r5a.so:--side-steps 0 --streams 1" > $out/config2_one_stream.txt 2>&1
for cfg in "config3 4000000" "config4 8000000" "config5 2000000"; do
  set -- $cfg
  C="--workload $1 --rays $2 --steps 40 --warmup 5 --side-steps 0 --reps 3"
  python3 tools/ab.py --reps 3 "$1-lean::$C" "$1-before:PRT_LIB=$L/libprt_hip_r5a.so:$C" > $out/$1.txt 2>&1
done
grep -h "passed\|failed" $out/tests.txt $out/gpu_suite.txt; cat $out/config2_overlap.txt $out/config2_one_stream.txt $out/config3.txt $out/config4.txt $out/config5.txt
