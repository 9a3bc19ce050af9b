#!/bin/bash
# the parity / fuzz / cull / multi-rank suites under every switch, then a fuzz soak, on the round's last library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r5_gpu7; mkdir -p $out
cd $R
python3 -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
bash tools/run_matrix.sh > $out/matrix.txt 2>&1
s0=$(date +%s)
PRT_FUZZ_FIRST=2200000 PRT_FUZZ_SEEDS=70000 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 > $out/fuzz_soak.txt 2>&1
echo "soak seconds $(( $(date +%s) - s0 ))" >> $out/fuzz_soak.txt
grep -h "passed\|failed" $out/gpu_suite.txt; cat $out/matrix.txt; tail -n 3 $out/fuzz_soak.txt
