#!/bin/bash
# The GPU suite, the option / flag matrix (tools/run_matrix.sh) and a fuzz soak on the library as built, one summary each:
#   bash tools/soak.sh [first seed] [seeds]      (defaults: 2300000, 70000 = 210 000 cases, about 16 minutes on 8 processes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/soak; mkdir -p $out
cd $R
python3 -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
bash tools/run_matrix.sh > $out/matrix.txt 2>&1
s0=$(date +%s)
PRT_FUZZ_FIRST=${1:-2300000} PRT_FUZZ_SEEDS=${2:-70000} python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 > $out/fuzz_soak.txt 2>&1
echo "soak seconds $(( $(date +%s) - s0 ))" >> $out/fuzz_soak.txt
grep -h "passed\|failed" $out/gpu_suite.txt; cat $out/matrix.txt; tail -n 3 $out/fuzz_soak.txt
