#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r5_gpu6; mkdir -p $out
cd $R
bash tools/gen_counters.sh config3 4000000 > /dev/null 2>&1
cp gpurun_out/gen_counters/config3_per_generation.txt $out/config3_clearance.txt
bash tools/gen_counters.sh config2 1000000 > /dev/null 2>&1
cp gpurun_out/gen_counters/config2_per_generation.txt $out/config2_clearance.txt
export PRT_LIB=$R/pyrayt_amd/csrc/libprt_hip_noclear.so
bash tools/gen_counters.sh config3 4000000 > /dev/null 2>&1
cp gpurun_out/gen_counters/config3_per_generation.txt $out/config3_without.txt
bash tools/gen_counters.sh config2 1000000 > /dev/null 2>&1
cp gpurun_out/gen_counters/config2_per_generation.txt $out/config2_without.txt
tail -n 16 $out/config3_clearance.txt $out/config3_without.txt; tail -n 8 $out/config2_clearance.txt $out/config2_without.txt
