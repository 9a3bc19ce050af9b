#!/bin/bash
# pruned vs unpruned generation kernel (tools/experiments/README.md): both libraries at ABI 200, run from a checkout of
# commit 783f99a (tmp_ab/, made by hand: git archive 783f99a + the two builds)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prune_ab; mkdir -p $out
cd $R/tmp_ab
L=$R/tmp_ab/pyrayt_amd/csrc
python3 tools/ab.py --reps 4 "unpruned::--streams 1 --side-steps 0" "pruned:PRT_LIB=$L/libprt_hip_pruned.so:--streams 1 --side-steps 0" > $out/config2.txt 2>&1
C3="--workload config3 --rays 4000000 --steps 50 --warmup 5 --side-steps 0"
python3 tools/ab.py --reps 4 "unpruned::$C3" "pruned:PRT_LIB=$L/libprt_hip_pruned.so:$C3" > $out/config3.txt 2>&1
cat $out/config2.txt $out/config3.txt
