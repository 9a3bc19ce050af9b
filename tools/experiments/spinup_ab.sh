#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/spinup_ab; mkdir -p $out
cd $R
python3 tools/ab.py --reps 4 "spin60::--steps 20 --warmup 5 --side-steps 0" "spin300::--steps 20 --warmup 5 --side-steps 0 --spinup-ms 300" "spin1000::--steps 20 --warmup 5 --side-steps 0 --spinup-ms 1000" "steps200::--side-steps 0" > $out/ab.txt 2>&1
cat $out/ab.txt
