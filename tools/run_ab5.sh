#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab5; mkdir -p $O; cd $R
L=$R/pyrayt_amd/csrc
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/ab.py --reps 3 "base:PRT_LIB=$L/libprt_hip_base.so" "new:" > $O/ab.txt 2>&1
cat $O/ab.txt
