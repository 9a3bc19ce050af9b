#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; L=$R/pyrayt_amd/csrc
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in "config2 1000000" "config3 4000000" "config4 8000000" "config5 2000000"; do set -- $w; python tools/ab.py --reps 2 "plain_$1:PRT_LIB=$L/libprt_hip_plain.so:--workload $1 --rays $2" "nt_$1::--workload $1 --rays $2" 2>&1 | tail -2; done
