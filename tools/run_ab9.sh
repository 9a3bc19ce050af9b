#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; L=$R/pyrayt_amd/csrc
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -5
for w in "config2 1000000" "config3 4000000" "config4 8000000" "config5 2000000"; do set -- $w; python tools/ab.py --reps 3 "nohints_$1:PRT_NO_HINTS=1:--workload $1 --rays $2" "hints_$1::--workload $1 --rays $2" 2>&1 | tail -2; done
