#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; L=$R/pyrayt_amd/csrc
for w in "config2 1000000" "config3 4000000" "config5 2000000"; do set -- $w; python tools/ab.py --reps 3 "l1_table_$1::--workload $1 --rays $2" "lds_waterfall_$1:PRT_LDS_PRIMS=1:--workload $1 --rays $2" 2>&1 | tail -2; done
