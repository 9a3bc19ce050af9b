#!/bin/bash
P=$GRAFT_REPO_ROOT/pyrayt_amd/csrc/libprt_hip_prev.so
python tools/ab.py --reps 3 "compact:" "full:PRT_FULL_ROWS=1" "prev:PRT_LIB=$P"
python tools/ab.py --reps 2 "compact3::--workload config3 --rays 4000000 --steps 50 --warmup 5" "prev3:PRT_LIB=$P:--workload config3 --rays 4000000 --steps 50 --warmup 5" "compact4::--workload config4 --rays 8000000 --steps 50 --warmup 5" "prev4:PRT_LIB=$P:--workload config4 --rays 8000000 --steps 50 --warmup 5" "compact5::--workload config5 --rays 2000000 --steps 50 --warmup 5" "prev5:PRT_LIB=$P:--workload config5 --rays 2000000 --steps 50 --warmup 5"
