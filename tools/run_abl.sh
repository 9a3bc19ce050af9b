#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; L=$R/pyrayt_amd/csrc
python tools/ab.py --reps 2 "full::--generation-limit 1" "no_lookback:PRT_LIB=$L/libprt_hip_ablate16.so:--generation-limit 1" "no_stores:PRT_LIB=$L/libprt_hip_ablate32.so:--generation-limit 1" "neither:PRT_LIB=$L/libprt_hip_ablate48.so:--generation-limit 1" "l2_stores:PRT_LIB=$L/libprt_hip_ablate64.so:--generation-limit 1" "no_hit_math:PRT_LIB=$L/libprt_hip_ablate6.so:--generation-limit 1" 2>&1 | tail -8
