#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab3; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
L=$R/pyrayt_amd/csrc
python tools/ab.py --reps 3 "base:PRT_LIB=$L/libprt_hip_base.so" "chain_cull3:PRT_CULL_MIN=3" "chain_cull2:" > $O/ab.txt 2>&1
python tools/ab.py --reps 2 "c5_cull3:PRT_CULL_MIN=3:--workload config5 --rays 2000000" "c5_cull2::--workload config5 --rays 2000000" >> $O/ab.txt 2>&1
cat $O/ab.txt
