#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab4; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
L=$R/pyrayt_amd/csrc
python tools/ab.py --reps 3 "base:PRT_LIB=$L/libprt_hip_base.so" "new:" > $O/ab.txt 2>&1
python tools/ab.py --reps 2 "base_c3:PRT_LIB=$L/libprt_hip_base.so:--workload config3 --rays 4000000" "new_c3::--workload config3 --rays 4000000" "base_c4:PRT_LIB=$L/libprt_hip_base.so:--workload config4 --rays 8000000" "new_c4::--workload config4 --rays 8000000" "base_c5:PRT_LIB=$L/libprt_hip_base.so:--workload config5 --rays 2000000" "new_c5::--workload config5 --rays 2000000" >> $O/ab.txt 2>&1
cat $O/ab.txt
