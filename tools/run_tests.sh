#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tests; mkdir -p $O; cd $R
PRT_FUZZ_SEEDS=${PRT_FUZZ_SEEDS:-24} python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -15
PRT_LIB=$R/pyrayt_amd/csrc/libprt_hip_count.so python tools/slow_paths.py > $O/slow_paths.txt 2>&1; cat $O/slow_paths.txt
python tools/ab.py --reps 2 "base:PRT_LIB=$R/pyrayt_amd/csrc/libprt_hip_base.so" "new:" 2>&1 | tail -3
