#!/bin/bash
# one GPU-box pass: instruction price list, shared-division check, GPU tests, A/B of the division variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab1; mkdir -p $O; cd $R
(cd tools/ubench && ./valu_costs) > $O/valu_costs.txt 2>&1
(cd tools/ubench && ./div_shared) > $O/div_shared.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
L=$R/pyrayt_amd/csrc
python tools/ab.py --reps 3 "base:PRT_LIB=$L/libprt_hip_base.so" "v1_div2+div3:" "v2_div2:PRT_LIB=$L/libprt_hip_v2.so" "v3_div3:PRT_LIB=$L/libprt_hip_v3.so" "v4_all_spill:PRT_LIB=$L/libprt_hip_v4.so" "v5_all_4waves:PRT_LIB=$L/libprt_hip_v5.so" > $O/ab.txt 2>&1
cat $O/div_shared.txt $O/ab.txt
