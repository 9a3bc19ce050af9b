#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cull; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_cull.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -6
python tools/cull_scaling.py > $O/cull_groups.txt 2>&1; PRT_NO_GROUPS=1 python tools/cull_scaling.py > $O/cull_flat.txt 2>&1
echo "== grouped"; grep lenses $O/cull_groups.txt; echo "== flat"; grep lenses $O/cull_flat.txt
