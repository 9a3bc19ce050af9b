#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hitab; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "surface_parallel" 2>&1 | tail -8
python tools/hit_ab.py > $O/hit_ab.txt 2>&1; cat $O/hit_ab.txt
