#!/bin/bash
# ablation: the two metadata rows that are the same for every ray of config 2 (intensity, wavelength) neither read nor
# written between generations (constants instead); rows stay exact for this workload
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/uniform_ab; mkdir -p $out
cd $R
L=$R/pyrayt_amd/csrc
python3 tools/ab.py --reps 5 "product::--side-steps 0" "uniform:PRT_LIB=$L/libprt_hip_uniform.so:--side-steps 0" > $out/overlap.txt 2>&1
python3 tools/ab.py --reps 4 "product::--side-steps 0 --streams 1" "uniform:PRT_LIB=$L/libprt_hip_uniform.so:--side-steps 0 --streams 1" > $out/one_stream.txt 2>&1
cat $out/overlap.txt $out/one_stream.txt
