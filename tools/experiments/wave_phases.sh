#!/bin/bash
# where a wave's life goes (PRT_TIMING builds: s_memrealtime stamps at eight points of k_generation), hints in force
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/wave_phases; mkdir -p $out
cd $R
for spec in "config2 1000000" "config3 4000000" "config4 1000000"; do
  set -- $spec
  for g in "" 1; do
    PRT_LIB=$R/pyrayt_amd/csrc/libprt_hip_timing$g.so python3 tools/wave_stamps.py /tmp/stamps.bin $2 $1 hints > /dev/null 2>&1
    echo "== $1, $2 rays, generation ${g:-0} (dense hints in force)" >> $out/phases.txt
    python3 tools/wave_timeline.py /tmp/stamps.bin >> $out/phases.txt 2>&1
  done
done
cat $out/phases.txt
