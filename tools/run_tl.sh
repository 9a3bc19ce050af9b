#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O; cd $R
PRT_LIB=$R/pyrayt_amd/csrc/libprt_hip_timing.so PRT_TIMING_FILE=$O/stamps.bin python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_timing.json 2>&1
python tools/wave_timeline.py $O/stamps.bin > $O/timeline.txt; cat $O/timeline.txt
python tools/lookback_analysis.py $O/stamps.bin > $O/lookback.txt; cat $O/lookback.txt
rm -f $O/stamps.bin
