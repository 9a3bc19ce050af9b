#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O; cd $R
PRT_LIB=$R/pyrayt_amd/csrc/libprt_hip_timing.so PRT_TIMING_FILE=$O/stamps.bin python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_timing.json 2>&1
python tools/wave_timeline.py $O/stamps.bin > $O/timeline.txt; cat $O/timeline.txt
rm -f $O/stamps.bin
bash tools/sq.sh gen python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/sq.py gpurun_out/sq gen > $O/sq_counters.txt; cat $O/sq_counters.txt
