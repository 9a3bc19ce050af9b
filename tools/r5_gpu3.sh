#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r5_gpu3; mkdir -p $out
cd $R
python3 -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
bash tools/experiments/prune_ab.sh > $out/prune_ab.txt 2>&1
L=$R/pyrayt_amd/csrc
python3 tools/ab.py --reps 4 "product::--side-steps 0" "nometa:PRT_LIB=$L/libprt_hip_nometa.so:--side-steps 0" > $out/nometa_overlap.txt 2>&1
python3 tools/ab.py --reps 3 "product::--side-steps 0 --streams 1" "nometa:PRT_LIB=$L/libprt_hip_nometa.so:--side-steps 0 --streams 1" > $out/nometa_one_stream.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o trace -- python3 $R/bench.py --steps 20 --warmup 5 --side-steps 0 --no-cpu-baseline > $out/trace.log 2>&1
cd $R
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/busy_union.py {} > $out/busy_union.txt 2>&1
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
python3 bench.py --workload config4 --rays 64000000 --generation-limit 4 --steps 6 --warmup 2 --reps 3 --side-steps 2 --no-cpu-baseline > $out/bench_config4_full.json 2> $out/bench_config4_full.err
tail -n 4 $out/gpu_suite.txt; cat $out/prune_ab.txt $out/nometa_overlap.txt $out/nometa_one_stream.txt $out/busy_union.txt; tail -c 600 $out/bench_config4_full.err; cut -c1-400 $out/bench_config4_full.json
