#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r5_gpu2; mkdir -p $out
cd $R
python3 -m pytest tests -m gpu -q -x > $out/gpu_suite.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
python3 bench.py > $out/bench.json 2> $out/bench.err
L=$R/pyrayt_amd/csrc
python3 tools/ab.py --reps 3 "pruned::--streams 1 --side-steps 0" "unpruned:PRT_LIB=$L/libprt_hip_unpruned.so:--streams 1 --side-steps 0" > $out/prune_ab_config2.txt 2>&1
C3="--workload config3 --rays 4000000 --steps 50 --warmup 5 --side-steps 0"
python3 tools/ab.py --reps 3 "pruned::$C3" "unpruned:PRT_LIB=$L/libprt_hip_unpruned.so:$C3" > $out/prune_ab_config3.txt 2>&1
tail -n 6 $out/gpu_suite.txt; cat $out/prune_ab_config2.txt $out/prune_ab_config3.txt; tail -c 1500 $out/bench.err; cut -c1-1500 $out/bench_driver_form.json
