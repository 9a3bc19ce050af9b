#!/bin/bash
# HBM traffic of the generation kernel from PMC counters, per the MI355X guide: FETCH_SIZE and
# WRITE_SIZE in separate passes, calibrated on a kernel with the same access pattern and a known
# byte count (tools/ubench/copy_f64).  Writes gpurun_out/traffic/*.csv ; parse with tools/traffic.py
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/traffic
mkdir -p $out; cd /tmp
(cd $GRAFT_REPO_ROOT/tools/ubench && make -s copy_f64 >/dev/null 2>&1)  # (built binaries do not travel)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o cal_$c -- $GRAFT_REPO_ROOT/tools/ubench/copy_f64 > $out/cal_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o bench_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --spinup-ms 0 --no-cpu-baseline --side-steps 0 --streams 1 > $out/bench_$c.log 2>&1
done
ls $out
