#!/bin/bash
# timing experiment: k_hit with parts of the hit computation compiled out (see csrc/Makefile)
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/abl
for a in "" _ablate1 _ablate2 _ablate3 _ablate4 _ablate7; do
  export PRT_LIB=$GRAFT_REPO_ROOT/pyrayt_amd/csrc/libprt_hip$a.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$a -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --flags 2 > /tmp/abl$a.log 2>&1)
  cp /tmp/abl$a/t_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/abl/stats$a.csv
done
