#!/bin/bash
# Workgroup size of the whole library, interleaved: the product (256 threads, four waves to a tile) against builds with
# -DPRT_BLOCK=128 / 64 (csrc: hipcc ... -DPRT_BLOCK=64 -o libprt_hip_b64.so).  Round 1 measured 128 at -7 % and 64 at -23 % when
# every generation compacted by look-back; the dense forms of rounds 3-5 have no look-back, hence the question again.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LIBS="libprt_hip.so libprt_hip_b128.so libprt_hip_b64.so"
echo "# config 2, two traces in flight (the bench line), 200 steps"; LIBS="$LIBS" bash tools/experiments/ab_lib.sh
echo "# config 2, hint-less (PRT_TRACE_NO_HINTS = 4: every generation by look-back)"; LIBS="$LIBS" BENCH_ARGS="--steps 200 --warmup 20 --flags 4" bash tools/experiments/ab_lib.sh
echo "# config 3, 4M rays"; LIBS="$LIBS" BENCH_ARGS="--workload config3 --rays 4000000 --steps 50 --warmup 5" bash tools/experiments/ab_lib.sh
