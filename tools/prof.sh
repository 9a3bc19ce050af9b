#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box through gpurun)
# kernel trace + stats, then PMC passes (each in its own run, kernel-trace only)
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/trace.log 2>&1
head -12 $out/trace_kernel_stats.csv | cut -c1-160
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out -o pmc_$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_$name.log 2>&1
done
ls $out
