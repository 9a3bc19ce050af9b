#!/bin/bash
# SQ issue / stall counters of one kernel, in passes of <= 8 counters (guide: PMC slots).
# usage: tools/sq.sh <tag> <program> [args...]   -> gpurun_out/sq/<tag>_<pass>_counter_collection.csv
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/sq
mkdir -p $out; cd /tmp
pass1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
pass2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
pass3="SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64"
pass4="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU"
n=1
for p in "$pass1" "$pass2" "$pass3" "$pass4"; do
  rocprofv3 --kernel-trace --pmc $p --output-format csv -d $out -o ${tag}_$n -- "$@" > $out/${tag}_$n.log 2>&1
  n=$((n+1))
done
ls $out | head -30
