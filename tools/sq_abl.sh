#!/bin/bash
# instruction counts of k_hit with parts of the hit computation compiled out
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/sqabl
mkdir -p $out; cd /tmp
for a in "" _ablate2 _ablate4 _ablate6; do
  export PRT_LIB=$GRAFT_REPO_ROOT/pyrayt_amd/csrc/libprt_hip$a.so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $out -o hit$a -- python3 $GRAFT_REPO_ROOT/tools/hit_only.py 4 > $out/hit$a.log 2>&1
done
cd $GRAFT_REPO_ROOT
for a in "" _ablate2 _ablate4 _ablate6; do echo "== hit$a"; python3 tools/sq.py gpurun_out/sqabl hit${a}_counter | grep -v "^k_hit"; done
