#!/bin/bash
# SQ issue / stall counters and calibrated HBM traffic (PMC) of the generation kernel on the other BASELINE
# configs at their per-GPU sizes, plus per-generation launch times.  One GPU-box pass:
#   bash tools/counters_cfg.sh [round]   -> gpurun_out/counters/{sq_counters_<cfg>.txt, traffic_<cfg>.json, gen_times_<cfg>.txt}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/counters
rm -rf $out $R/gpurun_out/sq $R/gpurun_out/traffic; mkdir -p $out $R/gpurun_out/traffic
cd $R
(cd tools/ubench && make -s copy_f64 >/dev/null 2>&1)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic -o cal_$c -- $R/tools/ubench/copy_f64 > $R/gpurun_out/traffic/cal_$c.log 2>&1
done
for cfg in "config3 4000000" "config4 8000000" "config5 2000000" ${EXTRA_CFG:+"$EXTRA_CFG"}; do
  set -- $cfg
  args="--workload $1 --rays $2 --steps 3 --warmup 1 --spinup-ms 0 --no-cpu-baseline --side-steps 0 --no-pipeline"
  cd $R
  bash tools/sq.sh $1 python3 $R/bench.py $args > /dev/null 2>&1
  python3 tools/sq.py gpurun_out/sq $1 > $out/sq_counters_$1.txt
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic -o $1_$c -- python3 $R/bench.py $args > $R/gpurun_out/traffic/$1_$c.log 2>&1
  done
  cd $R
  python3 tools/traffic.py gpurun_out/traffic $out/traffic_$1.json $1 > /dev/null 2>> $out/errors.txt
  python3 tools/gen_times.py $1 $2 2>&1 | grep -v amdgpu.ids > $out/gen_times_$1.txt
done
ls -la $out
