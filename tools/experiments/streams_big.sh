#!/bin/bash
# do two traces in flight pay for ray sets whose state does not fit the Infinity Cache even once?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/streams_big; mkdir -p $out
cd $R
for cfg in "config3 4000000" "config5 2000000" "config4 8000000"; do
  set -- $cfg
  C="--workload $1 --rays $2 --steps 40 --warmup 5 --side-steps 0 --reps 3"
  python3 tools/ab.py --reps 3 "$1-1::$C --streams 1" "$1-2::$C --streams 2" "$1-3::$C --streams 3" > $out/$1.txt 2>&1
  cat $out/$1.txt
done
time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_form.json 2> $out/driver_form.err
