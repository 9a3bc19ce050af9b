#!/bin/bash
# round 4's tree (commit 62dcf6e, library da652c215e89cfb5, unpacked by hand into tmp_r4/) against this one, interleaved on one box:
# the default bench command of each tree, 4 repetitions each; ms per step and rows/s as each tree's bench.py reports them
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/round_ab; mkdir -p $out
: > $out/ab.txt
for rep in 1 2 3 4; do
  for tree in tmp_r4 .; do
    (cd $R/$tree && python3 bench.py --no-cpu-baseline --side-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tree', 'ms_per_step %.4f value %.4g' % (d['ms_per_step'], d['value']))") >> $out/ab.txt
    (cd $R/$tree && python3 bench.py --no-cpu-baseline --side-steps 0 --streams 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tree', 'one stream: ms_per_step %.4f kernel_ms_per_step %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step']))") >> $out/ab.txt
    (cd $R/$tree && python3 bench.py --no-cpu-baseline --side-steps 0 --workload config3 --rays 4000000 --steps 50 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tree', 'config3: ms_per_step %.4f' % d['ms_per_step'])") >> $out/ab.txt
  done
done
sort $out/ab.txt
