#!/bin/bash
# What each speculative compaction form buys (review item 6): bench.py per BASELINE config with the form switched off by its
# trace flag -- 512 no per-tile records (mode 3), 1024 no sparse-loss keep (modes 4 / 5 / 6) -- against the default,
# timed region (rotating ray sets where the config has seeds) and the replay of one ray set.   usage: tools/mode_value.sh > out.txt
for w in config2 config3 config4 config5; do
  rays=""; [ $w = config3 ] && rays="--rays 4000000"; [ $w = config4 ] && rays="--rays 8000000"; [ $w = config5 ] && rays="--rays 2000000"
  for rep in 1 2; do
  for flags in 0 512 1024; do
    python bench.py --no-cpu-baseline --workload $w $rays --flags $flags --steps 100 --warmup 10 --side-steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=lambda k: ('%.4e'%d[k]) if d.get(k) else '-'
print('$w flags $flags', 'value', g('value'), 'ms/step %.4f'%d['ms_per_step'], 'replay', g('value_replay'), 'one_stream', g('value_one_stream'), 'telemetry', {k:v for k,v in d['config']['telemetry'].items() if v and k in ('speculation_misses','dense_launches','tile_record_launches','tile_record_misses','sparse_keep_launches')})
"
  done; done
done
