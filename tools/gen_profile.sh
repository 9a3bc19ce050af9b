#!/bin/bash
# per-generation launch durations of a bench workload (rocprofv3 kernel trace): tools/gen_profile.sh <workload> <rays> [ENV=V ...]
W=$1; N=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/genprof
rocprofv3 --kernel-trace -d /tmp/genprof -o gp --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --rays $N --no-cpu-baseline --steps 6 --warmup 3 --spinup-ms 20 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/genprof/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_generation' in r['Kernel_Name'] or 'reinit' in r['Kernel_Name']]
# split into traces at reinit
traces, cur = [], []
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if 'reinit' in r['Kernel_Name']:
        traces.append(cur); cur = []
    else:
        cur.append(d / 1e3)
last = traces[-4:]
for t in last: print(' '.join(f'{d:7.1f}' for d in t), ' | sum %.1f us' % sum(t))
PY
