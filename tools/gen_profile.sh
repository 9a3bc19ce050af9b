#!/bin/bash
# per-generation launch durations and the gaps between launches of a bench workload (rocprofv3 kernel
# trace): tools/gen_profile.sh <workload> <rays> [ENV=V ...]
W=$1; N=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/genprof
rocprofv3 --kernel-trace -d /tmp/genprof -o gp --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --rays $N --no-cpu-baseline --steps 6 --warmup 3 --spinup-ms 20 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/genprof/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_generation' in r['Kernel_Name'] or 'reinit' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
traces, cur, prev_end = [], [], None
for r in rows:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = b
    cur.append((gap, (b - a) / 1e3, 'reinit' in r['Kernel_Name']))
    if 'reinit' in r['Kernel_Name']:
        traces.append(cur); cur = []
for t in traces[-4:]:
    print(' '.join(f'[{g:5.1f}] {d:6.1f}{"r" if re else ""}' for g, d, re in t), ' | kernels %.1f us, gaps %.1f us' % (sum(d for _, d, re in t if not re), sum(g for g, _, _ in t[1:])))
PY
