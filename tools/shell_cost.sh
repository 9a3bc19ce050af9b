#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/shell
mkdir -p $out; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $out -o shell -- python3 $GRAFT_REPO_ROOT/tools/shell_cost.py > $out/shell.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections
rows = [r for r in csv.DictReader(open("gpurun_out/shell/shell_counter_collection.csv")) if "k_hit" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
names = ["1 plane", "2 planes", "4 planes", "lens", "lens+plane"]
ids = sorted(by)
for k, name in enumerate(names):
    d = by[ids[3 * k + 2]]
    print(f"{name:12s}", {c: round(v / 15628, 1) for c, v in sorted(d.items())})
PY
