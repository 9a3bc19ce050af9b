#!/bin/bash
# Per-dispatch instruction counters of the generation kernel (VALU / SALU / branches / scalar loads per wave), in
# launch order -- which generation costs what:  bash tools/gen_counters.sh config3 4000000
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/gen_counters
mkdir -p $out; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out -o $1 -- python3 $R/bench.py --workload $1 --rays $2 --steps 3 --warmup 1 --spinup-ms 0 --no-cpu-baseline --side-steps 0 --no-pipeline --ray-sets 1 > $out/$1.log 2>&1
python3 - "$out" "$1" <<'PY'
import csv, glob, sys, collections
out, name = sys.argv[1:3]
path = (glob.glob(f"{out}/**/{name}_counter_collection.csv", recursive=True) + glob.glob(f"{out}/{name}_counter_collection.csv"))[0]
by = collections.defaultdict(dict)
for r in csv.DictReader(open(path)):
    if "k_generation" not in r["Kernel_Name"]:
        continue
    d = by[int(r["Dispatch_Id"])]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(f"{out}/{name}_per_generation.txt", "w") as fh:
    fh.write(f"# {name}: k_generation dispatches in launch order; instruction counts per wave\n")
    fh.write("dispatch      us    waves    VALU    SALU  branch   SMEM  VALU-active/wave-cycles\n")
    for i in sorted(by):
        d = by[i]
        w = max(d.get("SQ_WAVES", 1.0), 1.0)
        fh.write(f"{i:8d} {d['us']:7.1f} {int(w):8d} {d.get('SQ_INSTS_VALU', 0) / w:7.1f} {d.get('SQ_INSTS_SALU', 0) / w:7.1f} "
                 f"{d.get('SQ_INSTS_BRANCH', 0) / w:7.1f} {d.get('SQ_INSTS_SMEM', 0) / w:6.1f}  "
                 f"{d.get('SQ_ACTIVE_INST_VALU', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1):.3f}\n")
print(open(f"{out}/{name}_per_generation.txt").read()[-2500:])
PY
