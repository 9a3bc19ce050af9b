#!/bin/bash
# Per-dispatch durations of the generation kernel for one workload (rocprofv3 --kernel-trace), in launch order:
#   bash tools/gen_trace.sh config3 4000000 -> gpurun_out/gen_trace/<workload>_durations.txt
#   EXTRA='--flags 1024' LABEL=nokeep bash tools/gen_trace.sh config3 4000000 -> ..._nokeep_durations.txt (more bench.py arguments)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/gen_trace
mkdir -p $out; cd /tmp
name=$1${LABEL:+_$LABEL}
rocprofv3 --kernel-trace --output-format csv -d $out -o $name -- python3 $R/bench.py --workload $1 --rays $2 --steps 4 --warmup 2 --spinup-ms 0 --no-cpu-baseline --side-steps 0 --no-pipeline --ray-sets 1 $EXTRA > $out/$name.log 2>&1
python3 - "$out" "$name" <<'PY'
import csv, glob, sys
out, name = sys.argv[1:3]
rows = []
for path in glob.glob(f"{out}/**/{name}_kernel_trace.csv", recursive=True) + glob.glob(f"{out}/{name}_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(path)) if "k_generation" in r["Kernel_Name"]]
    break
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
with open(f"{out}/{name}_durations.txt", "w") as fh:
    fh.write(f"# {name}: k_generation dispatches in launch order, microseconds ({len(durs)} launches)\n")
    fh.write(" ".join(f"{d:.1f}" for d in durs) + "\n")
print(open(f"{out}/{name}_durations.txt").read()[-1500:])
PY
