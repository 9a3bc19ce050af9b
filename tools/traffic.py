#!/usr/bin/env python3
"""Turn tools/traffic.sh's counter CSVs into profiles/<round>/traffic.json (bytes per launch of
the generation kernel, corrected by the calibration kernel's known byte count)."""
import csv, json, sys, collections
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traffic"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r1/traffic.json"
prefix = sys.argv[3] if len(sys.argv) > 3 else "bench"  # file prefix of the run (tools/counters_cfg.sh: the workload)
KNOWN = {"FETCH_SIZE": 1000000 * 13 * 8, "WRITE_SIZE": 1000000 * 28 * 8}
out = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; counter unit KiB; "
                 "scale = known bytes of tools/ubench/copy_f64 (same 8 B/lane row-major pattern) / its counter"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{src}/cal_{c}_counter_collection.csv"))
           if r["Kernel_Name"].startswith("copy_rows") and r["Counter_Name"] == c]
    cal_kib = sum(cal[1:]) / len(cal[1:])
    scale = KNOWN[c] / (cal_kib * 1024)
    vals = []
    for r in csv.DictReader(open(f"{src}/{prefix}_{c}_counter_collection.csv")):
        if "k_generation" in r["Kernel_Name"] and r["Counter_Name"] == c:
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            vals.append((float(r["Counter_Value"]), dur))
    real = [v for v, d in vals if d > 20000]  # launches that found rays (the 4th of a batch exits)
    out[c] = {"calibration_counter_KiB": cal_kib, "calibration_known_bytes": KNOWN[c], "scale": scale,
              "raw_KiB_per_working_launch": sum(real) / len(real),
              "bytes_per_working_launch": sum(real) / len(real) * 1024 * scale,
              "working_launches": len(real), "all_launches": len(vals),
              "bytes_per_launch_all": sum(v for v, _ in vals) * 1024 * scale / len(vals)}
import hashlib, os
_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyrayt_amd", "csrc", "libprt_hip.so")
out["library_sha16"] = hashlib.sha256(open(_lib, "rb").read()).hexdigest()[:16]  # the build the counters were read from (bench.py checks it)
out["hbm_bytes_per_launch"] = out["FETCH_SIZE"]["bytes_per_launch_all"] + out["WRITE_SIZE"]["bytes_per_launch_all"]
out["hbm_bytes_per_working_launch"] = out["FETCH_SIZE"]["bytes_per_working_launch"] + out["WRITE_SIZE"]["bytes_per_working_launch"]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
