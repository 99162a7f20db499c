#!/usr/bin/env python
"""Summarise the rocprofv3 passes of tools/diag/pmc_run.sh into profiles/<tag>_pmc_summary.json.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE
are collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced streaming read (the scan uses 16-byte loads), so the read side is doubled."""
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof2"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
kern = sys.argv[3] if len(sys.argv) > 3 else "cone_dense_kernel"
out = {"kernel": kern}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    fs = sorted(glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not fs:
        continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if kern in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
        out[k + "_launches"] = len(v)
fs = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if kern in r["Name"]:
            out["kernel_trace_avg_ns"] = float(r["AverageNs"])
            out["kernel_trace_calls"] = int(r["Calls"])
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_read_bytes_per_launch"] = out["FETCH_SIZE"] * 1024 * 2   # gfx950 half-count correction
    out["hbm_write_bytes_per_launch"] = out["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
out["workload"] = sys.argv[4] if len(sys.argv) > 4 else "tsp20_b1024_inner"
path = os.path.join("profiles", f"{tag}_pmc_summary.json")
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(out))
