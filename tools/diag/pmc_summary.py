#!/usr/bin/env python
"""Summarise the rocprofv3 passes of tools/diag/pmc_run.sh into profiles/<tag>_pmc_summary.json.

    python tools/diag/pmc_summary.py <rocprof dir> <tag> <kernel[,kernel...]> <workload>

The step may consist of several kernels (the split form of the dense operator launches a pack kernel and a
solve kernel): every listed kernel is summarised on its own and the per-STEP HBM traffic is their sum.
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of the touched 128-byte lines
(streams and gathers alike: profiles/r04_fetch_size_gather_check.txt), so the read side is doubled.
PMC_GRID=<threads>: only launches of that grid size (the fused step: 2048 workgroups x 128 threads = 262144)."""
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r04"
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
kerns = (sys.argv[3] if len(sys.argv) > 3 else "cone_step_kernel").split(",")
out = {"kernels": {}, "workload": sys.argv[4] if len(sys.argv) > 4 else "tsp20_b1024_inner",
       "split": any("cone_pack_kernel" in k for k in kerns), "fused": any("cone_step_kernel" in k for k in kerns),
       "fetch_size_note": "FETCH_SIZE x 2: the counter reports half of the bytes of the touched 128-byte lines for streams AND "
                          "gathers (profiles/r04_fetch_size_gather_check.txt)"}
tot_r = tot_w = 0.0
for kern in kerns:
    o = {}
    for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_ic", "pmc_inst", "pmc_ic2"):
        fs = sorted(glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
        if not fs:
            continue
        acc = {}
        for r in csv.DictReader(open(fs[0])):
            if kern in r["Kernel_Name"] and (not os.environ.get("PMC_GRID") or r.get("Grid_Size") == os.environ["PMC_GRID"]):
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            o[k] = sum(v) / len(v)
            o[k + "_launches"] = len(v)
    fs = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    if fs:
        for r in csv.DictReader(open(fs[0])):
            if kern in r["Name"]:
                o["kernel_trace_avg_ns"] = float(r["AverageNs"])
                o["kernel_trace_calls"] = int(r["Calls"])
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_read_bytes_per_launch"] = o["FETCH_SIZE"] * 1024 * 2   # gfx950 half-count correction
        o["hbm_write_bytes_per_launch"] = o["WRITE_SIZE"] * 1024
        tot_r += o["hbm_read_bytes_per_launch"]
        tot_w += o["hbm_write_bytes_per_launch"]
    if "SQ_WAVE_CYCLES" in o:
        o["wait_share"] = o.get("SQ_WAIT_ANY", 0.0) / o["SQ_WAVE_CYCLES"]
        o["issue_share"] = o.get("SQ_ACTIVE_INST_ANY", 0.0) / o["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_INST_ANY" in o:
            o["issue_stall_share"] = o["SQ_WAIT_INST_ANY"] / o["SQ_WAVE_CYCLES"]
    if o.get("SQC_ICACHE_REQ"):
        o["icache_miss_rate"] = o.get("SQC_ICACHE_MISSES", 0.0) / o["SQC_ICACHE_REQ"]
    out["kernels"][kern] = o
if tot_r:
    out["hbm_read_bytes_per_launch"], out["hbm_write_bytes_per_launch"] = tot_r, tot_w
    out["hbm_bytes_per_launch"] = tot_r + tot_w
path = os.path.join("profiles", f"{tag}_pmc_summary.json")
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(out))
