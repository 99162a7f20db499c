#!/usr/bin/env python
"""rocprofv3 --kernel-trace output -> per-kernel, per-launch-shape table (the `--stats` summary merges the three grid
shapes of cone_step_kernel -- fused, solve-only, pack-only -- into one row).
    python tools/diag/kernel_stats_by_grid.py gpurun_out/prof_r04 > profiles/r04_kernel_stats_by_grid.txt"""
import csv, glob, os, sys, collections
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r04"
fs = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
rows = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    name = r["Kernel_Name"]
    if "cone_" not in name and "lite_" not in name: continue
    name = name.split("(")[0].replace("void cave::", "")
    wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"])
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
    rows[(name, grid // wg, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --kernel-trace of `python bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-extras` (tools/diag/pmc_run.sh),")
print("# the cone kernels split by launch shape: cone_step_kernel with 2048 workgroups = the fused step (1024 solve + 1024 pack")
print("# blocks); with 1024 = the pack-only / solve-only launches (set-up, the back-to-back form timed beside the fused one).")
print(f"{'kernel':58s} {'blocks':>7s} {'threads':>7s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s}")
for (name, blocks, wg), v in sorted(rows.items()):
    print(f"{name[:58]:58s} {blocks:7d} {wg:7d} {len(v):6d} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f}")
