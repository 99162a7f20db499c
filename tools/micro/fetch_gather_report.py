#!/usr/bin/env python
"""FETCH_SIZE (KiB) of the six kernels of tools/micro/fetch_gather.hip against the bytes they are known to touch."""
import csv, glob, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fetch_gather"
GiB = 1 << 30
# kernel order = launch order; (label, distinct bytes requested, bytes of the touched 64-B lines, of the touched 128-B lines)
cases = [("stream16  16 B/lane consecutive", GiB, GiB, GiB), ("stream4    4 B/lane consecutive", GiB, GiB, GiB),
         ("stream2    2 B/lane consecutive", GiB, GiB, GiB), ("gather8_64   8 B per 64-B line", GiB // 8, GiB, GiB),
         ("gather8_128  8 B per 128-B line", GiB // 16, GiB // 2, GiB), ("gather2_32   2 B per 32 B", GiB // 16, GiB, GiB)]
rows = []
for f in glob.glob(src + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and ("stream" in r["Kernel_Name"] or "gather" in r["Kernel_Name"]):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024))
rows.sort()
print(f"{'case':34s} {'FETCH_SIZE bytes':>16s} {'/ requested':>12s} {'/ 64-B lines':>13s} {'/ 128-B lines':>14s}")
for (label, req, l64, l128), (_, name, val) in zip(cases, rows):
    print(f"{label:34s} {val:16.0f} {val / req:12.3f} {val / l64:13.3f} {val / l128:14.3f}")
