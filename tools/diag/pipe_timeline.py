"""Diagnostic: start / end times of the pack and solve kernels of `bench.py --pipeline` from a rocprofv3 kernel trace.
    python tools/diag/pipe_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = "pack" if "cone_pack_kernel" in n else ("solve" if "cone_packed_kernel" in n else None)
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r.get("LDS_Block_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)))
ev.sort()
ev = ev[-24:]
t0 = ev[0][0]
for s, e, k, lds, wg in ev: print(f"{k:5s} start {(s - t0) / 1e3:8.1f} us  end {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  lds {lds} wg {wg}")
