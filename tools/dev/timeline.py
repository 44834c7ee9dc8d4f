#!/usr/bin/env python3
"""Development aid: the kernel timeline of the LAST call in a rocprofv3 --kernel-trace csv (start / end in ms from the first
kernel of the call, by stream / queue), to see which kernels really ran side by side.
    python tools/dev/timeline.py gpurun_out/<dir>/..._kernel_trace.csv [n_last]"""
import csv, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = r["Kernel_Name"].replace("void lx::", "").split("(")[0][:48]
    print("%8.3f - %8.3f (%6.3f) q%-3s %s" % (a, b, b - a, r.get("Queue_Id", "?"), name))
