#!/usr/bin/env python3
"""Development aid: the kernels of ONE Level-2 call (the last complete one in a rocprofv3 --kernel-trace csv of bench.py --iterate), in launch order
with the idle time in front of each, consecutive launches of one kernel folded together.
    python tools/dev/l2_timeline.py gpurun_out/iterate_prof/k_kernel_trace.csv"""
import csv, re, sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("lx::", ""))
idx = [i for i, r in enumerate(rows) if "l2_keys" in r["Kernel_Name"]]
start, end = idx[-2], idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
agg, prev, last_end = [], None, t0
for r in rows[start:end]:
    n, a, b = name(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, a - last_end) / 1e3
    if prev and prev[0] == n and gap < 8:
        prev[2], prev[3], prev[4] = b, prev[3] + 1, prev[4] + (b - a) / 1e3
    else:
        prev = [n, a, b, 1, (b - a) / 1e3, gap]
        agg.append(prev)
    last_end = max(last_end, b)
for n, a, b, c, busy, gap in agg:
    print("%9.3f ms  idle %7.1f us in front  x%-3d busy %8.1f us  %s" % ((a - t0) / 1e6, gap, c, busy, n[:60]))
print("span %.3f ms" % ((last_end - t0) / 1e6))
