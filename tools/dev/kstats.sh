#!/bin/bash
# per-kernel durations of the headline step in launch order (development aid): bash tools/dev/kstats.sh [bench args]
D=$PWD/gpurun_out/kstats; rm -rf $D; mkdir -p $D; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $D/log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("$D/**/k_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "lx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last score kernel on
last = max(i for i, r in enumerate(rows) if "score_pair" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    n = r["Kernel_Name"].replace("void lx::", "").split("(")[0]
    print("%8.3f ms +%7.3f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, n[:60]))
PY
