#!/bin/bash
# per-kernel time of one ragged lx_extend_batch call (development aid): bash tools/dev/ragged_prof.sh
D=$PWD/gpurun_out/ragged_prof; rm -rf $D; mkdir -p $D; R=$PWD
LX_HOST_TIMING=1 python tools/quick_ragged.py > $D/plain.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python $R/tools/quick_ragged.py > $D/log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("$D/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lx::" in r["Name"]:
        print("%6s calls %10.3f ms total %8.3f avg  %s" % (r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Name"].replace("void lx::", "")[:90]))
PY
tail -5 $D/plain.log
