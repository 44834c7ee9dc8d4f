#!/bin/bash
# per-kernel time of any command (development aid): bash tools/dev/prof_cmd.sh <tag> <command ...>
TAG=$1; shift
D=$PWD/gpurun_out/$TAG; rm -rf $D; mkdir -p $D; R=$PWD
LX_HOST_TIMING=1 "$@" > $D/plain.log 2>&1
cd /tmp; export TMPDIR=/tmp
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- "$@" > $D/log 2>&1 )
cd $R
python - <<PY
import csv, glob
f = glob.glob("$D/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lx::" in r["Name"]:
        print("%6s calls %10.3f ms total %8.3f avg  %s" % (r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Name"].replace("void lx::", "")[:100]))
PY
tail -5 $D/plain.log
