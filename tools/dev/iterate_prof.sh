#!/bin/bash
# kernel timeline + host marks of one Level-2 call on the device match list (development aid):
#   bash tools/dev/iterate_prof.sh [dir under gpurun_out] [extra bench args, e.g. "--config 1"]
D=$PWD/gpurun_out/${1:-iterate_prof}; rm -rf $D; mkdir -p $D; R=$PWD; X="${2:-}"
LX_HOST_TIMING=1 python bench.py --iterate $X --no-cpu-baseline --steps 3 --warmup 2 > $D/plain.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -o k -- python $R/bench.py --iterate $X --no-cpu-baseline --steps 2 --warmup 2 > $D/log 2>&1
cd $R
python tools/dev/timeline.py $(find $D -name 'k_kernel_trace.csv' | head -1) ${3:-140} > $D/timeline.txt
python tools/dev/l2_timeline.py $(find $D -name 'k_kernel_trace.csv' | head -1) > $D/l2_timeline.txt
tail -80 $D/l2_timeline.txt; grep "lx host ms" $D/plain.log | tail -30
