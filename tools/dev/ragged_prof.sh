#!/bin/bash
# kernel timeline + host marks of one ragged lx_extend_batch call (development aid): bash tools/dev/ragged_prof.sh [dir under gpurun_out]
D=$PWD/gpurun_out/${1:-ragged_prof}; rm -rf $D; mkdir -p $D; R=$PWD
LX_HOST_TIMING=1 python tools/dev/quick_ragged.py > $D/plain.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o k -- python $R/tools/dev/quick_ragged.py > $D/log 2>&1
cd $R
python tools/dev/timeline.py $(find $D -name 'k_kernel_trace.csv' | head -1) 16 > $D/timeline.txt
cat $D/timeline.txt; tail -12 $D/plain.log
