#!/bin/bash
# kernel timeline of the strong-hits list (development aid): bash tools/dev/strong_prof.sh [dir under gpurun_out]
D=$PWD/gpurun_out/${1:-strong_prof}; rm -rf $D; mkdir -p $D; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o k -- python $R/bench.py --ragged --entry list --lq-range 500 800 --strong --no-cpu-baseline --steps 2 --warmup 1 > $D/log 2>&1
cd $R
python tools/dev/timeline.py $(find $D -name 'k_kernel_trace.csv' | head -1) 14 > $D/timeline.txt
cat $D/timeline.txt
