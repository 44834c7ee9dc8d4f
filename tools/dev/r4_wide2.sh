#!/bin/bash
D=gpurun_out/r4f; mkdir -p $D; R=$PWD
(LX_HOST_TIMING=1 timeout 300 python tools/dev/long_queries.py) > $D/long_queries.log 2>&1; tail -6 $D/long_queries.log
cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats_long -o long -- python $R/tools/dev/long_queries.py) > $R/$D/stats_long.log 2>&1
cd $R; f=$(find $D/stats_long -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
