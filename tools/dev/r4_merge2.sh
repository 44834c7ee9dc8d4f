#!/bin/bash
D=gpurun_out/r4h; mkdir -p $D
(LX_HOST_TIMING=1 timeout 300 python tools/dev/long_queries.py) > $D/long_queries.log 2>&1; tail -4 $D/long_queries.log
(LX_MQ_NO_MERGE=1 timeout 300 python tools/dev/long_queries.py) > $D/long_queries_nomerge.log 2>&1; tail -1 $D/long_queries_nomerge.log
(timeout 300 python tools/dev/long_queries.py 300 500) > $D/long_queries_300.log 2>&1; tail -1 $D/long_queries_300.log
(LX_MQ_NO_MERGE=1 timeout 300 python tools/dev/long_queries.py 300 500) > $D/long_queries_300_nomerge.log 2>&1; tail -1 $D/long_queries_300_nomerge.log
