#!/bin/bash
D=gpurun_out/r4e; mkdir -p $D
(timeout 1200 python -m pytest tests/test_gpu_mq.py -x -q -k "wide or solo or host_plan") > $D/pytest_wide.log 2>&1; tail -25 $D/pytest_wide.log
(timeout 300 python tools/dev/long_queries.py) > $D/long_queries.log 2>&1; tail -2 $D/long_queries.log
(LX_MQ_NO_WIDE=1 timeout 300 python tools/dev/long_queries.py) > $D/long_queries_nowide.log 2>&1; tail -1 $D/long_queries_nowide.log
(timeout 600 python bench.py --ragged --entry list --steps 5 --warmup 2) > $D/bench_ragged_list.log 2>&1; tail -1 $D/bench_ragged_list.log | cut -c1-300
