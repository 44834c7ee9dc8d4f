#!/bin/bash
D=gpurun_out/r4g; mkdir -p $D
(timeout 1800 python -m pytest tests/test_gpu_mq.py tests/test_gpu_list.py tests/test_gpu_level2.py -x -q) > $D/pytest_mq.log 2>&1; tail -12 $D/pytest_mq.log
(timeout 300 python tools/dev/long_queries.py) > $D/long_queries.log 2>&1; tail -1 $D/long_queries.log
(LX_MQ_NO_MERGE=1 timeout 300 python tools/dev/long_queries.py) > $D/long_queries_nomerge.log 2>&1; tail -1 $D/long_queries_nomerge.log
(LX_HOST_TIMING=1 timeout 600 python bench.py --ragged --entry list --steps 5 --warmup 2) > $D/bench_ragged_list.log 2>&1; tail -3 $D/bench_ragged_list.log | cut -c1-330
(LX_MQ_NO_MERGE=1 timeout 600 python bench.py --ragged --entry list --steps 5 --warmup 2) > $D/bench_ragged_list_nomerge.log 2>&1; tail -1 $D/bench_ragged_list_nomerge.log | cut -c1-330
