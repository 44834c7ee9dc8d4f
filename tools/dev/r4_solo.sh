#!/bin/bash
D=gpurun_out/r4d; mkdir -p $D
(timeout 1200 python -m pytest tests/test_gpu_mq.py -x -q -k "solo") > $D/pytest_solo.log 2>&1; tail -25 $D/pytest_solo.log
(timeout 1200 python -m pytest tests/test_gpu_level2.py tests/test_gpu_mq.py tests/test_gpu_list.py -x -q) > $D/pytest_l2mq.log 2>&1; tail -8 $D/pytest_l2mq.log
(LX_HOST_TIMING=1 timeout 900 python bench.py --iterate --steps 4 --warmup 2) > $D/iterate_dev.log 2>&1; tail -5 $D/iterate_dev.log | cut -c1-900
(LX_HOST_TIMING=1 timeout 600 python tools/cli_scale_nucl.py 1000000 100) > $D/cli_nucl_dev.log 2>&1; grep -v "pipeline of 2 chunks\|(1[0-9][0-9][0-9][0-9][0-9] matches)" $D/cli_nucl_dev.log | tail -8
