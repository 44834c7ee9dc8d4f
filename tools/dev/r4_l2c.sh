#!/bin/bash
D=gpurun_out/r4k; mkdir -p $D
(timeout 1200 python -m pytest tests/test_gpu_level2.py tests/test_gpu_trace.py -x -q -k "iterate or level2 or widen or device_list") > $D/pytest_l2.log 2>&1; tail -8 $D/pytest_l2.log
(LX_HOST_TIMING=1 timeout 900 python bench.py --iterate --entry host --steps 3 --warmup 2) > $D/iterate_host.log 2>&1; grep "device list work\|lx_iterate_matches_dev:" $D/iterate_host.log | tail -3; tail -1 $D/iterate_host.log | cut -c1-330
(timeout 900 python bench.py --iterate --steps 4 --warmup 2) > $D/iterate_dev.log 2>&1; tail -1 $D/iterate_dev.log | cut -c1-330
(timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu) > $D/pytest_cli.log 2>&1; tail -3 $D/pytest_cli.log
