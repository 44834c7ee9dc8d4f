#!/bin/bash
D=gpurun_out/r4c; mkdir -p $D
(timeout 900 python -m pytest tests/test_gpu_level2.py -x -q) > $D/pytest_l2.log 2>&1; tail -5 $D/pytest_l2.log
(LX_HOST_TIMING=1 timeout 900 python bench.py --iterate --steps 4 --warmup 2) > $D/iterate_dev.log 2>&1; grep -v "^\[lx host ms\]   pipeline" $D/iterate_dev.log | tail -14 | cut -c1-900
(timeout 900 python bench.py --iterate --entry host --steps 3 --warmup 1) > $D/iterate_host.log 2>&1; tail -1 $D/iterate_host.log | cut -c1-400
