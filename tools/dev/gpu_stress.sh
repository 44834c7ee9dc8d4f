#!/bin/bash
D=gpurun_out/stress; mkdir -p $D
(timeout 1500 python -m pytest tests/test_gpu_level2.py tests/test_gpu_mq.py tests/test_cli.py -x -q -m gpu) > $D/pytest.log 2>&1; tail -4 $D/pytest.log
(timeout 600 python bench.py --iterate --steps 5 --warmup 2) > $D/iterate_dev.log 2>&1; tail -1 $D/iterate_dev.log | cut -c1-300
(timeout 400 python tools/stress_parity.py 150 41) > $D/stress_a.log 2>&1; tail -2 $D/stress_a.log
(timeout 400 python tools/stress_parity.py 150 42 700 100) > $D/stress_b.log 2>&1; tail -2 $D/stress_b.log
(timeout 400 python tools/stress_parity.py 150 43 1000 450) > $D/stress_c.log 2>&1; tail -2 $D/stress_c.log
