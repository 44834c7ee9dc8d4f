#!/bin/bash
# Level-2-on-device lease: parity tests of the new path, then the CLI at configs[2] scale with the list on the device and (A/B) on the host
D=gpurun_out/r4b; mkdir -p $D
(timeout 900 python -m pytest tests/test_gpu_level2.py -x -q) > $D/pytest_l2.log 2>&1; tail -15 $D/pytest_l2.log
(LX_HOST_TIMING=1 timeout 600 python tools/cli_scale_nucl.py 1000000 100) > $D/cli_nucl_dev.log 2>&1; grep -v "pipeline of 2 chunks\|(1[0-9][0-9][0-9][0-9][0-9] matches)" $D/cli_nucl_dev.log | tail -12
(LAMBDA3_HOST_LIST=1 LX_HOST_TIMING=1 timeout 600 python tools/cli_scale_nucl.py 1000000 100) > $D/cli_nucl_host.log 2>&1; tail -4 $D/cli_nucl_host.log
(timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu) > $D/pytest_cli.log 2>&1; tail -5 $D/pytest_cli.log
