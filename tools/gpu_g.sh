#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_cli.py tests/test_golden.py -m gpu -x -q ) 2>&1 | tail -4
for th in 8 12 16; do echo threads $th; LX_HOST_THREADS=$th LX_HOST_TIMING=1 timeout 600 python tools/quick_host_extend.py 100000 4 2>&1 | grep -E "pipeline of|validate|GCUPS" | tail -5; done
