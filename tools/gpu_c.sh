#!/bin/bash
# quick loop: trace-related parity tests + headline/searchn bench phases
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
( time timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_golden.py -m gpu -x -q ) > gpurun_out/r2c/pytest.log 2>&1
tail -4 gpurun_out/r2c/pytest.log
for c in ${CONFIGS:-1 2}; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r2c/bench_c$c.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', d['value'], d['ms_per_step'], d['phase_ms_last_call'])"
done
