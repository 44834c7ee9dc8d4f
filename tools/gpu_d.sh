#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['phase_ms_last_call']['backtrace'])"; }
for ta in 64 56 48 44 40 32 24; do for ra in 4 12 32; do LX_BT_TILE_AT=$ta LX_BT_REFILL_AT=$ra run "tile_at=$ta refill_at=$ra"; done; done
( timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_golden.py -m gpu -x -q ) 2>&1 | tail -2
