#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['phase_ms_last_call'])"; }
run pass1 --pass1-only
run full ""
( timeout 900 python -m pytest tests/test_gpu_score.py -m gpu -x -q ) 2>&1 | tail -2
