#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -3
run() { env $3 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['phase_ms_last_call'], d['roofline']['kernel'][:40])"; }
run "lq100 narrow" "--lq 100 --queries 150000"
run "lq100 (8,19)" "--lq 100 --queries 150000" LX_NO_NARROW_SWEEP=1
run "headline" ""
python tools/quick_ragged.py 50000 2>&1 | grep ragged
