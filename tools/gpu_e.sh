#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2e
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2e/pytest.log 2>&1
tail -3 gpurun_out/r2e/pytest.log
timeout 900 python bench.py --band 64 --steps 5 --warmup 2 > gpurun_out/r2e/bench_band64.json 2> gpurun_out/r2e/bench_band64.err; echo rc=$?
python -c "
import json
d=json.loads(open('gpurun_out/r2e/bench_band64.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms_last_call'], d['roofline']['kernel'])"
