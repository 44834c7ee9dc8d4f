#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2b/pytest.log 2>&1
tail -5 gpurun_out/r2b/pytest.log
for c in 1 2; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_c$c.json 2> gpurun_out/r2b/bench_c$c.err
  echo "config $c rc=$?"; python -c "
import json,sys
d=json.loads(open('gpurun_out/r2b/bench_c$c.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms_last_call'])"
done
for w in 8 16; do LX_BT_WAVES_PER_CU=$w timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves/cu $w', d['value'], d['ms_per_step'], d['phase_ms_last_call'])"; done
