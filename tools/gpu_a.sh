#!/bin/bash
# round-2 first GPU pass: parity suite + the four BASELINE configs through bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a/pytest.log 2>&1
tail -5 gpurun_out/r2a/pytest.log
for c in 1 2 3 4; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 2 > gpurun_out/r2a/bench_c$c.json 2> gpurun_out/r2a/bench_c$c.err
  echo "config $c rc=$?"; head -c 600 gpurun_out/r2a/bench_c$c.json; echo
done
