#!/bin/bash
# headline step: phase times (HIP events) of the last call, three runs
for k in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms_last_call'])"; done
