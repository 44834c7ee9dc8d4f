#!/bin/bash
# one GPU lease: the Level-2 tests, a short randomised stress of them, the --iterate line with the host's phase marks, and the kernel
# statistics of the same command
D=gpurun_out/${1:-l2q}; mkdir -p $D; R=$PWD
(timeout 900 python -m pytest tests/test_gpu_level2.py -x -q) > $D/pytest.log 2>&1; tail -3 $D/pytest.log
(timeout 300 python tools/stress_level2.py 60 7) > $D/stress.log 2>&1; tail -2 $D/stress.log
(LX_HOST_TIMING=1 timeout 600 python bench.py --iterate --steps 10 --warmup 3 --no-cpu-baseline) > $D/iterate.log 2>&1
grep "lx host ms" $D/iterate.log | tail -24
tail -1 $D/iterate.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_min'], d['ms_max'])"
cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats_iterate -o iterate -- python $R/bench.py --iterate --steps 3 --warmup 2 --no-cpu-baseline) > $R/$D/stats_iterate.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('$R/$D/stats_iterate/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:22]:
    print(r['Name'][:80].ljust(82), r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
