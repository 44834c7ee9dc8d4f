#!/bin/bash
# the Level-2 lines three more times on this box, with its load (development aid): bash tools/dev/rerun_iterate.sh <dir under gpurun_out>
D=gpurun_out/${1:-iter_more}; mkdir -p $D
uptime | tee $D/uptime.txt
for i in 1 2 3; do (timeout 900 python bench.py --iterate --steps 5 --warmup 2) > $D/bench_iterate_dev.run$i.log 2>&1; (timeout 900 python bench.py --iterate --cold --steps 5 --warmup 1 --no-cpu-baseline) > $D/bench_iterate_cold.run$i.log 2>&1; done
python - <<PY
import json
for n in ("bench_iterate_dev", "bench_iterate_cold"):
    v = []
    for i in (1, 2, 3):
        l = [x for x in open("$D/%s.run%d.log" % (n, i)) if x.startswith("{")][-1]
        v.append(json.loads(l)["ms_per_step"])
    print(n, v)
PY
uptime
