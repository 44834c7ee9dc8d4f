#!/bin/bash
D=gpurun_out/check; mkdir -p $D
(timeout 2400 python -m pytest tests -m gpu -x -q) > $D/pytest.log 2>&1; tail -6 $D/pytest.log
(timeout 600 python bench.py --ragged --entry list --steps 5 --warmup 2) > $D/bench_ragged_list.log 2>&1; tail -1 $D/bench_ragged_list.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:900]); print(d['cpu_baseline'])"
(timeout 600 python bench.py --host-path --entry list --steps 5 --warmup 2) > $D/bench_host_list.log 2>&1; tail -1 $D/bench_host_list.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:600]); print(d['cpu_baseline'])"
(timeout 600 python bench.py --steps 10 --warmup 2) > $D/bench.log 2>&1; tail -1 $D/bench.log | cut -c1-400
(timeout 600 python bench.py --iterate --steps 10 --warmup 3) > $D/bench_iterate.log 2>&1; tail -1 $D/bench_iterate.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_min'], json.dumps(d['roofline'])[:700]); print(d['cpu_baseline'])"
