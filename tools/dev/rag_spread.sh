#!/bin/bash
# the ragged list line three times on this box, with what the host did (development aid): bash tools/dev/rag_spread.sh <tag>
D=gpurun_out/rag_spread_${1:-a}; mkdir -p $D
nproc; uptime
for i in 1 2 3; do (timeout 900 python bench.py --ragged --entry list --steps 5 --warmup 2) > $D/run$i.log 2>&1; python -c "
import json,sys
l=[x for x in open('$D/run$i.log') if x.startswith('{')][-1]; d=json.loads(l); r=d['roofline']; print('ragged', d['ms_per_step'], d['value'], r['kernel_ms_per_call'], r['backtrace_ms_per_call'])"; done
LX_HOST_TIMING=1 python bench.py --ragged --entry list --no-cpu-baseline --steps 3 --warmup 2 2>&1 | grep "lx_extend_batch_list:" | tail -3
