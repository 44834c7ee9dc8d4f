#!/bin/bash
# backtrace scheduling thresholds on the headline batch: ms of ckpt_backtrace_kernel<8,19> per (LX_BT_TILE_AT, LX_BT_REFILL_AT)
D=gpurun_out/bts; mkdir -p $D
for t in 0 24 32 48 56; do for r in 0 6 20 32; do
  (LX_BT_TILE_AT=$t LX_BT_REFILL_AT=$r timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline) > $D/t$t.r$r.log 2>&1
  echo tile_at $t refill_at $r $(tail -1 $D/t$t.r$r.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['phase_ms_last_call']['backtrace'], d['ms_per_step'])")
done; done
