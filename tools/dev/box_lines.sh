#!/bin/bash
# the headline and the Level-2 lines on THIS box (development aid; run as three separate gpurun calls = three boxes of the pool):
#   bash tools/dev/box_lines.sh <name under gpurun_out>
D=gpurun_out/${1:-box}; mkdir -p $D; uptime > $D/uptime.txt
(timeout 600 python bench.py --steps 10 --warmup 2) > $D/bench.log 2>&1
(timeout 900 python bench.py --iterate --config 1 --steps 8 --warmup 2) > $D/bench_iterate_protein.log 2>&1
(timeout 900 python bench.py --iterate --steps 8 --warmup 2) > $D/bench_iterate_dev.log 2>&1
(timeout 900 python bench.py --ragged --entry list --steps 5 --warmup 2) > $D/bench_ragged_list.log 2>&1
(timeout 900 python bench.py --ragged --entry list --lq-range 500 800 --strong --steps 5 --warmup 3) > $D/bench_ragged_long_strong.log 2>&1
(timeout 600 python bench.py --host-path --entry list --steps 5 --warmup 2) > $D/bench_host_path_list.log 2>&1
python - "$D" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench*.log")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline") or {}
        print("%-34s %9.2f %-6s ms %.3f sweep %s backtrace %s" % (f.split("/")[-1], d["value"], d["unit"][:6], d["ms_per_step"], r.get("kernel_ms_per_call") or r.get("kernel_ms_per_launch"), r.get("backtrace_ms_per_call")))
    except Exception as e:
        print(f, "failed:", e)
PY
