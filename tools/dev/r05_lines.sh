#!/bin/bash
# the secondary bench lines VERDICT r4 item 1 names, one summary row each (development aid): bash tools/dev/r05_lines.sh <dir under gpurun_out>
D=gpurun_out/${1:-r05_lines}; mkdir -p $D
python bench.py --ragged --entry list --no-cpu-baseline > $D/ragged.json 2>$D/err.txt
python bench.py --ragged --entry list --lq-range 500 800 --strong --no-cpu-baseline > $D/strong.json 2>>$D/err.txt
python bench.py --iterate --no-cpu-baseline > $D/iterate.json 2>>$D/err.txt
python - <<PY
import json
for f in ("ragged", "strong", "iterate"):
    try:
        d = json.load(open("$D/%s.json" % f)); r = d["roofline"]
        print("%-8s %9.2f %-5s ms/step %7.3f  frac %.4f exec %.4f  sweep %.3f ms x%s  bt %.3f ms" % (f, d["value"], d["unit"], d["ms_per_step"], r["frac"], r.get("frac_executed") or 0,
              r.get("kernel_ms_per_call") or 0, r.get("launches_per_call"), r.get("backtrace_ms_per_call") or 0))
    except Exception as e:
        print(f, "failed:", e)
PY
