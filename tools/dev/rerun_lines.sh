#!/bin/bash
# the secondary lines three times each on one box; the MEDIAN run becomes the line tools/collect_profiles.py picks up, all values go
# to <dir>/line_spread.txt (development aid): bash tools/dev/rerun_lines.sh <dir under gpurun_out>
D=gpurun_out/${1:-r05prof}; mkdir -p $D
run() { # name, args...
  n=$1; shift
  for i in 1 2 3; do (timeout 900 python bench.py "$@") > $D/$n.run$i.log 2>&1; done
  python - "$D" "$n" <<'PY'
import json, sys, shutil
d, n = sys.argv[1], sys.argv[2]
vals = []
for i in (1, 2, 3):
    try:
        line = [l for l in open(f"{d}/{n}.run{i}.log") if l.startswith("{")][-1]
        vals.append((json.loads(line)["ms_per_step"], i))
    except Exception as e:
        print(n, "run", i, "failed:", e)
vals.sort()
if vals:
    med = vals[len(vals) // 2][1]
    shutil.copy(f"{d}/{n}.run{med}.log", f"{d}/{n}.log")
    open(f"{d}/line_spread.txt", "a").write("%-28s ms per step of three runs on one box: %s -> the median run is the committed line\n" % (n, ", ".join("%.3f" % v for v, _ in vals)))
PY
}
rm -f $D/line_spread.txt
run bench_iterate_dev --iterate --steps 5 --warmup 2
run bench_iterate_cold --iterate --cold --steps 5 --warmup 1 --no-cpu-baseline
run bench_ragged_list --ragged --entry list --steps 5 --warmup 2
run bench_ragged_long_strong --ragged --entry list --lq-range 500 800 --strong --steps 5 --warmup 3
run bench_ragged_nucl --ragged --entry list --config 2 --steps 5 --warmup 2
cat $D/line_spread.txt
