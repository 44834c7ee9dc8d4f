D=gpurun_out/r6_prot; mkdir -p $D; R=$PWD
python bench.py --iterate --config 1 --steps 8 --warmup 2 --no-cpu-baseline > $D/line.json 2> $D/line.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats -o it -- python $R/bench.py --iterate --config 1 --steps 4 --warmup 2 --no-cpu-baseline > $R/$D/stats.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6_prot/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:45]: print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
cut -c1-400 $D/line.json
