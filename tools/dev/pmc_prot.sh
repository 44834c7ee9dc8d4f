D=gpurun_out/r6_prot_pmc; mkdir -p $D; R=$PWD
cd /tmp; export TMPDIR=/tmp
for p in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "sq_wait:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${p%%:*}; c=${p#*:}
  (timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$D/pmc_$n -o pmc -- python $R/bench.py --iterate --config 1 --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_$n.log 2>&1
done
cd $R
for e in 1 2 3 4; do LX_L2_RANGES=$e python bench.py --iterate --config 1 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ranges $e', d['value'], d['ms_min'], r['kernel_ms_per_call'], r['backtrace_ms_per_call'])"; done
python - <<'PY'
import csv,collections,glob
for n in ('sq','sq_wait','fetch','write'):
    f=glob.glob(f'gpurun_out/r6_prot_pmc/pmc_{n}/**/*counter_collection.csv',recursive=True)
    if not f: print(n,'missing'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); L=collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        if 'sweep_mq' in k or 'backtrace' in k:
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); L[k].add(r['Dispatch_Id'])
    for k,v in acc.items():
        print(n,k[:60],len(L[k]),{c:round(x/len(L[k])) for c,x in v.items()})
PY
