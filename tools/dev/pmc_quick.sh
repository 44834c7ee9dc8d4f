#!/bin/bash
# two SQ PMC passes of the headline step (instruction counts; wave-cycle breakdown) -> gpurun_out/$1
D=gpurun_out/${1:-pmcq}; mkdir -p $D; R=$PWD
cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/$D/pmc_sq -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}) > $R/$D/pmc_sq.log 2>&1
(timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $R/$D/pmc_sq_wait -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}) > $R/$D/pmc_sq_wait.log 2>&1
cd $R
python - <<PY
import csv, collections, glob
for name in ("pmc_sq", "pmc_sq_wait"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("$D/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "lx::" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        print(name, k[:70], len(n[k]), {c: "%.4g" % (x / len(n[k])) for c, x in v.items()})
PY
