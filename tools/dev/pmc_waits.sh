#!/bin/bash
# where the wavefronts of the sweeps wait: LDS / VMEM / instruction-fetch counters of the multi-query sweep (ragged list) beside the
# headline sweep's.  Separate passes, --pmc only (no tracing).  usage: bash tools/dev/pmc_waits.sh <dir under gpurun_out>
D=gpurun_out/${1:-pw}; mkdir -p $D; R=$PWD
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
P2="SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
P3="SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH"
P4="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  (timeout 600 rocprofv3 --pmc $P --output-format csv -d $R/$D/rag$i -o pmc -- python $R/bench.py --ragged --entry list --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/rag$i.log 2>&1
  (timeout 600 rocprofv3 --pmc $P --output-format csv -d $R/$D/head$i -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/head$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for tag in ("rag", "head"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob("$R/$D/%s*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sweep_mq" not in k and "score_pair" not in k and "backtrace" not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        print(tag, k[:70])
        for c in sorted(acc[k]):
            print("    %-34s %16.0f per dispatch (%d dispatches)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
PY
