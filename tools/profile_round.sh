#!/bin/bash
# Round profile of the headline step (BASELINE.json configs[1] through bench.py): the bench lines, rocprofv3 kernel stats, and
# the PMC passes (SQ instruction counts; SQ wave-cycle breakdown; FETCH_SIZE; WRITE_SIZE -- separate runs, never combined with
# tracing other than --kernel-trace).  usage: bash tools/profile_round.sh <dir under gpurun_out>; then
# python tools/collect_profiles.py gpurun_out/<dir> rNN
D=gpurun_out/${1:-r2prof}; mkdir -p $D; R=$PWD
# second argument: "profiles" = only the rocprofv3 passes, "lines" = only the bench lines (run them AFTER tools/collect_profiles.py has
# condensed the passes: a line's `traffic` is read from the committed profile of the kernel sources it runs), default both
M=${2:-all}
[ $M = profiles ] || { (timeout 600 python bench.py --steps 10 --warmup 2) > $D/bench.log 2>&1 ; }
[ $M = profiles ] || { (timeout 600 python bench.py --steps 10 --warmup 2 --pass1-only --no-cpu-baseline) > $D/bench_pass1.log 2>&1 ; }
[ $M = profiles ] || { for c in 2 3 4; do (timeout 900 python bench.py --config $c --steps 5 --warmup 2) > $D/bench_config$c.log 2>&1; done }
[ $M = profiles ] || { (timeout 600 python bench.py --band 64 --steps 5 --warmup 2) > $D/bench_band64.log 2>&1 ; }
[ $M = profiles ] || { (timeout 600 python bench.py --host-path --steps 5 --warmup 2) > $D/bench_host_path.log 2>&1 ; }
[ $M = profiles ] || { (timeout 900 python bench.py --ragged --steps 5 --warmup 2) > $D/bench_ragged.log 2>&1 ; }
[ $M = profiles ] || { for e in rle list; do (timeout 600 python bench.py --host-path --entry $e --steps 5 --warmup 2) > $D/bench_host_path_$e.log 2>&1; (timeout 900 python bench.py --ragged --entry $e --steps 5 --warmup 2) > $D/bench_ragged_$e.log 2>&1; done }
[ $M = profiles ] || { for r in 0.02 0.1; do (timeout 600 python bench.py --steps 10 --warmup 3 --survivor-rate $r --no-cpu-baseline) > $D/bench_survivors_$r.log 2>&1; done }
[ $M = profiles ] || { (timeout 600 python bench.py --steps 10 --warmup 3 --survivor-rate 0.02 --adapt-permille 0 --no-cpu-baseline) > $D/bench_survivors_0.02_sweep.log 2>&1 ; }
[ $M = profiles ] || { (timeout 600 python tools/host_curve.py 100) > $D/host_curve.jsonl 2>&1 ; }
cd /tmp; export TMPDIR=/tmp
[ $M = lines ] || { (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats -o bench -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline) > $R/$D/stats.log 2>&1 ; }
[ $M = lines ] || { (timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/$D/pmc_sq -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_sq.log 2>&1 ; }
[ $M = lines ] || { (timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $R/$D/pmc_sq_wait -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_sq_wait.log 2>&1 ; }
[ $M = lines ] || { (timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$D/pmc_fetch -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_fetch.log 2>&1 ; }
[ $M = lines ] || { (timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$D/pmc_write -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_write.log 2>&1 ; }
[ $M = lines ] || { (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats_ragged -o ragged -- python $R/bench.py --ragged --entry list --steps 5 --warmup 2) > $R/$D/stats_ragged.log 2>&1 ; }
[ $M = lines ] || { (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats_host -o host -- python $R/bench.py --host-path --entry list --steps 5 --warmup 2) > $R/$D/stats_host.log 2>&1 ; }
[ $M = lines ] || { (timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$D/pmc_write_lowsurv -o pmc -- python $R/bench.py --steps 2 --warmup 3 --survivor-rate 0.02 --no-cpu-baseline) > $R/$D/pmc_write_lowsurv.log 2>&1 ; }
# round 4: the Level-2 driver, the long strong-hit list, PMC passes of the multi-query sweep on the ragged list
cd $R
[ $M = profiles ] || { (timeout 900 python bench.py --iterate --steps 5 --warmup 2) > $D/bench_iterate_dev.log 2>&1 ; }
[ $M = profiles ] || { (timeout 900 python bench.py --iterate --entry host --steps 3 --warmup 2) > $D/bench_iterate_host.log 2>&1 ; }
[ $M = profiles ] || { (timeout 900 python bench.py --ragged --entry list --lq-range 500 800 --strong --steps 5 --warmup 3) > $D/bench_ragged_long_strong.log 2>&1 ; }
[ $M = profiles ] || { (LX_MQ_NO_WIDE=1 timeout 900 python bench.py --ragged --entry list --lq-range 500 800 --strong --steps 3 --warmup 2 --no-cpu-baseline) > $D/bench_ragged_long_strong_codes_only.log 2>&1 ; }
[ $M = profiles ] || { (timeout 900 python bench.py --ragged --entry list --config 2 --steps 5 --warmup 2) > $D/bench_ragged_nucl.log 2>&1 ; }
[ $M = profiles ] || { (LX_HOST_TIMING=1 timeout 900 python tools/cli_scale_nucl.py 1000000 100) > $D/cli_nucl.log 2>&1 ; }
[ $M = profiles ] || { (LAMBDA3_HOST_LIST=1 LX_ITERATE_ON_HOST=1 timeout 900 python tools/cli_scale_nucl.py 1000000 100) > $D/cli_nucl_host_list.log 2>&1 ; }
cd /tmp
for p in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "sq_wait:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${p%%:*}; c=${p#*:}
[ $M = lines ] || {   (timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$D/pmc_ragged_$n -o pmc -- python $R/bench.py --ragged --entry list --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_ragged_$n.log 2>&1 ; }
done
[ $M = lines ] || { (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats_iterate -o iterate -- python $R/bench.py --iterate --steps 3 --warmup 2) > $R/$D/stats_iterate.log 2>&1 ; }
# round 5: the Level-2 driver's first call of a fresh handle; PMC passes of the solo sweep (the lines that had no `traffic` in round 4)
cd $R
[ $M = profiles ] || { (timeout 900 python bench.py --iterate --cold --steps 5 --warmup 1 --no-cpu-baseline) > $D/bench_iterate_cold.log 2>&1 ; }
[ $M = profiles ] || { (LX_HOST_TIMING=1 timeout 600 python tools/dev/cold_iterate.py 1000000 reserve) > $D/cold_iterate.log 2>&1 ; }
cd /tmp
for p in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "sq_wait:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${p%%:*}; c=${p#*:}
[ $M = lines ] || {   (timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$D/pmc_iterate_$n -o pmc -- python $R/bench.py --iterate --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_iterate_$n.log 2>&1 ; }
done
for p in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${p%%:*}; c=${p#*:}
[ $M = lines ] || {   (timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$D/pmc_ragged_nucl_$n -o pmc -- python $R/bench.py --ragged --entry list --config 2 --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_ragged_nucl_$n.log 2>&1 ; }
done
# round 6: the Level-2 driver on the protein list of configs[1] (the plan of the free packing made on the device): lines, kernel stats, PMC
# passes; the PMC passes of the long strong-hit list (sweep_mq_kernel<19,true,true>: the line that had no `traffic`)
cd $R
[ $M = profiles ] || { (timeout 900 python bench.py --iterate --config 1 --steps 8 --warmup 2) > $D/bench_iterate_protein.log 2>&1 ; }
[ $M = profiles ] || { (timeout 900 python bench.py --iterate --config 1 --cold --steps 4 --warmup 1 --no-cpu-baseline) > $D/bench_iterate_protein_cold.log 2>&1 ; }
cd /tmp
[ $M = lines ] || { (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats_iterate_protein -o iterate -- python $R/bench.py --iterate --config 1 --steps 3 --warmup 2 --no-cpu-baseline) > $R/$D/stats_iterate_protein.log 2>&1 ; }
for p in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "sq_wait:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${p%%:*}; c=${p#*:}
[ $M = lines ] || {   (timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$D/pmc_iterate_protein_$n -o pmc -- python $R/bench.py --iterate --config 1 --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_iterate_protein_$n.log 2>&1 ; }
[ $M = lines ] || {   (timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$D/pmc_strong_$n -o pmc -- python $R/bench.py --ragged --entry list --lq-range 500 800 --strong --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_strong_$n.log 2>&1 ; }
done
cd $R; tail -1 $D/bench.log | cut -c1-300; ls $D $D/stats | head -60
