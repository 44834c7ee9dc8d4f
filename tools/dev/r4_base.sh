#!/bin/bash
# Round-4 baseline lease: GPU test suite on a fresh box, where lx_iterate_matches' time goes at configs[2] scale, PMC passes of the
# multi-query sweep on the ragged list, the long strong-hit query list.
D=gpurun_out/r4a; mkdir -p $D; R=$PWD
(timeout 1500 python -m pytest tests -m gpu -x -q) > $D/pytest.log 2>&1; tail -3 $D/pytest.log
(LX_HOST_TIMING=1 timeout 600 python tools/cli_scale_nucl.py 1000000 100) > $D/cli_nucl.log 2>&1; tail -12 $D/cli_nucl.log
(LX_HOST_TIMING=1 timeout 300 python tools/quick_iterate.py 20000) > $D/quick_iterate.log 2>&1; tail -8 $D/quick_iterate.log
(timeout 300 python tools/dev/long_queries.py) > $D/long_queries.log 2>&1; tail -2 $D/long_queries.log
(timeout 600 python bench.py --ragged --entry list --steps 5 --warmup 2) > $D/bench_ragged_list.log 2>&1; tail -1 $D/bench_ragged_list.log | cut -c1-400
cd /tmp; export TMPDIR=/tmp
for p in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "wait:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${p%%:*}; c=${p#*:}
  (timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$D/pmc_ragged_$n -o pmc -- python $R/bench.py --ragged --entry list --steps 2 --warmup 1) > $R/$D/pmc_ragged_$n.log 2>&1
done
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats_ragged -o ragged -- python $R/bench.py --ragged --entry list --steps 5 --warmup 2) > $R/$D/stats_ragged.log 2>&1
cd $R; ls $D
