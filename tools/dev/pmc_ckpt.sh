D=gpurun_out/ck1; mkdir -p $D; R=$PWD
cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/$D/pmc_sq -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_sq.log 2>&1
(timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$D/pmc_fetch -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $R/$D/pmc_fetch.log 2>&1
cd $R; ls $D/pmc_sq | head
