#!/bin/bash
# kernel traces (with the copies) of the host-buffer entry points, to read the GPU's idle gaps off: headline batch and ragged list
D=gpurun_out/${1:-th}; mkdir -p $D; R=$PWD
cd /tmp; export TMPDIR=/tmp
(LX_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$D/host -o host -- python $R/bench.py --host-path --entry list --steps 3 --warmup 2 --no-cpu-baseline) > $R/$D/host.log 2>&1
(LX_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$D/ragged -o ragged -- python $R/bench.py --ragged --entry list --steps 3 --warmup 2 --no-cpu-baseline) > $R/$D/ragged.log 2>&1
grep "lx host ms" $R/$D/host.log | tail -12
grep "lx host ms" $R/$D/ragged.log | tail -12
ls $R/$D/host $R/$D/ragged
