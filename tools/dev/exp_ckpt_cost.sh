#!/bin/bash
# what the headline sweep's checkpoint stores cost, piece by piece (development aid): kernel variants without the row checkpoints /
# without the boundary codes' flush -- their RESULTS are wrong (the backtrace reads what was never written), only the sweep's time counts
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in "" "-DLX_EXP_NO_ROWCK" "-DLX_EXP_NO_FLUSH" "-DLX_EXP_NO_ROWCK -DLX_EXP_NO_FLUSH"; do
  D=$R/gpurun_out/exp_ckpt/$(echo "v$v" | tr -d ' -'); rm -rf $D; mkdir -p $D
  (cd $R && LX_EXTRA_DEFINES="$v" python -c "import lambda_amd.build as b; b.build_product(force=True)") > $D/build.log 2>&1
  LX_EXTRA_DEFINES="$v" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $D/log 2>&1
  echo "variant [$v]: $(grep score_pair_kernel $D/k_kernel_stats.csv | cut -d, -f1-5 | cut -c1-160)"
done
(cd $R && python -c "import lambda_amd.build as b; b.build_product(force=True)") > /dev/null 2>&1
