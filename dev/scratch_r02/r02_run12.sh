#!/bin/bash
R=$PWD; cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf $R/gpurun_out/r02_prof_pair$mode
  CTR_MLP_PAIR=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_pair$mode -- python3 $R/bench.py --no-gather-leg --no-cpu-baseline --no-graph --steps 20 > /dev/null 2>$R/gpurun_out/r02_prof_pair$mode.err
  echo "mode $mode"; grep -E "mlp_|reduce_seg" $R/gpurun_out/r02_prof_pair$mode/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
done
