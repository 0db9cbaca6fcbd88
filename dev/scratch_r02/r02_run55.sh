#!/bin/bash
mkdir -p gpurun_out/r02; rm -rf gpurun_out/pmck_*
bash dev/pmc_kernel.sh seg_reduce bench.py --no-graph --no-cpu-baseline --no-gather-leg --steps 10 --warmup 3 > gpurun_out/r02/pmc_seg_reduce.txt 2>&1
tail -4 gpurun_out/r02/pmc_seg_reduce.txt
