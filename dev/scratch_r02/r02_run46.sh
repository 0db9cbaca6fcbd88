#!/bin/bash
mkdir -p gpurun_out/r02; rm -rf gpurun_out/pmck_*
bash dev/pmc_kernel.sh ncf16_fwd bench.py --no-graph --no-cpu-baseline --no-gather-leg --steps 10 --warmup 3 > gpurun_out/r02/pmc_ncf16_fwd.txt 2>&1
tail -4 gpurun_out/r02/pmc_ncf16_fwd.txt
rm -rf gpurun_out/pmck_*
bash dev/pmc_kernel.sh ncf16_bwd bench.py --no-graph --no-cpu-baseline --no-gather-leg --steps 10 --warmup 3 > gpurun_out/r02/pmc_ncf16_bwd_final.txt 2>&1
tail -4 gpurun_out/r02/pmc_ncf16_bwd_final.txt
