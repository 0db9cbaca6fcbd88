#!/bin/bash
mkdir -p gpurun_out/r02; rm -rf gpurun_out/pmck_*
bash dev/pmc_kernel.sh gru16_mfma_fwd bench.py --workload dien --no-graph --no-cpu-baseline --no-gather-leg --steps 5 --warmup 2 > gpurun_out/r02/pmc_gru_fwd.txt 2>&1
tail -4 gpurun_out/r02/pmc_gru_fwd.txt
