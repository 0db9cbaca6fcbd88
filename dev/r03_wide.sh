#!/bin/bash
# the 256 x 256 macro-tile forward (gemm_wide.hip): tests, then the GEMM microbenchmark with it on / off
set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "linear" > gpurun_out/r03/wide_tests.txt 2>&1 || { tail -30 gpurun_out/r03/wide_tests.txt; exit 1; }
tail -2 gpurun_out/r03/wide_tests.txt
for rep in 1 2; do
  echo "== wide on ($rep)"; python dev/gemm_bench.py 65536x256x512 65536x512x416 65536x256x128 3276800x256x64 2>&1 | grep -v amdgpu | cut -c1-60
  echo "== wide off ($rep)"; CTR_GEMM_WIDE=0 python dev/gemm_bench.py 65536x256x512 65536x512x416 65536x256x128 3276800x256x64 2>&1 | grep -v amdgpu | cut -c1-60
done | tee gpurun_out/r03/wide_ab.txt
python dev/gemm_bench.py 65536x256x512 2>&1 | grep -v amdgpu | tail -1
