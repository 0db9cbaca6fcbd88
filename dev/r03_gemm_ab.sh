#!/bin/bash
# A/B of two library builds on the GEMM microbenchmark (dev/gemm_bench.py), alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
SH="${@:-65536x256x512 65536x512x416 65536x128x256 3276800x128x64}"
for rep in 1 2; do
  echo "== new ($rep)"; python3 $R/dev/gemm_bench.py $SH
  echo "== old ($rep)"; CTRHIP_LIB=$R/dev/timing/libctrhip_old.so python3 $R/dev/gemm_bench.py $SH
done 2>&1 | tee $R/gpurun_out/r03/gemm_ab.txt
