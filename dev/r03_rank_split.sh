#!/bin/bash
# how the batch's rank atomics are split between the projection launch and the forward launch (per cent in the first)
set -e
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "neuralcf or ncf" > gpurun_out/r03/ncf_tests.txt 2>&1 || { tail -40 gpurun_out/r03/ncf_tests.txt; exit 1; }
tail -2 gpurun_out/r03/ncf_tests.txt
for sp in 0 30 40 50 60 100; do
  echo "== split $sp"
  CTR_NCFP_RANK_SPLIT=$sp bash dev/r03_probe.sh sp_$sp | grep "C step       ncfp_prep\|C step       ncfp_fwd"
  CTR_NCFP_RANK_SPLIT=$sp timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', round(d['value']/1e6,1),'M/s', round(d['ms_per_step']*1e3,2),'us')"
done
