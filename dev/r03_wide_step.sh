#!/bin/bash
# gemm_wide.hip inside the steps that have deep wide layers: on / off
set -e
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "linear_forward or linear_backward" > gpurun_out/r03/wide_tests.txt 2>&1 || { tail -30 gpurun_out/r03/wide_tests.txt; exit 1; }
tail -1 gpurun_out/r03/wide_tests.txt
for wl in deepfm26 deepfm widedeep; do
  for v in 1 0 1 0; do
    CTR_GEMM_WIDE=$v timeout -k 10 300 python bench.py --workload $wl --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k.startswith('linear_fwd') and ('x512x' in k or 'x256x512' in k or 'x768x' in k)}
print('$wl wide=$v', round(d['ms_per_step']*1e3,1),'us', ks)"
  done
done
