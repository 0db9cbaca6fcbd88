#!/bin/bash
# first-order (V, 1) gradients of small tables summed in LDS (ctr_rows1_scatter) against the per-sample atomics
set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -x -k "rows1 or ffm or lr or widedeep or wide or deepfm or nfm or afm or deepcross or golden or fixture" > gpurun_out/r03/rows1_tests.txt 2>&1 || { tail -40 gpurun_out/r03/rows1_tests.txt; exit 1; }
tail -2 gpurun_out/r03/rows1_tests.txt
for wl in ffm lr widedeep afm; do
  for v in 1 0; do
    CTR_ROWS1_LDS=$v timeout -k 10 300 python bench.py --workload $wl --no-gather-leg --no-cpu-baseline > gpurun_out/r03/rows1_${wl}_$v.json 2> gpurun_out/r03/rows1_${wl}_$v.err || { tail -20 gpurun_out/r03/rows1_${wl}_$v.err; exit 1; }
    python - $wl $v <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r03/rows1_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
ks={k:round(v["avg_us"],1) for k,v in d["kernels"].items() if any(t in k for t in ("ffm_fused_bwd","fm_wide_bwd","rows1","ffm_head_bwd"))}
print("%-10s CTR_ROWS1_LDS=%s %8.2f M/s %9.1f us/step  %s"%(sys.argv[1],sys.argv[2],d["value"]/1e6,d["ms_per_step"]*1e3,ks))
P
  done
done
