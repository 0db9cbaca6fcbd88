#!/bin/bash
# tail-column split of the wide GEMMs: op tests, then Deep & Cross / Deep Crossing / DeepFM-26 bench lines
set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "linear" > gpurun_out/r03/tail_tests.txt 2>&1 || { tail -30 gpurun_out/r03/tail_tests.txt; exit 1; }
tail -2 gpurun_out/r03/tail_tests.txt
for wl in deepcross deepcrossing; do
  timeout -k 10 300 python bench.py --workload $wl --no-gather-leg --no-cpu-baseline > gpurun_out/r03/tail_$wl.json 2> gpurun_out/r03/tail_$wl.err || { tail -20 gpurun_out/r03/tail_$wl.err; exit 1; }
  python - $wl <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r03/tail_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]/1e6,2),"M/s", round(d["ms_per_step"],3),"ms")
ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1].get("total_us",kv[1].get("avg_us",0)))[:8]
for k,v in ks: print("   ",k, {a:(round(v[a],1) if isinstance(v[a],float) else v[a]) for a in ("avg_us","calls","frac","bound") if a in v})
P
done
