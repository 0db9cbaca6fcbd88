#!/bin/bash
# full GPU suite + the script-shape leg (round 3)
set -e
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r03/suite_tests.txt 2>&1 || { tail -40 gpurun_out/r03/suite_tests.txt; exit 1; }
tail -3 gpurun_out/r03/suite_tests.txt
for v in 1 0; do
  CTR_NCF_PROJ=$v timeout -k 10 300 python bench.py --workload neuralcf_script --no-gather-leg --no-cpu-baseline > gpurun_out/r03/script2_$v.json 2> gpurun_out/r03/script2_$v.err || { tail -20 gpurun_out/r03/script2_$v.err; exit 1; }
  python - $v <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r03/script2_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("CTR_NCF_PROJ",sys.argv[1], round(d["value"]/1e6,2),"M/s", round(d["ms_per_step"]*1e3,1),"us")
ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1].get("total_us",kv[1].get("avg_us",0)))[:14]
for k,v in ks: print("   ",k, {a:(round(v[a],1) if isinstance(v[a],float) else v[a]) for a in ("avg_us","calls","frac","bound") if a in v})
P
done
