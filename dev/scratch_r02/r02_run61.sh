#!/bin/bash
mkdir -p gpurun_out/r02
for v in 0 1 0 1; do
CTR_NCF_OVERLAP_SORT=$v timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r02/bench_z.json 2> gpurun_out/r02/bench_z.err || tail -5 gpurun_out/r02/bench_z.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_z.json"))
print("overlap sort=$v:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4))
PY
done
