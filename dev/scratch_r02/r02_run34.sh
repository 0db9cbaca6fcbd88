#!/bin/bash
mkdir -p gpurun_out/r02
for sl in 1024 2048 4096 512; do
CTR_SORT_SLICE=$sl timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_ncf_sl.json 2> gpurun_out/r02/bench_ncf.err || tail -5 gpurun_out/r02/bench_ncf.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_ncf_sl.json"))
print("slice $sl:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
