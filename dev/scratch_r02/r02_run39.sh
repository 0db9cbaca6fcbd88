#!/bin/bash
mkdir -p gpurun_out/r02
for cfg in "4 16" "2 16" "8 16" "4 8" "4 32" "2 32" "8 8"; do
set -- $cfg
CTR_ROWS_UNROLL=$1 CTR_ROWS_GRID=$2 timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_y.json 2> gpurun_out/r02/bench_y.err || tail -5 gpurun_out/r02/bench_y.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_y.json"))
print("unroll $1 grid $2:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), d["kernels"]["embed_fwd"]["avg_us"])
PY
done
