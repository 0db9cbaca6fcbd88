#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -x -k "bce or loss or BCE or trainer or neuralcf" 2>&1 | tail -3 &&
for v in 1 2 3; do
timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r02/bench_z.json 2> gpurun_out/r02/bench_z.err || tail -5 gpurun_out/r02/bench_z.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_z.json"))
print("run $v:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
