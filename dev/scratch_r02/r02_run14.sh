#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline > gpurun_out/r02/bench_full.json 2> gpurun_out/r02/bench_full.err || tail -5 gpurun_out/r02/bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02/bench_full.json"))
print(round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: (v["avg_us"], v["frac"]) for k, v in d["kernels"].items()})
PY
