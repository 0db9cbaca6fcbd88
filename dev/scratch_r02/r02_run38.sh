#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -q -m gpu -k "mlp or dien or DIEN or stack or neuralcf or NeuralCF or head" 2>&1 | tail -6
for w in dien neuralcf; do
timeout -k 10 300 python bench.py --workload $w --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_x.json 2> gpurun_out/r02/bench_x.err || tail -5 gpurun_out/r02/bench_x.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_x.json"))
print("$w:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: (v["avg_us"], v["calls_per_step"]) for k, v in list(d["kernels"].items())[:5]})
PY
done
