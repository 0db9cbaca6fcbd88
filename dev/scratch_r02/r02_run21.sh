#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -k "embed or neuralcf or ncf or NeuralCF or sorted" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -k "graph" 2>&1 | tail -2
for run in 8; do
CTR_SEG_RUN=$run timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_ncf_run$run.json 2> gpurun_out/r02/bench_ncf.err || tail -5 gpurun_out/r02/bench_ncf.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_ncf_run$run.json"))
print("run $run:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
