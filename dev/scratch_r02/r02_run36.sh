#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -k "embed or neuralcf or ncf or NeuralCF or sorted or ffm or FFM or pnn" 2>&1 | tail -3
for one in 1 0; do
for w in neuralcf ffm; do
CTR_SORT_ONE=$one timeout -k 10 300 python bench.py --workload $w --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_one.json 2> gpurun_out/r02/bench_one.err || tail -5 gpurun_out/r02/bench_one.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_one.json"))
print("sort one=$one $w:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in list(d["kernels"].items())[:4]})
PY
done
done
