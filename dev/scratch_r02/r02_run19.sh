#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -q -m gpu -k "mlp or neuralcf or ncf or NeuralCF or head" 2>&1 | tail -5
for wv in 8 4; do
CTR_MLP_FWD_WAVES=$wv timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_ncf_w$wv.json 2> gpurun_out/r02/bench_ncf.err || tail -5 gpurun_out/r02/bench_ncf.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_ncf_w$wv.json"))
print("fwd waves $wv:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
