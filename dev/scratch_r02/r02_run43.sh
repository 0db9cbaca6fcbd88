#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -k "mlp or neuralcf or ncf or NeuralCF or head" 2>&1 | tail -4 &&
CTRHIP_LIB=dev/timing/libctrhip_stamps.so timeout -k 10 300 python dev/ncf16_stamps.py 2>&1 | tail -4 &&
for m16 in 1 0; do
CTR_MLP_16=$m16 timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_z.json 2> gpurun_out/r02/bench_z.err || tail -5 gpurun_out/r02/bench_z.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_z.json"))
print("mlp16=$m16:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
