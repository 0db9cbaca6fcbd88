#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -k "neuralcf or ncf or NeuralCF or graph" 2>&1 | tail -3
for ov in 1 0; do
CTR_NCF_OVERLAP_SORT=$ov timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_ncf_ov$ov.json 2> gpurun_out/r02/bench_ncf.err || tail -5 gpurun_out/r02/bench_ncf.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_ncf_ov$ov.json"))
print("overlap $ov:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "full", d["full_step"]["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
