#!/bin/bash
mkdir -p gpurun_out/r02
CTR_MLP_PAIR=1 timeout -k 10 300 python -m pytest tests/test_gpu_models.py -q -m gpu -k "neuralcf or graph or trainer" 2>&1 | tail -5
for mode in 1 0; do
  CTR_MLP_PAIR=$mode timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline > gpurun_out/r02/bench_pair$mode.json 2> gpurun_out/r02/bench_pair$mode.err || tail -5 gpurun_out/r02/bench_pair$mode.err
done
python - <<'PY'
import json
for mode in (1, 0):
    d = json.load(open(f"gpurun_out/r02/bench_pair{mode}.json"))
    print("pair" if mode else "single", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: (v["avg_us"], v["frac"]) for k, v in d["kernels"].items()})
PY
