#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_models.py -q -m gpu -k "neuralcf or graph or trainer" 2>&1 | tail -8 > gpurun_out/r02/gpu_tests_ncf.txt; tail -4 gpurun_out/r02/gpu_tests_ncf.txt
for mode in 1 0; do
  CTR_MLP_PAIR=$mode timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline > gpurun_out/r02/bench_pair$mode.json 2> gpurun_out/r02/bench_pair$mode.err || tail -5 gpurun_out/r02/bench_pair$mode.err
done
python - <<'PY'
import json
for mode in (1, 0):
    try:
        d = json.load(open(f"gpurun_out/r02/bench_pair{mode}.json"))
    except Exception as e:
        print(mode, "no json", e); continue
    print("pair" if mode else "single", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: (v["avg_us"], v["frac"]) for k, v in d["kernels"].items()})
PY
